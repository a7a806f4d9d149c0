#!/bin/bash
# round 3, call 43: bench.py on mid-size messages (--gib-per-gpu), default warmup / steps scaled to the step length
O=$PWD/gpurun_out/r03_run43; mkdir -p $O
for g in 0.015625 0.0625 0.25 1; do
  timeout 300 python bench.py --gib-per-gpu $g --no-cpu-baseline > $O/bench_g$g.json 2> $O/bench_g$g.err; echo "rc=$?"
done
timeout 300 python bench.py --config cfg2 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python - $O <<'PY'
import json,sys,glob,os
for p in sorted(glob.glob(sys.argv[1]+"/bench*.json")):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-24s %.1f GiB/s step %.4f ms (warmup %d steps %d) kernel %.4f ms %s frac %.3f" % (os.path.basename(p), d["value"], d["ms_per_step"], d["warmup"], d["steps"], r["avg_launch_ms"], r["kernel"][:28], r["frac"]))
    except Exception as e:
        print(p, "unreadable", e, open(p.replace('.json','.err')).read()[-300:])
PY
