#!/bin/bash
# round 3, call 36: clock ramp and the other configs: cfg5 (6 ms steps) and cfg3 (16 ms) with short and long warmups
O=$PWD/gpurun_out/r03_run36; mkdir -p $O
for rep in 1 2; do
  for w in 1 3 20 60; do timeout 300 python bench.py --config cfg5 --warmup $w --steps 20 --no-cpu-baseline > $O/cfg5_w${w}_$rep.json 2> $O/cfg5_w${w}_$rep.err; done
  for w in 1 3 10; do timeout 600 python bench.py --warmup $w --steps 10 --no-cpu-baseline > $O/cfg3_w${w}_$rep.json 2> $O/cfg3_w${w}_$rep.err; done
done
python - $O <<'PY'
import json,sys,glob,os
for p in sorted(glob.glob(sys.argv[1]+"/cfg*.json")):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-26s %.1f GiB/s step %.3f ms kernel %.3f ms tag_ok %s" % (os.path.basename(p), d["value"], d["ms_per_step"], r["avg_launch_ms"], d["tag_ok"]))
    except Exception as e:
        print(p, "unreadable", e)
PY
