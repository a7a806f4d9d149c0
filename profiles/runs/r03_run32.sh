#!/bin/bash
# round 3, call 32: priority rotation in the DEALT k_body (experiment): cfg3 and cfg2, alternating
O=$PWD/gpurun_out/r03_run32; mkdir -p $O
for rep in 1 2 3; do for pr in 0 2 8; do
  AESGCM_BODY_PRIO=$pr timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/cfg3_prio${pr}_$rep.json 2> $O/cfg3_prio${pr}_$rep.err
  AESGCM_BODY_PRIO=$pr timeout 300 python bench.py --config cfg2 --steps 20 --warmup 3 --no-cpu-baseline > $O/cfg2_prio${pr}_$rep.json 2> $O/cfg2_prio${pr}_$rep.err
done; done
python - $O <<'PY'
import json,sys,glob,os
for p in sorted(glob.glob(sys.argv[1]+"/cfg*.json")):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-26s %.1f GiB/s step %.3f ms kernel %.3f ms tag_ok %s" % (os.path.basename(p), d["value"], d["ms_per_step"], r["avg_launch_ms"], d["tag_ok"]))
    except Exception as e:
        print(p, "unreadable", e)
PY
