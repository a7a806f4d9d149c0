#!/bin/bash
# round 3, call 35: is cfg2's step (1 ms) measured while the clock is still ramping?  warmup 3 / 30 / 100, steps 20 / 100
O=$PWD/gpurun_out/r03_run35; mkdir -p $O
for rep in 1 2; do for ws in "3 20" "30 20" "100 20" "30 100" "3 100"; do
  set -- $ws
  timeout 300 python bench.py --config cfg2 --warmup $1 --steps $2 --no-cpu-baseline > $O/cfg2_w$1_s$2_$rep.json 2> $O/cfg2_w$1_s$2_$rep.err
done; done
python - $O <<'PY'
import json,sys,glob,os
for p in sorted(glob.glob(sys.argv[1]+"/cfg2_*.json")):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-26s %.1f GiB/s step %.3f ms kernel %.3f ms tag_ok %s sclk %s" % (os.path.basename(p), d["value"], d["ms_per_step"], r["avg_launch_ms"], d["tag_ok"], r.get("sclk_mhz")))
    except Exception as e:
        print(p, "unreadable", e)
PY
