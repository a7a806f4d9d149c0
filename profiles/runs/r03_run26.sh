#!/bin/bash
# round 3, call 26: does the 200 us poll budget cost the 1 GiB message anything (it ends in hipStreamSynchronize)?  + the timeline of the 1 GiB tail
O=$PWD/gpurun_out/r03_run26; mkdir -p $O
for rep in 1 2; do for us in 200 5000; do
  AESGCM_POLL_US=$us timeout 300 python bench.py --config cfg2 --steps 20 --warmup 3 --no-cpu-baseline > $O/cfg2_poll${us}_$rep.json 2> $O/cfg2_poll${us}_$rep.err
done; done
python - $O <<'PY'
import json,sys,glob,os
for p in sorted(glob.glob(sys.argv[1]+"/cfg2_*.json")):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-26s %.1f GiB/s step %.3f ms kernel %.3f ms tag_ok %s" % (os.path.basename(p), d["value"], d["ms_per_step"], r["avg_launch_ms"], d["tag_ok"]))
    except Exception as e:
        print(p, "unreadable", e)
PY
