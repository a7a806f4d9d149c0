#!/bin/bash
# round 2, call 36: rows per chunk of k_body at 16 GiB: 64 (forced by the 2^18-chunk limit) against 128 and 256
O=gpurun_out/r02_run36; mkdir -p $O
for rep in 1 2 3; do for tw in 0 128 256; do
  if [ $tw = 0 ]; then unset AESGCM_TW; else export AESGCM_TW=$tw; fi
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/tw$tw.$rep.json 2> $O/tw$tw.$rep.err
  python - $O/tw$tw.$rep.json $tw <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("tw %-4s %.1f GiB/s step %.3f ms kernel %.3f ms tag_ok %s" % (sys.argv[2], d["value"], d["ms_per_step"], r["avg_launch_ms"], d["tag_ok"]))
PY
done; done
