#!/bin/bash
# round 2, call 37: rows per chunk of k_body at 8 GiB, where 16 .. 128 are all allowed (does a shorter chunk than 64 pay in the kernel itself?)
O=gpurun_out/r02_run37; mkdir -p $O
for rep in 1 2 3; do for tw in 32 64 128 0; do
  if [ $tw = 0 ]; then unset AESGCM_TW; else export AESGCM_TW=$tw; fi
  timeout 300 python bench.py --gib-per-gpu 8 --steps 8 --warmup 2 --no-cpu-baseline > $O/tw$tw.$rep.json 2> $O/tw$tw.$rep.err
  python - $O/tw$tw.$rep.json $tw <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("tw %-4s %.1f GiB/s step %.3f ms kernel %.3f ms" % (sys.argv[2], d["value"], d["ms_per_step"], r["avg_launch_ms"]))
PY
done; done
