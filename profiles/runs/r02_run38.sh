#!/bin/bash
# round 2, call 38: issue priority raised for the waves that take the last chunks of their queue (AESGCM_TAIL_PRIO) against the build without
O=gpurun_out/r02_run38; mkdir -p $O
for rep in 1 2 3 4; do for v in _base _prio; do
  AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/ab$v$rep.json 2> $O/ab$v$rep.err
  python - $O/ab$v$rep.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("%-24s %.1f GiB/s step %.3f ms kernel %.3f ms tag_ok %s" % (sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], r["avg_launch_ms"], d["tag_ok"]))
PY
done; done
