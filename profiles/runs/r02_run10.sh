#!/bin/bash
O=gpurun_out/r02_run10; mkdir -p $O
for rep in 1 2; do for v in "" _prev; do
  AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/ab$v$rep.json 2> $O/ab$v$rep.err
  python - $O/ab$v$rep.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[1], "%.1f GiB/s kernel %.3f ms ceiling %s sclk %s" % (d["value"], r["avg_launch_ms"], (r.get("formulation_ceiling") or {}).get("value"), r.get("sclk_mhz")))
PY
done; done
