#!/bin/bash
O=gpurun_out/r02_run12; mkdir -p $O
for rep in 1 2; do for v in "" _fx_NTSTORE _fx_NTLOAD _fx_NTBOTH _fx_NOLOAD _fx_NOSTORE; do
  AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/ab$v$rep.json 2> $O/ab$v$rep.err
  python - $O/ab$v$rep.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; c=r.get("formulation_ceiling") or {}
print("%-44s %.1f GiB/s kernel %.3f ms sclk %s | ceiling %.3f ms sclk %s | tag_ok %s" % (sys.argv[1].split("/")[-1], d["value"], r["avg_launch_ms"], r.get("sclk_mhz"), c.get("ms",0), c.get("sclk_mhz"), d["tag_ok"]))
PY
done; done
