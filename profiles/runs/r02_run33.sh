#!/bin/bash
# round 2, call 33: k_body workgroup size (one workgroup per CU): 1024 lanes (4 waves per SIMD) against 896 and 768 (3 per SIMD)
O=gpurun_out/r02_run33; mkdir -p $O
for rep in 1 2; do for v in "" _wg896 _wg768; do
  AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/ab$v$rep.json 2> $O/ab$v$rep.err
  python - $O/ab$v$rep.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; c=r["formulation_ceiling"]
print("%-24s %.1f GiB/s kernel %.3f ms sclk %s  probe %.3f ms sclk %s lanes %s tag_ok %s" % (sys.argv[1].split("/")[-1], d["value"], r["avg_launch_ms"], r.get("sclk_mhz"), c["ms"], c["sclk_mhz"], d["config"]["wg_lanes"], d["tag_ok"]))
PY
done; done
