#!/bin/bash
# round 3, call 13: k_body<14> with the late rounds' keys in vector registers (no SGPR spill, no v_readlane in the row loop) against all keys in scalar registers (31 SGPRs spilled, 10 v_readlane per row)
O=$PWD/gpurun_out/r03_run13; mkdir -p $O
for rep in 1 2 3 4; do for v in _norkv "" _rkv12; do
  AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/ab$v$rep.json 2> $O/ab$v$rep.err
done; done
for rep in 1 2; do for v in _norkv ""; do
  AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python bench.py --decrypt --steps 8 --warmup 2 --no-cpu-baseline > $O/dec$v$rep.json 2> $O/dec$v$rep.err
done; done
python - $O <<'PY'
import json,sys,glob,os
for p in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d["roofline"]; c=r["formulation_ceiling"]
        print("%-20s %.1f GiB/s step %.3f ms kernel %.3f ms frac %.4f probe %.3f ms sclk %s/%s tag_ok %s" % (os.path.basename(p), d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], c["ms"], r["sclk_mhz"], c["sclk_mhz"], d["tag_ok"]))
    except Exception as e:
        print(p, "unreadable", e)
PY
