#!/bin/bash
# round 2, call 16: k_body GHASH through five-bit tables / ds_read_b64 (AESGCM_GH5) against the nibble / ds_read_b128 build, same box
O=gpurun_out/r02_run16; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -k "large or fuzz or parity or pipeline or fixtures" > $O/pytest_gh5.log 2>&1; grep -E "passed|failed" $O/pytest_gh5.log | tail -2
for rep in 1 2 3; do for v in _gh4 ""; do
  AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/ab$v$rep.json 2> $O/ab$v$rep.err
  python - $O/ab$v$rep.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("%-24s %.1f GiB/s kernel %.3f ms sclk %s ceiling %s tag_ok %s" % (sys.argv[1].split("/")[-1], d["value"], r["avg_launch_ms"], r.get("sclk_mhz"), r.get("formulation_ceiling"), d["tag_ok"]))
PY
done; done
for v in _gh4 ""; do AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python bench.py --decrypt --steps 8 --warmup 2 --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('dec $v', d['value'], r['avg_launch_ms'], d['tag_ok'])"; done
for v in _gh4 ""; do AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python bench.py --config cfg2 --steps 20 --warmup 3 --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('cfg2 $v', d['value'], r['avg_launch_ms'], d['tag_ok'])"; done
