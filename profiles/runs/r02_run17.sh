#!/bin/bash
# round 2, call 17: five-bit GHASH tables in every AES kernel (k_main, k_pkt, k_pktl as well as k_body): full GPU suite, then
# same-box A/B against the build with them in k_body only (gh5b) and the nibble build (gh4)
O=gpurun_out/r02_run17; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
for v in _gh4 _gh5b ""; do
  echo "== size sweep $v"; AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python profiles/size_sweep.py > $O/size_sweep$v.txt 2>&1; grep "AES-256" $O/size_sweep$v.txt
done
for v in _gh4 ""; do for k in pktw pktl; do
  echo "== $k $v"; AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python profiles/pkt_bench.py $k --len 1024 --key-bits 256 --steps 7 2>&1 | tail -1 | cut -c1-200
done; done
for rep in 1 2; do for v in _gh5b ""; do
  AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/ab$v$rep.json 2> $O/ab$v$rep.err
  python - $O/ab$v$rep.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("%-24s %.1f GiB/s kernel %.3f ms sclk %s tag_ok %s" % (sys.argv[1].split("/")[-1], d["value"], r["avg_launch_ms"], r.get("sclk_mhz"), d["tag_ok"]))
PY
done; done
