#!/bin/bash
# round 2, call 18: byte-1 table addresses from one v_bitop3 instead of v_perm (AESGCM_ADDR_B1) against the build before it (gh5c), same box
O=gpurun_out/r02_run18; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -k "large or fuzz or parity or pipeline or fixtures or batch or packets" > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
for rep in 1 2 3; do for v in _gh5c ""; do
  AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/ab$v$rep.json 2> $O/ab$v$rep.err
  python - $O/ab$v$rep.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; c=r["formulation_ceiling"]
print("%-24s %.1f GiB/s kernel %.3f ms sclk %s  probe %.3f ms sclk %s tag_ok %s" % (sys.argv[1].split("/")[-1], d["value"], r["avg_launch_ms"], r.get("sclk_mhz"), c["ms"], c["sclk_mhz"], d["tag_ok"]))
PY
done; done
for v in _gh5c ""; do AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python bench.py --config cfg2 --steps 20 --warmup 3 --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('cfg2 $v', d['value'], r['avg_launch_ms'], d['tag_ok'])"; done
for v in _gh5c ""; do AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python profiles/pkt_bench.py batch --steps 7 2>&1 | tail -1 | cut -c1-160; done
