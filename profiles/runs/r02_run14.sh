#!/bin/bash
O=gpurun_out/r02_run14; mkdir -p $O
export AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip_t4.so
timeout 1800 python -m pytest tests -m gpu -x -q -k "large or fuzz or parity or pipeline" > $O/pytest_t4.log 2>&1; grep -E "passed|failed" $O/pytest_t4.log | tail -2
unset AESGCM_LIB
for rep in 1 2 3; do for v in "" _t4; do
  AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/ab$v$rep.json 2> $O/ab$v$rep.err
  python - $O/ab$v$rep.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("%-24s %.1f GiB/s kernel %.3f ms tag_ok %s" % (sys.argv[1].split("/")[-1], d["value"], r["avg_launch_ms"], d["tag_ok"]))
PY
done; done
for v in "" _t4; do AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python bench.py --decrypt --steps 8 --warmup 2 --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('dec $v', d['value'], r['avg_launch_ms'], d['tag_ok'])"; done
