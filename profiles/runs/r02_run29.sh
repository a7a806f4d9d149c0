#!/bin/bash
# round 2, call 29: k_fold with 4 waves per workgroup against 8
O=$PWD/gpurun_out/r02_run29; mkdir -p $O
REPO=$PWD
AESGCM_LIB=$REPO/aes-gcm-128-192-256-bits_amd/libaesgcm_hip_fw4.so timeout 1200 python -m pytest tests -m gpu -x -q -k "fold or large or fuzz" > $O/pytest_fw4.log 2>&1; grep -E "passed|failed" $O/pytest_fw4.log | tail -2
for v in _fw8 _fw4; do
  export AESGCM_LIB=$REPO/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$v -- python3 $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/s$v.log 2>&1)
  f=$(find $O/s$v -name "*kernel_stats.csv" | head -1); echo "== lib '$v'"; grep -E "k_fold|k_combine" $f | cut -d, -f1-4 | cut -c1-140
done
unset AESGCM_LIB
for rep in 1 2 3; do for v in _fw8 _fw4; do
  AESGCM_LIB=$REPO/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/ab$v$rep.json 2> $O/ab$v$rep.err
  python - $O/ab$v$rep.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("%-24s %.1f GiB/s step %.3f ms kernel %.3f ms tag_ok %s" % (sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], r["avg_launch_ms"], d["tag_ok"]))
PY
done; done
for v in _fw8 _fw4; do echo "== sizes $v"; AESGCM_LIB=$REPO/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python profiles/size_sweep.py 2>&1 | grep "AES-256.*enc"; done
