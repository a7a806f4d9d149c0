#!/bin/bash
# round 2, GPU call 1: microbench, full -m gpu suite, workgroup-size A/B (scratch-free builds), default bench line
O=gpurun_out/r02_run1; mkdir -p $O
( cd profiles/microbench && timeout 120 ./overlap ) > $O/overlap.txt 2>&1
for v in "" _768 _896; do
  AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/ab$v.json 2> $O/ab$v.err
done
for v in "" _768 _896; do
  AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/ab2$v.json 2> $O/ab2$v.err
done
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --config cfg2 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 300 python bench.py --decrypt --no-cpu-baseline --steps 10 > $O/bench_dec.json 2> $O/bench_dec.err
timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
cat $O/overlap.txt
for f in $O/ab*.json $O/bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(" value %.1f GiB/s  ms/step %.3f  kernel %.3f ms  frac %.4f  sclk %s  ceiling %s  tag_ok %s" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], r.get("sclk_mhz"), (r.get("formulation_ceiling") or {}).get("value"), d["tag_ok"]))
    if "cpu_baseline" in d: print(" cpu:", {k:v for k,v in d["cpu_baseline"].items() if k in ("value","value_1core","cores","lib","cpu_model")})
except Exception as e:
    print(" parse failed", e)
PY
done
