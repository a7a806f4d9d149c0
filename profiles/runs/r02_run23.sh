#!/bin/bash
# round 2, call 23: dispensers with two alternating queue sets (no failing fetch per wave per queue): GPU suite, stress, Tw sweep, sizes, 16 GiB A/B
O=gpurun_out/r02_run23; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
timeout 600 python profiles/tw_sweep.py 1 4 8 16 32 64 100 127 > $O/tw_sweep.txt 2>&1; cat $O/tw_sweep.txt
for v in _fg ""; do echo "== sizes $v"; AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import aesgcm_amd
from aesgcm_amd import lib
MiB = 1 << 20
a, b = lib.DeviceBuffer(4096 * MiB), lib.DeviceBuffer(4096 * MiB)
a.fill_splitmix64(1)
ctx = lib.Context(bytes(range(32)))
for mib in (128, 256, 512, 1024, 2048, 4096):
    best = 1e9
    for it in range(9):
        t0 = time.perf_counter(); ctx.encrypt_dev(bytes(12), a.ptr, mib * MiB, b.ptr); best = min(best, time.perf_counter() - t0)
    print("%6d MiB %8.1f us %7.1f GiB/s" % (mib, best * 1e6, mib / 1024 / best))
PY
done
for rep in 1 2; do for v in _fg ""; do
  AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/ab$v$rep.json 2> $O/ab$v$rep.err
  python - $O/ab$v$rep.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("%-24s %.1f GiB/s step %.3f ms kernel %.3f ms sclk %s tag_ok %s" % (sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], r["avg_launch_ms"], r.get("sclk_mhz"), d["tag_ok"]))
PY
done; done
