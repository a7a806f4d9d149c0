#!/bin/bash
# round 2, call 21: k_fold with a runtime group size (fold_group): full GPU suite, then sizes 64 KiB .. 1 GiB against the build before (gh5c)
O=gpurun_out/r02_run21; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
for v in _gh5c ""; do echo "== sizes $v"; AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import aesgcm_amd
from aesgcm_amd import lib
MiB = 1 << 20
a, b = lib.DeviceBuffer(1024 * MiB), lib.DeviceBuffer(1024 * MiB)
a.fill_splitmix64(1)
ctx = lib.Context(bytes(range(32)))
for kib in (64, 256, 512, 1024, 4096, 16384, 32768, 65536, 130048, 262144, 1048576):
    best = 1e9
    for it in range(15):
        t0 = time.perf_counter(); ctx.encrypt_dev(bytes(12), a.ptr, kib * 1024, b.ptr); best = min(best, time.perf_counter() - t0)
    print("%8d KiB %8.1f us %7.1f GiB/s" % (kib, best * 1e6, kib / 1048576 / best))
PY
done
