#!/bin/bash
# round 2, call 19: where the k_body cut pays on the five-bit-table build; k_main-only sizes against the nibble build
O=gpurun_out/r02_run19; mkdir -p $O
timeout 600 python profiles/split_threshold.py > $O/split_threshold.txt 2>&1; cat $O/split_threshold.txt
for v in _gh4 ""; do echo "== k_main sizes $v"; AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import aesgcm_amd
from aesgcm_amd import lib
MiB = 1 << 20
a, b = lib.DeviceBuffer(128 * MiB), lib.DeviceBuffer(128 * MiB)
a.fill_splitmix64(1)
ctx = lib.Context(bytes(range(32)))
for mib in (4, 16, 32, 64, 96, 127):
    best = 1e9
    for it in range(9):
        t0 = time.perf_counter(); ctx.encrypt_dev(bytes(12), a.ptr, mib * MiB, b.ptr); best = min(best, time.perf_counter() - t0)
    print("%4d MiB %8.1f us %7.1f GiB/s" % (mib, best * 1e6, mib / 1024 / best))
PY
done
