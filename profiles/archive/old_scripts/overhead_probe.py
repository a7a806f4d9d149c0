#!/usr/bin/env python3
"""Where does the time go for mid-size messages?  wall time vs k_main event time, per chunk-size override."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib
GiB = 1 << 30
n = int(float(sys.argv[1]) * (1 << 20)) if len(sys.argv) > 1 else 256 << 20
a, b = lib.DeviceBuffer(n), lib.DeviceBuffer(n)
a.fill_splitmix64(1)
ctx = lib.Context(bytes(range(32)))
iv = bytes(12)
for it in range(3):
    ctx.encrypt_dev(iv, a.ptr, n, b.ptr)
ctx.timing_enable(True)
best = (1e9, 0)
for it in range(5):
    ctx.timing_read(reset=True)
    t0 = time.perf_counter()
    ctx.encrypt_dev(iv, a.ptr, n, b.ptr)
    wall = time.perf_counter() - t0
    nl, ms = ctx.timing_read(reset=True)
    best = min(best, (wall, ms))
tr = ctx.wg_trace()
chunks = sum(t[3] & 0xFFFFFFFF for t in tr)
print("TW=%s  %d MiB: wall %.1f us, k_main %.1f us (%.0f GiB/s kernel-only), %d chunks, %d workgroups" % (
    os.environ.get("AESGCM_TW", "auto"), n >> 20, best[0] * 1e6, best[1] * 1e3, n / (best[1] / 1e3) / GiB, chunks, len(tr)))
