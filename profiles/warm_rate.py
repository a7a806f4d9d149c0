#!/usr/bin/env python3
"""Cold and warm: the same message encrypted back to back for 300 ms (GPU box).  The first calls run on a chip whose clock is still ramping (what an
isolated call sees, and what the latency tables quote); the last third is the sustained rate.  encrypt_dev incl. tag, AES-256, us per call."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib
kb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
MiB = 1 << 20
a, b = lib.DeviceBuffer(1024 * MiB), lib.DeviceBuffer(1024 * MiB)
a.fill_splitmix64(1)
iv = bytes(12)
ctx = lib.Context(bytes(range(kb)))
print("AES-%d      KiB   first 5 (us)                      median of the last third   GiB/s sustained" % (kb * 8))
for kib in (64, 1024, 16384, 65536, 262144, 1048576):
    n = kib << 10
    lib.dev_sync(); time.sleep(0.5)                                  # let the chip go idle
    ts = []
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end or len(ts) < 12:
        t0 = time.perf_counter()
        ctx.encrypt_dev(iv, a.ptr, n, b.ptr)
        ts.append(time.perf_counter() - t0)
    last = ts[2 * len(ts) // 3:]
    m = statistics.median(last)
    print("         %8d   %-32s  %8.1f  (%d calls)           %8.1f" % (kib, " ".join("%.1f" % (t * 1e6) for t in ts[:5]), m * 1e6, len(ts), n / m / (1 << 30)), flush=True)
