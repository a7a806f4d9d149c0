#!/usr/bin/env python3
"""Device-resident throughput vs message size, key size and direction (GPU box).  Wall time around
encrypt_dev/decrypt_dev including the 16-byte tag readback; per size 4 untimed calls (the clock needs a few milliseconds of load to settle: best-of-5 from
cold read 1 GiB at 852 GiB/s where 24 calls give 978, profiles/r04/cyc_sweep_aes256.txt), then median and best of 12."""
import statistics
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib
GiB = 1 << 30
nmax = 4 * GiB
a, b = lib.DeviceBuffer(nmax), lib.DeviceBuffer(nmax)
a.fill_splitmix64(1)
iv = bytes(12)
print("%-10s %-8s %-4s %10s %10s %10s %10s" % ("size", "key", "dir", "median us", "GiB/s", "best us", "GiB/s"))
for kb in (16, 24, 32):
    ctx = lib.Context(bytes(range(kb)))
    sizes = [4 << 10, 64 << 10, 1 << 20, 16 << 20, 256 << 20, 1 * GiB, 4 * GiB] if kb == 32 else [1 * GiB, 4 * GiB]
    for n in sizes:
        for d in ("enc", "dec"):
            ts = []
            for it in range(16):
                t0 = time.perf_counter()
                if d == "enc":
                    ctx.encrypt_dev(iv, a.ptr, n, b.ptr)
                else:
                    ctx.decrypt_dev(iv, a.ptr, n, b.ptr)
                if it >= 4:
                    ts.append(time.perf_counter() - t0)
            med, best = statistics.median(ts), min(ts)
            print("%-10d AES-%-4d %-4s %10.1f %10.1f %10.1f %10.1f" % (n, kb * 8, d, med * 1e6, n / med / GiB, best * 1e6, n / best / GiB))
