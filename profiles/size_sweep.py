#!/usr/bin/env python3
"""Device-resident throughput vs message size, key size and direction (GPU box).  Wall time around
encrypt_dev/decrypt_dev including the 16-byte tag readback, best of 5."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib
GiB = 1 << 30
nmax = 4 * GiB
a, b = lib.DeviceBuffer(nmax), lib.DeviceBuffer(nmax)
a.fill_splitmix64(1)
iv = bytes(12)
print("%-10s %-8s %-4s %10s %10s" % ("size", "key", "dir", "us", "GiB/s"))
for kb in (16, 24, 32):
    ctx = lib.Context(bytes(range(kb)))
    sizes = [4 << 10, 64 << 10, 1 << 20, 16 << 20, 256 << 20, 1 * GiB, 4 * GiB] if kb == 32 else [1 * GiB, 4 * GiB]
    for n in sizes:
        for d in ("enc", "dec"):
            best = 1e9
            for it in range(5):
                t0 = time.perf_counter()
                if d == "enc":
                    ctx.encrypt_dev(iv, a.ptr, n, b.ptr)
                else:
                    ctx.decrypt_dev(iv, a.ptr, n, b.ptr)
                best = min(best, time.perf_counter() - t0)
            print("%-10d AES-%-4d %-4s %10.1f %10.1f" % (n, kb * 8, d, best * 1e6, n / best / GiB))
