#!/usr/bin/env python3
"""Where the head / k_body / tail cut starts to pay: wall time of encrypt_dev (incl. tag readback, best of 7) with the
cut forced (AESGCM_BODY_MIN=4096) and disabled, by message size (GPU box).  Contexts read the variable at creation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib
MiB = 1 << 20
nmax = 2048 * MiB
a, b = lib.DeviceBuffer(nmax), lib.DeviceBuffer(nmax)
a.fill_splitmix64(1)
iv = bytes(12)
ctxs = {}
for name, v in (("split", "4096"), ("single", str(1 << 60))):
    os.environ["AESGCM_BODY_MIN"] = v
    ctxs[name] = lib.Context(bytes(range(32)))
print("%8s %12s %12s %8s" % ("MiB", "single us", "split us", "gain %"))
for mib in (16, 32, 64, 128, 256, 512, 1024, 2048):
    n = mib * MiB
    t = {}
    for name, ctx in ctxs.items():
        best = 1e9
        for it in range(7):
            t0 = time.perf_counter()
            ctx.encrypt_dev(iv, a.ptr, n, b.ptr)
            best = min(best, time.perf_counter() - t0)
        t[name] = best
    print("%8d %12.1f %12.1f %8.2f" % (mib, t["single"] * 1e6, t["split"] * 1e6, 100 * (t["single"] / t["split"] - 1)))
