#!/usr/bin/env python3
"""A few encrypt_dev calls per size for `rocprofv3 --kernel-trace --stats` (per-kernel durations by message size)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib
sizes = [int(float(x) * (1 << 20)) for x in (sys.argv[1:] or ["1", "16", "256", "1024"])]
nmax = max(sizes)
a, b = lib.DeviceBuffer(nmax), lib.DeviceBuffer(nmax)
a.fill_splitmix64(1)
ctx = lib.Context(bytes(range(32)))
for n in sizes:
    for _ in range(10):
        ctx.encrypt_dev(bytes(12), a.ptr, n, b.ptr)
