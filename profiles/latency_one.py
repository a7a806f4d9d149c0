#!/usr/bin/env python3
"""One message size, many calls: for `rocprofv3 --kernel-trace --stats` (per-kernel split of the small-message path).
    python profiles/latency_one.py BYTES [CALLS]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa: E402,F401
from aesgcm_amd import lib  # noqa: E402
n, calls = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 300
ctx = lib.Context(bytes(range(32)))
a, b = lib.DeviceBuffer(max(n, 16)), lib.DeviceBuffer(max(n, 16))
a.fill_splitmix64(1); lib.dev_sync()
for _ in range(calls):
    ctx.encrypt_dev(bytes(12), a.ptr, n, b.ptr)
