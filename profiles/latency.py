#!/usr/bin/env python3
"""Small-message latency of the device-pointer path: wall time of aesgcm_encrypt_dev (launches + 16-byte tag copy +
stream sync) per message size, median and best of N calls.  Run bare for the latency table, under
`rocprofv3 --kernel-trace --stats` for the per-kernel split.   python profiles/latency.py [N]"""
import os
import statistics
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa: E402,F401
from aesgcm_amd import lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = lib.Context(bytes(range(32)))
iv = bytes(range(12))
buf_in, buf_out = lib.DeviceBuffer(4 << 20), lib.DeviceBuffer(4 << 20)
buf_in.fill_splitmix64(1)
d_aad = lib.DeviceBuffer(64); d_aad.upload(bytes(range(28)))
lib.dev_sync()
print("%10s %10s %10s   (AES-256-GCM, device-resident, tag to host; us per call)" % ("bytes", "median", "best"))
for n, aad in ((48, 28), (1024, 0), (16 << 10, 0), (64 << 10, 0), (256 << 10, 0), (1 << 20, 0), (4 << 20, 0)):
    for _ in range(20):
        ctx.encrypt_dev(iv, buf_in.ptr, n, buf_out.ptr, d_aad=d_aad.ptr if aad else None, aad_len=aad)
    ts = []
    for _ in range(N):
        t0 = time.perf_counter()
        ctx.encrypt_dev(iv, buf_in.ptr, n, buf_out.ptr, d_aad=d_aad.ptr if aad else None, aad_len=aad)
        ts.append(time.perf_counter() - t0)
    print("%10d %10.1f %10.1f" % (n, statistics.median(ts) * 1e6, min(ts) * 1e6))
