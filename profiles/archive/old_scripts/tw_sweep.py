#!/usr/bin/env python3
"""Rows per chunk (AESGCM_TW) against message size: wall time of encrypt_dev incl. tag readback, best of 7 (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib
MiB = 1 << 20
sizes = [int(x) for x in (sys.argv[1:] or ["1", "4", "16", "64", "256", "1024", "4096"])]
nmax = max(sizes) * MiB
a, b = lib.DeviceBuffer(nmax), lib.DeviceBuffer(nmax)
a.fill_splitmix64(1)
iv = bytes(12)
tws = [0, 1, 2, 4, 8, 16, 32, 64, 128]
print("%8s " % "MiB" + " ".join("%8s" % ("auto" if t == 0 else "tw%d" % t) for t in tws) + "   (us)")
ctxs = {}
for t in tws:
    if t: os.environ["AESGCM_TW"] = str(t)
    else: os.environ.pop("AESGCM_TW", None)
    ctxs[t] = lib.Context(bytes(range(32)))
for mib in sizes:
    n = mib * MiB
    row = []
    for t in tws:
        if t and (n // 1024) // t > (1 << 18):
            row.append(float("nan")); continue
        best = 1e9
        for it in range(7):
            t0 = time.perf_counter(); ctxs[t].encrypt_dev(iv, a.ptr, n, b.ptr); best = min(best, time.perf_counter() - t0)
        row.append(best * 1e6)
    print("%8d " % mib + " ".join("%8.1f" % v for v in row), flush=True)
