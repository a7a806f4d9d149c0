#!/usr/bin/env python3
"""What AAD and a ragged end cost at mid sizes (GPU box): the same message size as one aligned body, with 20 bytes of AAD,
with 5 extra bytes, with both.  encrypt_dev incl. tag readback, median of 30 calls (us)."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib
MiB = 1 << 20
a, b = lib.DeviceBuffer(513 * MiB), lib.DeviceBuffer(513 * MiB)
a.fill_splitmix64(1)
d_aad = lib.DeviceBuffer(64); d_aad.upload(bytes(range(64)))
iv = bytes(12)
ctx = lib.Context(bytes(range(32)))
print("   MiB    aligned   +aad20   +5bytes   +both   (us, AES-256)")
for mib in (1, 4, 16, 64, 256, 320, 384, 448, 512):
    row = []
    for al, extra in ((0, 0), (20, 0), (0, 5), (20, 5)):
        n = mib * MiB + extra
        ts = []
        for it in range(30):
            t0 = time.perf_counter()
            ctx.encrypt_dev(iv, a.ptr, n, b.ptr, d_aad=d_aad.ptr if al else None, aad_len=al)
            ts.append(time.perf_counter() - t0)
        row.append(statistics.median(ts[3:]) * 1e6)
    print("%6d  %9.1f %9.1f %9.1f %9.1f" % (mib, *row), flush=True)
old = lib.Context(bytes(range(32))).set_option("cyc_min", 0).set_option("cyc_max", 0)
print("round-2 paths (options cyc_min = cyc_max = 0)")
for mib in (256, 320, 384, 448, 512):
    row = []
    for al, extra in ((0, 0), (20, 5)):
        n = mib * MiB + extra
        ts = []
        for it in range(30):
            t0 = time.perf_counter()
            old.encrypt_dev(iv, a.ptr, n, b.ptr, d_aad=d_aad.ptr if al else None, aad_len=al)
            ts.append(time.perf_counter() - t0)
        row.append(statistics.median(ts[3:]) * 1e6)
    print("%6d  %9.1f %9.1f" % (mib, *row), flush=True)
