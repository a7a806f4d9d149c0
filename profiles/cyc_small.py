#!/usr/bin/env python3
"""Where the fused cyclic launch starts to pay: small messages, the k_main paths against the cyclic launch forced down (GPU box).
encrypt_dev incl. tag readback, median of 60 calls (us); shapes: aligned / 20 B of AAD and 5 odd bytes."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib
KiB = 1 << 10
a, b = lib.DeviceBuffer(8192 * KiB + 64), lib.DeviceBuffer(8192 * KiB + 64)
a.fill_splitmix64(1)
d_aad = lib.DeviceBuffer(64); d_aad.upload(bytes(range(64)))
iv = bytes(12)
for kb in (32, 16):
    old = lib.Context(bytes(range(kb))).set_option("cyc_min", 0).set_option("cyc_max", 0)
    cyc = lib.Context(bytes(range(kb))).set_option("cyc_min", 16 * KiB).set_option("cyc_max", 1 << 50)
    print("AES-%d   KiB   k_main   cyclic   k_main+pieces  cyclic+pieces  (us)" % (kb * 8))
    for kib in (16, 32, 64, 128, 192, 256, 384, 512, 768, 1024, 2048, 4096):
        row = []
        for al, extra in ((0, 0), (20, 5)):
            for ctx in (old, cyc):
                n = kib * KiB + extra
                ts = []
                for it in range(60):
                    t0 = time.perf_counter()
                    ctx.encrypt_dev(iv, a.ptr, n, b.ptr, d_aad=d_aad.ptr if al else None, aad_len=al)
                    ts.append(time.perf_counter() - t0)
                row.append(statistics.median(ts[5:]) * 1e6)
        print("       %6d  %7.1f  %7.1f  %12.1f  %12.1f" % (kib, *row), flush=True)
