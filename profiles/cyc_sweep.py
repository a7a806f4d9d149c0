#!/usr/bin/env python3
"""k_body as cyclic rows (with k_fold + k_combine behind the launch, and with the fused closing) against the round-2 paths (k_main below 256 MiB, dealt k_body from there), by message size (GPU box).
Device-resident encrypt_dev incl. the tag readback, median and best of N calls, alternating the two settings call by call
group so that clock drift hits both.  The three contexts differ in their options (aesgcm_ctx_set_option).
    python profiles/cyc_sweep.py [key_bytes]"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib
kb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
MiB = 1 << 20
nmax = 4096 * MiB
a, b = lib.DeviceBuffer(nmax), lib.DeviceBuffer(nmax)
a.fill_splitmix64(1)
iv = bytes(12)
old = lib.Context(bytes(range(kb))).set_option("cyc_min", 0).set_option("cyc_max", 0)
cyc = lib.Context(bytes(range(kb))).set_option("cyc_min", 1 * MiB).set_option("cyc_max", 1 << 50).set_option("cyc_close", 0)
fus = lib.Context(bytes(range(kb))).set_option("cyc_min", 1 * MiB).set_option("cyc_max", 1 << 50).set_option("cyc_close", 1)
print("AES-%d   MiB    old_med   old_best    cyc_med   cyc_best  fused_med fused_best  (us)   fused GiB/s   tags" % (kb * 8))
for mib in (1, 2, 4, 8, 16, 32, 64, 100, 128, 256, 512, 1024, 2048, 4096):
    n = mib * MiB
    ts = {"old": [], "cyc": [], "fus": []}
    tags = {}
    for rep in range(4):
        for name, ctx in (("old", old), ("cyc", cyc), ("fus", fus)):
            for it in range(6):
                t0 = time.perf_counter()
                tags[name] = ctx.encrypt_dev(iv, a.ptr, n, b.ptr)
                ts[name].append(time.perf_counter() - t0)
    o, c, f = ts["old"], ts["cyc"], ts["fus"]
    print("       %6d  %9.1f  %9.1f  %9.1f  %9.1f  %9.1f  %9.1f          %8.1f   %s" % (mib, statistics.median(o) * 1e6, min(o) * 1e6, statistics.median(c) * 1e6, min(c) * 1e6,
          statistics.median(f) * 1e6, min(f) * 1e6, n / statistics.median(f) / (1 << 30), "same" if tags["old"] == tags["cyc"] == tags["fus"] else "DIFFERENT"), flush=True)
