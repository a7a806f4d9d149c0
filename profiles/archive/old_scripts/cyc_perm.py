#!/usr/bin/env python3
"""Workgroup -> item group permutation of the cyclic launch (AESGCM_CYC_PERM) against the identity and against the dealt chunks (GPU box).
encrypt_dev incl. tag on a warm chip, median of 12 (us)."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib
kb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
MiB = 1 << 20
nmax = (int(sys.argv[2]) if len(sys.argv) > 2 else 4096) * MiB
a, b = lib.DeviceBuffer(nmax), lib.DeviceBuffer(nmax)
a.fill_splitmix64(1)
iv = bytes(12)
ctxs = []
os.environ["AESGCM_BODY_CYC"] = "0:0"
ctxs.append(("dealt", lib.Context(bytes(range(kb)))))
os.environ["AESGCM_BODY_CYC"] = "%d:%d" % (1 * MiB, 1 << 50)
for k in (0, 1):
    os.environ["AESGCM_CYC_PERM"] = str(k)
    ctxs.append(("perm%d" % k, lib.Context(bytes(range(kb)))))
os.environ.pop("AESGCM_BODY_CYC"); os.environ.pop("AESGCM_CYC_PERM")
print("AES-%d   MiB  " % (kb * 8) + "  ".join("%8s" % n for n, _ in ctxs) + "   (us)")
for mib in (64, 256, 512, 1024, 2048, 4096, 8192, 16384):
    n = mib * MiB
    if n > nmax: break
    ts = {name: [] for name, _ in ctxs}
    tags = set()
    for rep in range(3):
        for name, ctx in ctxs:
            for it in range(4):
                t0 = time.perf_counter()
                tags.add(ctx.encrypt_dev(iv, a.ptr, n, b.ptr))
                ts[name].append(time.perf_counter() - t0)
    print("       %6d  " % mib + "  ".join("%8.1f" % (statistics.median(ts[name]) * 1e6) for name, _ in ctxs) + ("   tags same" if len(tags) == 1 else "   TAGS DIFFER"), flush=True)
