#!/usr/bin/env python3
"""Does rotating the waves' issue priorities keep the equal shares of a cyclic launch together (GPU box)?  Cyclic rows forced at every size,
AESGCM_CYC_PRIO = rows between rotations (0 = off), against the dealt chunks; encrypt_dev incl. tag, median of 12 (us)."""
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
ctxs = []
os.environ["AESGCM_BODY_CYC"] = "0:0"
ctxs.append(("dealt", lib.Context(bytes(range(kb)))))
os.environ["AESGCM_BODY_CYC"] = "%d:%d" % (1 * MiB, 1 << 50)
for k in ([int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else (0, 1, 2, 4, 8, 32)):
    os.environ["AESGCM_CYC_PRIO"] = str(k)
    ctxs.append(("prio%d" % k, lib.Context(bytes(range(kb)))))
os.environ.pop("AESGCM_BODY_CYC"); os.environ.pop("AESGCM_CYC_PRIO")
print("AES-%d   MiB  " % (kb * 8) + "  ".join("%8s" % n for n, _ in ctxs) + "   (us)")
for mib in ([int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (16, 64, 256, 512, 1024, 2048, 4096)):
    n = mib * MiB
    ts = {name: [] for name, _ in ctxs}
    tags = set()
    for rep in range(3):
        for name, ctx in ctxs:
            for it in range(4):
                t0 = time.perf_counter()
                tags.add(ctx.encrypt_dev(iv, a.ptr, n, b.ptr))
                ts[name].append(time.perf_counter() - t0)
    print("       %6d  " % mib + "  ".join("%8.1f" % (statistics.median(ts[name]) * 1e6) for name, _ in ctxs) + ("   tags same" if len(tags) == 1 else "   TAGS DIFFER"), flush=True)
