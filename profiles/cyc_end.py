#!/usr/bin/env python3
"""The end of a fused cyclic launch (GPU box): the tag is published from inside the launch; what does it cost to have the ciphertext in memory by then?
  wb   (default) every workgroup writes its XCD's L2 back before it counts itself arrived
  wt   (AESGCM_CYC_FUSE=4) the rows store their ciphertext through the L2 (sc0 sc1), nothing to write back
  end  (AESGCM_CYC_FUSE=2) no write-back in the launch; the host waits for the end of the launch behind the tag
  3l   (AESGCM_CYC_FUSE=0) k_fold + k_combine behind the launch
encrypt_dev incl. tag readback, median of 60 calls (us)."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib
KiB = 1 << 10
a, b = lib.DeviceBuffer(256 * 1024 * KiB), lib.DeviceBuffer(256 * 1024 * KiB)
a.fill_splitmix64(1)
iv = bytes(12)
ctxs = []
for name, v in (("wb", "1"), ("wt", "4"), ("end", "2"), ("3l", "0")):
    os.environ["AESGCM_CYC_FUSE"] = v
    ctxs.append((name, lib.Context(bytes(range(32)))))
os.environ.pop("AESGCM_CYC_FUSE")
print("AES-256   KiB       wb       wt      end       3l   (us)")
for kib in (64, 256, 1024, 4096, 16384, 65536, 262144):
    row = []
    for name, ctx in ctxs:
        n = kib * KiB
        ts = []
        for it in range(60):
            t0 = time.perf_counter()
            ctx.encrypt_dev(iv, a.ptr, n, b.ptr)
            ts.append(time.perf_counter() - t0)
        row.append(statistics.median(ts[5:]) * 1e6)
    print("       %8d  %7.1f  %7.1f  %7.1f  %7.1f" % (kib, *row), flush=True)

# the dealt k_body with write-through stores: steady-state throughput
print("dealt k_body, 1 GiB AES-256: us per message, default stores / through the L2")
big_a, big_b = lib.DeviceBuffer(1 << 30), lib.DeviceBuffer(1 << 30)
big_a.fill_splitmix64(2)
res = {}
for name, v in (("default", "0"), ("wt", "1")):
    os.environ["AESGCM_BODY_WT"] = v
    res[name] = (lib.Context(bytes(range(32))), [])
os.environ.pop("AESGCM_BODY_WT")
for rep in range(5):
    for name in res:
        ctx, ts = res[name]
        for it in range(4):
            t0 = time.perf_counter(); ctx.encrypt_dev(iv, big_a.ptr, 1 << 30, big_b.ptr); ts.append(time.perf_counter() - t0)
print("   ".join("%s %.1f (best %.1f)" % (k, statistics.median(v[1]) * 1e6, min(v[1]) * 1e6) for k, v in res.items()))
