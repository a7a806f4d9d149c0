#!/usr/bin/env python3
"""The end of a fused cyclic launch (GPU box): the tag is published from inside the launch; what does it cost to have the ciphertext in memory by then?
  wt      (default) the rows store their ciphertext through the L2 (sc0 sc1): nothing to write back, nothing to wait for
  wt+wb   (AESGCM_CYC_FUSE=1) ... and every workgroup writes its XCD's L2 back before it counts itself arrived (what plain stores would need)
  wt+end  (AESGCM_CYC_FUSE=2) ... and the host waits for the end of the launch behind the tag (the other thing plain stores would allow)
(profiles/r03c/cyc_end.txt was taken when the store kind was a run-time switch: its wb and end columns are with plain stores.)
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
for name, v in (("wt", "4"), ("wb", "1"), ("end", "2"), ("3l", "0")):
    os.environ["AESGCM_CYC_FUSE"] = v
    ctxs.append((name, lib.Context(bytes(range(32)))))
os.environ.pop("AESGCM_CYC_FUSE")
print("AES-256   KiB       wt    wt+wb   wt+end       3l   (us)")
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

