#!/usr/bin/env python3
"""The one-wave-per-packet shape of the per-packet-key batch (k_batch3<.., 6>; until round 4: k_batch) in the regime where the library picks it: 1024 packets of 16 KiB, AES-128 (GPU box; for rocprofv3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib
n, pkt, kb = 1024, 16384, 16
d_keys, d_ivs, d_tags = lib.DeviceBuffer(kb * n), lib.DeviceBuffer(12 * n), lib.DeviceBuffer(16 * n)
d_keys.fill_splitmix64(1); d_ivs.fill_splitmix64(2, nbytes=12 * n // 8 * 8)
d_pt, d_ct = lib.DeviceBuffer(pkt * n), lib.DeviceBuffer(pkt * n)
d_pt.fill_splitmix64(3)
best = 1e9
for it in range(8):
    lib.dev_sync(); t0 = time.perf_counter()
    lib.batch_crypt_dev(False, n, kb, d_keys.ptr, d_ivs.ptr, d_pt.ptr, pkt, d_ct.ptr, d_tags.ptr)
    lib.dev_sync(); best = min(best, time.perf_counter() - t0)
print("wave per packet, 1024 x 16 KiB AES-128: %.1f GiB/s, %.1f us" % (n * pkt / best / 2**30, best * 1e6))
