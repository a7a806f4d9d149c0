#!/usr/bin/env python3
"""cfg5 timing on the GPU box: 2^20 x 4 KiB packets, per-packet key/IV (AES-128), wall time around the launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib
n, pkt = 1 << 20, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
klen = int(sys.argv[2]) if len(sys.argv) > 2 else 16
d_keys, d_ivs = lib.DeviceBuffer(klen * n), lib.DeviceBuffer(12 * n)
d_keys.fill_splitmix64(1); d_ivs.fill_splitmix64(2, nbytes=12 * n // 8 * 8)
d_pt, d_ct, d_tags = lib.DeviceBuffer(pkt * n), lib.DeviceBuffer(pkt * n), lib.DeviceBuffer(16 * n)
d_pt.fill_splitmix64(3)
for it in range(4):
    lib.dev_sync()
    t0 = time.perf_counter()
    lib.batch_crypt_dev(False, n, klen, d_keys.ptr, d_ivs.ptr, d_pt.ptr, pkt, d_ct.ptr, d_tags.ptr)
    lib.dev_sync()
    dt = time.perf_counter() - t0
    print("pkt %d B, AES-%d: %.3f ms  %.2f Mpkt/s  %.1f GiB/s" % (pkt, klen * 8, dt * 1e3, n / dt / 1e6, n * pkt / dt / (1 << 30)))
