#!/usr/bin/env python3
"""2^20 x 4 KiB packets under ONE key (per-packet IV), AES-128 and AES-256 (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib
n, pkt = 1 << 20, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d_ivs = lib.DeviceBuffer(12 * n); d_ivs.fill_splitmix64(2, nbytes=12 * n // 8 * 8)
d_pt, d_ct, d_tags = lib.DeviceBuffer(pkt * n), lib.DeviceBuffer(pkt * n), lib.DeviceBuffer(16 * n)
d_pt.fill_splitmix64(3)
for kb in (16, 32):
    ctx = lib.Context(bytes(range(kb)))
    for it in range(3):
        lib.dev_sync()
        t0 = time.perf_counter()
        ctx.packets_crypt_dev(False, n, d_ivs.ptr, d_pt.ptr, d_ct.ptr, d_tags.ptr, pkt_len=pkt)
        lib.dev_sync()
        dt = time.perf_counter() - t0
    print("one key, pkt %d B, AES-%d: %.3f ms  %.2f Mpkt/s  %.1f GiB/s" % (pkt, kb * 8, dt * 1e3, n / dt / 1e6, n * pkt / dt / (1 << 30)))
