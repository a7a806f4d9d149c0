#!/usr/bin/env python3
"""Packets under one key: the kernel shapes (one wave / 16, 8, 4 lanes / one lane per packet) over packet count and size (GPU box).
Prints GiB/s; the library's own choice is the last column.  Shapes are forced through the debug build (libaesgcm_hip_dbg.so, include/aesgcm_debug.h)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib
kb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
_dbg = lib.debug_library(); _dbg.__enter__()
ctx = lib.Context(bytes(range(kb)))
nmax = 1 << 20
d_ivs = lib.DeviceBuffer(12 * nmax); d_ivs.fill_splitmix64(2, nbytes=12 * nmax // 8 * 8)
d_tags = lib.DeviceBuffer(16 * nmax)
print("AES-%d   n_pkts  pkt_B     wave  group16   group8   group4     lane     auto   (GiB/s)" % (kb * 8))
for pkt in (64, 256, 1024, 4096, 16384):
    nm = min(nmax, (1 << 32) // pkt)
    d_pt, d_ct = lib.DeviceBuffer(pkt * nm), lib.DeviceBuffer(pkt * nm)
    d_pt.fill_splitmix64(3)
    for ln in range(10, 21, 2):
        n = 1 << ln
        if n > nm: break
        row = []
        for lanes in (64, 16, 8, 4, 1, 0):
            _dbg.force(pkt_lanes=lanes)
            best = 1e9
            for it in range(4):
                lib.dev_sync(); t0 = time.perf_counter()
                ctx.packets_crypt_dev(False, n, d_ivs.ptr, d_pt.ptr, d_ct.ptr, d_tags.ptr, pkt_len=pkt)
                lib.dev_sync(); best = min(best, time.perf_counter() - t0)
            row.append(n * pkt / best / (1 << 30))
        print("        %8d %6d %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f" % (n, pkt, *row), flush=True)
    del d_pt, d_ct
