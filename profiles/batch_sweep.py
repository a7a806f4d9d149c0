#!/usr/bin/env python3
"""Per-packet-key batch (aesgcm_batch_crypt_dev): the kernel shapes over packet count and size (GPU box) -- one wave per packet (k_batch3<.., 6>), 16 and 8 lanes per
packet (k_batch3), and the library's own choice.  Shapes are forced through the debug build (libaesgcm_hip_dbg.so).  GiB/s, best of 4.
python profiles/batch_sweep.py [key bytes]      (round 3's version also had the two-phase k_batch2, deleted in round 4)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib
kb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
nmax = 1 << 20
d_keys, d_ivs, d_tags = lib.DeviceBuffer(kb * nmax), lib.DeviceBuffer(12 * nmax), lib.DeviceBuffer(16 * nmax)
d_keys.fill_splitmix64(1); d_ivs.fill_splitmix64(2, nbytes=12 * nmax // 8 * 8)
print("AES-%d   n_pkts  pkt_B   k_batch  k_batch3/16 k_batch3/8    auto   (GiB/s)" % (kb * 8))
_dbg = lib.debug_library(); _dbg.__enter__()
for pkt in (64, 256, 1024, 1500, 4096, 16384):
    nm = min(nmax, (1 << 32) // pkt)
    d_pt, d_ct = lib.DeviceBuffer(pkt * nm), lib.DeviceBuffer(pkt * nm)
    d_pt.fill_splitmix64(3)
    for ln in range(8, 21, 2):
        n = 1 << ln
        if n > nm: break
        row = []
        for lanes in (64, 16, 8, 0):
            _dbg.force(batch_lanes=lanes)
            best = 1e9
            for it in range(4):
                lib.dev_sync(); t0 = time.perf_counter()
                lib.batch_crypt_dev(False, n, kb, d_keys.ptr, d_ivs.ptr, d_pt.ptr, pkt, d_ct.ptr, d_tags.ptr)
                lib.dev_sync(); best = min(best, time.perf_counter() - t0)
            row.append(n * pkt / best / (1 << 30))
        print("        %8d %6d %9.1f %9.1f %9.1f %8.1f" % (n, pkt, *row), flush=True)
    del d_pt, d_ct
