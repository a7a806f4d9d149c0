#!/usr/bin/env python3
"""Per-packet-key batch (aesgcm_batch_crypt_dev): the three kernel shapes over packet count and size (GPU box) -- one wave per
packet (k_batch, AESGCM_BATCH_LG=6), 16 lanes per packet in two phases (k_batch2, LG=4 FUSED=0) and in one pass (k_batch3,
LG=4 FUSED=1), 8 lanes per packet in one pass (k_batch3, LG=3), and the library's own choice.  GiB/s, best of 4.   python profiles/batch_sweep.py [key bytes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib
kb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
nmax = 1 << 20
d_keys, d_ivs, d_tags = lib.DeviceBuffer(kb * nmax), lib.DeviceBuffer(12 * nmax), lib.DeviceBuffer(16 * nmax)
d_keys.fill_splitmix64(1); d_ivs.fill_splitmix64(2, nbytes=12 * nmax // 8 * 8)
print("AES-%d   n_pkts  pkt_B   k_batch  k_batch2  k_batch3 k_batch3/8    auto   (GiB/s)" % (kb * 8))
for pkt in (64, 256, 1024, 1500, 4096, 16384):
    nm = min(nmax, (1 << 32) // pkt)
    d_pt, d_ct = lib.DeviceBuffer(pkt * nm), lib.DeviceBuffer(pkt * nm)
    d_pt.fill_splitmix64(3)
    for ln in range(8, 21, 2):
        n = 1 << ln
        if n > nm: break
        row = []
        for env in ({"AESGCM_BATCH_LG": "6"}, {"AESGCM_BATCH_LG": "4", "AESGCM_BATCH_FUSED": "0"}, {"AESGCM_BATCH_LG": "4", "AESGCM_BATCH_FUSED": "1"}, {"AESGCM_BATCH_LG": "3", "AESGCM_BATCH_FUSED": "1"}, {}):
            for k in ("AESGCM_BATCH_LG", "AESGCM_BATCH_FUSED"):
                os.environ.pop(k, None)
            os.environ.update(env)
            best = 1e9
            for it in range(4):
                lib.dev_sync(); t0 = time.perf_counter()
                lib.batch_crypt_dev(False, n, kb, d_keys.ptr, d_ivs.ptr, d_pt.ptr, pkt, d_ct.ptr, d_tags.ptr)
                lib.dev_sync(); best = min(best, time.perf_counter() - t0)
            row.append(n * pkt / best / (1 << 30))
        print("        %8d %6d %9.1f %9.1f %9.1f %9.1f %8.1f" % (n, pkt, *row), flush=True)
    del d_pt, d_ct
