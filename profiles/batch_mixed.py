#!/usr/bin/env python3
"""Packets with a key each and MIXED lengths (aesgcm_batch_crypt_var_dev; frames of 64 .. 1514 bytes at 16-byte aligned starts, mean ~ 700) over the packet
count: array order against the order by falling length class (debug build: batch_order 2 / 1), and the library's own rule (GPU box).  GiB/s."""
import os, random, struct, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib
kb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
nmax = 1 << 20
rng = random.Random(5)
lens = [rng.choice((64, 128, 256, 512, 1000, 1500, 1514, rng.randrange(64, 1515))) for _ in range(nmax)]
off = [0]
for x in lens: off.append(off[-1] + (x + 15) // 16 * 16)
d_off = lib.DeviceBuffer(8 * (nmax + 1)); d_off.upload(struct.pack("<%dQ" % (nmax + 1), *off))
d_keys, d_ivs, d_tags = lib.DeviceBuffer(kb * nmax), lib.DeviceBuffer(12 * nmax), lib.DeviceBuffer(16 * nmax)
d_keys.fill_splitmix64(1); d_ivs.fill_splitmix64(2, nbytes=12 * nmax // 8 * 8)
d_pt, d_ct = lib.DeviceBuffer(off[-1] + 64), lib.DeviceBuffer(off[-1] + 64)
d_pt.fill_splitmix64(3)
print("AES-%d, key per packet, mixed frames   n_pkts   array order   by length class   library   lanes per packet (library)" % (kb * 8))
with lib.debug_library() as dbg:
    for n in sorted({1 << k for k in range(12, 21, 2)} | {1 << k for k in range(15, 19)} | {3 << (k - 1) for k in range(15, 19)}):
        row = []
        for order in (2, 1, 0):
            dbg.force(batch_order=order)
            best = 1e9
            for it in range(4):
                lib.dev_sync(); t0 = time.perf_counter()
                lib.batch_crypt_var_dev(False, n, kb, d_keys.ptr, d_ivs.ptr, d_pt.ptr, d_off.ptr, d_ct.ptr, d_tags.ptr)
                lib.dev_sync(); best = min(best, time.perf_counter() - t0)
            row.append(off[n] / best / (1 << 30))
        print("%46d %13.1f %17.1f %9.1f %8d" % (n, *row, lib.batch_shape(n, var_len=True)), flush=True)
