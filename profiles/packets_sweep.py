#!/usr/bin/env python3
"""Packets under one key: the kernel shapes (one wave / 16, 8, 4 lanes / one lane per packet) over packet count and size (GPU box).
Prints GiB/s; the library's own choice is the last column.  Shapes are forced through the debug build (libaesgcm_hip_dbg.so, include/aesgcm_debug.h)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib
kb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
var = len(sys.argv) > 2 and sys.argv[2] in ("var", "packed")       # frames of mixed length (MACsec-shaped: 64 .. 1514 bytes) through offset arrays: starts rounded up to 16 bytes,
packed = len(sys.argv) > 2 and sys.argv[2] == "packed"             # or back to back as they come (starts at any byte)
_dbg = lib.debug_library(); _dbg.__enter__()
ctx = lib.Context(bytes(range(kb)))
aad_len = int(os.environ.get("SWEEP_AAD", "0"))        # a header of this many bytes per frame as AAD (MACsec: 20 or 28), fixed-size records
if len(sys.argv) > 3:
    ctx.set_option("pkt_order", {"noorder": 0, "order": 1}[sys.argv[3]])      # never / always by length class (default: from 98304 packets)
nmax = 1 << 20
# counts: powers of two from 2^10, and 1.5 x 2^k where the shapes cross (2^15 .. 2^18)
counts = sorted({1 << k for k in range(10, 21, 2)} | {1 << k for k in range(15, 19)} | {3 << (k - 1) for k in range(15, 19)})
d_ivs = lib.DeviceBuffer(12 * nmax); d_ivs.fill_splitmix64(2, nbytes=12 * nmax // 8 * 8)
d_tags = lib.DeviceBuffer(16 * nmax)
print("AES-%d   n_pkts  pkt_B     wave  group16   group8   group4     lane     auto   (GiB/s)   lane: 768-lane form / ILP form" % (kb * 8))
if var:
    import random, struct
    rng = random.Random(5)
    print("AES-%d   n_pkts  mixed     wave  group16   group8   group4     lane     auto   (GiB/s; frames of 64 .. 1514 bytes, mean ~ 700%s)" % (kb * 8, "; " + sys.argv[3] if len(sys.argv) > 3 else "; by length class from 98304 packets"))
    lens = [rng.choice((64, 128, 256, 512, 1000, 1500, 1514, rng.randrange(64, 1515))) for _ in range(nmax)]
    off = [0]
    for x in lens: off.append(off[-1] + (x if packed else (x + 15) // 16 * 16))
    d_off = lib.DeviceBuffer(8 * (nmax + 1)); d_off.upload(struct.pack("<%dQ" % (nmax + 1), *off))
    d_pt, d_ct = lib.DeviceBuffer(off[-1] + 64), lib.DeviceBuffer(off[-1] + 64)
    d_pt.fill_splitmix64(3)
    d_aad = lib.DeviceBuffer(max(aad_len, 1) * nmax + 64); d_aad.fill_splitmix64(4)
    if aad_len:
        print("        (with %d bytes of AAD per frame)" % aad_len)
    for n in counts:
        row = []
        for lanes in (64, 16, 8, 4, 1, 0):
            _dbg.force(pkt_lanes=lanes)
            best = 1e9
            for it in range(4):
                lib.dev_sync(); t0 = time.perf_counter()
                ctx.packets_crypt_dev(False, n, d_ivs.ptr, d_pt.ptr, d_ct.ptr, d_tags.ptr, d_data_off=d_off.ptr, d_aad=d_aad.ptr if aad_len else None, aad_len=aad_len)
                lib.dev_sync(); best = min(best, time.perf_counter() - t0)
            row.append(off[n] / best / (1 << 30))
        print("        %8d  mixed %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f" % (n, *row), flush=True)
    sys.exit(0)
for pkt in (64, 256, 1024, 4096, 16384):
    nm = min(nmax, (1 << 32) // pkt)
    d_pt, d_ct = lib.DeviceBuffer(pkt * nm), lib.DeviceBuffer(pkt * nm)
    d_pt.fill_splitmix64(3)
    for n in counts:
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
        extra = []
        for ilp in (2, 1):
            _dbg.force(pkt_lanes=1, pkt_ilp=ilp)
            best = 1e9
            for it in range(4):
                lib.dev_sync(); t0 = time.perf_counter()
                ctx.packets_crypt_dev(False, n, d_ivs.ptr, d_pt.ptr, d_ct.ptr, d_tags.ptr, pkt_len=pkt)
                lib.dev_sync(); best = min(best, time.perf_counter() - t0)
            extra.append(n * pkt / best / (1 << 30))
        _dbg.force(pkt_ilp=0)
        print("        %8d %6d %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f   %8.1f %8.1f" % (n, pkt, *row, *extra), flush=True)
    del d_pt, d_ct
