#!/usr/bin/env python3
"""Where should a routed call send its SHORT messages?  (GPU box)  python profiles/route_sweep.py [--key-bits B]
Populations of short messages only -- U-shaped below 8 KiB (what is left of the reference's distribution below the mark), U-shaped below 1500, MACsec-shaped frames
(uniform 64 .. 1514), 1 KiB +- 25 % -- at counts from 2^12 to 2^20, each through one offset-array call forced by rows (the closing launch walks them block by block)
and forced through the packet kernels (the routed shape for the count, by falling size class).  HIP-event time per call on the context's stream, median of 5.
One line per population: n, bytes, ms by rows, ms by the packet kernels, ms by the product library's own rule (k_len_scan) and how that compares with the better of the two."""
import argparse
import json
import os
import random
import statistics
import struct
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa: E402,F401
from aesgcm_amd import lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--key-bits", type=int, default=256)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--counts", default="4096,16384,65536,131072,262144,524288,1048576")
ap.add_argument("--kinds", default="u8k,u1500,frames,1k,tiny")
ap.add_argument("--only", default="", help="rows | pkt | lib: that way alone (for the profiler)")
a = ap.parse_args()
kb = a.key_bits // 8


def draw(kind, n, rng):
    if kind == "u8k":
        return [int(rng.betavariate(0.1, 0.1) * 8191) for _ in range(n)]
    if kind == "u1500":
        return [int(rng.betavariate(0.1, 0.1) * 1500) for _ in range(n)]
    if kind == "frames":
        return [rng.randrange(64, 1515) for _ in range(n)]
    if kind == "1k":
        return [rng.randrange(768, 1281) for _ in range(n)]
    if kind == "band8_16":                                       # the band above the high mark, ragged (what decides whether the mark should rise with the count)
        return [rng.randrange(8192, 16384) for _ in range(n)]
    if kind == "band8_16w":                                      # ... in whole rows of 1 KiB
        return [1024 * rng.randrange(8, 16) for _ in range(n)]
    if kind == "band16_32":
        return [rng.randrange(16384, 32768) for _ in range(n)]
    if kind == "u16k":
        return [int(rng.betavariate(0.1, 0.1) * 16383) for _ in range(n)]
    if kind == "tiny":
        return [rng.randrange(0, 129) for _ in range(n)]
    raise ValueError(kind)


def run(ls, force):
    """force = None: the product library's own rule"""
    m = len(ls)
    off = [0]
    for x in ls:
        off.append(off[-1] + x)
    total = off[-1]
    d_in, d_out = lib.DeviceBuffer(total + 64), lib.DeviceBuffer(total + 64)
    d_in.fill_splitmix64(0xAE5C0067, nbytes=(total + 64) // 8 * 8)
    d_ivs, d_tags = lib.DeviceBuffer(12 * m + 16), lib.DeviceBuffer(16 * m)
    d_ivs.fill_splitmix64(0x4956, nbytes=(12 * m + 16) // 8 * 8)
    d_off = lib.DeviceBuffer(8 * (m + 1)); d_off.upload(struct.pack("<%dQ" % (m + 1), *off))
    import contextlib
    with (lib.debug_library() if force else contextlib.nullcontext()) as dbg:
        if force:
            dbg.force(**force)
        ctx = lib.Context(bytes(range(kb)))
        t = lib.Timer()
        for _ in range(2):
            ctx.packets_crypt_dev(False, m, d_ivs.ptr, d_in.ptr, d_out.ptr, d_tags.ptr, d_data_off=d_off.ptr)
        lib.dev_sync()
        ts = []
        for _ in range(a.steps):
            t.start(ctx.stream())
            ctx.packets_crypt_dev(False, m, d_ivs.ptr, d_in.ptr, d_out.ptr, d_tags.ptr, d_data_off=d_off.ptr)
            t.stop(ctx.stream())
            ts.append(t.ms())
    for d in (d_in, d_out):
        d.free()
    return statistics.median(ts), total


for kind in a.kinds.split(","):
    for n in [int(x) for x in a.counts.split(",")]:
        rng = random.Random(99 + n)
        ls = draw(kind, n, rng)
        if a.only:
            t_, total = run(ls, dict(pkt_rows=1) if a.only == "rows" else dict(pkt_rows=2) if a.only == "pkt" else None)
            print(json.dumps({"kind": kind, "n": n, "only": a.only, "ms": round(t_, 4), "gib_s": round(total / t_ / 1e-3 / 2**30, 1)}), flush=True)
            continue
        r, total = run(ls, dict(pkt_rows=1))
        p, _ = run(ls, dict(pkt_rows=2))
        own, _ = run(ls, None)
        blocks = sum((x + 15) // 16 for x in ls)
        print(json.dumps({"kind": kind, "n": n, "bytes": total, "blocks": blocks, "ms_rows": round(r, 4), "ms_pkt": round(p, 4), "ms_lib": round(own, 4), "lib_vs_best": round(min(r, p) / own, 3),
                          "gib_s_rows": round(total / r / 1e-3 / 2**30, 1), "gib_s_pkt": round(total / p / 1e-3 / 2**30, 1), "gib_s_lib": round(total / own / 1e-3 / 2**30, 1)}), flush=True)
