#!/usr/bin/env python3
"""A call that holds messages of both kinds (GPU box):  python profiles/mixed_bench.py [--n N] [--max-len L] [--aad A] [--key-bits B] [--steps K] [--scattered] [--only WHAT]

N lengths drawn as the reference's harness draws them -- int(betavariate(0.1, 0.1) * L), tb/gcm_gctr.py:279-281: U-shaped, near 0 and near L in the same stream --
through aesgcm_packets_crypt_dev with offset arrays (or aesgcm_messages_crypt_dev: --scattered), timed with HIP events on the context's stream:

  mixed      the product library's call: every message routed by its own size on the device (round 6)
  all_rows   the same call with everything forced by rows      (debug library: what round 5 did with a hint of "large")
  all_pkt    ... with everything forced through the packet kernels (what round 5 did with a hint of "frames")
  short      the messages below the mark alone, as a call of their own (they all take the packet kernels)      } the two PURE paths: their times add up to the
  long       the messages at or above the mark alone (they all go by rows)                                      } byte-weighted combination the mixed call is held to

One JSON line.  `vs_combination` = (time of short + time of long) / time of mixed: 1.0 = the mixed call costs what its two halves cost apart."""
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
ap.add_argument("--n", type=int, default=1 << 18)
ap.add_argument("--max-len", type=int, default=65535)
ap.add_argument("--aad", type=int, default=0, help="bytes of AAD per message (0 = no AAD array)")
ap.add_argument("--key-bits", type=int, default=256)
ap.add_argument("--steps", type=int, default=7)
ap.add_argument("--mark", type=int, default=8192, help="where `short` and `long` are cut (the library's own mark for many messages)")
ap.add_argument("--scattered", action="store_true")
ap.add_argument("--dec", action="store_true")
ap.add_argument("--seed", type=int, default=2026)
ap.add_argument("--dist", default="u", choices=("u", "frames"), help="u: betavariate(.1, .1) x max-len; frames: uniform 64 .. 1514 (MACsec-shaped)")
ap.add_argument("--only", default="", help="comma list of mixed,all_rows,all_pkt,short,long (default: all): for profiling one of them")
a = ap.parse_args()
rng = random.Random(a.seed)
n, kb = a.n, a.key_bits // 8
lens = [int(rng.betavariate(0.1, 0.1) * a.max_len) for _ in range(n)] if a.dist == "u" else [rng.randrange(64, 1515) for _ in range(n)]
want = [w for w in a.only.split(",") if w] or ["mixed", "all_rows", "all_pkt", "short", "long"]


def layout(ls):
    off = [0]
    for x in ls:
        off.append(off[-1] + x)
    return off


def bench(ls, force=None):
    m = len(ls)
    if not m:
        return None
    off = layout(ls)
    total = off[-1]
    d_in, d_out = lib.DeviceBuffer(total + 64), lib.DeviceBuffer(total + 64)
    d_in.fill_splitmix64(0xAE5C0066, nbytes=(total + 64) // 8 * 8)
    d_ivs, d_tags = lib.DeviceBuffer(12 * m + 16), lib.DeviceBuffer(16 * m)
    d_ivs.fill_splitmix64(0x4956, nbytes=(12 * m + 16) // 8 * 8)
    d_off = lib.DeviceBuffer(8 * (m + 1)); d_off.upload(struct.pack("<%dQ" % (m + 1), *off))
    d_aad = d_aoff = None
    if a.aad:
        d_aad = lib.DeviceBuffer(a.aad * m + 16); d_aad.fill_splitmix64(0x414144, nbytes=(a.aad * m + 16) // 8 * 8)
        d_aoff = lib.DeviceBuffer(8 * (m + 1)); d_aoff.upload(struct.pack("<%dQ" % (m + 1), *[a.aad * i for i in range(m + 1)]))
    dbg = None
    if force:
        dbg = lib.debug_library(); dbg.__enter__(); dbg.force(**force)
    ctx = lib.Context(bytes(range(kb)))
    if a.scattered:
        d_ip, d_op, d_ln = lib.DeviceBuffer(8 * m), lib.DeviceBuffer(8 * m), lib.DeviceBuffer(4 * m)
        d_ip.upload(struct.pack("<%dQ" % m, *[d_in.ptr + x for x in off[:-1]])); d_op.upload(struct.pack("<%dQ" % m, *[d_out.ptr + x for x in off[:-1]]))
        d_ln.upload(struct.pack("<%dI" % m, *ls))
        d_ap = d_al = None
        if a.aad:
            d_ap, d_al = lib.DeviceBuffer(8 * m), lib.DeviceBuffer(4 * m)
            d_ap.upload(struct.pack("<%dQ" % m, *[d_aad.ptr + a.aad * i for i in range(m)])); d_al.upload(struct.pack("<%dI" % m, *([a.aad] * m)))

        def go():
            ctx.messages_crypt_dev(a.dec, m, d_ivs.ptr, d_ip.ptr, d_ln.ptr, d_op.ptr, d_tags.ptr, d_aad_ptr=d_ap.ptr if d_ap else None, d_aad_len=d_al.ptr if d_al else None)
    else:
        def go():
            ctx.packets_crypt_dev(a.dec, m, d_ivs.ptr, d_in.ptr, d_out.ptr, d_tags.ptr, d_data_off=d_off.ptr, d_aad=d_aad.ptr if d_aad else None, d_aad_off=d_aoff.ptr if d_aoff else None)
    t = lib.Timer()
    go(); go(); lib.dev_sync()
    assert ctx.status() == (lib.STATUS_OK, 0)
    ts = []
    for _ in range(a.steps):
        t.start(ctx.stream()); go(); t.stop(ctx.stream())
        ts.append(t.ms())
    if dbg:
        dbg.__exit__(None, None, None)
    for d in (d_in, d_out):
        d.free()
    return {"messages": m, "bytes": total, "ms_median": round(statistics.median(ts), 4), "ms_best": round(min(ts), 4), "gib_per_s": round(total / statistics.median(ts) / 1e-3 / 2**30, 1)}


short = [x for x in lens if x + a.aad < a.mark]
long_ = [x for x in lens if x + a.aad >= a.mark]
res = {"n": n, "max_len": a.max_len, "aad": a.aad, "key_bits": a.key_bits, "scattered": a.scattered, "decrypt": a.dec, "mark": a.mark,
       "short_share_of_messages": round(len(short) / n, 3), "short_share_of_bytes": round(sum(short) / max(sum(lens), 1), 4), "device": lib.device_name(0)}
if "mixed" in want:
    res["mixed"] = bench(lens)
if "all_rows" in want:
    res["all_rows"] = bench(lens, dict(pkt_rows=1))
if "all_pkt" in want:
    res["all_pkt"] = bench(lens, dict(pkt_rows=2))
if "short" in want:
    res["short"] = bench(short)
if "long" in want:
    res["long"] = bench(long_)
if all(res.get(k) for k in ("mixed", "short", "long")):
    res["combination_ms"] = round(res["short"]["ms_median"] + res["long"]["ms_median"], 4)
    res["vs_combination"] = round(res["combination_ms"] / res["mixed"]["ms_median"], 4)
    res["combination_gib_per_s"] = round(res["mixed"]["bytes"] / res["combination_ms"] / 1e-3 / 2**30, 1)
print(json.dumps(res))
