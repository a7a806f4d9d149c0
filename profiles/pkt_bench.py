#!/usr/bin/env python3
"""Packet-path workloads for profiling (GPU box):  python profiles/pkt_bench.py KIND [--n N] [--len L] [--key-bits B] [--steps K]
  KIND = batch  BASELINE config 5: N packets of L bytes, per-packet key and IV (k_batch3), SplitMix64 inputs of SURVEY 8(d)
         pktw   N packets under ONE key, one wave per packet (k_pktg<.., 6>)
         pktg   N packets under ONE key, 16 lanes per packet (k_pktg<.., 4>)
         pktl   N packets under ONE key, one lane per packet (k_pktl)
         pkt    N packets under ONE key, the library's own choice (product library): by rows (k_rows) from 8 KiB per packet, from 2 KiB when few
         rows / norows   by rows always / never (debug library)
Prints one JSON line: ms per launch (median and best of K, HIP-synchronised wall time; ms_queued: K calls enqueued back to back and waited for once, per
call -- what a caller that keeps the stream busy sees), packets/s, GiB/s and the algorithmic HBM bytes per launch (32 B per block + key/IV/tag traffic)."""
import argparse
import json
import os
import statistics
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa: E402,F401
from aesgcm_amd import lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("kind", choices=("batch", "pkt", "pktw", "pktg", "pktg8", "pktg4", "pktl", "rows", "norows"))
ap.add_argument("--scatter", action="store_true", help="one key: the same packets as messages wherever they live (aesgcm_messages_crypt_dev: arrays of addresses and lengths; always by rows)")
ap.add_argument("--var", action="store_true", help="one key: the same packets through offset arrays (pkt_len then is the caller's hint)")
ap.add_argument("--aad", type=int, default=0, help="one key: bytes of AAD per packet")
ap.add_argument("--opt", action="append", default=[], help="context option key=value (aesgcm_ctx_set_option), repeatable")
ap.add_argument("--rows-block", type=int, default=0, help="context option rows_block (units per dealt block of k_rows; 0 = the library's cut)")
ap.add_argument("--n", type=int, default=1 << 20)
ap.add_argument("--len", type=int, default=4096)
ap.add_argument("--key-bits", type=int, default=128)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--dec", action="store_true", help="decrypt + tag (the launch of the decrypt instance over the same bytes; tags are computed, not compared)")
ap.add_argument("--inplace", action="store_true", help="write the output over the input (traffic probe: no separate output lines)")
a = ap.parse_args()
n, pkt, kb = a.n, a.len, a.key_bits // 8
d_ivs = lib.DeviceBuffer(16 * n)
d_ivs.fill_splitmix64(0x4956)
d_pt, d_ct, d_tags = lib.DeviceBuffer(pkt * n), lib.DeviceBuffer(pkt * n), lib.DeviceBuffer(16 * n)
d_pt.fill_splitmix64(0xAE5C0005)
if a.inplace:
    d_ct = d_pt
if a.kind == "batch":
    d_keys = lib.DeviceBuffer(kb * n)
    d_keys.fill_splitmix64(0x4B4559)

    def go():
        lib.batch_crypt_dev(a.dec, n, kb, d_keys.ptr, d_ivs.ptr, d_pt.ptr, pkt, d_ct.ptr, d_tags.ptr)
else:
    # a forced shape is a function of the debug build only (libaesgcm_hip_dbg.so, include/aesgcm_debug.h); "pkt" = the library's own choice, product build
    _dbg = None
    if a.kind != "pkt":
        _dbg = lib.debug_library()
        _dbg.__enter__()
        if a.kind in ("rows", "norows"):
            _dbg.force(pkt_rows=1 if a.kind == "rows" else 2)
        else:
            _dbg.force(pkt_lanes={"pktw": 64, "pktg": 16, "pktg8": 8, "pktg4": 4, "pktl": 1}[a.kind], pkt_rows=2)
    ctx = lib.Context(bytes(range(kb)))
    if a.rows_block:
        ctx.set_option("rows_block", a.rows_block)
    for kv in a.opt:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    d_off = d_aoff = d_aad = None
    if a.var:
        import struct
        d_off = lib.DeviceBuffer(8 * (n + 1)); d_off.upload(struct.pack("<%dQ" % (n + 1), *[pkt * i for i in range(n + 1)]))
        if a.aad:
            d_aoff = lib.DeviceBuffer(8 * (n + 1)); d_aoff.upload(struct.pack("<%dQ" % (n + 1), *[a.aad * i for i in range(n + 1)]))
    if a.aad:
        d_aad = lib.DeviceBuffer(a.aad * n); d_aad.fill_splitmix64(0x414144)

    if a.scatter:
        import struct
        d_ip, d_op, d_ln = lib.DeviceBuffer(8 * n), lib.DeviceBuffer(8 * n), lib.DeviceBuffer(4 * n)
        d_ip.upload(struct.pack("<%dQ" % n, *[d_pt.ptr + pkt * i for i in range(n)])); d_op.upload(struct.pack("<%dQ" % n, *[d_ct.ptr + pkt * i for i in range(n)]))
        d_ln.upload(struct.pack("<%dI" % n, *([pkt] * n)))
        d_ap = d_al = None
        if a.aad:
            d_ap, d_al = lib.DeviceBuffer(8 * n), lib.DeviceBuffer(4 * n)
            d_ap.upload(struct.pack("<%dQ" % n, *[d_aad.ptr + a.aad * i for i in range(n)])); d_al.upload(struct.pack("<%dI" % n, *([a.aad] * n)))

    def go():
        if a.scatter:
            return ctx.messages_crypt_dev(a.dec, n, d_ivs.ptr, d_ip.ptr, d_ln.ptr, d_op.ptr, d_tags.ptr, d_aad_ptr=d_ap.ptr if d_ap else None, d_aad_len=d_al.ptr if d_al else None)
        ctx.packets_crypt_dev(a.dec, n, d_ivs.ptr, d_pt.ptr, d_ct.ptr, d_tags.ptr, pkt_len=pkt, d_data_off=d_off.ptr if d_off else None,
                              d_aad=d_aad.ptr if d_aad else None, aad_len=0 if d_aoff else a.aad, d_aad_off=d_aoff.ptr if d_aoff else None)
go(); lib.dev_sync()
ts = []
for _ in range(a.steps):
    lib.dev_sync()
    t0 = time.perf_counter()
    go()
    lib.dev_sync()
    ts.append(time.perf_counter() - t0)
med, best = statistics.median(ts), min(ts)
lib.dev_sync()
t0 = time.perf_counter()
for _ in range(a.steps):
    go()
lib.dev_sync()
queued = (time.perf_counter() - t0) / a.steps
alg = n * (2 * pkt + 16 + 12 + (kb if a.kind == "batch" else 0))
print(json.dumps({"kind": a.kind, "n_pkts": n, "pkt_len": pkt, "key_bits": a.key_bits, "ms_median": round(med * 1e3, 4), "ms_best": round(best * 1e3, 4),
                  "ms_queued": round(queued * 1e3, 4), "gib_per_s_queued": round(n * pkt / queued / 2**30, 1),
                  "shape": (ctx.packets_shape(n, pkt, a.var) if a.kind != "batch" else None),
                  "mpkt_per_s": round(n / med / 1e6, 2), "gib_per_s": round(n * pkt / med / 2**30, 1), "alg_bytes_per_launch": alg,
                  "alg_gb_per_s": round(alg / med / 1e9, 1), "frac_of_hbm_peak": round(alg / med / 8e12, 4)}))
