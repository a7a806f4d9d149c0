#!/usr/bin/env python3
"""Which lane shape should a routed call give n frames?  (GPU box)  python3 profiles/pkt_shape_sweep.py [--counts ...]
n MACsec-shaped frames (64 .. 1514 bytes, 28 B of AAD) through offset arrays with every packet-kernel shape forced (debug library: 1 / 4 / 8 / 16 / 64 lanes per frame) and
by the library's own rule; HIP-event ms per call, median of 20 back-to-back calls."""
import argparse, contextlib, json, os, random, statistics, struct, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa: E402,F401
from aesgcm_amd import lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--counts", default="512,1024,2048,4096,8192,16384,32768,65536,131072,262144")
ap.add_argument("--lo", type=int, default=64); ap.add_argument("--hi", type=int, default=1514); ap.add_argument("--aad", type=int, default=28)
a = ap.parse_args()
for n in [int(x) for x in a.counts.split(",")]:
    rng = random.Random(5 + n)
    ls = [rng.randrange(a.lo, a.hi + 1) for _ in range(n)]
    off = [0]
    for x in ls: off.append(off[-1] + x)
    total = off[-1]
    d_in, d_out = lib.DeviceBuffer(total + 64), lib.DeviceBuffer(total + 64)
    d_in.fill_splitmix64(0xAE5C0068, nbytes=(total + 64) // 8 * 8)
    d_ivs, d_tags, d_aad = lib.DeviceBuffer(12 * n + 16), lib.DeviceBuffer(16 * n), lib.DeviceBuffer(a.aad * n + 64)
    d_ivs.fill_splitmix64(0x4956, nbytes=(12 * n + 16) // 8 * 8)
    d_off = lib.DeviceBuffer(8 * (n + 1)); d_off.upload(struct.pack("<%dQ" % (n + 1), *off))
    d_aoff = lib.DeviceBuffer(8 * (n + 1)); d_aoff.upload(struct.pack("<%dQ" % (n + 1), *[a.aad * i for i in range(n + 1)]))
    row = {"n": n, "bytes": total}
    for lanes in (0, 1, 4, 8, 16, 64):
        with (lib.debug_library() if lanes else contextlib.nullcontext()) as dbg:
            if lanes: dbg.force(pkt_lanes=lanes)
            ctx = lib.Context(bytes(range(32)))
            t = lib.Timer()
            for _ in range(3):
                ctx.packets_crypt_dev(False, n, d_ivs.ptr, d_in.ptr, d_out.ptr, d_tags.ptr, d_data_off=d_off.ptr, d_aad=d_aad.ptr, d_aad_off=d_aoff.ptr)
            lib.dev_sync()
            ts = []
            for _ in range(20):
                t.start(ctx.stream()); ctx.packets_crypt_dev(False, n, d_ivs.ptr, d_in.ptr, d_out.ptr, d_tags.ptr, d_data_off=d_off.ptr, d_aad=d_aad.ptr, d_aad_off=d_aoff.ptr); t.stop(ctx.stream())
                ts.append(t.ms())
            row["lib" if not lanes else "l%d" % lanes] = round(statistics.median(ts) * 1e3, 1)
            if not lanes: row["lib_lanes"] = ctx.last_route()["lanes"]
            ctx.close()
    best = min((row[k], k) for k in ("l1", "l4", "l8", "l16", "l64"))
    row["best"] = best[1]; row["lib_vs_best"] = round(best[0] / row["lib"], 3)
    print(json.dumps(row), flush=True)
    for d in (d_in, d_out): d.free()
