#!/usr/bin/env python3
"""The frames workload of bench.py --config frames as a bare loop for the profiler (GPU box):  python3 profiles/frames_one.py [--probe] [--n N] [--steps K] [--key-bits B] [--dec] [--aad A]
N MACsec-shaped frames (64 .. 1514 bytes, lengths stream 0x4C454E, A bytes of AAD each) under one key through the offset arrays of one aesgcm_packets_crypt_dev call,
K calls; --probe: aesgcm_frames_ceiling_probe_dev instead (the packet kernel without the data's loads and stores).  One JSON line: ms per call (HIP events), the route."""
import argparse
import json
import os
import statistics
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import aesgcm_amd  # noqa: E402,F401
from aesgcm_amd import lib, sharding  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--probe", action="store_true")
ap.add_argument("--dec", action="store_true")
ap.add_argument("--n", type=int, default=1 << 20)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--key-bits", type=int, default=256)
ap.add_argument("--aad", type=int, default=28)
ap.add_argument("--fixed", type=int, default=0, help="every frame this many bytes instead of 64 .. 1514")
ap.add_argument("--placed", type=int, default=0, help="every frame (and its AAD) in a buffer of its own at a multiple of this many bytes, through aesgcm_messages_crypt_dev (arrays of addresses and lengths) instead of offset arrays")
a = ap.parse_args()
n, al = a.n, a.aad
d_w = lib.DeviceBuffer(8 * n)
d_w.fill_splitmix64(0x4C454E, 0)
w = np.frombuffer(bytes(d_w.download()), dtype="<u8")
lens = (64 + (w % np.uint64(1451))).astype(np.int64) if not a.fixed else np.full(n, a.fixed, dtype=np.int64)
P = a.placed
pad = (lambda x: (x + P - 1) // P * P) if P else (lambda x: x)
doff = np.zeros(n + 1, dtype=np.uint64); doff[1:] = np.cumsum(pad(lens))
aoff = np.arange(n + 1, dtype=np.uint64) * np.uint64(pad(al) if al else 0)
total = int(doff[-1])
d_ivs, d_pt, d_ct, d_tags = lib.DeviceBuffer(12 * n + 16), lib.DeviceBuffer(total + 64), lib.DeviceBuffer(total + 64), lib.DeviceBuffer(16 * n)
d_ivs.fill_splitmix64(0x4956, nbytes=(12 * n + 16) // 8 * 8)
d_pt.fill_splitmix64(0xAE5C0006, nbytes=(total + 64) // 8 * 8)
d_aad = lib.DeviceBuffer(int(aoff[-1]) + 64); d_aad.fill_splitmix64(0x414144, nbytes=(int(aoff[-1]) + 64) // 8 * 8)
d_doff, d_aoff = lib.DeviceBuffer(8 * (n + 1)), lib.DeviceBuffer(8 * (n + 1))
d_doff.upload(doff.tobytes()); d_aoff.upload(aoff.tobytes())
akw = dict(d_aad=d_aad.ptr, d_aad_off=d_aoff.ptr) if al else {}
ctx = lib.Context(sharding.splitmix64_bytes(0x4B4559, a.key_bits // 8))


if P:
    assert not a.probe, "the probe takes offset arrays"
    d_in_ptr, d_out_ptr, d_aad_ptr, d_len, d_alen = tuple(lib.DeviceBuffer(8 * n) for _ in range(3)) + tuple(lib.DeviceBuffer(4 * n) for _ in range(2))
    d_in_ptr.upload((np.uint64(d_pt.ptr) + doff[:-1]).tobytes()); d_out_ptr.upload((np.uint64(d_ct.ptr) + doff[:-1]).tobytes())
    d_aad_ptr.upload((np.uint64(d_aad.ptr) + aoff[:-1]).tobytes())
    d_len.upload(lens.astype(np.uint32).tobytes()); d_alen.upload(np.full(n, al, dtype=np.uint32).tobytes())


def go():
    if P:
        return ctx.messages_crypt_dev(a.dec, n, d_ivs.ptr, d_in_ptr.ptr, d_len.ptr, d_out_ptr.ptr, d_tags.ptr, **(dict(d_aad_ptr=d_aad_ptr.ptr, d_aad_len=d_alen.ptr) if al else {}))
    if a.probe:
        return ctx.frames_ceiling_probe_dev(n, d_ivs.ptr, d_doff.ptr, d_tags.ptr, **akw)
    ctx.packets_crypt_dev(a.dec, n, d_ivs.ptr, d_pt.ptr, d_ct.ptr, d_tags.ptr, d_data_off=d_doff.ptr, **akw)


go(); go(); lib.dev_sync()
route = ctx.last_route()
t = lib.Timer()
ts = []
for _ in range(a.steps):
    t.start(ctx.stream()); go(); t.stop(ctx.stream())
    ts.append(t.ms())
payload = int(lens.sum())
print(json.dumps({"probe": a.probe, "n": n, "bytes": payload, "aad": al, "key_bits": a.key_bits, "decrypt": a.dec, "ms_median": round(statistics.median(ts), 4), "ms_best": round(min(ts), 4),
                  "placed": P, "gib_per_s": round(int(lens.sum()) / statistics.median(ts) / 1e-3 / 2**30, 1), "alg_bytes": int(2 * payload + n * (al + 28)), "route": route}))
