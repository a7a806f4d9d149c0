#!/usr/bin/env python3
"""A call of MIXED message sizes by rows, for profiling (GPU box): 3000 messages of 64 KiB .. 16 MiB (log-uniform, ragged ends) with 20 bytes of AAD each
through offset arrays -- k_rows_plan, k_rows, k_rows_close.  Prints GiB/s; profiles/collect.sh rows_mixed 'k_rows<' profiles/rows_mixed.py"""
import os, random, struct, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib
rng = random.Random(5)
n = 3000
lens = [int(65536 * 256 ** rng.random()) + rng.randrange(0, 1024) for _ in range(n)]
doff, aoff = [0], [0]
for a in lens:
    doff.append(doff[-1] + a); aoff.append(aoff[-1] + 20)
total = doff[-1]
d_in, d_out = lib.DeviceBuffer(total + 16), lib.DeviceBuffer(total + 16)
d_in.fill_splitmix64(7, nbytes=total // 8 * 8)
d_ivs, d_aad, d_tags = lib.DeviceBuffer(12 * n), lib.DeviceBuffer(20 * n), lib.DeviceBuffer(16 * n)
d_ivs.fill_splitmix64(8); d_aad.fill_splitmix64(9)
d_doff, d_aoff = lib.DeviceBuffer(8 * (n + 1)), lib.DeviceBuffer(8 * (n + 1))
d_doff.upload(struct.pack("<%dQ" % (n + 1), *doff)); d_aoff.upload(struct.pack("<%dQ" % (n + 1), *aoff))
ctx = lib.Context(bytes(range(32)))
def go():
    ctx.packets_crypt_dev(False, n, d_ivs.ptr, d_in.ptr, d_out.ptr, d_tags.ptr, pkt_len=1 << 20, d_data_off=d_doff.ptr, d_aad=d_aad.ptr, d_aad_off=d_aoff.ptr)
for _ in range(3):
    go()
lib.dev_sync()
t0 = time.perf_counter()
K = 12
for _ in range(K):
    go()
lib.dev_sync()
dt = (time.perf_counter() - t0) / K
print("%d messages, %.2f GiB, %.3f ms per call, %.1f GiB/s" % (n, total / 2**30, dt * 1e3, total / dt / 2**30))
