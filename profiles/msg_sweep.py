#!/usr/bin/env python3
"""Messages of 8 KiB .. 16 MiB under ONE key, every way the library offers to run them, from one box in one process (GPU box):

    python profiles/msg_sweep.py [--total-gib 4] [--key-bits 256] > profiles/r05/size_sweep.txt

  waited     one context, aesgcm_encrypt_dev with the tag waited for, message after message (the reference model's call order, tb/gcm_model.py:21-35)
  K = 3      three contexts with a stream each, calls enqueued with tag = NULL, the tag collected when the context comes round again (bench.py --inflight 3)
  packets    ALL the messages as the packets of one aesgcm_packets_crypt_dev call: by rows (k_rows, round 5: the library's rule from 8 KiB per packet) ...
  pkt kernels   ... and with rows switched off (context option rows_min = 0): the wave-per-packet / lane-group kernels of round 4
The buffers of one measurement total --total-gib (beyond the 256 MiB Infinity Cache), every form runs for at least 0.25 s after a warm-up of its own (the chip
takes some milliseconds of load to reach its clock), the packet calls are queued back to back and waited for once.  GiB/s of plaintext; `shape` is what
aesgcm_packets_shape says the packets call takes."""
import argparse
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa: E402,F401
from aesgcm_amd import lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--total-gib", type=float, default=4.0)
ap.add_argument("--key-bits", type=int, default=256)
ap.add_argument("--sizes-kib", type=int, nargs="*", default=[8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384])
ap.add_argument("--min-s", type=float, default=0.25)
a = ap.parse_args()
GiB = 1 << 30
total = int(a.total_gib * GiB)
key = bytes(range(a.key_bits // 8))
d_in, d_out = lib.DeviceBuffer(total), lib.DeviceBuffer(total)
d_in.fill_splitmix64(0xAE5C0055)
nmax = total // (min(a.sizes_kib) << 10)
d_ivs, d_tags = lib.DeviceBuffer(12 * nmax + 16), lib.DeviceBuffer(16 * nmax)
d_ivs.fill_splitmix64(0x4956, nbytes=(12 * nmax) // 8 * 8)
lib.dev_sync()


def timed(step, units_per_step, sync, queued=1):
    """units per second over at least min_s seconds.  queued > 1: asynchronous steps, enqueued `queued` at a time and waited for (never more: an unbounded loop
    enqueues thousands of 5 ms calls in the time it measures -- the first run of this script spent 25 GPU-minutes draining such a queue)"""
    def burst():
        for _ in range(queued):
            step()
        if queued > 1:
            sync()
    burst()
    sync()
    t_end = time.perf_counter() + 0.05
    while time.perf_counter() < t_end:                           # warm-up under load
        burst()
    sync()
    n = 0
    t0 = time.perf_counter()
    while True:
        burst()
        n += queued
        if time.perf_counter() - t0 > a.min_s:
            break
    sync()
    return units_per_step * n / (time.perf_counter() - t0)


print("# AES-%d-GCM, %.3g GiB of buffers per measurement, >= %.2f s each; GiB/s of plaintext" % (a.key_bits, a.total_gib, a.min_s))
print("%-10s %8s %10s %10s %12s %14s  %s" % ("message", "count", "waited", "K = 3", "packets", "pkt kernels", "shape of the packets call"))
for kib in a.sizes_kib:
    size = kib << 10
    n = total // size
    iv = bytes(12)
    # waited
    one = lib.Context(key)
    state = {"i": 0}

    def waited():
        i = state["i"] % n
        one.encrypt_dev(iv, d_in.ptr + i * size, size, d_out.ptr + i * size)
        state["i"] += 1
    r_wait = timed(waited, size, lib.dev_sync) / GiB
    one.close()
    # three in flight
    ctxs = [lib.Context(key) for _ in range(3)]
    busy = [False] * 3

    def flight():
        k = state["i"] % 3
        i = state["i"] % n
        if busy[k]:
            ctxs[k].last_tag()
        ctxs[k].encrypt_dev(iv, d_in.ptr + i * size, size, d_out.ptr + i * size, want_tag=False)
        busy[k] = True
        state["i"] += 1
    r_k3 = timed(flight, size, lib.dev_sync) / GiB
    for c in ctxs:
        c.close()
    # all of them as packets of one call: by rows, and by the packet kernels
    res = []
    for rows_min in (8192, 0):                                       # the library's own rule / never
        ctx = lib.Context(key)
        if rows_min == 0:
            ctx.set_option("rows_min", 0)
        shape = ctx.packets_shape(n, size)

        def pk():
            ctx.packets_crypt_dev(False, n, d_ivs.ptr, d_in.ptr, d_out.ptr, d_tags.ptr, pkt_len=size)
        res.append((timed(pk, n * size, lib.dev_sync, queued=4) / GiB, shape))
        ctx.close()
    print("%-10s %8d %10.1f %10.1f %12.1f %14.1f  %s / %s" % ("%d KiB" % kib, n, r_wait, r_k3, res[0][0], res[1][0],
          "rows" if res[0][1] == lib.SHAPE_ROWS else "%d lanes" % res[0][1], "%d lanes per packet" % res[1][1]))
    sys.stdout.flush()
