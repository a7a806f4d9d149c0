#!/usr/bin/env python3
"""Where the time of ONE cyclic launch goes (GPU box): per-workgroup timestamps of k_body<.., true> in timing mode -- start, tables staged, last row done, closing
done (aesgcm_ctx_wg_trace; 100 MHz wall clock) -- for a few message sizes.  Prints, in microseconds relative to the first workgroup's start: the spread of the
starts (dispatch), the staging time, the row phase, the closing, the end of the last workgroup; beside it the waited call as the host sees it.
    python profiles/cyc_timeline.py [key_bytes] [MiB ...]"""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib
kb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
sizes = [float(x) for x in sys.argv[2:]] or [0.0625, 1, 4, 16, 64, 256]
MiB = 1 << 20
nmax = int(max(sizes) * MiB)
a, b = lib.DeviceBuffer(nmax + 64), lib.DeviceBuffer(nmax + 64)
a.fill_splitmix64(1)
ctx = lib.Context(bytes(range(kb)))
iv = bytes(12)
med = statistics.median
print("AES-%d   MiB | host us (waited call) | WG starts: spread us | staged +us (med, max) | rows us (med) | closing us (med, max) | last WG done at us | first done at us" % (kb * 8))
for m in sizes:
    n = int(m * MiB) // 16 * 16
    for _ in range(20):
        ctx.encrypt_dev(iv, a.ptr, n, b.ptr)
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); ctx.encrypt_dev(iv, a.ptr, n, b.ptr); ts.append(time.perf_counter() - t0)
    ctx.timing_enable(True)
    rows = []
    for _ in range(5):
        ctx.encrypt_dev(iv, a.ptr, n, b.ptr)
        lib.dev_sync()
        tr = ctx.wg_trace()
        s0 = min(t[0] for t in tr)
        start = [(t[0] - s0) / 100.0 for t in tr]
        staged = [((t[2] >> 40) & 0xFFF) / 100.0 for t in tr]
        rws = [(t[1] - t[0]) / 100.0 - st for t, st in zip(tr, staged)]
        close = [((t[2] >> 52) & 0xFFF) / 100.0 for t in tr]
        done = [(t[1] - s0) / 100.0 + c for t, c in zip(tr, close)]
        rows.append((max(start), med(staged), max(staged), med(rws), med(close), max(close), max(done), min(done)))
    ctx.timing_enable(False)
    r = [med(x) for x in zip(*rows)]
    print("%12.4g | %8.1f            | %8.2f             | %6.2f %6.2f          | %8.2f      | %6.2f %6.2f          | %8.2f          | %8.2f" % (m, med(ts) * 1e6, *r), flush=True)
