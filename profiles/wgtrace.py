#!/usr/bin/env python3
"""Per-workgroup timeline of one fused-kernel launch (run on the GPU box): which CU/XCD each workgroup ran
on, when it started and ended.  Used to find load imbalance."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(gib * (1 << 30))
ctx = lib.Context(bytes(range(32)))
a, b = lib.DeviceBuffer(n), lib.DeviceBuffer(n)
a.fill_splitmix64(1)
iv = bytes(12)
for _ in range(2):
    ctx.encrypt_dev(iv, a.ptr, n, b.ptr)
ctx.timing_enable(True)
ctx.encrypt_dev(iv, a.ptr, n, b.ptr)
tr = ctx.wg_trace()
nl, ms = ctx.timing_read()
t0 = min(t[0] for t in tr)
ends = sorted((t[1] - t0) / 100.0 for t in tr)          # us
starts = sorted((t[0] - t0) / 100.0 for t in tr)
print("kernel %.3f ms, %d workgroups" % (ms, len(tr)))
print("start  us: min %.1f med %.1f max %.1f" % (starts[0], starts[len(starts) // 2], starts[-1]))
print("end    us: min %.1f p10 %.1f med %.1f p90 %.1f max %.1f" % (ends[0], ends[len(ends) // 10], ends[len(ends) // 2], ends[9 * len(ends) // 10], ends[-1]))
dur = sorted((t[1] - t[0]) / 100.0 for t in tr)
print("dur    us: min %.1f med %.1f max %.1f" % (dur[0], dur[len(dur) // 2], dur[-1]))
per_cu = collections.Counter()
per_xcc = collections.defaultdict(list)
for i, t in enumerate(tr):
    hw, xcc = t[2] & 0xFFFFFFFF, (t[2] >> 32) & 0xF
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    per_cu[(xcc, se, sh, cu)] += 1
    per_xcc[xcc].append((t[1] - t0) / 100.0)
print("distinct CUs used:", len(per_cu), " workgroups per CU histogram:", collections.Counter(per_cu.values()))
for x in sorted(per_xcc):
    v = sorted(per_xcc[x])
    print("xcc %d: %3d wgs, end med %.1f max %.1f us" % (x, len(v), v[len(v) // 2], v[-1]))
chunks = sorted(t[3] & 0xFFFFFFFF for t in tr)
kc = sum(t[3] >> 32 for t in tr) * 1024.0          # shader cycles summed over all waves
waves = len(tr) * 16
print("mean shader clock while resident: %.0f MHz" % (kc / waves / (ms * 1e3)))
print("chunks per workgroup: min %d med %d max %d (total %d)" % (chunks[0], chunks[len(chunks) // 2], chunks[-1], sum(chunks)))
