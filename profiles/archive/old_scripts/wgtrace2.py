#!/usr/bin/env python3
"""Per-workgroup end times of k_body launches inside a loop of waited calls (GPU box): how long the launch really is in the loop (first start to last end
on the GPU's wall clock, against the HIP-event time) and how the workgroups' ends spread over XCDs -- dealt chunks or cyclic rows (AESGCM_BODY_CYC).
    python profiles/wgtrace2.py KEY_BYTES MIB [calls]"""
import collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib
kb, mib = int(sys.argv[1]), int(sys.argv[2])
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 5
n = mib << 20
ctx = lib.Context(bytes(range(kb)))
a, b = lib.DeviceBuffer(n), lib.DeviceBuffer(n)
a.fill_splitmix64(1)
iv = bytes(12)
for _ in range(3):
    ctx.encrypt_dev(iv, a.ptr, n, b.ptr)
ctx.timing_enable(True)
print("AES-%d %d MiB  AESGCM_BODY_CYC=%s" % (kb * 8, mib, os.environ.get("AESGCM_BODY_CYC")))
for it in range(calls):
    t0 = time.perf_counter()
    ctx.encrypt_dev(iv, a.ptr, n, b.ptr)
    wall = (time.perf_counter() - t0) * 1e6
    tr = ctx.wg_trace()
    nl, ms = ctx.timing_read()
    s0 = min(t[0] for t in tr)
    ends = sorted((t[1] - s0) / 100.0 for t in tr)
    starts = sorted((t[0] - s0) / 100.0 for t in tr)
    per_xcc = collections.defaultdict(list)
    for t in tr:
        per_xcc[(t[2] >> 32) & 0xF].append((t[1] - s0) / 100.0)
    xs = " ".join("x%d %.0f/%.0f" % (x, sorted(v)[len(v) // 2], max(v)) for x, v in sorted(per_xcc.items()))
    print("call %d: host %.1f us  event %.1f us  gpu clock first start -> last end %.1f us (starts spread %.1f) | ends: min %.1f p10 %.1f med %.1f p90 %.1f max %.1f | per XCD med/max: %s" % (
        it, wall, ms * 1e3, ends[-1], starts[-1], ends[0], ends[len(ends) // 10], ends[len(ends) // 2], ends[9 * len(ends) // 10], ends[-1], xs))
