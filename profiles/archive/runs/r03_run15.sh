#!/bin/bash
# round 3, call 15: the timeline of ONE message at mid sizes (where do the microseconds between k_main/k_body and the tag go?)
O=$PWD/gpurun_out/r03_run15; mkdir -p $O
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for n in 1048576 16777216 268435456 1073741824; do
  rocprofv3 --kernel-trace --output-format csv -d $O/trace_$n -- python3 $REPO/profiles/latency_one.py $n 12 > $O/trace_$n.out 2> $O/trace_$n.err
  t=$(find $O/trace_$n -name "*kernel_trace.csv" | head -1)
  python3 - $t $n <<'PY' | tee -a $O/timeline.txt
import csv,sys
rows=sorted(csv.DictReader(open(sys.argv[1])), key=lambda r:int(r["Start_Timestamp"]))
rows=[r for r in rows if not r["Kernel_Name"].startswith(("k_fill","k_setup","k_init"))]
# last two calls: find fused kernels
ks=[i for i,r in enumerate(rows) if "k_main" in r["Kernel_Name"] or "k_body" in r["Kernel_Name"]]
print("== %s bytes" % sys.argv[2])
first=ks[-3]
t0=int(rows[first]["Start_Timestamp"]); prev_end=None
for r in rows[first:]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("   %-44s start %9.1f us  dur %8.1f us  gap %6.1f us" % (r["Kernel_Name"][:44], (s-t0)/1e3, (e-s)/1e3, 0 if prev_end is None else (s-prev_end)/1e3))
    prev_end=e
PY
done
cd $REPO
python - <<'PY' | tee $O/host_times.txt
import os, sys, time, statistics
sys.path.insert(0, os.getcwd())
import aesgcm_amd
from aesgcm_amd import lib
ctx = lib.Context(bytes(range(32)))
for n in (1 << 20, 4 << 20, 16 << 20, 64 << 20, 256 << 20, 1 << 30):
    a, b = lib.DeviceBuffer(n), lib.DeviceBuffer(n); a.fill_splitmix64(1); lib.dev_sync()
    ts = []
    for i in range(40):
        t0 = time.perf_counter(); ctx.encrypt_dev(bytes(12), a.ptr, n, b.ptr); ts.append(time.perf_counter() - t0)
    print("%11d B  median %8.1f us  best %8.1f us" % (n, statistics.median(ts[5:]) * 1e6, min(ts) * 1e6), flush=True)
    del a, b
PY
