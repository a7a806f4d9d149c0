#!/bin/bash
# round 3, call 27: timeline of the cfg2 step (1 GiB AES-128): where are the 130 us between the kernel's event time and the step?
O=$PWD/gpurun_out/r03_run27; mkdir -p $O
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $REPO/bench.py --config cfg2 --steps 6 --warmup 2 --no-cpu-baseline > $O/trace.json 2> $O/trace.err
t=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - $t <<'PY' | tee $O/timeline.txt
import csv,sys
rows=sorted(csv.DictReader(open(sys.argv[1])), key=lambda r:int(r["Start_Timestamp"]))
ks=[i for i,r in enumerate(rows) if "k_body<10, 0" in r["Kernel_Name"]]
print(len(ks), "k_body launches")
first=ks[3]
t0=int(rows[first]["Start_Timestamp"]); prev=None
for r in rows[first:first+30]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("   %-44s start %9.1f us  dur %8.1f us  gap %6.1f us" % (r["Kernel_Name"][:44], (s-t0)/1e3, (e-s)/1e3, 0 if prev is None else (s-prev)/1e3))
    prev=e
PY
tail -1 $O/trace.json | cut -c1-300
