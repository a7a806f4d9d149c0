#!/bin/bash
# round 2, call 30: per-kernel split of 256 MiB and 1 GiB messages (k_body path) under rocprofv3 --kernel-trace
O=$PWD/gpurun_out/r02_run30; mkdir -p $O
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for mib in 256 1024; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$mib -- python3 $REPO/profiles/latency_one.py $((mib*1048576)) 12 > $O/s$mib.log 2>&1
  t=$(find $O/s$mib -name "*kernel_trace.csv" | head -1); echo "== $mib MiB"; python3 - $t <<'PY'
import csv,sys
rows=sorted(csv.DictReader(open(sys.argv[1])), key=lambda r:int(r["Start_Timestamp"]))
ks=[i for i,r in enumerate(rows) if "k_body" in r["Kernel_Name"]]
i=ks[-1]; t0=int(rows[i]["Start_Timestamp"])
for r in rows[i-1:i+5]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("   %-34s start %8.1f us  dur %8.1f us  grid %s" % (r["Kernel_Name"][:34], (s-t0)/1e3, (e-s)/1e3, r.get("Grid_Size_X") or r.get("Grid_Size")))
PY
done
