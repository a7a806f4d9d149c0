#!/bin/bash
# round 3, call 14: per-kernel split of mid-size messages (1, 16, 64, 256 MiB, 1 GiB) under rocprofv3 --kernel-trace
O=$PWD/gpurun_out/r03_run14; mkdir -p $O
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for mib in 1 16 64 256 1024; do
  rocprofv3 --kernel-trace --output-format csv -d $O/s$mib -- python3 $REPO/profiles/latency_one.py $((mib*1048576)) 12 > $O/s$mib.log 2>&1
  t=$(find $O/s$mib -name "*kernel_trace.csv" | head -1); echo "== $mib MiB  $(tail -1 $O/s$mib.log)"; python3 - $t <<'PY'
import csv,sys
rows=sorted(csv.DictReader(open(sys.argv[1])), key=lambda r:int(r["Start_Timestamp"]))
ks=[i for i,r in enumerate(rows) if ("k_body" in r["Kernel_Name"] or "k_main" in r["Kernel_Name"])]
i=ks[-1]; t0=int(rows[i]["Start_Timestamp"])
for r in rows[i:i+6]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("   %-34s start %8.1f us  dur %8.1f us  grid %s wg %s" % (r["Kernel_Name"][:34], (s-t0)/1e3, (e-s)/1e3, r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size")))
PY
  rm -rf $O/s$mib
done
