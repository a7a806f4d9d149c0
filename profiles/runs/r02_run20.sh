#!/bin/bash
# round 2, call 20: per-kernel split of mid-size messages (k_main path) under rocprofv3 --kernel-trace --stats
O=$PWD/gpurun_out/r02_run20; mkdir -p $O
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for mib in 1 16 64; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$mib -- python3 $REPO/profiles/latency_one.py $((mib*1048576)) 50 > $O/s$mib.log 2>&1
  f=$(find $O/s$mib -name "*kernel_stats.csv" | head -1); echo "== $mib MiB"; cut -d, -f1-7 $f | cut -c1-150
  t=$(find $O/s$mib -name "*kernel_trace.csv" | head -1); python3 - $t <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last call: print kernels with start offsets
tail=rows[-8:]
t0=int(tail[0]["Start_Timestamp"])
for r in tail: print("   %-40s start %8.1f us  dur %8.1f us  grid %s wg %s" % (r["Kernel_Name"][:40], (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r.get("Grid_Size"), r.get("Workgroup_Size")))
PY
done
