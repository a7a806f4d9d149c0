#!/bin/bash
# round 6: the counting sort in front of the packet kernels (k_len_hist / k_len_scan / k_len_scatter), shipped library of the collection (old=<path>) against the in-tree one:
# kernel times from rocprofv3 --kernel-trace --stats, and the call's time, at 2^20 / 65536 / 16384 frames.  -> profiles/r06/len_sort_ab.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_len_sort; mkdir -p $O; : > $O/ab.txt
for v in new=$R/aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so "$@"; do
  for n in 1048576 65536 16384; do
    export AESGCM_LIB=${v#*=}
    echo "== ${v%%=*} n=$n" >> $O/ab.txt
    python3 $R/profiles/frames_one.py --n $n --steps 20 >> $O/ab.txt 2>&1
    rm -rf $O/prof; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/profiles/frames_one.py --n $n --steps 20 > /dev/null 2>&1
    f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
    python3 - "$f" >> $O/ab.txt <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith(("k_len_","void k_pkt","k_rows_plan_sums","void k_rows<")): print("   %-60s calls %4s avg %9.1f ns" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])))
PY
  done
done
cat $O/ab.txt
