#!/bin/bash
# round 2, call 24: which kernel got slower with the two-set dispensers (4 GiB message, per-kernel times under rocprofv3)
O=$PWD/gpurun_out/r02_run24; mkdir -p $O
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for v in _fg ""; do
  export AESGCM_LIB=$REPO/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$v -- python3 $REPO/profiles/latency_one.py $((4096*1048576)) 6 > $O/s$v.log 2>&1
  f=$(find $O/s$v -name "*kernel_stats.csv" | head -1); echo "== lib '$v'"; cut -d, -f1-4 $f | head -6 | cut -c1-160
  t=$(find $O/s$v -name "*kernel_trace.csv" | head -1); python3 - $t <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "k_body" in r["Kernel_Name"]]
print("   k_body durations us:", [round((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3) for r in rows])
PY
done
