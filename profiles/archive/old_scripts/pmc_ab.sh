#!/bin/bash
# HBM traffic counters of k_main vs rows-per-chunk (GPU box): separate rocprofv3 --pmc passes per counter.
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for TW in 32 64 128 256; do
  for C in WRITE_SIZE FETCH_SIZE; do
    rm -rf /tmp/pm_$C
    AESGCM_TW=$TW rocprofv3 --pmc $C --output-format csv -d /tmp/pm_$C -- python3 $REPO/profiles/wgtrace.py 16 > /tmp/pm.log 2>&1
    python3 - <<PY
import csv,glob
for p in glob.glob("/tmp/pm_$C/**/*counter_collection.csv", recursive=True):
    rows=[r for r in csv.DictReader(open(p)) if "k_main" in r["Kernel_Name"]]
    by={}
    for r in rows: by[r["Dispatch_Id"]]=by.get(r["Dispatch_Id"],0.0)+float(r["Counter_Value"])
    print("TW=$TW $C (1e6 KiB per launch)", [round(v/1e6,2) for v in by.values()])
PY
  done
  grep kernel /tmp/pm.log
done
