#!/bin/bash
# HBM traffic counters of k_main for the default 16 GiB launch (GPU box): separate rocprofv3 --pmc passes.
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for C in WRITE_SIZE FETCH_SIZE; do
  rm -rf /tmp/pm_$C
  rocprofv3 --pmc $C --output-format csv -d /tmp/pm_$C -- python3 $REPO/profiles/wgtrace.py 16 > /tmp/pm.log 2>&1
  python3 - <<PY
import csv,glob
for p in glob.glob("/tmp/pm_$C/**/*counter_collection.csv", recursive=True):
    rows=[r for r in csv.DictReader(open(p)) if "k_main" in r["Kernel_Name"]]
    by={}
    for r in rows: by[r["Dispatch_Id"]]=by.get(r["Dispatch_Id"],0.0)+float(r["Counter_Value"])
    print("$C (1e6 KiB per launch)", [round(v/1e6,2) for v in by.values()], "scratch/vgpr", set((r["Scratch_Size"] if "Scratch_Size" in r else "?", r.get("VGPR_Count")) for r in rows))
PY
done
grep -E "kernel|clock" /tmp/pm.log
