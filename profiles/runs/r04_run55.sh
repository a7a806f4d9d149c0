#!/bin/bash
# round 4, call 55: the idle-workgroup exit, same box A/B: waited latency by size (python wall time), FETCH_SIZE of 1 MiB messages, in-flight small messages
O=$PWD/gpurun_out/r04_run55; mkdir -p $O
E=$PWD/experiments
for V in noidle idle noidle idle; do
  echo "== $V"; AESGCM_LIB=$E/lib_$V.so timeout 300 python3 profiles/latency.py 1000 2>&1 | tail -7
done | tee $O/latency_ab.txt
for V in noidle idle; do
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/f55_$V && AESGCM_LIB=$E/lib_$V.so timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/f55_$V -- python3 $OLDPWD/profiles/latency_one.py 1048576 200 > /dev/null 2> $O/pmc_$V.err )
  python3 - /tmp/f55_$V $V <<'PY'
import csv, glob, sys
tot = n = 0
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "k_body" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            tot += float(r["Counter_Value"]); n += 1
print("%s: 1 MiB messages, k_body launches %d, HBM reads per launch %.3e B (FETCH_SIZE x 2048; the message is 1.049e6)" % (sys.argv[2], n, tot / max(n, 1) * 2048))
PY
done | tee $O/fetch_ab.txt
