#!/bin/bash
# A/B of alternative builds of the row kernel on ONE box (GPU box, from the repo root):  bash profiles/rows_ab.sh out.txt libX.so libY.so ...
# Every build runs the product path (kind "pkt": the library's own rule sends packets of 64 KiB and more by rows) over the shapes of the round-5 marks.
OUT=$1; shift
rm -f $OUT
for L in "$@"; do
  for cfg in "4096 65536" "4096 1048576" "1024 65536" "256 1048576" "64 16777216"; do set -- $cfg
    echo -n "$L " >> $OUT
    AESGCM_LIB=$PWD/$L timeout 120 python profiles/pkt_bench.py pkt --n $1 --len $2 --key-bits 256 --steps 9 2>&1 | cut -c1-215 >> $OUT
  done
done
cat $OUT
