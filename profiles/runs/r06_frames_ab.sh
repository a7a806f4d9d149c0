#!/bin/bash
# round 6: which side of the data traffic the cycles between k_pktl's probe and the real kernel belong to (GPU box, through gpurun).  experiments/libaesgcm_<V>.so are builds
# of the same sources with -DAESGCM_PKT_<V> (aesgcm_pkt.h: NO_LOADS, NO_STORES, NT_LOADS, NT_STORES, NT_BOTH, WT_STORES); three interleaved rounds against the shipped library
# and its probe, so that the box's clock drift lands on every variant alike.  -> profiles/r06/frames/loads_stores_ab.txt
O=$PWD/gpurun_out/r06_ab; mkdir -p $O; : > $O/ab.jsonl
for r in 1 2 3; do
  for a in "" "--fixed 1024 --aad 0"; do
    for v in base probe NO_LOADS NO_STORES NT_LOADS NT_STORES NT_BOTH WT_STORES; do
      case $v in base) L=; X=;; probe) L=; X=--probe;; *) L=$PWD/experiments/libaesgcm_$v.so; X=;; esac
      echo -n "{\"variant\": \"$v\", \"round\": $r, \"args\": \"$a\", \"line\": " >> $O/ab.jsonl
      AESGCM_LIB=$L timeout 300 python3 profiles/frames_one.py --steps 12 $X $a >> $O/ab.jsonl 2>> $O/ab.err || echo null >> $O/ab.jsonl
      sed -i '$ s/$/}/' $O/ab.jsonl
    done
  done
done
python3 - $O/ab.jsonl <<'PY'
import json, sys, collections, statistics
t = collections.defaultdict(list)
for l in open(sys.argv[1]):
    d = json.loads(l)
    if d["line"]: t[(d["args"] or "frames 64..1514 + 28 AAD", d["variant"])].append(d["line"]["ms_median"])
for (a, v), ms in t.items():
    print("%-28s %-10s %s  median %.4f ms" % (a, v, " ".join("%.4f" % m for m in ms), statistics.median(ms)))
PY
