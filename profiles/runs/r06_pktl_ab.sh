#!/bin/bash
# round 6: k_pktl builds side by side on one box (GPU box, through gpurun): bash profiles/runs/r06_pktl_ab.sh <name>=<library> ...   ("new" = the in-tree library is always there)
# 2^20 frames of 64 .. 1514 bytes with 28 bytes of AAD, and fixed sizes without AAD; each as the real call and as its no-data twin; two interleaved rounds
O=$PWD/gpurun_out/r06_pktl_ab; mkdir -p $O; : > $O/ab.jsonl
for r in 1 2; do
  for a in "" "--dec" "--fixed 1024 --aad 0" "--fixed 1025 --aad 0" "--fixed 1500 --aad 16"; do
    for v in new=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so "$@"; do
      for x in "" "--probe"; do
        [ "$x" = "--probe" ] && [ "$a" = "--dec" ] && continue
        echo -n "{\"variant\": \"${v%%=*}\", \"probe\": \"$x\", \"args\": \"$a\", \"line\": " >> $O/ab.jsonl
        AESGCM_LIB=${v#*=} timeout 300 python3 profiles/frames_one.py --steps 12 $x $a >> $O/ab.jsonl 2>> $O/ab.err || echo null >> $O/ab.jsonl
        sed -i '$ s/$/}/' $O/ab.jsonl
      done
    done
  done
done
python3 - $O/ab.jsonl <<'PY'
import json, sys, collections, statistics
t = collections.defaultdict(list)
for l in open(sys.argv[1]):
    d = json.loads(l)
    if d["line"]: t[(d["args"] or "frames", d["variant"], d["probe"])].append(d["line"]["ms_median"])
for (a, v, x), ms in t.items():
    print("%-24s %-8s %-8s %s" % (a, v, x, " ".join("%.4f" % m for m in ms)))
PY
