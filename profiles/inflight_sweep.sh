#!/bin/bash
# Sustained rate of mid-size AES-256-GCM messages with K in flight (GPU box, from the repo root):  bash profiles/inflight_sweep.sh <out dir> [sizes in MiB ...]
# One bench.py --inflight line per (size, K); prints a table: MiB, K, GiB/s, us per message, tags ok.
O=$1; shift
mkdir -p $O
SIZES=${@:-1 4 16 64 256}
for M in $SIZES; do
  for K in ${INFLIGHT_KS:-1 2 4}; do
    G=$(python3 -c "print($M/1024)")
    timeout 300 python3 bench.py --gib-per-gpu $G --inflight $K --no-cpu-baseline ${INFLIGHT_ARGS:-} > $O/inflight_${M}m_k$K.json 2> $O/inflight_${M}m_k$K.err
    python3 - $O/inflight_${M}m_k$K.json $M $K <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%6s MiB  K=%s  %8.1f GiB/s  %9.2f us/message  kernel alone %8.2f us  tags ok %s (%d checked)" % (sys.argv[2], sys.argv[3], d["value"], d["config"]["us_per_message"], d["roofline"]["avg_launch_ms"] * 1e3, d["tag_ok"], d["tags_checked"]))
except Exception as e:
    print("%6s MiB  K=%s  unreadable: %r" % (sys.argv[2], sys.argv[3], e))
PY
  done
done
