#!/bin/bash
# A/B of packet-kernel builds on ONE box (GPU box, from the repo root):  bash profiles/pkt_ab.sh <out dir> <kind> <kernel regex> <lib> [<lib> ...]
# kind = pkt_bench.py's (pktl, pktg, pktg8, pktg4, pktw, pkt, batch); every <lib> must be a -DAESGCM_DEBUG_KNOBS build (forced shapes).  Per library and
# PKT_AB_LENS (default "1024"): the timing line of profiles/pkt_bench.py, then FETCH_SIZE / WRITE_SIZE in passes of their own (never with tracing).
O=$1; KIND=$2; RE=$3; shift 3
mkdir -p $O
REPO=$PWD
for LIB in "$@"; do
  N=$(basename $LIB .so)
  for LEN in ${PKT_AB_LENS:-1024}; do
    T=${N}_${KIND}_$LEN
    ARGS="$KIND --len $LEN --key-bits ${PKT_AB_KEYBITS:-256} --steps 7 ${PKT_AB_ARGS:-}"
    ( cd $REPO && AESGCM_LIB=$LIB AESGCM_LIB_DEBUG=$LIB timeout 300 python3 profiles/pkt_bench.py $ARGS > $O/bench_$T.json 2> $O/bench_$T.err )
    ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pab_$T && for C in FETCH_SIZE WRITE_SIZE; do
        AESGCM_LIB=$LIB AESGCM_LIB_DEBUG=$LIB timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pab_$T/$C -- python3 $REPO/profiles/pkt_bench.py $ARGS > /dev/null 2>> $O/pmc_$T.err; done )
    python3 - $O $T /tmp/pab_$T "$RE" <<'PY'
import csv, glob, json, re, sys
from collections import defaultdict
O, T, D, RE = sys.argv[1:5]
acc, disp = defaultdict(float), defaultdict(set)
for p in glob.glob(D + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if re.search(RE, r["Kernel_Name"]):
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp[r["Counter_Name"]].add(r["Dispatch_Id"])
pm = {c: acc[c] / max(1, len(disp[c])) for c in acc}
try:
    b = json.loads(open("%s/bench_%s.json" % (O, T)).read().strip().splitlines()[-1])
except Exception as e:
    b = {"gib_per_s": float("nan"), "ms_median": float("nan"), "ms_best": float("nan"), "alg_bytes_per_launch": 0, "err": repr(e)}
hbm = pm.get("FETCH_SIZE", 0) * 2048 + pm.get("WRITE_SIZE", 0) * 1024
alg = b.get("alg_bytes_per_launch") or 0
print("%-28s %7.1f GiB/s  median %.3f ms  best %.3f ms  HBM %.4e B = %.3f x algorithmic (reads %.4e, writes %.4e)" % (
    T, b["gib_per_s"], b["ms_median"], b["ms_best"], hbm, hbm / alg if alg else 0, pm.get("FETCH_SIZE", 0) * 2048, pm.get("WRITE_SIZE", 0) * 1024))
PY
  done
done
