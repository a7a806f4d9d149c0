#!/bin/bash
# A/B of k_batch3 builds on ONE box (GPU box, from the repo root):  bash profiles/batch_ab.sh <out dir> <lib> [<lib> ...]
# Every <lib> is a -DAESGCM_DEBUG_KNOBS build of a variant (it serves as product and as debug library: AESGCM_LIB = AESGCM_LIB_DEBUG).  For each library and each
# forced lane-group size (BATCH_AB_LGS, default "3 4" -> bench.py --batch-lanes 8 / 16): the cfg5 bench line (tags checked against the fixture),
# then two counter passes of the same command (never combined with tracing): the LDS / instruction counters, and GRBM_GUI_ACTIVE.
# Prints one row per variant: GiB/s, kernel ms, LDS-array cycles, conflict cycles, their ratio, LDS busy, VALU and LDS instructions.
O=$1; shift
mkdir -p $O
REPO=$PWD
EXTRA=${BATCH_AB_ARGS:-}
for LIB in "$@"; do
  N=$(basename $LIB .so)
  for LG in ${BATCH_AB_LGS:-3 4}; do
    T=${N}_lg$LG
    ( cd $REPO && AESGCM_LIB=$LIB AESGCM_LIB_DEBUG=$LIB timeout 300 python3 bench.py --config cfg5 --batch-lanes $((1 << LG)) --no-cpu-baseline $EXTRA > $O/bench_$T.json 2> $O/bench_$T.err )
    ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ab_$T && \
      AESGCM_LIB=$LIB AESGCM_LIB_DEBUG=$LIB timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/ab_$T/sq -- python3 $REPO/bench.py --config cfg5 --batch-lanes $((1 << LG)) --steps 4 --warmup 1 --no-cpu-baseline $EXTRA > /dev/null 2> $O/pmc_$T.err; \
      AESGCM_LIB=$LIB AESGCM_LIB_DEBUG=$LIB timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/ab_$T/grbm -- python3 $REPO/bench.py --config cfg5 --batch-lanes $((1 << LG)) --steps 4 --warmup 1 --no-cpu-baseline $EXTRA > /dev/null 2>> $O/pmc_$T.err; \
      for C in FETCH_SIZE WRITE_SIZE; do AESGCM_LIB=$LIB AESGCM_LIB_DEBUG=$LIB timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/ab_$T/$C -- python3 $REPO/bench.py --config cfg5 --batch-lanes $((1 << LG)) --steps 4 --warmup 1 --no-cpu-baseline $EXTRA > /dev/null 2>> $O/pmc_$T.err; done )
    python3 - $O $T /tmp/ab_$T <<'PY'
import csv, glob, json, sys
from collections import defaultdict
O, T, D = sys.argv[1:4]
acc, disp = defaultdict(float), defaultdict(set)
for p in glob.glob(D + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "k_batch" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp[r["Counter_Name"]].add(r["Dispatch_Id"])
pm = {c: acc[c] / max(1, len(disp[c])) for c in acc}
try:
    b = json.loads(open("%s/bench_%s.json" % (O, T)).read().strip().splitlines()[-1])
    v, ms, ok, k = b["value"], b["roofline"]["avg_launch_ms"], b.get("tag_ok"), b["roofline"].get("kernel")
except Exception as e:
    v, ms, ok, k = float("nan"), float("nan"), "unreadable: %s" % e, None
ia, bc, g = pm.get("SQ_LDS_IDX_ACTIVE", 0), pm.get("SQ_LDS_BANK_CONFLICT", 0), pm.get("GRBM_GUI_ACTIVE", 0)
hbm = pm.get("FETCH_SIZE", 0) * 2048 + pm.get("WRITE_SIZE", 0) * 1024
row = {"variant": T, "kernel": k, "GiB_s": v, "kernel_ms": ms, "tag_ok": ok, "lds_idx_active": ia, "lds_bank_conflict": bc,
       "conflict_frac": bc / ia if ia else None, "lds_busy": ia / (256 * g / 8) if g else None, "insts_valu": pm.get("SQ_INSTS_VALU"), "insts_lds": pm.get("SQ_INSTS_LDS"),
       "wait_inst_lds": pm.get("SQ_WAIT_INST_LDS"), "grbm_gui_active": g, "hbm_bytes": hbm}
json.dump(row, open("%s/row_%s.json" % (O, T), "w"))
print("%-16s %-22s %7.1f GiB/s  kernel %.3f ms  tag_ok %s  LDS array %.3e  conflicts %.3e (%.1f %%)  busy %.3f  VALU %.3e  LDS insts %.3e  HBM %.4e B" % (
    T, k, v, ms, ok, ia, bc, 100 * bc / ia if ia else 0, row["lds_busy"] or 0, pm.get("SQ_INSTS_VALU", 0), pm.get("SQ_INSTS_LDS", 0), hbm))
PY
  done
done
