#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash profiles/collect.sh <round-tag> [bench args]
# Produces under gpurun_out/prof_<tag>/ : kernel-trace + stats of bench.py, and separate PMC passes
# (never combined with tracing; FETCH_SIZE and WRITE_SIZE in their own runs -- MI355X_MICROARCH.md).
TAG=${1:-r1}; shift
ARGS=${@:---steps 4 --warmup 1 --no-cpu-baseline}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py $ARGS > $OUT/stats_bench.json 2> $OUT/stats.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq1 -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq1.json 2> $OUT/pmc_sq1.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq2 -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq2.json 2> $OUT/pmc_sq2.err
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_grbm -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_grbm.json 2> $OUT/pmc_grbm.err
cd $REPO
find $OUT -name "*.csv" | head -40
python3 profiles/summarize.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
