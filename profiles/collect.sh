#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:
#     bash profiles/collect.sh <tag> <hot-kernel-regex> <python script and args ...>
# e.g. bash profiles/collect.sh cfg3_n1 'k_body|k_main' bench.py --steps 4 --warmup 1 --no-cpu-baseline
# Produces gpurun_out/prof_<tag>/ : rocprofv3 kernel-trace + stats of the command, then separate PMC passes of the same
# command (never combined with tracing; FETCH_SIZE and WRITE_SIZE in their own runs -- MI355X_MICROARCH.md), and
# summary.txt / summary.json / pmc_<tag>.json stamped with the git hash and the SHA-256 of the library that ran.
TAG=$1; HOT=$2; shift 2
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/"$@" > $OUT/stats_run.json 2> $OUT/stats.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $REPO/"$@" > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
done
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $REPO/"$@" > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err
# second SQ pass (round 3, what binds the fused kernel): cycles with an instruction of each class in flight, wave and busy cycles
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD --output-format csv -d $OUT/pmc_sq2 -- python3 $REPO/"$@" > $OUT/pmc_sq2.json 2> $OUT/pmc_sq2.err
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_LDS --output-format csv -d $OUT/pmc_sq3 -- python3 $REPO/"$@" > $OUT/pmc_sq3.json 2> $OUT/pmc_sq3.err
# request sizes at the L2's memory side: FETCH_SIZE is only calibrated for wide coalesced streams (it tallies 128-byte requests at 64 B)
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/pmc_tcc_rd -- python3 $REPO/"$@" > $OUT/pmc_tcc_rd.json 2> $OUT/pmc_tcc_rd.err
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc_wr -- python3 $REPO/"$@" > $OUT/pmc_tcc_wr.json 2> $OUT/pmc_tcc_wr.err
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm -- python3 $REPO/"$@" > $OUT/pmc_grbm.json 2> $OUT/pmc_grbm.err
cd $REPO
python3 profiles/summarize.py $OUT "$TAG" "$HOT" > $OUT/summary.txt 2>&1
tail -40 $OUT/summary.txt
