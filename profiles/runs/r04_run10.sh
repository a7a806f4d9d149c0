#!/bin/bash
# round 4, call 10: where the half shape stops paying: 8 .. 96 MiB, full shape K = 2 against half shape K = 2, 3 on one box; its per-workgroup timeline
O=$PWD/gpurun_out/r04_run10; mkdir -p $O
echo "== full shape"; INFLIGHT_KS="2 3" INFLIGHT_ARGS="--half 0" bash profiles/inflight_sweep.sh $O/full 8 24 32 48 64 96 128 2>&1 | tee $O/inflight_full.txt
echo "== half shape"; INFLIGHT_KS="2 3" INFLIGHT_ARGS="--half 1" bash profiles/inflight_sweep.sh $O/half 8 24 32 48 64 96 128 2>&1 | tee $O/inflight_half.txt
