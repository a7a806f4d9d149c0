#!/bin/bash
# round 4, call 14: messages in flight through the DEALT kernel (dispensers absorb CUs that are busy with another message's fold) against the cyclic rows:
# 32 .. 256 MiB, K = 2, 3, one box
O=$PWD/gpurun_out/r04_run14; mkdir -p $O
echo "== cyclic rows, full shape"; INFLIGHT_KS="2 3" INFLIGHT_ARGS="--half 0" bash profiles/inflight_sweep.sh $O/cyc 32 64 128 256 2>&1 | tee $O/inflight_cyc.txt
echo "== dealt chunks (cyc_max 0, body_min 16 MiB)"; INFLIGHT_KS="1 2 3" INFLIGHT_ARGS="--half 0 --opt cyc_min=0 --opt cyc_max=0 --opt body_min=16777216" bash profiles/inflight_sweep.sh $O/dealt 32 64 128 256 2>&1 | tee $O/inflight_dealt.txt
echo "== dealt chunks, no FoldClose"; INFLIGHT_KS="2 3" INFLIGHT_ARGS="--half 0 --opt cyc_min=0 --opt cyc_max=0 --opt body_min=16777216 --opt fold_close=0" bash profiles/inflight_sweep.sh $O/dealt_nofc 64 128 2>&1 | tee $O/inflight_dealt_nofc.txt
