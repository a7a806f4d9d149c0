#!/bin/bash
# round 4, call 21: messages in flight with the closing OUTSIDE the row launch (context option cyc_close=0: k_fold + k_combine behind k_body on the context's
# stream, where they can overlap the next message's rows on another stream) against the in-launch closing, full and half shape; same box
O=$PWD/gpurun_out/r04_run21; mkdir -p $O
for V in "fused:" "three:--opt cyc_close=0" "half:--half 1" "half_three:--half 1 --opt cyc_close=0"; do
  N=${V%%:*}; A=${V#*:}
  echo "== $N ($A)"
  INFLIGHT_KS="1 2 3 4" INFLIGHT_ARGS="$A" bash profiles/inflight_sweep.sh $O/$N 16 64
done 2>&1 | tee $O/inflight_closing_outside.txt
