#!/bin/bash
# round 4, call 48: does the runtime's hardware-queue limit (GPU_MAX_HW_QUEUES, default 4) explain why four messages in flight are slower than three?
O=$PWD/gpurun_out/r04_run48; mkdir -p $O
for Q in default 8; do
  echo "== GPU_MAX_HW_QUEUES $Q"
  if [ $Q != default ]; then export GPU_MAX_HW_QUEUES=$Q; fi
  INFLIGHT_KS="3 4 6 8" bash profiles/inflight_sweep.sh $O/q$Q 4 16 64
done 2>&1 | tee $O/inflight_hw_queues.txt
