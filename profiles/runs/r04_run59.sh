#!/bin/bash
# round 4, call 59: small messages from several host threads (examples/mt_stream.c): does the sustained rate scale beyond one thread's ~6 us per launch?
O=$PWD/gpurun_out/r04_run59; mkdir -p $O
for S in 64 1024 4096 16384; do
  for T in 1 2 4 8; do
    N=$((S <= 1024 ? 20000 : S <= 4096 ? 8000 : 3000))
    timeout 120 ./examples/mt_stream $S $T 3 $N
  done
done 2>&1 | tee $O/mt_stream.txt
