#!/bin/bash
# round 4, call 47: the whole GPU suite three times over on one box (flakiness check of the final state)
O=$PWD/gpurun_out/r04_run47; mkdir -p $O
for i in 1 2 3; do
  timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/pytest_$i.txt 2>&1; echo "pass $i rc=$? $(tail -1 $O/pytest_$i.txt)"
done | tee $O/summary.txt
