#!/bin/bash
# round 3, call 23: what it costs to have the ciphertext in memory when the in-launch tag appears
O=$PWD/gpurun_out/r03_run23; mkdir -p $O
timeout 600 python profiles/cyc_end.py | tee $O/cyc_end.txt
timeout 600 python -m pytest tests/test_gpu_cyclic.py tests/test_gpu_pinned.py -x -q -m gpu 2>&1 | tail -3
