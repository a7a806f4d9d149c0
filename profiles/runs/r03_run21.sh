#!/bin/bash
# round 3, call 21: lower end of the fused cyclic launch
O=$PWD/gpurun_out/r03_run21; mkdir -p $O
timeout 600 python profiles/cyc_small.py | tee $O/cyc_small.txt
timeout 600 python profiles/cyc_sweep.py 16 | tee $O/cyc_sweep_aes128.txt
