#!/bin/bash
# round 3, call 40: batch shapes over count and size with the 8-lane k_batch3
O=$PWD/gpurun_out/r03_run40; mkdir -p $O
timeout 1200 python profiles/batch_sweep.py 16 | tee $O/batch_sweep_aes128.txt
