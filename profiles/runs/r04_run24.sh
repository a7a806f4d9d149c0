#!/bin/bash
# round 4, call 24: the new shape rule of packets_pick_lg in the `auto` column, and frames of mixed length (offset arrays) over the shapes
O=$PWD/gpurun_out/r04_run24; mkdir -p $O
timeout 900 python3 profiles/packets_sweep.py 32 var 2>&1 | tee $O/packets_sweep_mixed_aes256.txt
timeout 900 python3 profiles/packets_sweep.py 16 var 2>&1 | tee $O/packets_sweep_mixed_aes128.txt
timeout 900 python3 profiles/packets_sweep.py 32 2>&1 | tee $O/packets_sweep_aes256.txt
