#!/bin/bash
# round 4, call 23: the one-key packet shapes over count and size again (the shape rule packets_pick_lg dates from round 3; k_pktl is 10 % faster since)
O=$PWD/gpurun_out/r04_run23; mkdir -p $O
timeout 900 python3 profiles/packets_sweep.py 32 2>&1 | tee $O/packets_sweep_aes256.txt
timeout 900 python3 profiles/packets_sweep.py 16 2>&1 | tee $O/packets_sweep_aes128.txt
