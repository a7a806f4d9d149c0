#!/bin/bash
# round 3, call 30: priority rotation in the cyclic launch
O=$PWD/gpurun_out/r03_run30; mkdir -p $O
timeout 900 python profiles/cyc_prio.py 32 | tee $O/cyc_prio_aes256.txt
timeout 900 python profiles/cyc_prio.py 16 | tee $O/cyc_prio_aes128.txt
