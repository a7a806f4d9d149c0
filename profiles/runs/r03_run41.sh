#!/bin/bash
# round 3, call 41: soak of the in-launch closings
O=$PWD/gpurun_out/r03_run41; mkdir -p $O
timeout 1500 python profiles/cyc_soak.py 30000 | tee $O/cyc_soak.txt
