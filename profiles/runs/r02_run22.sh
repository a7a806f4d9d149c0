#!/bin/bash
# round 2, call 22: rows per chunk against message size on the build with the runtime fold group
O=gpurun_out/r02_run22; mkdir -p $O
timeout 600 python profiles/tw_sweep.py 1 2 4 8 16 32 64 100 127 > $O/tw_sweep.txt 2>&1; cat $O/tw_sweep.txt
