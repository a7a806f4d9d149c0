#!/bin/bash
O=gpurun_out/r02_run4; mkdir -p $O
( cd profiles/microbench && timeout 300 ./bs_ctr 4096 ) > $O/bs_ctr.txt 2>&1
timeout 300 python profiles/ks_time.py 4096 > $O/ks_time.txt 2>&1
cat $O/bs_ctr.txt $O/ks_time.txt
