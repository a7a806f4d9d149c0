#!/bin/bash
# round 2, call 31: rows per chunk for k_body at mid sizes (the cut forced with AESGCM_BODY_MIN=4096), against k_main alone
O=gpurun_out/r02_run31; mkdir -p $O
echo "k_body forced"; AESGCM_BODY_MIN=4096 timeout 600 python profiles/tw_sweep.py 64 128 192 256 384 512 1024 2048 > $O/tw_body.txt 2>&1; cat $O/tw_body.txt
echo "k_main only"; AESGCM_BODY_MIN=1152921504606846976 timeout 600 python profiles/tw_sweep.py 64 128 192 256 384 512 > $O/tw_main.txt 2>&1; cat $O/tw_main.txt
