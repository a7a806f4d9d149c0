#!/bin/bash
# round 3, call 31: where the cyclic launch with rotating priorities stops paying (384 .. 1280 MiB)
O=$PWD/gpurun_out/r03_run31; mkdir -p $O
timeout 900 python profiles/cyc_prio.py 32 384,512,640,768,896,1024,1280 0,2 | tee $O/cyc_prio_fine_aes256.txt
timeout 900 python profiles/cyc_prio.py 16 384,512,640,768,896,1024,1280 0,2 | tee $O/cyc_prio_fine_aes128.txt
timeout 900 python profiles/cyc_prio.py 32 384,512,640,768,896,1024,1280 0,2 | tee -a $O/cyc_prio_fine_aes256.txt
