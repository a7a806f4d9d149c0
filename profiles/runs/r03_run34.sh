#!/bin/bash
# round 3, call 34: where a launch's workgroups end -- cfg2's dealt kernel inside the loop, and cyclic rows at 1 and 4 GiB
O=$PWD/gpurun_out/r03_run34; mkdir -p $O
timeout 300 python profiles/wgtrace2.py 16 1024 | tee $O/wg_dealt_1g_aes128.txt
AESGCM_BODY_CYC=1048576:1125899906842624 timeout 300 python profiles/wgtrace2.py 16 1024 | tee $O/wg_cyc_1g_aes128.txt
AESGCM_BODY_CYC=1048576:1125899906842624 timeout 300 python profiles/wgtrace2.py 32 4096 3 | tee $O/wg_cyc_4g_aes256.txt
timeout 300 python profiles/wgtrace2.py 32 4096 3 | tee $O/wg_dealt_4g_aes256.txt
