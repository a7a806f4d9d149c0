#!/bin/bash
# round 3, call 38: does a workgroup -> item group permutation take the XCD imbalance out of the cyclic launch?
O=$PWD/gpurun_out/r03_run38; mkdir -p $O
timeout 900 python profiles/cyc_perm.py 32 16384 | tee $O/cyc_perm_aes256.txt
timeout 900 python profiles/cyc_perm.py 16 4096 | tee $O/cyc_perm_aes128.txt
AESGCM_CYC_PERM=1 AESGCM_BODY_CYC=1048576:1125899906842624 timeout 300 python profiles/wgtrace2.py 32 4096 3 | tee $O/wg_cyc_perm_4g_aes256.txt
