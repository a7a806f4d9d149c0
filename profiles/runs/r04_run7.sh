#!/bin/bash
# round 4, call 7: the two test files call 6 did not reach (soak with the corrected sizes, the stress child on the debug build), then k_pktl with the whole
# 128-byte line fetched at once (AESGCM_PKTL_LINE) against the 64-byte steps, at 1024 and 768 lanes per workgroup: time and HBM traffic, same box
O=$PWD/gpurun_out/r04_run7; mkdir -p $O
sha256sum aes-gcm-128-192-256-bits_amd/*.so experiments/*.so > $O/so_sha256.txt
timeout 2400 python -m pytest tests/test_gpu_soak.py tests/test_gpu_stress.py -x -q -m gpu --durations=5 > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -12 $O/pytest.txt
PKT_AB_LENS="1024 256 4096" bash profiles/pkt_ab.sh $O pktl k_pktl $PWD/experiments/lib_pl0.so $PWD/experiments/lib_pl1.so $PWD/experiments/lib_pl0w768.so $PWD/experiments/lib_pl1w768.so 2>&1 | tee $O/pktl_ab.txt
PKT_AB_LENS="1024" PKT_AB_KEYBITS=128 bash profiles/pkt_ab.sh $O/aes128 pktl k_pktl $PWD/experiments/lib_pl0.so $PWD/experiments/lib_pl1.so $PWD/experiments/lib_pl1w768.so 2>&1 | tee $O/pktl_ab_aes128.txt
PKT_AB_LENS="1024" bash profiles/pkt_ab.sh $O/g4 pktg4 k_pktg $PWD/experiments/lib_pl0.so 2>&1 | tee $O/pktg4.txt
