#!/bin/bash
# round 4, call 79: tests only (library of call 70): the two message soaks of tests/test_gpu_soak.py (3000 cyclic messages with the in-launch closing; 800
# messages through two contexts in the half shape) with seeds 31 .. 60 (AESGCM_SOAK_SEED): the ordering argument of the in-launch closing is partly
# empirical, so it gets volume
O=$PWD/gpurun_out/r04_run79; mkdir -p $O
sha256sum aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so > $O/so_sha256.txt
for k in $(seq 31 60); do
  AESGCM_SOAK_SEED=$k timeout 600 python -m pytest tests/test_gpu_soak.py -x -q -m gpu -k "cyclic_messages or half_shape" > $O/soak_seed$k.txt 2>&1; echo "seed $k rc=$? $(tail -1 $O/soak_seed$k.txt)" | tee -a $O/soak_seeds.txt
done
