#!/bin/bash
# round 4, call 78: tests only (library of call 70): the randomized differential tests, seeds 21 .. 32 at five times the iteration count, packet counts up to
# 140 000 in the draw (tests/test_gpu_fuzz.py, AESGCM_FUZZ_SEED / AESGCM_FUZZ_SCALE)
O=$PWD/gpurun_out/r04_run78; mkdir -p $O
sha256sum aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so > $O/so_sha256.txt
for k in 21 22 23 24 25 26 27 28 29 30 31 32; do
  AESGCM_FUZZ_SEED=$k AESGCM_FUZZ_SCALE=5 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "fuzz" > $O/fuzz_seed$k.txt 2>&1; echo "seed $k rc=$?" | tee -a $O/fuzz_seeds.txt; tail -3 $O/fuzz_seed$k.txt | tee -a $O/fuzz_seeds.txt
done
