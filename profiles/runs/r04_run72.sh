#!/bin/bash
# round 4, call 72: tests only (library of call 70): the randomized differential tests with eight more seeds (tests/test_gpu_fuzz.py, AESGCM_FUZZ_SEED),
# the in-flight and batch tests once more on another box
O=$PWD/gpurun_out/r04_run72; mkdir -p $O
sha256sum aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so > $O/so_sha256.txt
for k in 1 2 3 4 5 6 7 8; do
  AESGCM_FUZZ_SEED=$k timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "fuzz" > $O/fuzz_seed$k.txt 2>&1; echo "seed $k rc=$?" | tee -a $O/fuzz_seeds.txt; tail -3 $O/fuzz_seed$k.txt | tee -a $O/fuzz_seeds.txt
done
