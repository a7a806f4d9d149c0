#!/bin/bash
# round 4, call 3: same-box A/B of the three k_batch3 builds (round-3 form / PERM + DR / PERM + DR + PAIR, first k group of the multiplies peeled), AES-128 and AES-256, 8 and 16 lanes
O=$PWD/gpurun_out/r04_run3; mkdir -p $O
sha256sum aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so experiments/*.so > $O/so_sha256.txt
timeout 1500 python -m pytest tests/test_gpu_batch.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -4 $O/pytest.txt
bash profiles/batch_ab.sh $O $PWD/experiments/lib_base2.so $PWD/experiments/lib_nopair2.so $PWD/experiments/lib_pair2.so 2>&1 | tee $O/batch_ab.txt
BATCH_AB_LGS=3 bash profiles/batch_ab.sh $O/again $PWD/experiments/lib_nopair2.so $PWD/experiments/lib_pair2.so $PWD/experiments/lib_base2.so 2>&1 | tee $O/batch_ab_again.txt
BATCH_AB_LGS=3 BATCH_AB_ARGS="--key-bits 256" bash profiles/batch_ab.sh $O/aes256 $PWD/experiments/lib_base2.so $PWD/experiments/lib_nopair2.so $PWD/experiments/lib_pair2.so 2>&1 | tee $O/batch_ab_aes256.txt
BATCH_AB_LGS=3 BATCH_AB_ARGS="--decrypt" bash profiles/batch_ab.sh $O/dec $PWD/experiments/lib_base2.so $PWD/experiments/lib_nopair2.so $PWD/experiments/lib_pair2.so 2>&1 | tee $O/batch_ab_dec.txt
