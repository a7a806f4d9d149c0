#!/bin/bash
# round 4, call 2: k_batch3 at 8 lanes per packet with every multiply split over a lane pair (BATCH3_PAIR) against the build of call 1 (PERM + DR)
O=$PWD/gpurun_out/r04_run2; mkdir -p $O
sha256sum aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so experiments/*.so > $O/so_sha256.txt
timeout 1500 python -m pytest tests/test_gpu_batch.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -4 $O/pytest.txt
BATCH_AB_LGS=3 bash profiles/batch_ab.sh $O $PWD/experiments/lib_p1d1.so $PWD/experiments/lib_pair.so 2>&1 | tee $O/batch_ab.txt
BATCH_AB_LGS=3 bash profiles/batch_ab.sh $O/again $PWD/experiments/lib_p1d1.so $PWD/experiments/lib_pair.so 2>&1 | tee $O/batch_ab_again.txt
BATCH_AB_LGS=3 BATCH_AB_ARGS="--key-bits 256" bash profiles/batch_ab.sh $O/aes256 $PWD/experiments/lib_p0d0.so $PWD/experiments/lib_p1d1.so $PWD/experiments/lib_pair.so 2>&1 | tee $O/batch_ab_aes256.txt
