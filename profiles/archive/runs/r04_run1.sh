#!/bin/bash
# round 4, call 1: k_batch3 with packets on ds_read_b128 service groups (BATCH3_PERM) and the delayed-reduction table multiply (BATCH3_DR):
# parity of the batch tests on the new default build, then the four builds A/B on this box with the LDS counters beside the time
O=$PWD/gpurun_out/r04_run1; mkdir -p $O
sha256sum aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so experiments/*.so > $O/so_sha256.txt
timeout 1500 python -m pytest tests/test_gpu_batch.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -4 $O/pytest.txt
bash profiles/batch_ab.sh $O $PWD/experiments/lib_p0d0.so $PWD/experiments/lib_p1d0.so $PWD/experiments/lib_p0d1.so $PWD/experiments/lib_p1d1.so 2>&1 | tee $O/batch_ab.txt
