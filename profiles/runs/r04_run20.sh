#!/bin/bash
# round 4, call 20: the pair multiply's 32 table reads as one software-pipelined stream (shoup2_two_halves_dr) against the two rolled half loops; same box
O=$PWD/gpurun_out/r04_run20; mkdir -p $O
sha256sum experiments/*.so > $O/so_sha256.txt
E=$PWD/experiments
AESGCM_LIB=$E/lib_b3_pipe_dbg.so AESGCM_LIB_DEBUG=$E/lib_b3_pipe_dbg.so timeout 900 python3 -m pytest tests/test_gpu_batch.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -3 $O/pytest.txt
BATCH_AB_LGS=3 bash profiles/batch_ab.sh $O/aes128 $E/lib_b3_perm_dbg.so $E/lib_b3_pipe_dbg.so $E/lib_b3_perm_dbg.so $E/lib_b3_pipe_dbg.so 2>&1 | tee $O/batch_ab_aes128.txt
BATCH_AB_LGS=3 BATCH_AB_ARGS="--key-bits 256" bash profiles/batch_ab.sh $O/aes256 $E/lib_b3_perm_dbg.so $E/lib_b3_pipe_dbg.so 2>&1 | tee $O/batch_ab_aes256.txt
