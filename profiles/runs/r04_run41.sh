#!/bin/bash
# round 4, call 41: k_batch3's loop for plain records (one size, whole wave-iterations, no AAD, aligned: cfg5) without the general loop's per-iteration tests;
# parity of the batch paths, then cfg5 A/B with the issue counters, same box
O=$PWD/gpurun_out/r04_run41; mkdir -p $O
sha256sum experiments/*.so > $O/so_sha256.txt
timeout 900 python3 -m pytest tests/test_gpu_batch.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -3 $O/pytest.txt
E=$PWD/experiments
bash profiles/batch_ab.sh $O/aes128 $E/lib_b3_before_dbg.so $E/lib_b3_plain_dbg.so $E/lib_b3_before_dbg.so $E/lib_b3_plain_dbg.so 2>&1 | tee $O/batch_ab_aes128.txt
BATCH_AB_LGS=3 BATCH_AB_ARGS="--key-bits 256" bash profiles/batch_ab.sh $O/aes256 $E/lib_b3_before_dbg.so $E/lib_b3_plain_dbg.so 2>&1 | tee $O/batch_ab_aes256.txt
BATCH_AB_LGS=3 BATCH_AB_ARGS="--decrypt" bash profiles/batch_ab.sh $O/dec $E/lib_b3_before_dbg.so $E/lib_b3_plain_dbg.so 2>&1 | tee $O/batch_ab_dec.txt
