#!/bin/bash
# round 4, call 28: variable-length batches with a key per packet (k_batch3) taken by falling length class: parity, then counts x order on one box
O=$PWD/gpurun_out/r04_run28; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_batch.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -3 $O/pytest.txt
timeout 900 python3 profiles/batch_mixed.py 16 2>&1 | tee $O/batch_mixed_aes128.txt
timeout 900 python3 profiles/batch_mixed.py 32 2>&1 | tee $O/batch_mixed_aes256.txt
