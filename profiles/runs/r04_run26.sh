#!/bin/bash
# round 4, call 26 (the sort with LDS atomics only): packets of mixed length taken by falling length class (k_len_hist / k_len_scan / k_len_scatter in front of the launch) against array order;
# parity first (every shape, both orders), then the sweep over counts, both orders, same box
O=$PWD/gpurun_out/r04_run26; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_batch.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -5 $O/pytest.txt
timeout 900 python3 profiles/packets_sweep.py 32 var 2>&1 | tee $O/packets_sweep_mixed_aes256_ordered.txt
