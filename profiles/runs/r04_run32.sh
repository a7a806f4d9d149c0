#!/bin/bash
# round 4, call 32: the order slots with their events (reuse across streams): packet and batch parity, fuzz, stress
O=$PWD/gpurun_out/r04_run32; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_batch.py tests/test_gpu_fuzz.py tests/test_gpu_stress.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -5 $O/pytest.txt
