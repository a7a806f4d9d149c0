#!/bin/bash
# round 4, call 13: the half shape under the early read (about 1200 more calls) and in the soak (800 queued messages over two contexts)
O=$PWD/gpurun_out/r04_run13; mkdir -p $O
sha256sum aes-gcm-128-192-256-bits_amd/*.so > $O/so_sha256.txt
timeout 2400 python -m pytest tests/test_gpu_cyclic.py tests/test_gpu_soak.py -x -q -m gpu --durations=6 > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -14 $O/pytest.txt
timeout 600 ./examples/early_read 1 > $O/early_read.txt 2>&1; tail -30 $O/early_read.txt
