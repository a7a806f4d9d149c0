#!/bin/bash
O=gpurun_out/r02_run7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q > $O/pytest_batch.log 2>&1; tail -8 $O/pytest_batch.log
for lg in 6 4; do for kb in 128 256; do AESGCM_BATCH_LG=$lg timeout 120 python profiles/pkt_bench.py batch --key-bits $kb --steps 5; done; done 2>&1 | tee $O/batch_shapes.txt
for lg in 6 4; do AESGCM_BATCH_LG=$lg timeout 120 python profiles/pkt_bench.py batch --len 1024 --steps 5; AESGCM_BATCH_LG=$lg timeout 120 python profiles/pkt_bench.py batch --len 256 --n 4194304 --steps 5; AESGCM_BATCH_LG=$lg timeout 120 python profiles/pkt_bench.py batch --len 16384 --n 262144 --steps 5; done 2>&1 | tee -a $O/batch_shapes.txt
