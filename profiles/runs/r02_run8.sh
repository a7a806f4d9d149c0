#!/bin/bash
O=gpurun_out/r02_run8; mkdir -p $O
for lg in 4 5; do echo "LG=$lg"; for args in "--len 4096" "--len 4096 --key-bits 256" "--len 1024" "--len 256 --n 4194304" "--len 64 --n 4194304" "--len 16384 --n 262144" "--len 1500 --n 1048576"; do AESGCM_BATCH_LG=$lg timeout 120 python profiles/pkt_bench.py batch $args --steps 4 | cut -c1-150; done; done 2>&1 | tee $O/batch_lg.txt
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q 2>&1 | tail -3
