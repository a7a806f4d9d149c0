#!/bin/bash
# round 3, call 29: the scratch-free closing: cyclic tests (early read, stress), latency, counters of the 1 MiB and 64 MiB launches
O=$PWD/gpurun_out/r03_run29; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cyclic.py -x -q -m gpu 2>&1 | tail -4
./examples/early_read 100 | tee $O/early_read.txt
timeout 600 python profiles/cyc_end.py | tee $O/cyc_end.txt
timeout 600 python profiles/cyc_small.py | tee $O/cyc_small.txt
bash profiles/collect.sh cyc_1m 'k_body<14, 0, true>' profiles/latency_one.py 1048576 200 > $O/collect_cyc_1m.txt 2>&1
grep -E "hot_avg_ns|hbm_bytes_per_launch|lds_busy_frac|WRITE_SIZE |FETCH_SIZE |SQ_INSTS_VMEM_WR" gpurun_out/prof_cyc_1m/summary.txt | head
