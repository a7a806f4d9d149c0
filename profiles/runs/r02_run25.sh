#!/bin/bash
# round 2, call 25: after the chunking-rule change: GPU suite, split threshold, sizes
O=gpurun_out/r02_run25; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
timeout 600 python profiles/split_threshold.py > $O/split_threshold.txt 2>&1; cat $O/split_threshold.txt
timeout 300 python profiles/size_sweep.py > $O/size_sweep.txt 2>&1; grep "AES-256" $O/size_sweep.txt
timeout 300 python profiles/latency.py 300 > $O/latency.txt 2>&1; tail -12 $O/latency.txt
