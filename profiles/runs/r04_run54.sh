#!/bin/bash
# round 4, call 54: workgroups of a cyclic launch that have no row (whole messages below 4 MiB) arrive with nothing and go (no staging, no closing): parity of
# the message paths, the timeline, the in-flight sweep at small sizes with K up to 6, the waited latencies
O=$PWD/gpurun_out/r04_run54; mkdir -p $O
timeout 1800 python3 -m pytest tests/test_gpu_cyclic.py tests/test_gpu_inflight.py tests/test_gpu_soak.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -3 $O/pytest.txt
timeout 600 python3 profiles/cyc_timeline.py 32 0.0625 0.25 1 2 4 16 2>&1 | tee $O/cyc_timeline_aes256.txt
INFLIGHT_KS="1 2 3 6" bash profiles/inflight_sweep.sh $O/inflight 0.0625 0.25 1 4 16 2>&1 | tee $O/inflight_small.txt
timeout 300 ./examples/latency 500 > $O/latency_c.txt 2>&1; head -10 $O/latency_c.txt
