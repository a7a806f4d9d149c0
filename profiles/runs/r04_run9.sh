#!/bin/bash
# round 4, call 9: the half shape of the cyclic rows (k_bodyh, two workgroups per CU): parity, then the sustained rate of mid-size messages in flight with
# and without it on one box
O=$PWD/gpurun_out/r04_run9; mkdir -p $O
sha256sum aes-gcm-128-192-256-bits_amd/*.so > $O/so_sha256.txt
timeout 1800 python -m pytest tests/test_gpu_cyclic.py tests/test_gpu_inflight.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -6 $O/pytest.txt
echo "== full shape (cyc_half 0)"; INFLIGHT_KS="1 2" INFLIGHT_ARGS="--half 0" bash profiles/inflight_sweep.sh $O/full 1 4 16 64 256 2>&1 | tee $O/inflight_full.txt
echo "== half shape (cyc_half 1)"; INFLIGHT_KS="1 2 3 4" INFLIGHT_ARGS="--half 1" bash profiles/inflight_sweep.sh $O/half 1 4 16 64 256 2>&1 | tee $O/inflight_half.txt
echo "== AES-128"; INFLIGHT_KS="2" INFLIGHT_ARGS="--half 0 --key-bits 128" bash profiles/inflight_sweep.sh $O/full128 16 64 2>&1 | tee $O/inflight_full_aes128.txt
INFLIGHT_KS="2 4" INFLIGHT_ARGS="--half 1 --key-bits 128" bash profiles/inflight_sweep.sh $O/half128 16 64 2>&1 | tee $O/inflight_half_aes128.txt
