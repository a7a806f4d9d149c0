#!/bin/bash
# round 4, call 5: GPU tests of what changed (batch shapes, queued launches / last_tag through the host slot, self-launch with child cleanup, multi-device tests skipping here),
# then the sustained rate of mid-size messages with K = 1, 2, 4 in flight
O=$PWD/gpurun_out/r04_run5; mkdir -p $O
sha256sum aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so > $O/so_sha256.txt
timeout 2400 python -m pytest tests/test_gpu_inflight.py tests/test_gpu_cyclic.py tests/test_gpu_batch.py tests/test_gpu_selflaunch.py tests/test_gpu_multiproc.py tests/test_gpu_multidevice.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -6 $O/pytest.txt
bash profiles/inflight_sweep.sh $O 1 4 16 64 256 2>&1 | tee $O/inflight_sweep.txt
INFLIGHT_KS="1 2" INFLIGHT_ARGS="--key-bits 128" bash profiles/inflight_sweep.sh $O/aes128 16 64 2>&1 | tee $O/inflight_sweep_aes128.txt
