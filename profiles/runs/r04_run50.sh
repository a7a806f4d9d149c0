#!/bin/bash
# round 4, call 50: k_pktl's ILP form by the library's rule (auto column), the new 1 KiB mark of the shape rule; parity of the packet paths with both forms forced
O=$PWD/gpurun_out/r04_run50; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_batch.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -3 $O/pytest.txt
timeout 900 python3 profiles/packets_sweep.py 32 2>&1 | tee $O/packets_sweep_aes256.txt
timeout 900 python3 profiles/packets_sweep.py 16 2>&1 | tee $O/packets_sweep_aes128.txt
