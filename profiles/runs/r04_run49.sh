#!/bin/bash
# round 4, call 49: k_pktl's ILP form (512-lane workgroups, four keystream chains side by side) for batches that do not fill the chip: parity, then the sweep
# with the lane kernel in both forms beside the group shapes
O=$PWD/gpurun_out/r04_run49; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_batch.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -3 $O/pytest.txt
timeout 900 python3 profiles/packets_sweep.py 32 2>&1 | tee $O/packets_sweep_aes256.txt
