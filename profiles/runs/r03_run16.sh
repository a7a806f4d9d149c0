#!/bin/bash
# round 3, call 16: k_body as cyclic rows (no dispenser, 4096 items whatever the size): parity, then the size sweep against the round-2 paths
O=$PWD/gpurun_out/r03_run16; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_large.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest.txt
timeout 600 python profiles/cyc_sweep.py 32 | tee $O/cyc_sweep_aes256.txt
timeout 600 python profiles/cyc_sweep.py 16 | tee $O/cyc_sweep_aes128.txt
