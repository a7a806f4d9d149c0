#!/bin/bash
# round 3, call 20: the fused closing of the cyclic launch (one launch per message): parity under a watchdog, then A/B against k_fold + k_combine
O=$PWD/gpurun_out/r03_run20; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_cyclic.py -x -q -m gpu > $O/pytest_cyc.txt 2>&1; echo "pytest cyc rc=$?"
tail -15 $O/pytest_cyc.txt
timeout 600 python profiles/cyc_sweep.py 32 | tee $O/cyc_sweep_aes256.txt
timeout 300 python profiles/general_shape.py | tee $O/general_shape.txt
timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_model.py tests/test_gpu_pinned.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest.txt
