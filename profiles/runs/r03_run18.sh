#!/bin/bash
# round 3, call 18: cyclic rows with the whole range in one launch (AAD and head blocks as front rows, partial last row as item 4096): parity, then shapes
O=$PWD/gpurun_out/r03_run18; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_cyclic.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?"
tail -15 $O/pytest.txt
timeout 900 python profiles/general_shape.py | tee $O/general_shape.txt
