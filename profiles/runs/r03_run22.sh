#!/bin/bash
# round 3, call 22: fused cyclic launch from 64 KiB with the end-of-launch wait behind the tag: latency sweep, whole suite
O=$PWD/gpurun_out/r03_run22; mkdir -p $O
timeout 600 python profiles/cyc_small.py | tee $O/cyc_small.txt
timeout 300 python profiles/general_shape.py | tee $O/general_shape.txt
timeout 2700 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest.txt
