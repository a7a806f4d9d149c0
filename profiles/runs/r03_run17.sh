#!/bin/bash
# round 3, call 17: full GPU suite with cyclic rows as the default for 4..384 MiB; what AAD / ragged ends cost at mid sizes; threshold check 256..512 MiB
O=$PWD/gpurun_out/r03_run17; mkdir -p $O
timeout 900 python profiles/general_shape.py | tee $O/general_shape.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest.txt
