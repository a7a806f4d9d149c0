#!/bin/bash
# round 4, call 33: the hand-over of launcher-started ranks to the single-process path when no RCCL communicator forms (one-GPU box: up to the child's refusal)
O=$PWD/gpurun_out/r04_run33; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_multiproc.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -30 $O/pytest.txt
