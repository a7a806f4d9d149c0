#!/bin/bash
O=gpurun_out/r02_run9; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_gpu.log 2>&1; tail -14 $O/pytest_gpu.log
timeout 300 python profiles/latency.py 300 > $O/latency.txt 2>&1; cat $O/latency.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python3 $GRAFT_REPO_ROOT/profiles/latency.py 100 > $GRAFT_REPO_ROOT/$O/latency_prof.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cat $f | cut -c1-160 | head -8
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline | cut -c1-300
