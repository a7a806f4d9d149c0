#!/bin/bash
O=gpurun_out/r02_run6; mkdir -p $O
timeout 300 python profiles/latency.py 300 > $O/latency.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python3 $GRAFT_REPO_ROOT/profiles/latency.py 100 > $GRAFT_REPO_ROOT/$O/latency_prof.txt 2>&1
cd $GRAFT_REPO_ROOT
cat $O/latency.txt
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cat $f | cut -c1-200
