#!/bin/bash
# round 2, call 35: two plain launches against a two-node hipGraph (small-message launch path)
mkdir -p gpurun_out/r02f
cd profiles/microbench && hipcc --offload-arch=gfx950 -O3 -w -o /tmp/graph_launch graph_launch.hip || exit 1
timeout 120 /tmp/graph_launch > ../../gpurun_out/r02f/graph_launch.txt 2>&1; cat ../../gpurun_out/r02f/graph_launch.txt
