#!/bin/bash
# round 2, call 15: gather microbenchmark (vector-memory path beside the LDS)
set -x
mkdir -p gpurun_out/r02f
cd profiles/microbench
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/gather gather.hip || exit 1
timeout 120 /tmp/gather > ../../gpurun_out/r02f/gather.txt 2>&1
cat ../../gpurun_out/r02f/gather.txt
