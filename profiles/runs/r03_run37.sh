#!/bin/bash
# round 3, call 37: cold and warm rates by message size; cfg2 with its new defaults
O=$PWD/gpurun_out/r03_run37; mkdir -p $O
timeout 300 python profiles/warm_rate.py 32 | tee $O/warm_rate_aes256.txt
timeout 300 python profiles/warm_rate.py 16 | tee $O/warm_rate_aes128.txt
timeout 300 python bench.py --config cfg2 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err; cut -c1-330 $O/bench_cfg2.json
