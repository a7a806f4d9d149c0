#!/bin/bash
# round 4, call 34: stdout is the JSON line's alone (RCCL's banner and anything else a library prints go to stderr): the multi-process tests, and the plain bench
# lines with stdout and stderr kept apart
O=$PWD/gpurun_out/r04_run34; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_multiproc.py tests/test_gpu_inflight.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -5 $O/pytest.txt
timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/b1.out 2> $O/b1.err; echo "rc=$? stdout lines: $(wc -l < $O/b1.out)"; cut -c1-300 $O/b1.out
timeout 600 python3 bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline > $O/b5.out 2> $O/b5.err; echo "rc=$? stdout lines: $(wc -l < $O/b5.out)"; cut -c1-200 $O/b5.out
timeout 600 python3 bench.py --gib-per-gpu 0.0625 --inflight 3 --no-cpu-baseline > $O/bi.out 2> $O/bi.err; echo "rc=$? stdout lines: $(wc -l < $O/bi.out)"; cut -c1-200 $O/bi.out
timeout 900 python3 bench.py --gpus 2 --one-device --backend file --allow-file-exchange --gib-per-gpu 0.25 --steps 2 --warmup 1 --no-cpu-baseline > $O/b2.out 2> $O/b2.err; echo "rc=$? stdout lines: $(wc -l < $O/b2.out)"; cut -c1-200 $O/b2.out
python3 -c "
import json,sys
for f in ('b1','b5','bi','b2'):
    t=open('$O/'+f+'.out').read().strip().splitlines()
    d=json.loads(t[-1]); print(f, len(t), d['value'], d['roofline'].get('traffic'))
"
