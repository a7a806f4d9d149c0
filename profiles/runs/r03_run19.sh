#!/bin/bash
# round 3, call 19: the whole GPU suite on the cyclic-rows build, smoke, default bench
O=$PWD/gpurun_out/r03_run19; mkdir -p $O
timeout 2700 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?"
tail -4 $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
cut -c1-400 $O/bench_default.json
