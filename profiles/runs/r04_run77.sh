#!/bin/bash
# round 4, call 77: the GPU suite and smoke once more after the test files learned their seeds (library of call 70), the default bench line on this box
O=$PWD/gpurun_out/r04_run77; mkdir -p $O
sha256sum aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so > $O/so_sha256.txt
timeout 3000 python -m pytest tests -x -q -m gpu --durations=5 > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -9 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-400 $O/bench_default.json
