#!/bin/bash
# round 4, call 6: the whole GPU suite on the refactored library (no environment knobs, debug build for forced shapes, k_batch2 gone), with the new soak /
# early-read / large tests; smoke; default bench line
O=$PWD/gpurun_out/r04_run6; mkdir -p $O
sha256sum aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so aes-gcm-128-192-256-bits_amd/libaesgcm_hip_dbg.so > $O/so_sha256.txt
timeout 5400 python -m pytest tests -x -q -m gpu --durations=15 > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -25 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; cat $O/smoke.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-400 $O/bench_default.json
timeout 600 python bench.py --config cfg5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err; cut -c1-300 $O/bench_cfg5.json
