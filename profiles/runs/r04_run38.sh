#!/bin/bash
# round 4, call 38: the half shape by the library's own rule (another context has a message under way): parity of the cyclic / in-flight / soak paths, then
# the in-flight sweep with the rule against always / never, and the kernel names of a default --inflight 3 run
O=$PWD/gpurun_out/r04_run38; mkdir -p $O
timeout 1800 python3 -m pytest tests/test_gpu_cyclic.py tests/test_gpu_inflight.py tests/test_gpu_soak.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -4 $O/pytest.txt
timeout 600 ./examples/early_read 0.3 > $O/early_read.txt 2>&1; tail -2 $O/early_read.txt
for V in "rule:" "always:--half 1" "never:--half 0"; do
  N=${V%%:*}; A=${V#*:}
  echo "== $N ($A)"
  INFLIGHT_KS="1 2 3" INFLIGHT_ARGS="$A" bash profiles/inflight_sweep.sh $O/$N 4 16 64
done 2>&1 | tee $O/inflight_half_rule.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt38 && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt38 -- python3 $OLDPWD/bench.py --gib-per-gpu 0.015625 --inflight 3 --steps 200 --warmup 50 --no-cpu-baseline > /dev/null 2> $O/kt.err
find /tmp/kt38 -name "*kernel_stats.csv" -exec head -6 {} \; | cut -c1-160 | tee $O/kernel_names_inflight3_default.txt
