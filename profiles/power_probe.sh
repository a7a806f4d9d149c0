#!/bin/bash
# sample rocm-smi power/clock while the bench runs (GPU box)
(python bench.py --steps 600 --warmup 2 --no-cpu-baseline > gpurun_out/power_bench.json 2>gpurun_out/power_bench.err) &
BP=$!
sleep 6
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)" | head -8
  echo "--"
  sleep 0.4
done
wait $BP
cat gpurun_out/power_bench.json | cut -c1-400
