mkdir -p gpurun_out/r05
for L in 16 1008 1024 2048 4096 65536 65552; do
  n=$((1<<18)); if [ $L -ge 65536 ]; then n=32768; fi
  echo "n=$n len=$L rows $(python profiles/pkt_bench.py rows --n $n --len $L --key-bits 256 --steps 9 | cut -c1-330)"
done > gpurun_out/r05/rows_tail_cost.txt 2>&1
cat gpurun_out/r05/rows_tail_cost.txt
