mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_rows.py tests/test_gpu_wipe.py -x -q 2>&1 | tail -3 > gpurun_out/r05/rows_lds_pytest.txt
for L in 16 1024 4096 16384 32768 65536 1048576; do
  n=262144; if [ $L -ge 16384 ]; then n=$(( (1<<32) / L )); fi
  echo "len=$L n=$n $(python profiles/pkt_bench.py rows --n $n --len $L --key-bits 256 --steps 9 | cut -c1-200)"
done > gpurun_out/r05/rows_tail_cost3.txt 2>&1
cat gpurun_out/r05/rows_lds_pytest.txt gpurun_out/r05/rows_tail_cost3.txt
