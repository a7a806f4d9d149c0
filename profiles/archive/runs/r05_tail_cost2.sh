mkdir -p gpurun_out/r05
for e in 0 1 2 4 7; do
  if [ $e = 0 ]; then export AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so; else export AESGCM_LIB=$PWD/profiles/lab/libexp_$e.so; fi
  for L in 16 1024 4096; do
    echo "exp=$e len=$L $(python profiles/pkt_bench.py pkt --opt rows_min=16 --n 262144 --len $L --key-bits 256 --steps 9 | cut -c1-200)"
  done
done > gpurun_out/r05/rows_tail_cost2.txt 2>&1
cat gpurun_out/r05/rows_tail_cost2.txt
