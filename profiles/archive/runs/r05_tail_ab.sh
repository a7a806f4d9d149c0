# A/B on one box: long tails as units of their own (profiles/lab/libold.so) and taken along by the run that ends on the message's last whole row (the library)
mkdir -p gpurun_out/r05
for cfg in "262144 9000 13" "131072 16656 0" "131072 20000 0" "65536 33000 13" "32768 66000 13" "32768 131000 0" "4096 1048000 0" "65536 65536 0" "4096 1048576 0" "262144 16384 13" "524288 8192 0"; do set -- $cfg
  for lib in old new; do
    if [ $lib = old ]; then L=$PWD/profiles/lab/libold.so; else L=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so; fi
    echo "n=$1 len=$2 aad=$3 $lib $(AESGCM_LIB=$L timeout 100 python profiles/pkt_bench.py pkt --opt rows_min=2048 --var --n $1 --len $2 --aad $3 --key-bits 256 --steps 15 | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["gib_per_s_queued"], d["gib_per_s"], d["shape"])')"
  done
done > gpurun_out/r05/rows_tail_ab.txt 2>&1
cat gpurun_out/r05/rows_tail_ab.txt
