# A/B on one box: the library before (profiles/lab/libold.so) and after the piece-end change (16 table loads in flight, DPP wave reduction)
mkdir -p gpurun_out/r05
for cfg in "1048576 4096" "524288 8192" "262144 16384" "65536 65536" "4096 1048576" "4096 65536" "16384 4096"; do set -- $cfg
  for lib in old new; do
    if [ $lib = old ]; then L=$PWD/profiles/lab/libold.so; else L=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so; fi
    echo "n=$1 len=$2 $lib $(AESGCM_LIB=$L timeout 100 python profiles/pkt_bench.py pkt --opt rows_min=2048 --n $1 --len $2 --key-bits 256 --steps 15 | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["gib_per_s_queued"], d["gib_per_s"], d["shape"])')"
  done
done > gpurun_out/r05/rows_piece_end_ab.txt 2>&1
cat gpurun_out/r05/rows_piece_end_ab.txt
