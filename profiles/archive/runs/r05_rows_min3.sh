mkdir -p gpurun_out/r05
for cfg in "16384 4096" "4096 4096" "1024 4096" "16384 2048" "4096 2048" "1024 2048" "699050 6144" "65536 6144" "8192 6144" "32768 4096" "4096 1024" "131072 4096"; do set -- $cfg
  for k in rows norows; do
    echo "n=$1 len=$2 $k $(timeout 100 python profiles/pkt_bench.py $k --n $1 --len $2 --key-bits 256 --steps 9 | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["gib_per_s_queued"], d["gib_per_s"], d["shape"])')"
  done
done > gpurun_out/r05/rows_min_sweep3.txt 2>&1
cat gpurun_out/r05/rows_min_sweep3.txt
