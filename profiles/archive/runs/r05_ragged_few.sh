mkdir -p gpurun_out/r05
for cfg in "16384 2047 0" "16384 3000 0" "4096 5000 13" "16384 8191 0" "16384 7000 13" "8192 2500 0" "1024 4000 20" "262144 9000 13" "524288 8200 0"; do set -- $cfg
  for k in rows norows; do
    echo "n=$1 len=$2 aad=$3 $k $(timeout 100 python profiles/pkt_bench.py $k --n $1 --len $2 --aad $3 --key-bits 256 --steps 9 | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["gib_per_s_queued"], d["gib_per_s"], d["shape"])')"
  done
done > gpurun_out/r05/rows_ragged_few.txt 2>&1
cat gpurun_out/r05/rows_ragged_few.txt
