mkdir -p gpurun_out/r05
for cfg in "262144 16384 0" "262144 16384 13" "262144 16400 13" "65536 65536 0" "65536 65536 13" "65536 65549 13" "4096 1048576 13" "4096 1048589 13"; do set -- $cfg
  for k in rows norows; do
    echo "n=$1 len=$2 aad=$3 $k $(timeout 100 python profiles/pkt_bench.py $k --n $1 --len $2 --aad $3 --key-bits 256 --steps 9 | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["gib_per_s_queued"], d["gib_per_s"], d["shape"])')"
  done
done > gpurun_out/r05/rows_aad_cost.txt 2>&1
cat gpurun_out/r05/rows_aad_cost.txt
