mkdir -p gpurun_out/r05
for cfg in "262144 8448 0" "262144 8464 0" "262144 9000 13" "262144 12544 0" "262144 13000 13" "131072 16656 0" "131072 17200 13" "131072 20000 0" "65536 33000 13" "32768 66000 13" "16384 2047 0" "16384 3000 0" "16384 7000 13" "8192 2500 0" "4096 5000 13"; do set -- $cfg
  for k in rows norows; do
    echo "n=$1 len=$2 aad=$3 $k $(timeout 100 python profiles/pkt_bench.py $k --n $1 --len $2 --aad $3 --key-bits 256 --steps 9 | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["gib_per_s_queued"], d["gib_per_s"], d["shape"])')"
  done
  echo "n=$1 len=$2 aad=$3 rule $(timeout 100 python profiles/pkt_bench.py pkt --n $1 --len $2 --aad $3 --key-bits 256 --steps 9 | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["gib_per_s_queued"], d["gib_per_s"], d["shape"])')"
done > gpurun_out/r05/rows_ragged_many.txt 2>&1
cat gpurun_out/r05/rows_ragged_many.txt
