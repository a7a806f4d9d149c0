# rows against the packet kernels around the routing rule, one box (after E_K(J0) moved to the closing launch): sizes that end on a row and sizes that do not
mkdir -p gpurun_out/r05
for cfg in "1048576 4096" "1048576 4112" "524288 8192" "524288 8208" "262144 16384" "262144 16400" "131072 32768" "131072 32784" \
           "65536 4096" "32768 8192" "32768 8208" "16384 16384" "16384 16400" "8192 8192" "4096 8192" "4096 16384" "4096 16400" "1024 8192" "1024 16384" "262144 2048" "1048576 2048" "262144 1024"; do set -- $cfg
  for k in rows norows; do
    echo "n=$1 len=$2 $k $(timeout 100 python profiles/pkt_bench.py $k --n $1 --len $2 --key-bits 256 --steps 9 | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["gib_per_s_queued"], d["gib_per_s"], d["shape"])')"
  done
done > gpurun_out/r05/rows_min_sweep2.txt 2>&1
cat gpurun_out/r05/rows_min_sweep2.txt
