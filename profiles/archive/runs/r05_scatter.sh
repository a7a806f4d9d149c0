# the same messages as fixed-size records of one buffer (aesgcm_packets_crypt_dev), through offset arrays, and wherever they live (aesgcm_messages_crypt_dev): one box
mkdir -p gpurun_out/r05
for cfg in "4096 1048576 0" "65536 65536 0" "262144 16384 13" "524288 8192 0" "4096 65536 0" "256 1048576 0"; do set -- $cfg
  for form in "" "--var" "--scatter"; do
    echo "n=$1 len=$2 aad=$3 form=${form:-fixed} $(timeout 100 python profiles/pkt_bench.py pkt $form --n $1 --len $2 --aad $3 --key-bits 256 --steps 15 | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["gib_per_s_queued"], d["gib_per_s"])')"
  done
done > gpurun_out/r05/rows_scatter.txt 2>&1
cat gpurun_out/r05/rows_scatter.txt
