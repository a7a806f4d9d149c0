# A/B on one box: the row path with write-through stores (the library) and with plain stores (profiles/lab/libwt0.so: -DAESGCM_BODY_WT=0)
mkdir -p gpurun_out/r05
for cfg in "1048576 4096" "524288 8192" "262144 16384" "65536 65536" "4096 1048576" "4096 65536" "16384 4096"; do set -- $cfg
  for lib in old wt0; do
    L=$PWD/profiles/lab/lib$lib.so
    echo "n=$1 len=$2 $lib $(AESGCM_LIB=$L timeout 100 python profiles/pkt_bench.py pkt --opt rows_min=2048 --n $1 --len $2 --key-bits 256 --steps 15 | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["gib_per_s_queued"], d["gib_per_s"], d["shape"])')"
  done
done > gpurun_out/r05/rows_wt_ab.txt 2>&1
cat gpurun_out/r05/rows_wt_ab.txt
