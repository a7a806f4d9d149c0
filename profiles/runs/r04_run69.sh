#!/bin/bash
# round 4, call 69: addendum to the final collection (same library): counters for the two shapes that are new since -- k_batch3<.., 6> on 4096 x 1 MiB packets
# with a key each, k_pktl's ILP form on 131072 x 1 KiB packets under one key
O=$PWD/gpurun_out/r04_run69; mkdir -p $O
export GIT_HEAD=$(cat .git_head 2>/dev/null)
sha256sum aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so > $O/so_sha256.txt
bash profiles/collect.sh batchw_1m 'k_batch3' profiles/pkt_bench.py batch --len 1048576 --n 4096 --key-bits 256 --steps 5 > $O/collect_batchw_1m.txt 2>&1
bash profiles/collect.sh pktl_ilp_1k 'k_pktl' profiles/pkt_bench.py pkt --len 1024 --n 131072 --key-bits 256 --steps 7 > $O/collect_pktl_ilp_1k.txt 2>&1
bash profiles/collect.sh pktw_1m 'k_pktg' profiles/pkt_bench.py pkt --len 1048576 --n 4096 --key-bits 256 --steps 5 > $O/collect_pktw_1m.txt 2>&1
for t in batchw_1m pktl_ilp_1k pktw_1m; do
  mkdir -p $O/prof_$t; cp gpurun_out/prof_$t/summary*.txt gpurun_out/prof_$t/pmc_$t.json gpurun_out/prof_$t/stats_run.json $O/prof_$t/ 2>/dev/null
  find gpurun_out/prof_$t/stats -name "*kernel_stats.csv" -exec cp {} $O/prof_$t/kernel_stats.csv \;
  echo "== $t"; grep -E "hot_kernel|hot_avg_ns|hbm_bytes_per_launch|lds_busy_frac" gpurun_out/prof_$t/summary.txt | head -6
  rm -rf gpurun_out/prof_$t
done
