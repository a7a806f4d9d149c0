#!/bin/bash
# round 4, call 15: k_pktl with four T-tables in LDS (no rotates in rounds 2 .. NR-1) against the two-table round, with and without the whole-line fetch; same box
O=$PWD/gpurun_out/r04_run15; mkdir -p $O
sha256sum experiments/*.so > $O/so_sha256.txt
for KB in 256 128; do
  echo "== AES-$KB"
  PKT_AB_LENS="1024 4096 256" PKT_AB_KEYBITS=$KB bash profiles/pkt_ab.sh $O/aes$KB pktl k_pktl $PWD/experiments/lib_pktl_t2.so $PWD/experiments/lib_pktl_t4.so $PWD/experiments/lib_pktl_t4_noline.so $PWD/experiments/lib_pktl_t2.so $PWD/experiments/lib_pktl_t4.so 2>&1
done | tee $O/pktl_t4_ab.txt
