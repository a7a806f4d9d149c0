#!/bin/bash
# round 4, call 17: k_pktl's stores.  The final collection (call 16) showed WRITE_SIZE 3.2 x the ciphertext for k_pktl<14, 0>: the build it ran stores every
# block as soon as it is encrypted (16 bytes at a time, a fifth of a millisecond between a lane's first and last block of a line).  Same box: that build
# (store16), the four stores of a half line back to back (store64: the build measured in call 15), and 768-lane workgroups (no AES-256 scratch) with 64- and
# 128-byte store groups.
O=$PWD/gpurun_out/r04_run17; mkdir -p $O
sha256sum experiments/*.so > $O/so_sha256.txt
E=$PWD/experiments
for KB in 256 128; do
  echo "== AES-$KB"
  PKT_AB_LENS="1024 4096 256" PKT_AB_KEYBITS=$KB bash profiles/pkt_ab.sh $O/aes$KB pktl k_pktl $E/lib_pktl_store16.so $E/lib_pktl_store64.so $E/lib_pktl_768_store64.so $E/lib_pktl_768_store128.so $E/lib_pktl_store16.so $E/lib_pktl_store64.so $E/lib_pktl_768_store64.so $E/lib_pktl_768_store128.so 2>&1
done | tee $O/pktl_store_ab.txt
