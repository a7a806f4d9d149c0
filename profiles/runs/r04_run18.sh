#!/bin/bash
# round 4, call 18: k_pktl at 768 lanes per workgroup (3 waves per SIMD: no scratch in any instance), the eight stores of a line back to back, and the
# whole-line fetch for decrypt too -- parity first (the packet and batch tests), then against the 1024-lane build with 64-byte store groups (call 15's), same
# box, encrypt and decrypt.
O=$PWD/gpurun_out/r04_run18; mkdir -p $O
sha256sum experiments/*.so aes-gcm-128-192-256-bits_amd/*.so > $O/so_sha256.txt
timeout 900 python3 -m pytest tests/test_gpu_batch.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -3 $O/pytest.txt
E=$PWD/experiments
for KB in 256 128; do
  for DIR in "" "--dec"; do
    echo "== AES-$KB $DIR"
    PKT_AB_ARGS="$DIR" PKT_AB_LENS="1024 4096 256" PKT_AB_KEYBITS=$KB bash profiles/pkt_ab.sh $O/aes$KB$DIR pktl k_pktl $E/lib_pktl_store64.so $E/lib_pktl_new.so $E/lib_pktl_store64.so $E/lib_pktl_new.so 2>&1
  done
done | tee $O/pktl_768_ab.txt
