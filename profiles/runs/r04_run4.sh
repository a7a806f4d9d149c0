#!/bin/bash
# round 4, call 4: same-box A/B, PERM + DR against PERM + DR + PAIR (8 lanes per packet), AES-128 / AES-256 / decrypt; base = round-3 form
O=$PWD/gpurun_out/r04_run4; mkdir -p $O
sha256sum aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so experiments/*.so > $O/so_sha256.txt
BATCH_AB_LGS=3 bash profiles/batch_ab.sh $O $PWD/experiments/lib_base2.so $PWD/experiments/lib_nopair2.so $PWD/experiments/lib_zpair2.so $PWD/experiments/lib_nopair2.so $PWD/experiments/lib_zpair2.so 2>&1 | tee $O/batch_ab.txt
BATCH_AB_LGS=3 BATCH_AB_ARGS="--key-bits 256" bash profiles/batch_ab.sh $O/aes256 $PWD/experiments/lib_nopair2.so $PWD/experiments/lib_zpair2.so 2>&1 | tee $O/batch_ab_aes256.txt
BATCH_AB_LGS=3 BATCH_AB_ARGS="--pkt-len 1024" bash profiles/batch_ab.sh $O/p1k $PWD/experiments/lib_base2.so $PWD/experiments/lib_nopair2.so $PWD/experiments/lib_zpair2.so 2>&1 | tee $O/batch_ab_1k.txt
BATCH_AB_LGS=3 BATCH_AB_ARGS="--pkt-len 256 --n-pkts 4194304" bash profiles/batch_ab.sh $O/p256 $PWD/experiments/lib_base2.so $PWD/experiments/lib_nopair2.so $PWD/experiments/lib_zpair2.so 2>&1 | tee $O/batch_ab_256.txt
