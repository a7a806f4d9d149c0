#!/bin/bash
# round 4, call 76: tests only (library of call 70): examples/early_read at 200 times its call counts (about 800 000 calls)
O=$PWD/gpurun_out/r04_run76; mkdir -p $O
sha256sum aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so > $O/so_sha256.txt
( time timeout 2000 ./examples/early_read 200 ) > $O/early_read_x200.txt 2>&1; echo "early_read rc=$?" | tee -a $O/early_read_x200.txt; tail -8 $O/early_read_x200.txt
