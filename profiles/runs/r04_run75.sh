#!/bin/bash
# round 4, call 75: tests only (library of call 70): examples/early_read at four times its call counts (16 000 calls: is the whole result in memory
# when the call returns with the tag?), the plain-C examples once more
O=$PWD/gpurun_out/r04_run75; mkdir -p $O
sha256sum aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so > $O/so_sha256.txt
timeout 1800 ./examples/early_read 4 > $O/early_read_x4.txt 2>&1; echo "early_read rc=$?" | tee -a $O/early_read_x4.txt; tail -4 $O/early_read_x4.txt
timeout 120 ./examples/kat > $O/kat.txt 2>&1; echo "kat rc=$?" | tee -a $O/kat.txt
timeout 300 ./examples/frames > $O/frames.txt 2>&1; echo "frames rc=$?" | tee -a $O/frames.txt; tail -3 $O/frames.txt
timeout 300 ./examples/mt_stream > $O/mt_stream.txt 2>&1; echo "mt_stream rc=$?" | tee -a $O/mt_stream.txt; tail -3 $O/mt_stream.txt
