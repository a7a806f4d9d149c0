#!/bin/bash
# round 2, call 34: packet dispensers with a per-workgroup dry flag (and two alternating counters for k_pkt / k_pktl): parity, then A/B against the build before
O=$PWD/gpurun_out/r02_run34; mkdir -p $O
REPO=$PWD
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
for rep in 1 2; do for v in _old ""; do
  for shape in "pktl --len 1024 --key-bits 256" "pktl --len 256 --key-bits 256 --n 4194304" "pktl --len 64 --key-bits 128 --n 4194304" "pktw --len 1024 --key-bits 256" "pktw --len 16384 --key-bits 256 --n 65536" "batch --len 4096 --key-bits 128" "batch --len 1024 --key-bits 128"; do
  echo -n "lib '$v' $shape: "; AESGCM_LIB=$REPO/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python profiles/pkt_bench.py $shape --steps 9 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_median'], d['ms_best'], d['gib_per_s'])"
done; done; done
