#!/bin/bash
# round 2, call 32: k_batch2 Shoup tables as 8-byte halves in a bank-disjoint layout (ds_read_b64) against 16-byte entries (ds_read_b128): parity, then A/B
O=$PWD/gpurun_out/r02_run32; mkdir -p $O
REPO=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q -k "batch or packets or abi or kat" > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
for rep in 1 2; do for v in _b128 ""; do for shape in "--len 4096 --key-bits 128" "--len 4096 --key-bits 256" "--len 1024 --key-bits 128" "--len 256 --key-bits 128 --n 4194304" "--len 16384 --key-bits 128 --n 262144"; do
  echo -n "lib '$v' $shape: "; AESGCM_LIB=$REPO/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python profiles/pkt_bench.py batch $shape --steps 7 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_median'], d['ms_best'], d['gib_per_s'])"
done; done; done
export AESGCM_LIB=$REPO/aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d $O/pmc -- python3 $REPO/profiles/pkt_bench.py batch --steps 3 > $O/pmc.log 2>&1
f=$(find $O/pmc -name "*counter_collection.csv" | head -1); python3 - $f <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_batch2" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(k, sum(v)/len(v))
PY
