#!/bin/bash
# round 3, call 11: ds_swizzle instead of ds_bpermute for the xor-lane exchanges (k_pktg, k_batch3): parity, then the shipped geometry (768 lanes at 16 / 64 lanes per packet,
# 1024 at 8 / 4) against 1024 lanes everywhere, and the read traffic of the 8-lane shape
O=$PWD/gpurun_out/r03_run11; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_batch.py tests/test_gpu_selflaunch.py tests/test_gpu_stress.py tests/test_replay.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -3 $O/pytest.txt
for rep in 1 2; do for v in "" _all1024; do for kind in pktg4 pktg8 pktg pktw; do for len in 1024 4096; do
  echo -n "lib$v $kind rep$rep " >> $O/ab.txt
  AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python profiles/pkt_bench.py $kind --len $len --key-bits 256 --steps 7 >> $O/ab.txt 2>> $O/ab.err
done; done; done; done
python - $O/ab.txt <<'PY'
import sys,json
for l in open(sys.argv[1]):
    a=l.split(" ",3); d=json.loads(a[3])
    print("%-14s %-6s %s len %5d  %7.1f GiB/s  %.3f ms" % (a[0], a[1], a[2], d["pkt_len"], d["gib_per_s"], d["ms_median"]))
PY
REPO=$PWD; cd /tmp; export TMPDIR=/tmp
for kind in pktg8 pktg4; do
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/tcc_$kind -- python3 $REPO/profiles/pkt_bench.py $kind --len 1024 --key-bits 256 --steps 3 > $O/tcc_$kind.json 2> $O/tcc_$kind.err
  python3 - $O/tcc_$kind "$kind" <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(float); disp=set()
for p in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "k_pktg" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
n=max(1,len(disp))
print(sys.argv[2], "read 128B-request bytes %.4g write 64B-request bytes %.4g (algorithmic 1.086e9 each way)" % (128*acc.get("TCC_EA0_RDREQ_128B_sum",0)/n, 64*acc.get("TCC_EA0_WRREQ_64B_sum",0)/n))
PY
  rm -rf $O/tcc_$kind
done
