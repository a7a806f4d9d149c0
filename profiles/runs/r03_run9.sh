#!/bin/bash
# round 3, call 9: k_batch3 with 768-lane workgroups (3 waves per SIMD, 160 - 168 VGPRs, no scratch at AES-128) against 1024 (128 VGPRs, 104 - 124 B of scratch); hybrid k_pktg parity
O=$PWD/gpurun_out/r03_run9; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_batch.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -3 $O/pytest.txt
for rep in 1 2 3; do for v in "" _b768; do
  AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python bench.py --config cfg5 --steps 10 --warmup 2 --no-cpu-baseline > $O/cfg5$v$rep.json 2> $O/cfg5$v$rep.err
  AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python bench.py --config cfg5 --key-bits 256 --steps 10 --warmup 2 --no-cpu-baseline > $O/cfg5_aes256$v$rep.json 2>> $O/cfg5$v$rep.err
  AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python bench.py --config cfg5 --pkt-len 1024 --steps 10 --warmup 2 --no-cpu-baseline > $O/cfg5_1k$v$rep.json 2>> $O/cfg5$v$rep.err
done; done
python - $O <<'PY'
import json,sys,glob,os
for p in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-26s %.1f GiB/s step %.3f ms kernel %.3f ms frac %.4f tag_ok %s" % (os.path.basename(p), d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], d["tag_ok"]))
    except Exception as e:
        print(p, "unreadable", e)
PY
REPO=$PWD; cd /tmp; export TMPDIR=/tmp
for v in "" _b768; do
  AESGCM_LIB=$REPO/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/tcc$v -- python3 $REPO/bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline > $O/tcc$v.json 2> $O/tcc$v.err
  python3 - $O/tcc$v "lib$v" <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(float); disp=set()
for p in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "k_batch3" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
n=max(1,len(disp))
print(sys.argv[2], {k: round(v/n) for k,v in acc.items()}, "read 128B-request bytes %.4g write 64B-request bytes %.4g (algorithmic 4.32e9 each way)" % (128*acc.get("TCC_EA0_RDREQ_128B_sum",0)/n, 64*acc.get("TCC_EA0_WRREQ_64B_sum",0)/n))
PY
  rm -rf $O/tcc$v
done
