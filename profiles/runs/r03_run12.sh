#!/bin/bash
# round 3, call 12: scratch-free k_pktg / k_batch3 at 1024 lanes (bookkeeping out of registers): parity, speed and traffic against the 768-lane build of the same sources
O=$PWD/gpurun_out/r03_run12; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_batch.py tests/test_gpu_stress.py tests/test_gpu_selflaunch.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -3 $O/pytest.txt
for rep in 1 2; do for v in "" _w768; do
  for kind in pktg4 pktg8 pktg pktw; do for len in 1024 4096; do
    echo -n "lib$v $kind rep$rep " >> $O/ab.txt
    AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python profiles/pkt_bench.py $kind --len $len --key-bits 256 --steps 7 >> $O/ab.txt 2>> $O/ab.err
  done; done
  AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python bench.py --config cfg5 --steps 10 --warmup 2 --no-cpu-baseline > $O/cfg5$v$rep.json 2> $O/cfg5$v$rep.err
  AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python bench.py --config cfg5 --key-bits 256 --steps 10 --warmup 2 --no-cpu-baseline > $O/cfg5_aes256$v$rep.json 2>> $O/cfg5$v$rep.err
  AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python bench.py --config cfg5 --pkt-len 1024 --steps 10 --warmup 2 --no-cpu-baseline > $O/cfg5_1k$v$rep.json 2>> $O/cfg5$v$rep.err
done; done
python - $O/ab.txt <<'PY'
import sys,json
for l in open(sys.argv[1]):
    a=l.split(" ",3); d=json.loads(a[3])
    print("%-14s %-6s %s len %5d  %7.1f GiB/s  %.3f ms" % (a[0], a[1], a[2], d["pkt_len"], d["gib_per_s"], d["ms_median"]))
PY
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
for kind in pktg pktg8 pktg4 batch batch256; do
  if [ $kind = batch ]; then cmd="$REPO/bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline"; k="k_batch3";
  elif [ $kind = batch256 ]; then cmd="$REPO/bench.py --config cfg5 --key-bits 256 --steps 3 --warmup 1 --no-cpu-baseline"; k="k_batch3";
  else cmd="$REPO/profiles/pkt_bench.py $kind --len 1024 --key-bits 256 --steps 3"; k="k_pktg"; fi
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/tcc_$kind -- python3 $cmd > $O/tcc_$kind.json 2> $O/tcc_$kind.err
  python3 - $O/tcc_$kind "$kind" $k <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(float); disp=set()
for p in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if sys.argv[3] in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
n=max(1,len(disp))
print(sys.argv[2], "read 128B-request bytes %.4g write 64B-request bytes %.4g" % (128*acc.get("TCC_EA0_RDREQ_128B_sum",0)/n, 64*acc.get("TCC_EA0_WRREQ_64B_sum",0)/n))
PY
  rm -rf $O/tcc_$kind
done
