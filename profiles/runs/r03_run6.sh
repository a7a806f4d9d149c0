#!/bin/bash
# round 3, call 6: where the rank step's 0.36 ms outside k_body go (kernel trace of --emulate-rank), 16-row chunks for the 4 GiB shards, shape rule check
O=$PWD/gpurun_out/r03_run6; mkdir -p $O
REPO=$PWD
for rep in 1 2; do for tw in 0 16; do
  AESGCM_TW=$tw timeout 300 python bench.py --emulate-rank 3 --of 8 --steps 8 --warmup 2 > $O/emu_tw${tw}_$rep.json 2> $O/emu_tw${tw}_$rep.err
done; done
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/n1.json 2> $O/n1.err
python - $O <<'PY'
import json,sys,glob,os
for p in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-26s %.1f GiB/s step %.3f ms kernel %.3f ms x%d tag_ok %s" % (os.path.basename(p), d["value"], d["ms_per_step"], r["avg_launch_ms"], r["launches_timed"], d["tag_ok"]))
    except Exception as e:
        print(p, "unreadable", e)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $REPO/bench.py --emulate-rank 3 --of 8 --steps 3 --warmup 1 > $O/trace.json 2> $O/trace.err
t=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - $t <<'PY'
import csv,sys
rows=sorted(csv.DictReader(open(sys.argv[1])), key=lambda r:int(r["Start_Timestamp"]))
ks=[i for i,r in enumerate(rows) if "k_body<14, 0>" in r["Kernel_Name"]]
# the timed region: warmup 1 step (4 bodies) + prepass 28; take the 3 timed steps = bodies after the first 32
first=ks[32+4] if len(ks)>36 else ks[0]
t0=int(rows[first]["Start_Timestamp"])
n=0
for r in rows[first:]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("   %-40s start %9.1f us  dur %8.1f us  queue %s" % (r["Kernel_Name"][:40], (s-t0)/1e3, (e-s)/1e3, r.get("Queue_Id")))
    n+=1
    if n>60: break
PY
cd $REPO
timeout 600 python - <<'PY' | tee $O/shape_rule.txt
import os, sys, time
sys.path.insert(0, os.getcwd())
import aesgcm_amd
from aesgcm_amd import lib
ctx = lib.Context(bytes(range(32)))
d_ivs = lib.DeviceBuffer(12 << 20); d_ivs.fill_splitmix64(2, nbytes=(12 << 20))
d_tags = lib.DeviceBuffer(16 << 20)
os.environ.pop("AESGCM_PKT_SHAPE", None)
print("auto shape rule, AES-256, GiB/s")
for pkt in (64, 256, 1024, 4096, 16384):
    nm = min(1 << 20, (1 << 32) // pkt)
    a, b = lib.DeviceBuffer(pkt * nm), lib.DeviceBuffer(pkt * nm); a.fill_splitmix64(3)
    row = []
    for ln in range(10, 21, 2):
        n = 1 << ln
        if n > nm: break
        best = 1e9
        for it in range(4):
            lib.dev_sync(); t0 = time.perf_counter()
            ctx.packets_crypt_dev(False, n, d_ivs.ptr, a.ptr, b.ptr, d_tags.ptr, pkt_len=pkt)
            lib.dev_sync(); best = min(best, time.perf_counter() - t0)
        row.append("%7.1f" % (n * pkt / best / 2**30))
    print("%6d B: %s" % (pkt, " ".join(row)), flush=True)
    del a, b
PY
