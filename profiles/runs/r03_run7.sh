#!/bin/bash
# round 3, call 7: batched finalize (one launch for the M tags of a step) and the direct weighted partial of a pure-body shard: parity, then the rank step with / without
O=$PWD/gpurun_out/r03_run7; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py tests/test_gpu_multiproc.py tests/test_gpu_selflaunch.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -5 $O/pytest.txt
for rep in 1 2 3; do
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/n1_$rep.json 2> $O/n1_$rep.err
  timeout 300 python bench.py --emulate-rank 3 --of 8 --steps 8 --warmup 2 > $O/emu_batch_$rep.json 2> $O/emu_batch_$rep.err
  timeout 300 python bench.py --emulate-rank 3 --of 8 --steps 8 --warmup 2 --no-batch-finalize > $O/emu_nobatch_$rep.json 2> $O/emu_nobatch_$rep.err
  timeout 300 python bench.py --emulate-rank 3 --of 8 --steps 8 --warmup 2 --contexts 4 > $O/emu_batch_k4_$rep.json 2> $O/emu_batch_k4_$rep.err
done
python - $O <<'PY'
import json,sys,glob,os
for p in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-26s %.1f GiB/s step %.3f ms kernel %.3f ms x%d tag_ok %s" % (os.path.basename(p), d["value"], d["ms_per_step"], r["avg_launch_ms"], r["launches_timed"], d["tag_ok"]))
    except Exception as e:
        print(p, "unreadable", e)
PY
REPO=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $REPO/bench.py --emulate-rank 3 --of 8 --steps 3 --warmup 1 > $O/trace.json 2> $O/trace.err
t=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - $t <<'PY' | tee $O/rank_step_trace.txt
import csv,sys
rows=sorted(csv.DictReader(open(sys.argv[1])), key=lambda r:int(r["Start_Timestamp"]))
ks=[i for i,r in enumerate(rows) if "k_body<14, 0>" in r["Kernel_Name"]]
first=ks[32+4] if len(ks)>36 else ks[0]
t0=int(rows[first]["Start_Timestamp"])
n=0
for r in rows[first:]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("   %-40s start %9.1f us  dur %8.1f us  queue %s" % (r["Kernel_Name"][:40], (s-t0)/1e3, (e-s)/1e3, r.get("Queue_Id")))
    n+=1
    if n>24: break
PY
rm -rf $O/trace
