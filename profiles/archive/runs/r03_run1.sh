#!/bin/bash
# round 3, call 1: new tests (pinned-output race, self-launch, multiproc), default bench line, copy-kernel variants,
# --emulate-rank r --of 8 for r = 0, 3, 7 with 1 / 2 / 4 contexts per rank, counter names for task 6
O=gpurun_out/r03_run1; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_pinned.py tests/test_gpu_selflaunch.py tests/test_gpu_multiproc.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -5 $O/pytest.txt
timeout 600 python bench.py --steps 10 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 300 ./profiles/microbench/copy_variants 16 > $O/copy_variants.txt 2>&1; cat $O/copy_variants.txt
for r in 0 3 7; do for k in 1 2 4; do
  timeout 600 python bench.py --emulate-rank $r --of 8 --contexts $k --steps 10 --warmup 2 > $O/emu_r${r}_k$k.json 2> $O/emu_r${r}_k$k.err; echo "emu r=$r k=$k rc=$?"
done; done
python - $O <<'PY'
import json,sys,glob,os
for p in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-22s %.1f GiB/s step %.3f ms kernel %.3f ms x%d tag_ok %s copy %s" % (os.path.basename(p), d["value"], d["ms_per_step"], r["avg_launch_ms"], r["launches_timed"], d["tag_ok"], r["measured_copy_kernel"]["value"]))
    except Exception as e:
        print(p, "unreadable", e)
PY
(cd /tmp && rocprofv3 --list-avail > $OLDPWD/$O/list_avail.txt 2>&1); grep -c . $O/list_avail.txt
