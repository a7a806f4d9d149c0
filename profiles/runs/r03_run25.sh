#!/bin/bash
# round 3, call 25: whole suite + smoke + bench on the write-through build; latency from C
O=$PWD/gpurun_out/r03_run25; mkdir -p $O
timeout 2700 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?"
tail -4 $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
make -C examples -s && ./examples/latency 2>&1 | tee $O/latency_c.txt | tail -30
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 600 python bench.py --config cfg2 > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "bench cfg2 rc=$?"
python - $O <<'PY'
import json,sys,glob,os
for p in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-26s %.1f GiB/s step %.3f ms kernel %.3f ms tag_ok %s frac %.4f" % (os.path.basename(p), d["value"], d["ms_per_step"], r["avg_launch_ms"], d["tag_ok"], r["frac"]))
    except Exception as e:
        print(p, "unreadable", e)
PY
