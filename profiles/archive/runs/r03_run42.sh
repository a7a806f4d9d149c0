#!/bin/bash
# round 3, call 42: the second k_fold level closes 16 GiB messages (FoldClose at any level): parity, cfg3 A/B
O=$PWD/gpurun_out/r03_run42; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_large.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2 3; do for fc in 1 0; do
  AESGCM_FOLD_CLOSE=$fc timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/cfg3_fc${fc}_$rep.json 2> $O/cfg3_fc${fc}_$rep.err
done; done
python - $O <<'PY'
import json,sys,glob,os
for p in sorted(glob.glob(sys.argv[1]+"/cfg*.json")):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-26s %.1f GiB/s step %.3f ms kernel %.3f ms tag_ok %s" % (os.path.basename(p), d["value"], d["ms_per_step"], r["avg_launch_ms"], d["tag_ok"]))
    except Exception as e:
        print(p, "unreadable", e)
PY
