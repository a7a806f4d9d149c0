#!/bin/bash
# round 3, call 24: stores through the L2 -- parity of the cyclic launch in every closing mode, latency, and the dealt k_body at 16 GiB with and without
O=$PWD/gpurun_out/r03_run24; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_cyclic.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python profiles/cyc_end.py | tee $O/cyc_end.txt
for rep in 1 2; do for wt in 0 1; do
  AESGCM_BODY_WT=$wt timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_wt${wt}_$rep.json 2> $O/bench_wt${wt}_$rep.err
done; done
python - $O <<'PY'
import json,sys,glob,os
for p in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-26s %.1f GiB/s step %.3f ms kernel %.3f ms tag_ok %s sclk %s" % (os.path.basename(p), d["value"], d["ms_per_step"], r["avg_launch_ms"], d["tag_ok"], r.get("sclk_mhz")))
    except Exception as e:
        print(p, "unreadable", e)
PY
