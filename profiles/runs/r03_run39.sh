#!/bin/bash
# round 3, call 39: k_batch3 with 8 lanes per packet (eight packets per wave): parity, then cfg5 A/B and other packet sizes
O=$PWD/gpurun_out/r03_run39; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_batch.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do for lg in 4 3; do
  AESGCM_BATCH_LG=$lg timeout 300 python bench.py --config cfg5 --no-cpu-baseline > $O/cfg5_lg${lg}_$rep.json 2> $O/cfg5_lg${lg}_$rep.err
  AESGCM_BATCH_LG=$lg timeout 300 python bench.py --config cfg5 --key-bits 256 --no-cpu-baseline > $O/cfg5k256_lg${lg}_$rep.json 2> $O/cfg5k256_lg${lg}_$rep.err
  AESGCM_BATCH_LG=$lg timeout 300 python bench.py --config cfg5 --decrypt --no-cpu-baseline > $O/cfg5dec_lg${lg}_$rep.json 2> $O/cfg5dec_lg${lg}_$rep.err
done; done
python - $O <<'PY'
import json,sys,glob,os
for p in sorted(glob.glob(sys.argv[1]+"/cfg5*.json")):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-26s %.1f GiB/s step %.3f ms kernel %.3f ms tag_ok %s" % (os.path.basename(p), d["value"], d["ms_per_step"], r["avg_launch_ms"], d["tag_ok"]))
    except Exception as e:
        print(p, "unreadable", e)
PY
for lg in 4 3; do echo "== AESGCM_BATCH_LG=$lg"; for len in 256 1024 4096 16384; do AESGCM_BATCH_LG=$lg timeout 120 python profiles/pkt_bench.py batch --len $len --n $((1<<30 / len > 1048576 ? 1048576 : 1<<30 / len)) --steps 5 | cut -c1-200; done; done | tee $O/batch_lens.txt
