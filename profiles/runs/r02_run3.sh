#!/bin/bash
O=gpurun_out/r02_run3; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest_gpu.log 2>&1
tail -25 $O/pytest_gpu.log
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench768.json 2> $O/bench768.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02_run3/bench768.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("768 default: %.1f GiB/s kernel %.3f ms frac %.4f ceiling %s" % (d["value"], r["avg_launch_ms"], r["frac"], r["formulation_ceiling"].get("value")))
PY
