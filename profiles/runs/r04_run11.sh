#!/bin/bash
# round 4, call 11: cfg3 energy A/B (the kernel loses 14 % to the shader clock under HBM load and nothing to stalls): 1024-lane workgroups (4 waves per SIMD,
# what ships) against 768 (3 waves per SIMD, 168 registers), alternating on one box; time, shader cycles and sclk of both
O=$PWD/gpurun_out/r04_run11; mkdir -p $O
for rep in 1 2 3; do for V in wg1024 wg768; do
  AESGCM_LIB=$PWD/experiments/lib_$V.so timeout 600 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_${V}_$rep.json 2> $O/bench_${V}_$rep.err
  python3 - $O/bench_${V}_$rep.json $V $rep <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]; c = r.get("formulation_ceiling") or {}
print("%-7s rep %s: %.1f GiB/s  step %.3f ms  kernel %.3f ms  sclk %s MHz -> %.3e shader cycles per launch | no-HBM twin %.3f ms at %s MHz  tag_ok %s" % (
    sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], r["avg_launch_ms"], r.get("sclk_mhz"), r["avg_launch_ms"] * 1e-3 * (r.get("sclk_mhz") or 0) * 1e6, c.get("ms", 0), c.get("sclk_mhz"), d["tag_ok"]))
PY
done; done 2>&1 | tee $O/energy_ab.txt
