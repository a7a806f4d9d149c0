#!/bin/bash
# round 4, call 71: measurement only (library of call 70): the emulated rank steps of the 2- and 4-GPU jobs beside the 8-GPU ones, so that every N the
# driver's scaling run asks for (1, 2, 4, 8) has its one-GPU prediction; N = 1 on the same box for the ratio
O=$PWD/gpurun_out/r04_run71; mkdir -p $O
sha256sum aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so > $O/so_sha256.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err
for w in 2 4 8; do for r in 0 $((w-1)); do
  timeout 600 python bench.py --emulate-rank $r --of $w --no-cpu-baseline > $O/bench_emu_r${r}_of$w.json 2> $O/bench_emu_r${r}_of$w.err
done; done
python - $O <<'PY' | tee $O/emulated_scaling.txt
import json,sys,glob,os
O=sys.argv[1]
n1=json.loads(open(O+"/bench_n1.json").read().strip().splitlines()[-1])
print("N = 1 step: %.1f GiB/s, %.3f ms (sclk under load: see the .err files)" % (n1["value"], n1["ms_per_step"]))
print("%-8s %-6s %10s %10s %8s %s" % ("of W", "rank", "GiB/s", "ms/step", "vs N=1", "tags"))
for w in (2,4,8):
    for r in (0,w-1):
        d=json.loads(open(O+"/bench_emu_r%d_of%d.json"%(r,w)).read().strip().splitlines()[-1])
        print("%-8d %-6d %10.1f %10.3f %8.3f %s" % (w, r, d["value"], d["ms_per_step"], d["value"]/n1["value"], d.get("tag_ok")))
PY
