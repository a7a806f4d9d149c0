#!/bin/bash
# round 3, call 2: k_pktg (lane groups) replaces k_pkt -- full GPU suite, packet sweep over the three shapes, 2^20 x 1 KiB / 4 KiB through every
# shape; --emulate-rank with the fused kernels chained across 1 / 2 / 4 contexts (and unchained), new copy kernel in the bench line
O=gpurun_out/r03_run2; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -5 $O/pytest.txt
for kind in pktw pktg pktl; do for len in 1024 4096; do for kb in 128 256; do
  timeout 300 python profiles/pkt_bench.py $kind --len $len --key-bits $kb --steps 7 >> $O/pkt_bench.txt 2>> $O/pkt_bench.err
done; done; done
cat $O/pkt_bench.txt
timeout 900 python profiles/packets_sweep.py 32 > $O/packets_sweep_aes256.txt 2>&1; cat $O/packets_sweep_aes256.txt
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
for r in 3; do for k in 1 2 4; do
  timeout 600 python bench.py --emulate-rank $r --of 8 --contexts $k --steps 10 --warmup 2 > $O/emu_r${r}_k$k.json 2> $O/emu_r${r}_k$k.err; echo "emu r=$r k=$k rc=$?"
done; done
timeout 600 python bench.py --emulate-rank 3 --of 8 --contexts 4 --no-chain --steps 10 --warmup 2 > $O/emu_r3_k4_nochain.json 2> $O/emu_r3_k4_nochain.err
python - $O <<'PY'
import json,sys,glob,os
for p in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-26s %.1f GiB/s step %.3f ms kernel %.3f ms x%d tag_ok %s copy %s" % (os.path.basename(p), d["value"], d["ms_per_step"], r["avg_launch_ms"], r["launches_timed"], d["tag_ok"], r["measured_copy_kernel"]["value"]))
    except Exception as e:
        print(p, "unreadable", e)
PY
