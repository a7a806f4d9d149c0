#!/bin/bash
# round 3, call 4: k_body with a fine-grained region B at the end of the launch (plan_body_tail) against AESGCM_BODY_TAIL=0, N = 1 and --emulate-rank;
# parity of the large paths; size sweep both ways; examples/latency (pageable / pinned host output); batch shapes at small counts; request sizes for the packet kernels
O=gpurun_out/r03_run4; mkdir -p $O
export GIT_HEAD=$(cat .git_head 2>/dev/null)
timeout 2400 python -m pytest tests/test_gpu_large.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_multiproc.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -5 $O/pytest.txt
for rep in 1 2 3; do for tw in 0 4096; do
  AESGCM_BODY_TAIL=$tw timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/n1_tail${tw}_$rep.json 2> $O/n1_tail${tw}_$rep.err
  AESGCM_BODY_TAIL=$tw timeout 300 python bench.py --emulate-rank 3 --of 8 --steps 8 --warmup 2 > $O/emu_tail${tw}_$rep.json 2> $O/emu_tail${tw}_$rep.err
done; done
AESGCM_BODY_TAIL=8192 timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/n1_tail8192_1.json 2> $O/n1_tail8192_1.err
AESGCM_BODY_TAIL=2048 timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/n1_tail2048_1.json 2> $O/n1_tail2048_1.err
AESGCM_BODY_TAIL=8192 timeout 300 python bench.py --emulate-rank 3 --of 8 --steps 8 --warmup 2 > $O/emu_tail8192_1.json 2> $O/emu_tail8192_1.err
python - $O <<'PY'
import json,sys,glob,os
for p in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-26s %.1f GiB/s step %.3f ms kernel %.3f ms x%d tag_ok %s" % (os.path.basename(p), d["value"], d["ms_per_step"], r["avg_launch_ms"], r["launches_timed"], d["tag_ok"]))
    except Exception as e:
        print(p, "unreadable", e)
PY
for tw in 0 4096; do echo "== size sweep AESGCM_BODY_TAIL=$tw AESGCM_TAIL_MIN=0"; AESGCM_BODY_TAIL=$tw AESGCM_TAIL_MIN=0 timeout 600 python profiles/size_sweep.py 2>&1 | tee $O/size_sweep_tail$tw.txt | grep -E "AES-256|size"; done
timeout 300 ./examples/latency 500 > $O/latency_c.txt 2>&1; cat $O/latency_c.txt
timeout 900 python profiles/batch_sweep.py 16 > $O/batch_sweep_aes128.txt 2>&1; cat $O/batch_sweep_aes128.txt
bash profiles/collect.sh pktg_1k 'k_pktg' profiles/pkt_bench.py pktg --len 1024 --key-bits 256 --steps 5 > $O/collect_pktg_1k.txt 2>&1
grep -A12 tcc_read gpurun_out/prof_pktg_1k/pmc_pktg_1k.json
mkdir -p $O/prof_pktg_1k; cp gpurun_out/prof_pktg_1k/summary*.txt gpurun_out/prof_pktg_1k/*.json $O/prof_pktg_1k/; rm -rf gpurun_out/prof_pktg_1k
