#!/bin/bash
# round 3, call 3: k_batch3 (single-pass cfg5 kernel) parity + A/B against the two-phase k_batch2; cfg5 bench line with its CPU baseline;
# two-rows-in-flight k_body (AESGCM_BODY_ILP=2 build) against the shipped build; rocprof collections for k_pktg, k_batch3 and k_body
O=gpurun_out/r03_run3; mkdir -p $O
export GIT_HEAD=$(cat .git_head 2>/dev/null)
timeout 1800 python -m pytest tests/test_gpu_batch.py tests/test_gpu_stress.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -5 $O/pytest.txt
for rep in 1 2 3; do for f in 0 1; do
  AESGCM_BATCH_FUSED=$f timeout 300 python bench.py --config cfg5 --steps 10 --warmup 2 --no-cpu-baseline > $O/cfg5_fused${f}_$rep.json 2> $O/cfg5_fused${f}_$rep.err
  AESGCM_BATCH_FUSED=$f timeout 300 python bench.py --config cfg5 --key-bits 256 --steps 10 --warmup 2 --no-cpu-baseline > $O/cfg5_aes256_fused${f}_$rep.json 2>> $O/cfg5_fused${f}_$rep.err
  AESGCM_BATCH_FUSED=$f timeout 300 python bench.py --config cfg5 --decrypt --steps 10 --warmup 2 --no-cpu-baseline > $O/cfg5_dec_fused${f}_$rep.json 2>> $O/cfg5_fused${f}_$rep.err
done; done
timeout 600 python bench.py --config cfg5 --steps 10 --warmup 2 > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "cfg5 bench rc=$?"
for rep in 1 2 3; do for v in "" _ilp2; do
  AESGCM_LIB=$PWD/aes-gcm-128-192-256-bits_amd/libaesgcm_hip$v.so timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/ab_body$v$rep.json 2> $O/ab_body$v$rep.err
done; done
python - $O <<'PY'
import json,sys,glob,os
for p in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d["roofline"]
        extra = ""
        if "formulation_ceiling" in r and r["formulation_ceiling"]: extra = " probe %.3f ms sclk %s/%s" % (r["formulation_ceiling"].get("ms", 0), r.get("sclk_mhz"), r["formulation_ceiling"].get("sclk_mhz"))
        print("%-30s %.1f GiB/s step %.3f ms kernel %.3f ms frac %.4f tag_ok %s%s" % (os.path.basename(p), d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], d["tag_ok"], extra))
    except Exception as e:
        print(p, "unreadable", e)
PY
bash profiles/collect.sh pktg_1k 'k_pktg' profiles/pkt_bench.py pktg --len 1024 --key-bits 256 --steps 5 > $O/collect_pktg_1k.txt 2>&1
bash profiles/collect.sh pktw_1k 'k_pktg' profiles/pkt_bench.py pktw --len 1024 --key-bits 256 --steps 5 > $O/collect_pktw_1k.txt 2>&1
bash profiles/collect.sh pktl_1k 'k_pktl' profiles/pkt_bench.py pktl --len 1024 --key-bits 256 --steps 5 > $O/collect_pktl_1k.txt 2>&1
bash profiles/collect.sh cfg5_n1 'k_batch3' bench.py --config cfg5 --steps 5 --warmup 1 --no-cpu-baseline > $O/collect_cfg5.txt 2>&1
bash profiles/collect.sh cfg3_n1 'k_body<14, 0>' bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/collect_cfg3.txt 2>&1
python3 profiles/summarize.py gpurun_out/prof_cfg3_n1 cfg3_probe 'k_body<14, 4>' > gpurun_out/prof_cfg3_n1/summary_probe.txt 2>&1
for t in pktg_1k pktw_1k pktl_1k cfg5_n1 cfg3_n1; do echo "== $t"; tail -45 gpurun_out/prof_$t/summary.txt | head -60; done
tail -30 gpurun_out/prof_cfg3_n1/summary_probe.txt
# keep the merged output small: the raw rocprof trees stay on the box except the csv summaries
for t in pktg_1k pktw_1k pktl_1k cfg5_n1 cfg3_n1; do mkdir -p $O/prof_$t; cp gpurun_out/prof_$t/summary*.txt gpurun_out/prof_$t/*.json $O/prof_$t/ 2>/dev/null; find gpurun_out/prof_$t/stats -name "*kernel_stats.csv" -exec cp {} $O/prof_$t/kernel_stats.csv \; ; rm -rf gpurun_out/prof_$t; done
