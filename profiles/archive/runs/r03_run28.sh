#!/bin/bash
# round 3, call 28: FINAL collection (second: after the cyclic rows and the fused closing) -- full GPU suite, smoke, the bench lines (cfg3, cfg2, decrypt, cfg5 with their CPU baselines), the emulated rank steps,
# rocprofv3 stats + counter passes for every measured kernel (profiles/collect.sh), latency, size sweep
O=$PWD/gpurun_out/r03_run28; mkdir -p $O
export GIT_HEAD=$(cat .git_head 2>/dev/null)
sha256sum aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so > $O/so_sha256.txt
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -4 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; cat $O/smoke.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 600 python bench.py --config cfg2 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 600 python bench.py --decrypt --no-cpu-baseline > $O/bench_dec.json 2> $O/bench_dec.err
timeout 600 python bench.py --config cfg5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 600 python bench.py --config cfg5 --decrypt --no-cpu-baseline > $O/bench_cfg5_dec.json 2> $O/bench_cfg5_dec.err
for r in 0 3 7; do timeout 600 python bench.py --emulate-rank $r --of 8 > $O/bench_emu_r$r.json 2> $O/bench_emu_r$r.err; done
python - $O <<'PY'
import json,sys,glob,os
for p in sorted(glob.glob(sys.argv[1]+"/bench*.json")):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d["roofline"]; c=d.get("cpu_baseline") or {}
        print("%-22s %.1f GiB/s step %.3f ms kernel %.3f ms frac %.4f tag_ok %s cpu %s GiB/s on %s cores (1 core %s)" % (os.path.basename(p), d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], d["tag_ok"], c.get("value"), c.get("cores"), c.get("value_1core")))
    except Exception as e:
        print(p, "unreadable", e)
PY
timeout 300 ./examples/latency 500 > $O/latency_c.txt 2>&1
timeout 600 python profiles/size_sweep.py > $O/size_sweep.txt 2>&1
bash profiles/collect.sh cfg3_n1 'k_body<14, 0, false>' bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/collect_cfg3.txt 2>&1
python3 profiles/summarize.py gpurun_out/prof_cfg3_n1 cfg3_probe 'k_body<14, 4, false>' > gpurun_out/prof_cfg3_n1/summary_probe.txt 2>&1
python3 profiles/summarize.py gpurun_out/prof_cfg3_n1 cfg3_n1 'k_body<14, 0, false>' > gpurun_out/prof_cfg3_n1/summary.txt 2>&1
bash profiles/collect.sh cyc_64m 'k_body<14, 0, true>' profiles/latency_one.py 67108864 60 > $O/collect_cyc_64m.txt 2>&1
bash profiles/collect.sh cyc_1m 'k_body<14, 0, true>' profiles/latency_one.py 1048576 200 > $O/collect_cyc_1m.txt 2>&1
timeout 600 python profiles/cyc_sweep.py 32 > $O/cyc_sweep_aes256.txt 2>&1
timeout 600 python profiles/cyc_small.py > $O/cyc_small.txt 2>&1
timeout 600 python profiles/general_shape.py > $O/general_shape.txt 2>&1
timeout 600 python profiles/cyc_end.py > $O/cyc_end.txt 2>&1
bash profiles/collect.sh cfg3_dec 'k_body<14, 1, false>' bench.py --decrypt --steps 4 --warmup 1 --no-cpu-baseline > $O/collect_cfg3_dec.txt 2>&1
bash profiles/collect.sh cfg2_n1 'k_body<10, 0, false>' bench.py --config cfg2 --steps 8 --warmup 2 --no-cpu-baseline > $O/collect_cfg2.txt 2>&1
bash profiles/collect.sh cfg5_n1 'k_batch3' bench.py --config cfg5 --steps 5 --warmup 1 --no-cpu-baseline > $O/collect_cfg5.txt 2>&1
bash profiles/collect.sh cfg5_aes256 'k_batch3' bench.py --config cfg5 --key-bits 256 --steps 5 --warmup 1 --no-cpu-baseline > $O/collect_cfg5_aes256.txt 2>&1
bash profiles/collect.sh cfg5_dec 'k_batch3' bench.py --config cfg5 --decrypt --steps 5 --warmup 1 --no-cpu-baseline > $O/collect_cfg5_dec.txt 2>&1
bash profiles/collect.sh pktg_1k 'k_pktg' profiles/pkt_bench.py pktg --len 1024 --key-bits 256 --steps 5 > $O/collect_pktg_1k.txt 2>&1
bash profiles/collect.sh pktg8_1k 'k_pktg' profiles/pkt_bench.py pktg8 --len 1024 --key-bits 256 --steps 5 > $O/collect_pktg8_1k.txt 2>&1
bash profiles/collect.sh pktg4_1k 'k_pktg' profiles/pkt_bench.py pktg4 --len 1024 --key-bits 256 --steps 5 > $O/collect_pktg4_1k.txt 2>&1
bash profiles/collect.sh pktw_1k 'k_pktg' profiles/pkt_bench.py pktw --len 1024 --key-bits 256 --steps 5 > $O/collect_pktw_1k.txt 2>&1
bash profiles/collect.sh pktl_1k 'k_pktl' profiles/pkt_bench.py pktl --len 1024 --key-bits 256 --steps 5 > $O/collect_pktl_1k.txt 2>&1
bash profiles/collect.sh batch1w_4k 'k_batch<' profiles/batch_one.py > $O/collect_batch1w.txt 2>&1
for t in cyc_64m cyc_1m cfg3_n1 cfg3_dec cfg2_n1 cfg5_n1 cfg5_aes256 cfg5_dec pktg_1k pktg8_1k pktg4_1k pktw_1k pktl_1k batch1w_4k; do
  mkdir -p $O/prof_$t; cp gpurun_out/prof_$t/summary*.txt gpurun_out/prof_$t/pmc_$t.json gpurun_out/prof_$t/pmc_cfg3_probe.json gpurun_out/prof_$t/stats_run.json $O/prof_$t/ 2>/dev/null
  find gpurun_out/prof_$t/stats -name "*kernel_stats.csv" -exec cp {} $O/prof_$t/kernel_stats.csv \;
  echo "== $t"; grep -E "hot_kernel|hot_avg_ns|hbm_bytes_per_launch|lds_busy_frac|bytes_by_request_size|bytes_64B_full" gpurun_out/prof_$t/summary.txt | head -8
  rm -rf gpurun_out/prof_$t
done
