#!/bin/bash
# round 5: FINAL collection -- full GPU suite, smoke, the bench lines (cfg3, cfg2, decrypt, cfg5 with its ceiling, msgs = 4096 x 1 MiB by rows, 64 KiB messages, emulated rank steps,
# messages in flight), the message-size sweep, rocprofv3 stats + counter passes for the measured kernels (profiles/collect.sh)
O=$PWD/gpurun_out/r05_final; mkdir -p $O
export GIT_HEAD=$(cat .git_head 2>/dev/null)
sha256sum aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so aes-gcm-128-192-256-bits_amd/libaesgcm_hip_dbg.so > $O/so_sha256.txt
timeout 3000 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -14 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; cat $O/smoke.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 600 python bench.py --config cfg2 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 600 python bench.py --decrypt --no-cpu-baseline > $O/bench_dec.json 2> $O/bench_dec.err
timeout 600 python bench.py --config cfg5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 600 python bench.py --config cfg5 --decrypt --no-cpu-baseline > $O/bench_cfg5_dec.json 2> $O/bench_cfg5_dec.err
timeout 600 python bench.py --config cfg5 --key-bits 256 --no-cpu-baseline > $O/bench_cfg5_aes256.json 2> $O/bench_cfg5_aes256.err
timeout 600 python bench.py --config msgs > $O/bench_msgs.json 2> $O/bench_msgs.err
timeout 600 python bench.py --config msgs --decrypt --no-cpu-baseline > $O/bench_msgs_dec.json 2> $O/bench_msgs_dec.err
timeout 600 python bench.py --config msgs --key-bits 128 --no-cpu-baseline > $O/bench_msgs_aes128.json 2> $O/bench_msgs_aes128.err
timeout 600 python bench.py --config msgs --n-pkts 65536 --pkt-len 65536 --no-cpu-baseline > $O/bench_msgs_64k.json 2> $O/bench_msgs_64k.err
timeout 600 python bench.py --config msgs --n-pkts 4096 --pkt-len 65536 --steps 400 --warmup 100 --no-cpu-baseline > $O/bench_msgs_64k_4096.json 2> $O/bench_msgs_64k_4096.err
for r in 0 7; do timeout 600 python bench.py --emulate-rank $r --of 8 --no-cpu-baseline > $O/bench_emu_r$r.json 2> $O/bench_emu_r$r.err; done
python - $O <<'PY'
import json,sys,glob,os
for p in sorted(glob.glob(sys.argv[1]+"/bench*.json")):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d["roofline"]; c=d.get("cpu_baseline") or {}
        print("%-28s %.1f GiB/s step %.3f ms kernel %.3f ms frac %.4f tag_ok %s ceiling %s cpu %s GiB/s on %s cores" % (os.path.basename(p), d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], d["tag_ok"], r.get("achieved_over_ceiling"), c.get("value"), c.get("cores")))
    except Exception as e:
        print(p, "unreadable", e)
PY
timeout 900 python profiles/msg_sweep.py > $O/size_sweep.txt 2> $O/size_sweep.err; cat $O/size_sweep.txt
timeout 600 python profiles/msg_sweep.py --total-gib 0.25 --sizes-kib 64 256 1024 4096 16384 > $O/size_sweep_256m.txt 2>> $O/size_sweep.err; cat $O/size_sweep_256m.txt
INFLIGHT_KS="1 3" bash profiles/inflight_sweep.sh $O/inflight 1 16 64 2>&1 | tee $O/inflight_sweep.txt
timeout 600 python profiles/packets_sweep.py 32 var > $O/packets_sweep_mixed_aes256.txt 2>&1
timeout 300 python profiles/ctx_time.py > $O/ctx_time.txt 2>&1
timeout 600 python bench.py --config msgs --n-pkts 262144 --pkt-len 16384 --aad-len 13 --no-cpu-baseline > $O/bench_msgs_tls.json 2> $O/bench_msgs_tls.err
timeout 300 ./examples/latency 500 > $O/latency_c.txt 2>&1; tail -6 $O/latency_c.txt
bash profiles/collect.sh cfg3_n1 'k_body<14, 0, false>' bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/collect_cfg3.txt 2>&1
python3 profiles/summarize.py gpurun_out/prof_cfg3_n1 cfg3_probe 'k_body<14, 4, false>' > gpurun_out/prof_cfg3_n1/summary_probe.txt 2>&1
python3 profiles/summarize.py gpurun_out/prof_cfg3_n1 cfg3_n1 'k_body<14, 0, false>' > gpurun_out/prof_cfg3_n1/summary.txt 2>&1
bash profiles/collect.sh cfg2_n1 'k_body<10, 0, false>' bench.py --config cfg2 --steps 8 --warmup 2 --no-cpu-baseline > $O/collect_cfg2.txt 2>&1
bash profiles/collect.sh cfg5_n1 'k_batch3<10, 0' bench.py --config cfg5 --steps 5 --warmup 1 --no-cpu-baseline > $O/collect_cfg5.txt 2>&1
bash profiles/collect.sh rows_1m 'k_rows<' bench.py --config msgs --steps 8 --warmup 2 --no-cpu-baseline > $O/collect_rows_1m.txt 2>&1
bash profiles/collect.sh rows_1m_dec 'k_rows<' bench.py --config msgs --decrypt --steps 8 --warmup 2 --no-cpu-baseline > $O/collect_rows_1m_dec.txt 2>&1
bash profiles/collect.sh rows_64k 'k_rows<' bench.py --config msgs --n-pkts 4096 --pkt-len 65536 --steps 100 --warmup 20 --no-cpu-baseline > $O/collect_rows_64k.txt 2>&1
bash profiles/collect.sh rows_mixed 'k_rows<' profiles/rows_mixed.py > $O/collect_rows_mixed.txt 2>&1
bash profiles/collect.sh rows_tls 'k_rows<' profiles/pkt_bench.py pkt --n 262144 --len 16400 --aad 13 --key-bits 256 --steps 9 > $O/collect_rows_tls.txt 2>&1
bash profiles/archive/runs/r05_rows_min2.sh > /dev/null 2>&1; bash profiles/archive/runs/r05_rows_min3.sh > /dev/null 2>&1; bash profiles/archive/runs/r05_aad_cost.sh > /dev/null 2>&1; bash profiles/archive/runs/r05_ragged_many.sh > /dev/null 2>&1; bash profiles/archive/runs/r05_few_large_stats.sh > /dev/null 2>&1; bash profiles/archive/runs/r05_rows_8k_stats.sh > /dev/null 2>&1; bash profiles/archive/runs/r05_var_stats.sh > /dev/null 2>&1; bash profiles/archive/runs/r05_scatter.sh > /dev/null 2>&1
cp gpurun_out/r05/rows_min_sweep2.txt gpurun_out/r05/rows_min_sweep3.txt gpurun_out/r05/rows_ragged_many.txt gpurun_out/r05/rows_few_large_stats.txt gpurun_out/r05/rows_small_end_stats.txt gpurun_out/r05/rows_var_stats.txt gpurun_out/r05/rows_scatter.txt $O/; cp gpurun_out/r05/rows_aad_cost.txt $O/rows_aad_cost_after.txt
for t in cfg3_n1 cfg2_n1 cfg5_n1 rows_1m rows_1m_dec rows_64k rows_mixed rows_tls; do
  mkdir -p $O/prof_$t; cp gpurun_out/prof_$t/summary*.txt gpurun_out/prof_$t/pmc_$t.json gpurun_out/prof_$t/pmc_cfg3_probe.json gpurun_out/prof_$t/stats_run.json $O/prof_$t/ 2>/dev/null
  find gpurun_out/prof_$t/stats -name "*kernel_stats.csv" -exec cp {} $O/prof_$t/kernel_stats.csv \;
  echo "== $t"; grep -E "hot_kernel|hot_avg_ns|hbm_bytes_per_launch|lds_busy_frac|SQ_LDS_BANK_CONFLICT|SQ_LDS_IDX_ACTIVE " gpurun_out/prof_$t/summary.txt | head -8
  rm -rf gpurun_out/prof_$t
done
