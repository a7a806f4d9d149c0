#!/bin/bash
# round 6: FINAL collection (run ON THE GPU BOX through gpurun, from the repo root): full GPU suite, smoke, the bench lines -- cfg3, cfg2, cfg5, msgs, and the round's new
# ones: frames (with its ceiling), frames decrypt / AES-128 / 65536 frames --, the mixed call (U-shaped lengths, tb/gcm_gctr.py:279-281) against the two pure paths, the
# routing sweep, rocprofv3 stats + counter passes (profiles/collect.sh) for the measured kernels; adopted into profiles/r06/ by profiles/runs/r06_adopt.sh
O=$PWD/gpurun_out/r06_final; mkdir -p $O
sha256sum aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so aes-gcm-128-192-256-bits_amd/libaesgcm_hip_dbg.so > $O/so_sha256.txt
timeout 3000 python -m pytest tests -q -m gpu --durations=8 > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -14 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; cat $O/smoke.txt
bash profiles/collect.sh cfg3_n1 'k_body<14, 0, false>' bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/collect_cfg3.txt 2>&1
bash profiles/collect.sh cfg2_n1 'k_body<10, 0, false>' bench.py --config cfg2 --steps 8 --warmup 2 --no-cpu-baseline > $O/collect_cfg2.txt 2>&1
bash profiles/collect.sh cfg5_n1 'k_batch3<10, 0' bench.py --config cfg5 --steps 5 --warmup 1 --no-cpu-baseline > $O/collect_cfg5.txt 2>&1
bash profiles/collect.sh rows_1m 'k_rows<' bench.py --config msgs --steps 8 --warmup 2 --no-cpu-baseline > $O/collect_rows_1m.txt 2>&1
bash profiles/collect.sh frames 'k_pktl<14, 0' profiles/frames_one.py > $O/collect_frames.txt 2>&1
bash profiles/collect.sh frames_probe 'k_pktl<14, 2' profiles/frames_one.py --probe > $O/collect_frames_probe.txt 2>&1
bash profiles/collect.sh pktl_1k 'k_pktl<14, 0' profiles/frames_one.py --fixed 1024 --aad 0 > $O/collect_pktl_1k.txt 2>&1
bash profiles/collect.sh pktg_1k 'k_pktg<14, 0' profiles/frames_one.py --fixed 1024 --aad 0 --n 65536 > $O/collect_pktg_1k.txt 2>&1
bash profiles/collect.sh mixed_u 'k_rows<' profiles/mixed_bench.py --only mixed --n 262144 --max-len 65535 --aad 28 > $O/collect_mixed_u.txt 2>&1
bash profiles/collect.sh mixed_u_1m 'k_rows<' profiles/mixed_bench.py --only mixed --n 16384 --max-len 1048576 > $O/collect_mixed_u_1m.txt 2>&1
# (the counters of THIS library first, so that the bench lines behind them carry roofline.traffic: bench.py reads profiles/pmc_<tag>.json only when its hash is the running library's)
for t in cfg2_n1 cfg3_n1 cfg5_n1 rows_1m frames; do cp gpurun_out/prof_$t/pmc_$t.json profiles/ 2>/dev/null; done
b() { n=$1; shift; timeout 900 python bench.py "$@" > $O/bench_$n.json 2> $O/bench_$n.err; }
b default
b cfg2 --config cfg2
b dec --decrypt --no-cpu-baseline
b cfg5 --config cfg5
b msgs --config msgs
b msgs_scattered --config msgs --scattered --no-cpu-baseline
b frames --config frames
b frames_dec --config frames --decrypt --no-cpu-baseline
b frames_aes128 --config frames --key-bits 128 --no-cpu-baseline
b frames_64k --config frames --n-pkts 65536 --no-cpu-baseline
b frames_noaad --config frames --aad-len 0 --no-cpu-baseline
python - $O <<'PY'
import json,sys,glob,os
for p in sorted(glob.glob(sys.argv[1]+"/bench*.json")):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d["roofline"]; c=d.get("cpu_baseline") or {}
        print("%-28s %.1f GiB/s step %.3f ms kernel %.3f ms frac %.4f tag_ok %s ceiling %s cpu %s GiB/s on %s cores" % (os.path.basename(p), d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], d["tag_ok"], r.get("achieved_over_ceiling"), c.get("value"), c.get("cores")))
    except Exception as e:
        print(p, "unreadable", e)
PY
for a in "--n 262144 --max-len 65535" "--n 262144 --max-len 65535 --aad 28" "--n 16384 --max-len 1048576" "--n 16384 --max-len 1048576 --aad 28" "--n 262144 --max-len 65535 --scattered" "--n 262144 --max-len 65535 --dec" "--n 262144 --max-len 65535 --key-bits 128" "--n 1048576 --max-len 16384"; do
  timeout 600 python profiles/mixed_bench.py $a >> $O/mixed_u.jsonl 2>> $O/mixed_u.err
done
python - $O <<'PY'
import json,sys
for l in open(sys.argv[1]+"/mixed_u.jsonl"):
    d=json.loads(l)
    print("n %7d max %8d aad %2d %s%s%s: mixed %6.1f GiB/s (%.3f ms)  all_rows %6.1f  all_pkt %6.1f  short %.3f ms + long %.3f ms = %.3f -> vs combination %.3f" % (d["n"], d["max_len"], d["aad"], "AES-%d" % d["key_bits"], " scattered" if d["scattered"] else "", " dec" if d["decrypt"] else "",
          d["mixed"]["gib_per_s"], d["mixed"]["ms_median"], d["all_rows"]["gib_per_s"], d["all_pkt"]["gib_per_s"], d["short"]["ms_median"], d["long"]["ms_median"], d["combination_ms"], d["vs_combination"]))
PY
timeout 1200 python profiles/route_sweep.py > $O/route_sweep.jsonl 2> $O/route_sweep.err
python - $O <<'PY'
import json,sys
print("kind n blocks ms_rows ms_pkt ms_lib lib_vs_best")
for l in open(sys.argv[1]+"/route_sweep.jsonl"):
    d=json.loads(l); print(d["kind"], d["n"], d["blocks"], d["ms_rows"], d["ms_pkt"], d["ms_lib"], d["lib_vs_best"])
PY
timeout 900 python profiles/route_sweep.py --kinds band8_16,band8_16w,band16_32,u16k --counts 131072,262144,393216,458752,524288,1048576 > $O/route_band.jsonl 2> $O/route_band.err
timeout 900 python profiles/msg_sweep.py > $O/size_sweep.txt 2> $O/size_sweep.err; tail -14 $O/size_sweep.txt
for x in "" "--placed 1" "--placed 64" "--placed 64 --dec"; do timeout 200 python3 profiles/frames_one.py --steps 12 $x >> $O/placed.jsonl 2>> $O/placed.err; done
for n in 4096 16384 65536 1048576; do timeout 120 examples/graph_replay $n 200 >> $O/graph_replay.jsonl 2>> $O/graph_replay.err; done
for t in cfg3_n1 cfg2_n1 cfg5_n1 rows_1m frames frames_probe pktl_1k pktg_1k mixed_u mixed_u_1m; do
  mkdir -p $O/prof_$t; cp gpurun_out/prof_$t/summary*.txt gpurun_out/prof_$t/pmc_$t.json gpurun_out/prof_$t/stats_run.json $O/prof_$t/ 2>/dev/null
  find gpurun_out/prof_$t/stats -name "*kernel_stats.csv" -exec cp {} $O/prof_$t/kernel_stats.csv \;
  echo "== $t"; grep -E "hot_kernel|hot_avg_ns|hbm_bytes_per_launch|lds_busy_frac" gpurun_out/prof_$t/summary.txt | head -6
  rm -rf gpurun_out/prof_$t
done
