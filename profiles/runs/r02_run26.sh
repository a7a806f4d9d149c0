#!/bin/bash
# round 2, call 26: final collection (rocprofv3 stats + PMC for every measured workload) on the build with five-bit GHASH tables,
# runtime fold groups and the new dispensers; then the default bench line and smoke
O=gpurun_out/r02_run26; mkdir -p $O
bash profiles/collect.sh cfg3_n1 'k_body|k_main' bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/cfg3.log 2>&1
bash profiles/collect.sh cfg2_n1 'k_body|k_main' bench.py --config cfg2 --steps 6 --warmup 1 --no-cpu-baseline > $O/cfg2.log 2>&1
bash profiles/collect.sh cfg3_dec 'k_body|k_main' bench.py --decrypt --steps 4 --warmup 1 --no-cpu-baseline > $O/dec.log 2>&1
bash profiles/collect.sh cfg5_batch 'k_batch' profiles/pkt_bench.py batch --steps 4 > $O/cfg5.log 2>&1
bash profiles/collect.sh pktw_1k 'k_pkt' profiles/pkt_bench.py pktw --len 1024 --key-bits 256 --steps 4 > $O/pktw.log 2>&1
bash profiles/collect.sh pktl_1k 'k_pktl' profiles/pkt_bench.py pktl --len 1024 --key-bits 256 --steps 4 > $O/pktl.log 2>&1
for t in cfg3_n1 cfg2_n1 cfg3_dec cfg5_batch pktw_1k pktl_1k; do echo "=== $t"; python3 - $t <<'PY'
import json,sys
t=sys.argv[1]
try:
    d=json.load(open("gpurun_out/prof_%s/pmc_%s.json"%(t,t)))
    print({k:d.get(k) for k in ("kernel","kernel_avg_ns_under_rocprof","hbm_read_bytes_per_launch","hbm_write_bytes_per_launch","so_sha256")}, (d.get("lds") or {}).get("lds_busy_frac"))
except Exception as e: print("no pmc json", e)
PY
done
cp gpurun_out/prof_cfg3_n1/pmc_cfg3_n1.json profiles/pmc_cfg3_n1.json; cp gpurun_out/prof_cfg2_n1/pmc_cfg2_n1.json profiles/pmc_cfg2_n1.json
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -1 $O/bench_default.json | cut -c1-1500
timeout 300 python bench.py --config cfg2 > $O/bench_cfg2.json 2>/dev/null; tail -1 $O/bench_cfg2.json | cut -c1-300
timeout 300 python bench.py --decrypt --no-cpu-baseline > $O/bench_dec.json 2>/dev/null; tail -1 $O/bench_dec.json | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
