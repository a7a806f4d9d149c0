#!/bin/bash
# Copy what a final-collection call (profiles/archive/runs/r04_run*.sh, merged into gpurun_out/<run>/) left into profiles/<round>/ and profiles/pmc_cfg{2,3,5}_n1.json:
#   bash profiles/adopt_collection.sh gpurun_out/r04_run29 r04
O=$1; R=$2
for d in $O/prof_*; do t=$(basename $d); t=${t#prof_}; mkdir -p profiles/$R/$t; cp $d/* profiles/$R/$t/; done
cp $O/bench_*.json profiles/$R/
for f in inflight_sweep.txt inflight_sweep_half.txt latency_c.txt size_sweep.txt size_sweep_256m.txt so_sha256.txt smoke.txt pipeline_time.txt ctx_time.txt packets_sweep_mixed_aes256.txt packets_sweep_packed_aes256.txt packets_sweep_aes256.txt batch_mixed_aes128.txt; do
  [ -f $O/$f ] && cp $O/$f profiles/$R/
done
[ -f $O/cyc_timeline_aes256.txt ] && cp $O/cyc_timeline_aes256.txt profiles/$R/cyc_timeline_aes256_final.txt
tail -25 $O/pytest.txt > profiles/$R/pytest_tail.txt
for c in 2 3 5; do cp $O/prof_cfg${c}_n1/pmc_cfg${c}_n1.json profiles/; done
[ -f $O/prof_rows_1m/pmc_rows_1m.json ] && cp $O/prof_rows_1m/pmc_rows_1m.json profiles/
python3 tools/isa_census.py > profiles/$R/isa_census.txt
echo "library of the collection: $(cut -c1-12 $O/so_sha256.txt | head -1); in tree: $(sha256sum aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so | cut -c1-12); pmc stamps: $(grep -o '"so_sha256": "[0-9a-f]\{12\}' profiles/pmc_cfg3_n1.json) $(grep -o '"git": "[0-9a-f]*' profiles/pmc_cfg3_n1.json)"
