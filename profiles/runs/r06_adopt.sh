#!/bin/bash
# copy what profiles/runs/r06_final.sh left (merged into gpurun_out/r06_final/) into profiles/r06/ and the pmc stamps bench.py reads into profiles/
O=gpurun_out/r06_final; R=profiles/r06
mkdir -p $R
for d in $O/prof_*; do t=$(basename $d); t=${t#prof_}; mkdir -p $R/$t; cp $d/* $R/$t/; done
cp $O/bench_*.json $O/so_sha256.txt $O/smoke.txt $O/mixed_u.jsonl $O/route_sweep.jsonl $O/size_sweep.txt $O/graph_replay.jsonl $O/route_band.jsonl $O/placed.jsonl $R/ 2>/dev/null
tail -25 $O/pytest.txt > $R/pytest_tail.txt
for t in cfg2_n1 cfg3_n1 cfg5_n1 rows_1m frames; do [ -f $O/prof_$t/pmc_$t.json ] && cp $O/prof_$t/pmc_$t.json profiles/; done
python3 tools/isa_census.py > $R/isa_census.txt 2>/dev/null
echo "library of the collection: $(cut -c1-12 $O/so_sha256.txt | head -1); in tree: $(sha256sum aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so | cut -c1-12); pmc stamps: $(grep -o '"so_sha256": "[0-9a-f]\{12\}' profiles/pmc_cfg3_n1.json) $(grep -o '"so_sha256": "[0-9a-f]\{12\}' profiles/pmc_frames.json)"
