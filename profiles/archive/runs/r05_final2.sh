# after the collection is adopted (profiles/pmc_*.json carry the library's hash): the bench lines once more, with their traffic, and the tests added since
O=$PWD/gpurun_out/r05_final2; mkdir -p $O
sha256sum aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so > $O/so_sha256.txt
timeout 900 python -m pytest tests/test_gpu_rows.py tests/test_gpu_wipe.py tests/test_gpu_parity.py -x -q 2>&1 | tail -4 > $O/pytest_rows.txt; cat $O/pytest_rows.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --config cfg2 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 600 python bench.py --config cfg5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 600 python bench.py --config msgs > $O/bench_msgs.json 2> $O/bench_msgs.err
for f in default cfg2 cfg5 msgs; do python3 -c "
import json,sys
d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$f', d['value'], r['frac'], r['traffic'], r.get('traffic_build',{}).get('match'), d.get('tag_ok'))"; done
