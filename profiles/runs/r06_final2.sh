#!/bin/bash
# round 6: the bench lines once more after the collection's counters were adopted (profiles/pmc_*.json now carry this library's hash, so roofline.traffic is filled in);
# another box than r06_final.sh's.  -> profiles/r06/second_box/
O=$PWD/gpurun_out/r06_final2; mkdir -p $O
sha256sum aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so > $O/so_sha256.txt
b() { n=$1; shift; timeout 900 python bench.py "$@" > $O/bench_$n.json 2> $O/bench_$n.err; }
b default
b cfg2 --config cfg2
b cfg5 --config cfg5
b msgs --config msgs
b frames --config frames
b frames_dec --config frames --decrypt --no-cpu-baseline
python - $O <<'PY'
import json,sys,glob,os
for p in sorted(glob.glob(sys.argv[1]+"/bench*.json")):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1]); r=d["roofline"]; c=d.get("cpu_baseline") or {}
        print("%-24s %.1f GiB/s step %.3f ms frac %.4f traffic %s (alg %s) tag_ok %s ceiling %s cpu %s" % (os.path.basename(p), d["value"], d["ms_per_step"], r["frac"], r.get("traffic"), r.get("alg_bytes_per_launch"), d["tag_ok"], r.get("achieved_over_ceiling"), c.get("value")))
    except Exception as e:
        print(p, "unreadable", e)
PY
