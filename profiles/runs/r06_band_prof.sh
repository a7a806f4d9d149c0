cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_band; mkdir -p $O
for w in lib pkt; do
rm -rf $O/prof_$w; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -- python3 $R/profiles/route_sweep.py --kinds band8_16 --counts 393216 --only $w > $O/only_$w.txt 2>&1
cat $O/only_$w.txt | tail -1
python3 - "$(find $O/prof_$w -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["AverageNs"])>3000 and not r["Name"].startswith("k_fill"): print("   %-62s calls %4s avg %10.1f ns" % (r["Name"][:62], r["Calls"], float(r["AverageNs"])))
PY
done
