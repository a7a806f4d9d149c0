cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_sort1; mkdir -p $O
for n in 2048 4096 8192 16384; do
rm -rf $O/prof; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/profiles/frames_one.py --n $n --steps 30 > $O/out_$n.txt 2>&1
tail -1 $O/out_$n.txt | cut -c1-120
python3 - "$(find $O/prof -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith(("k_len_","void k_pktg","void k_pktl")) and float(r["AverageNs"])>3000: print("   %-50s calls %4s avg %9.1f ns" % (r["Name"][:50], r["Calls"], float(r["AverageNs"])))
PY
done
