cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_placed; mkdir -p $O
for x in "" "--placed 64"; do
rm -rf $O/prof; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/profiles/frames_one.py --steps 12 $x > $O/out.txt 2>&1
tail -1 $O/out.txt | cut -c1-100
python3 - "$(find $O/prof -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["AverageNs"])>4000 and not r["Name"].startswith(("k_fill","k_setup","__amd")): print("   %-56s calls %4s avg %10.1f ns" % (r["Name"][:56], r["Calls"], float(r["AverageNs"])))
PY
done
