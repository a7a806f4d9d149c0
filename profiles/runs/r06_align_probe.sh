O=$PWD/gpurun_out/r06_align; mkdir -p $O; : > $O/a.txt
for r in 1 2; do for f in 1024 1040 1028 1025 1152 1088; do for x in "" "--probe"; do
echo -n "fixed $f $x: " >> $O/a.txt; timeout 300 python3 profiles/frames_one.py --steps 12 --fixed $f --aad 0 $x 2>>$O/err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_median'], d['gib_per_s'])" >> $O/a.txt
done; done; done; cat $O/a.txt
