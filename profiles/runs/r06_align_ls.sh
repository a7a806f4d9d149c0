O=$PWD/gpurun_out/r06_align_ls; mkdir -p $O; : > $O/a.txt
for r in 1 2; do for f in 1024 1028 1025; do for v in base probe NO_LOADS NO_STORES; do
case $v in base) L=; X=;; probe) L=; X=--probe;; *) L=$PWD/experiments/libaesgcm_$v.so; X=;; esac
echo -n "fixed $f $v: " >> $O/a.txt; AESGCM_LIB=$L timeout 300 python3 profiles/frames_one.py --steps 12 --fixed $f --aad 0 $X 2>>$O/err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_median'])" >> $O/a.txt
done; done; done; cat $O/a.txt
