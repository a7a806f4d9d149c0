#!/bin/bash
# round 3, call 5: lane groups of 8 and 4 (k_pktg<.., 3> / <.., 2>) -- parity, sweep against the other shapes; where k_pktg's extra read traffic comes from
# (in place vs separate output); full GPU suite on the build without region B
O=gpurun_out/r03_run5; mkdir -p $O
export GIT_HEAD=$(cat .git_head 2>/dev/null)
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -5 $O/pytest.txt
timeout 1200 python profiles/packets_sweep.py 32 > $O/packets_sweep_aes256.txt 2>&1; cat $O/packets_sweep_aes256.txt
timeout 900 python profiles/packets_sweep.py 16 > $O/packets_sweep_aes128.txt 2>&1; cat $O/packets_sweep_aes128.txt
REPO=$PWD; cd /tmp; export TMPDIR=/tmp
for v in "" "--inplace"; do for kind in pktg pktg8 pktw; do
  tag=${kind}_1k$(echo $v | tr -d ' -')
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $REPO/$O/tcc_$tag -- python3 $REPO/profiles/pkt_bench.py $kind --len 1024 --key-bits 256 --steps 3 $v > $REPO/$O/tcc_$tag.json 2> $REPO/$O/tcc_$tag.err
  python3 - $REPO/$O/tcc_$tag $tag <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(float); disp=set()
for p in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "k_pkt" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
n=max(1,len(disp))
print(sys.argv[2], {k: round(v/n) for k,v in acc.items()}, "read bytes per launch %.4g (algorithmic 1.086e9)" % ((128*acc.get("TCC_EA0_RDREQ_128B_sum",0)+64*acc.get("TCC_EA0_RDREQ_64B_sum",0)+32*acc.get("TCC_EA0_RDREQ_32B_sum",0))/n))
PY
  rm -rf $REPO/$O/tcc_$tag
done; done
