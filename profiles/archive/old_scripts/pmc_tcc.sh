#!/bin/bash
# Run ON THE GPU BOX: request-size breakdown of the fused kernels' HBM reads (k_main vs k_body), one PMC pass each.
# bytes = TCC_BUBBLE*128 + (TCC_EA0_RDREQ - TCC_BUBBLE - TCC_EA0_RDREQ_32B)*64 + TCC_EA0_RDREQ_32B*32 if TCC_BUBBLE counts
# the 128-byte requests as on gfx942; printed raw so the calibration is visible.
OUT=$PWD/gpurun_out/pmc_tcc; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for v in main body; do
  if [ $v = main ]; then export AESGCM_BODY_MIN=99999999999999; else export AESGCM_BODY_MIN=4096; fi
  for C in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $C | tr ' ' '_')
    rocprofv3 --pmc $C --output-format csv -d $OUT/${v}_$tag -- python3 $REPO/profiles/kernel_mix.py 4096 > /dev/null 2> $OUT/${v}_$tag.err
  done
done
cd $REPO
python3 - <<'PY'
import csv, glob, collections, os
out = os.path.join(os.getcwd(), "gpurun_out", "pmc_tcc")
for d in sorted(glob.glob(out + "/*/")):
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].split("(")[0]
            if "k_main" in k or "k_body" in k:
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
        for k in acc:
            print(os.path.basename(d.rstrip("/")), k, {c: v / len(disp[k]) for c, v in acc[k].items()})
PY
