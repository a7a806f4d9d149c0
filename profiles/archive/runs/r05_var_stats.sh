# kernel times of the planned forms (offset arrays; addresses and lengths) against fixed-size records: 524 288 x 8 KiB, 262 144 x 16 KiB + 13 B, 65 536 x 64 KiB
mkdir -p gpurun_out/r05
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "524288 8192 0" "262144 16384 13" "65536 65536 0"; do set -- $cfg
  for form in fixed var scatter; do
    f2=""; [ $form = var ] && f2="--var"; [ $form = scatter ] && f2="--scatter"
    timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1_$2_$form -o x -- python3 $R/profiles/pkt_bench.py pkt $f2 --n $1 --len $2 --aad $3 --key-bits 256 --steps 20 > /tmp/out_$1_$2_$form.txt 2>&1 < /dev/null
    echo "== n=$1 len=$2 aad=$3 $form: $(grep -o '"gib_per_s_queued": [0-9.]*' /tmp/out_$1_$2_$form.txt | tail -1)"
    f=$(find /tmp/prof_$1_$2_$form -name '*kernel_stats.csv' | head -1)
    if [ -n "$f" ]; then python3 -c "
import csv,sys
for r in csv.reader(open('$f')):
    if r and 'k_rows' in r[0]: print('   %-22s calls %s  avg %.1f us' % (r[0].split('(')[0].replace('void ',''), r[1], float(r[3]) / 1e3))"; else echo 'no kernel_stats.csv'; fi
  done
done > $R/gpurun_out/r05/rows_var_stats.txt 2>&1
cat $R/gpurun_out/r05/rows_var_stats.txt
