# kernel times of the row path at the small end of its range (rocprofv3 --kernel-trace --stats): 524 288 x 8 KiB, 262 144 x 16 KiB, 16 384 x 2 KiB
mkdir -p gpurun_out/r05
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "524288 8192 0" "262144 16384 13" "16384 2048 0" "262144 9000 13"; do set -- $cfg
  kind=pkt; if [ $2 = 9000 ]; then kind=rows; fi
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1_$2 -o x -- python3 $R/profiles/pkt_bench.py $kind --n $1 --len $2 --aad $3 --key-bits 256 --steps 20 > /tmp/out_$1_$2.txt 2>&1 < /dev/null
  echo "== n=$1 len=$2 aad=$3 ($kind): $(grep -o '"gib_per_s_queued": [0-9.]*' /tmp/out_$1_$2.txt | tail -1)"
  f=$(find /tmp/prof_$1_$2 -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then python3 -c "
import csv,sys
for r in csv.reader(open('$f')):
    if r and 'k_rows' in r[0]: print('   %-22s calls %s  avg %.1f us' % (r[0].split('(')[0].replace('void ',''), r[1], float(r[3]) / 1e3))"; else echo 'no kernel_stats.csv'; fi
done > $R/gpurun_out/r05/rows_small_end_stats.txt 2>&1
cat $R/gpurun_out/r05/rows_small_end_stats.txt
