# 256 MiB per call as 4096 x 64 KiB, 256 x 1 MiB, 16 x 16 MiB: kernel times (rocprofv3 --kernel-trace --stats)
mkdir -p gpurun_out/r05
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "4096 65536" "256 1048576" "16 16777216"; do set -- $cfg
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1_$2 -o x -- python3 $R/profiles/pkt_bench.py pkt --n $1 --len $2 --key-bits 256 --steps 200 > /tmp/out_$1_$2.txt 2>&1 < /dev/null
  echo "== n=$1 len=$2: $(grep -o '"gib_per_s_queued": [0-9.]*' /tmp/out_$1_$2.txt | tail -1)"
  f=$(find /tmp/prof_$1_$2 -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then python3 -c "
import csv,sys
for r in csv.reader(open('$f')):
    if r and 'k_rows' in r[0]: print('   %-22s calls %s  avg %.1f us  min %.1f  max %.1f' % (r[0].split('(')[0].replace('void ',''), r[1], float(r[3]) / 1e3, float(r[5]) / 1e3, float(r[6]) / 1e3))"; else echo 'no kernel_stats.csv'; fi
done > $R/gpurun_out/r05/rows_few_large_stats.txt 2>&1
cat $R/gpurun_out/r05/rows_few_large_stats.txt
