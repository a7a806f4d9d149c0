# A/B of two builds of the library on one box: kernel times of the row path (rocprofv3 --kernel-trace --stats), tail-only packets and 64 KiB messages
mkdir -p gpurun_out/r05/rows_ab2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in old new; do
  if [ $lib = old ]; then export AESGCM_LIB=$R/profiles/lab/libold_dbg.so; else export AESGCM_LIB=$R/aes-gcm-128-192-256-bits_amd/libaesgcm_hip.so; fi
  for L in 16 4096 65536; do
    n=262144; if [ $L -ge 65536 ]; then n=65536; fi
    timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${lib}_$L -o x -- python3 $R/profiles/pkt_bench.py pkt --opt rows_min=16 --n $n --len $L --key-bits 256 --steps 20 > /tmp/out_${lib}_$L.txt 2>&1
    echo "== $lib len=$L: $(cut -c1-160 /tmp/out_${lib}_$L.txt | tail -1)"
    f=$(find /tmp/prof_${lib}_$L -name '*kernel_stats.csv' | head -1)
    if [ -n "$f" ]; then head -6 "$f" | cut -d, -f1-8; else echo 'no kernel_stats.csv'; fi < /dev/null
  done
done > $R/gpurun_out/r05/rows_ab2/summary.txt 2>&1
cat $R/gpurun_out/r05/rows_ab2/summary.txt
