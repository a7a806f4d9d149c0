mkdir -p gpurun_out/r05
for rb in 32 64 128 256 512; do
  echo "16384 x 1 MiB rows_block=$rb $(timeout 200 python profiles/pkt_bench.py pkt --n 16384 --len 1048576 --key-bits 256 --steps 7 --rows-block $rb | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["gib_per_s_queued"], d["gib_per_s"], d["shape"])')"
done > gpurun_out/r05/rows_block_sweep_16g.txt 2>&1
echo "cfg3 one 16 GiB message: $(python bench.py --steps 7 --warmup 3 2>&1 | tail -1 | cut -c1-200)" >> gpurun_out/r05/rows_block_sweep_16g.txt
cat gpurun_out/r05/rows_block_sweep_16g.txt
