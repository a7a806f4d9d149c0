#!/bin/bash
# round 4, call 27: packets of mixed length -- the library's own rule (by length class from 98304 packets, shape rule aware of it) in the `auto` column against
# always / never, both key sizes; parity of the packet paths first
O=$PWD/gpurun_out/r04_run27; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_batch.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -3 $O/pytest.txt
timeout 900 python3 profiles/packets_sweep.py 32 var 2>&1 | tee $O/packets_sweep_mixed_aes256.txt
timeout 900 python3 profiles/packets_sweep.py 32 var noorder 2>&1 | tee $O/packets_sweep_mixed_aes256_array_order.txt
timeout 900 python3 profiles/packets_sweep.py 32 var order 2>&1 | tee $O/packets_sweep_mixed_aes256_by_class.txt
timeout 900 python3 profiles/packets_sweep.py 16 var 2>&1 | tee $O/packets_sweep_mixed_aes128.txt
timeout 900 python3 profiles/packets_sweep.py 16 var noorder 2>&1 | tee $O/packets_sweep_mixed_aes128_array_order.txt
