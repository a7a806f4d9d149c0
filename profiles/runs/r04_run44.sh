#!/bin/bash
# round 4, call 44: whole blocks of packets that start at any byte address as ONE 16-byte access (gload16_any / gstore16_any: the target runs with unaligned
# access mode on) instead of sixteen byte loads and stores: parity of every packet path (offsets aligned or not), then frames packed back to back, frames at
# aligned starts and fixed-size records over the shapes
O=$PWD/gpurun_out/r04_run44; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_batch.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -3 $O/pytest.txt
timeout 600 python3 profiles/packets_sweep.py 32 packed 2>&1 | tee $O/packets_sweep_packed_aes256.txt
timeout 600 python3 profiles/packets_sweep.py 32 var 2>&1 | tee $O/packets_sweep_mixed_aes256.txt
timeout 600 python3 profiles/packets_sweep.py 32 2>&1 | tee $O/packets_sweep_aes256.txt
