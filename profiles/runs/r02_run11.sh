#!/bin/bash
O=gpurun_out/r02_run11; mkdir -p $O
examples/latency 500 | tee $O/latency_c.txt
timeout 300 python profiles/latency.py 300 | tee $O/latency_py.txt
timeout 1200 python -m pytest tests -m gpu -x -q -k "not cfg4 and not 16GiB and not maximum" 2>&1 | tail -4
