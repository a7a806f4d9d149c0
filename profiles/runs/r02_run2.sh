#!/bin/bash
O=gpurun_out/r02_run2; mkdir -p $O
NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,ENV,NET,GRAPH timeout 900 python profiles/rccl_probe.py > $O/probe_default.txt 2>&1
echo "=== second process, same box (files now cached)" >> $O/probe_default.txt
timeout 900 python profiles/rccl_probe.py >> $O/probe_default.txt 2>&1
grep "s\] " $O/probe_default.txt
