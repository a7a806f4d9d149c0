#!/usr/bin/env python3
"""GMAC shape: a large AAD and no (or little) data through aesgcm_encrypt_dev (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
ctx = lib.Context(bytes(range(32)))
d_aad = lib.DeviceBuffer(n + 64); d_aad.fill_splitmix64(7)
d_pt, d_ct = lib.DeviceBuffer(4096), lib.DeviceBuffer(4096)
iv = bytes(12)
for name, off, dl in (("aligned AAD, no data", 0, 0), ("AAD base + 1 (unaligned), no data", 1, 0), ("aligned AAD, 4 KiB data", 0, 4096)):
    for it in range(3):
        lib.dev_sync(); t0 = time.perf_counter()
        tag = ctx.encrypt_dev(iv, d_pt.ptr, dl, d_ct.ptr, d_aad=d_aad.ptr + off, aad_len=n)
        dt = time.perf_counter() - t0
    print("%-36s %8.3f ms  %7.1f GiB/s  tag %s" % (name, dt * 1e3, n / dt / (1 << 30), tag.hex()))
