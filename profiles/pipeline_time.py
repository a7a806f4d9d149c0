#!/usr/bin/env python3
"""PCIe-inclusive rate of the pipelined host-buffer path (GPU box): 4 GiB AES-256-GCM from/to pinned host memory."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib
n = int(float(sys.argv[1]) * (1 << 30)) if len(sys.argv) > 1 else 4 << 30
ctx = lib.Context(bytes(range(32)))
src, dst = lib.PinnedBuffer(n), lib.PinnedBuffer(n)
for chunk in (16 << 20, 64 << 20, 256 << 20):
    for it in range(3):
        t0 = time.perf_counter()
        _, tag = ctx.encrypt_pipelined(bytes(12), b"", src.view, out=dst.view, chunk_bytes=chunk)
        dt = time.perf_counter() - t0
    print("pinned, chunk %3d MiB: %.1f ms  %.1f GiB/s plaintext (H2D + kernel + D2H overlapped)" % (chunk >> 20, dt * 1e3, n / dt / (1 << 30)))
page_in, page_out = bytearray(n), bytearray(n)
t0 = time.perf_counter(); ctx.encrypt_pipelined(bytes(12), b"", page_in, out=page_out, chunk_bytes=64 << 20); dt = time.perf_counter() - t0
print("pageable, chunk 64 MiB: %.1f ms  %.1f GiB/s" % (dt * 1e3, n / dt / (1 << 30)))
t0 = time.perf_counter(); ctx.encrypt(bytes(12), b"", page_in, out=page_out); dt = time.perf_counter() - t0
print("one-shot aesgcm_encrypt (no overlap, pageable): %.1f ms  %.1f GiB/s" % (dt * 1e3, n / dt / (1 << 30)))
