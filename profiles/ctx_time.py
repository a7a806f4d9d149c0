#!/usr/bin/env python3
"""What a key costs (GPU box): creating a context, destroying it, loading a new key into an existing one (aesgcm_ctx_rekey).  Median of 20, microseconds."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib
lib.Context(bytes(32)).close()
cr, de, rk = [], [], []
keep = lib.Context(bytes(32))
for i in range(20):
    t0 = time.perf_counter(); c = lib.Context(bytes([i]) * 32); t1 = time.perf_counter(); c.close(); t2 = time.perf_counter()
    keep.rekey(bytes([i + 1]) * 32); t3 = time.perf_counter()
    cr.append(t1 - t0); de.append(t2 - t1); rk.append(t3 - t2)
print("aesgcm_ctx_create %.0f us, aesgcm_ctx_destroy %.0f us, aesgcm_ctx_rekey %.0f us (medians of 20, AES-256)" % tuple(statistics.median(x) * 1e6 for x in (cr, de, rk)))
