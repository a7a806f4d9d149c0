#!/usr/bin/env python3
"""Beat-by-beat drop-in class (gcm_model.gcm): 16-byte beats per second (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import gcm_model
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
hx = lambda b: {'data': b.hex() or '0', 'n_bytes': len(b)}
key, iv, pt = bytes(range(32)), bytes(12), bytes(n)
for rep in range(2):
    m = gcm_model.gcm(hx(key), hx(iv), 'enc')
    t0 = time.perf_counter()
    for o in range(0, n, 16):
        m.load_plain_text(pt[o:o + 16])
    m.get_tag(bytes(16))
    dt = time.perf_counter() - t0
print("%d beats in %.3f s: %.0f beats/s, %.2f MiB/s; tag %s" % (n // 16, dt, n / 16 / dt, n / dt / (1 << 20), m.tag[0].hex()))
print("one-shot tag %s" % gcm_model.encrypt(key, iv, b"", pt)[1].hex())
