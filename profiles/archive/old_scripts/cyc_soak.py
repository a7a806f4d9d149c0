#!/usr/bin/env python3
"""Soak of the in-launch closings (GPU box): N random messages (64 KiB .. 24 MiB, random AAD, odd lengths, random offsets) through a context whose cyclic
launch closes the tag itself and through one that keeps k_fold + k_combine behind it (AESGCM_CYC_FUSE=0); tags must agree call by call, and every 500th pair is
checked against the oracle-free third path (k_main / dealt, AESGCM_BODY_CYC=0:0).  Also N/10 dealt whole messages of 1 GiB + k x 4 KiB with and without FoldClose.
    python profiles/cyc_soak.py [N]"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aesgcm_amd  # noqa
from aesgcm_amd import lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
MiB = 1 << 20
rng = random.Random(424242)
key = bytes(rng.randrange(256) for _ in range(32))
fused = lib.Context(key)
os.environ["AESGCM_CYC_FUSE"] = "0"
three = lib.Context(key)
os.environ.pop("AESGCM_CYC_FUSE")
os.environ["AESGCM_BODY_CYC"] = "0:0"
old = lib.Context(key)
os.environ.pop("AESGCM_BODY_CYC")
nmax = 1056 * MiB
a, b1, b2 = lib.DeviceBuffer(nmax), lib.DeviceBuffer(nmax), lib.DeviceBuffer(nmax)
a.fill_splitmix64(7)
d_aad = lib.DeviceBuffer(8192); d_aad.upload(bytes(rng.randrange(256) for _ in range(8192)))
t0 = time.time()
bad = 0
for it in range(N):
    n = rng.choice((rng.randint(64 << 10, 24 * MiB), rng.randint(64 << 10, 2 * MiB), 1024 * rng.randint(64, 24576) + rng.choice((0, 1, 16, 1008, 1023))))
    al = rng.choice((0, 0, 0, 1, 16, 20, 1000, rng.randint(1, 8000)))
    off = rng.randrange(0, 64 * MiB, 16)
    iv = bytes(rng.randrange(256) for _ in range(12))
    t1 = fused.encrypt_dev(iv, a.ptr + off, n, b1.ptr, d_aad=d_aad.ptr if al else None, aad_len=al)
    t2 = three.encrypt_dev(iv, a.ptr + off, n, b2.ptr, d_aad=d_aad.ptr if al else None, aad_len=al)
    ok = t1 == t2
    if ok and it % 500 == 0:
        t3 = old.encrypt_dev(iv, a.ptr + off, n, b2.ptr, d_aad=d_aad.ptr if al else None, aad_len=al)
        ok = t3 == t1 and bytes(b1.download(min(n, 1 << 16), max(0, n - (1 << 16)))) == bytes(b2.download(min(n, 1 << 16), max(0, n - (1 << 16))))
    if not ok:
        bad += 1
        print("MISMATCH at", it, n, al, off, t1.hex(), t2.hex(), flush=True)
        if bad > 5: break
print("cyclic: %d messages, %d mismatches, %.0f s" % (it + 1, bad, time.time() - t0), flush=True)
os.environ["AESGCM_FOLD_CLOSE"] = "0"
nofc = lib.Context(key)
os.environ.pop("AESGCM_FOLD_CLOSE")
bad2 = 0
for it in range(max(10, N // 100)):
    n = 1024 * MiB + 4096 * rng.randint(0, 4096)
    iv = bytes(rng.randrange(256) for _ in range(12))
    t1 = fused.encrypt_dev(iv, a.ptr, n, b1.ptr)
    t2 = nofc.encrypt_dev(iv, a.ptr, n, b2.ptr)
    if t1 != t2:
        bad2 += 1; print("FOLDCLOSE MISMATCH", it, n, flush=True)
print("dealt + FoldClose: %d messages, %d mismatches" % (it + 1, bad2))
print("SOAK OK" if not bad and not bad2 else "SOAK FAILED")
