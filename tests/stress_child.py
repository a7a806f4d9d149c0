"""Child process of tests/test_gpu_stress.py: N randomised packet-kernel launches, a sample verified against the oracle.
Prints progress lines and one JSON result line.  usage: stress_child.py <launches> <seed>
Runs on the debug build of the library (libaesgcm_hip_dbg.so: the same sources plus aesgcm_debug_force_shape), because it forces
every packet-kernel shape and deal size on every input; "pkt_auto" and "batch" leave the choice to the library's own rule."""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    n_launch, seed = int(sys.argv[1]), int(sys.argv[2])
    import numpy as np
    import aesgcm_amd  # noqa: F401
    from aesgcm_amd import lib
    from oracle import oracle as O
    rng = random.Random(seed)
    nprng = np.random.default_rng(seed)
    with lib.debug_library() as dbg:
        return run(n_launch, rng, nprng, np, lib, O, dbg)


def run(n_launch, rng, nprng, np, lib, O, dbg):
    MAXP, MAXB = 4200, 1 << 20                       # packets per launch, bytes per launch
    d_in, d_out = lib.DeviceBuffer(MAXB + 4096), lib.DeviceBuffer(MAXB + 4096)
    d_aad = lib.DeviceBuffer(MAXP * 32 + 64)
    d_ivs, d_keys, d_tags = lib.DeviceBuffer(12 * MAXP), lib.DeviceBuffer(32 * MAXP), lib.DeviceBuffer(16 * MAXP)
    d_doff, d_aoff = lib.DeviceBuffer(8 * (MAXP + 1)), lib.DeviceBuffer(8 * (MAXP + 1))
    data = nprng.integers(0, 256, MAXB + 4096, dtype=np.uint8)
    aadb = nprng.integers(0, 256, MAXP * 32 + 64, dtype=np.uint8)
    d_in.upload(data); d_aad.upload(aadb)
    keys = {kb: bytes(nprng.integers(0, 256, kb, dtype=np.uint8)) for kb in (16, 24, 32)}
    ctxs = {kb: lib.Context(k) for kb, k in keys.items()}
    orcs = {kb: O.Fast(k) for kb, k in keys.items()}
    counts = [1, 2, 63, 64, 65, 127, 129, 1000, 4097, 4200]
    launches = verified = mismatches = 0
    t0 = time.time()
    while launches < n_launch:
        n = rng.choice(counts) if rng.random() < 0.5 else rng.randrange(1, 300)
        kind = rng.choice(("pkt_w", "pkt_g", "pkt_g8", "pkt_g4", "pkt_l", "pkt_auto", "batch", "batch"))
        kb = rng.choice((16, 24, 32))
        # packet lengths: many zero-length and tiny ones, a few long; total bounded by MAXB
        mean = max(1, min(2000, MAXB // n))
        lens = [0 if rng.random() < 0.08 else min(rng.randrange(0, 2 * mean), 60000) for _ in range(n)]
        if rng.random() < 0.3:
            lens = [(x + 15) // 16 * 16 for x in lens]                       # aligned packets take the fast path
        alens = [rng.choice((0, 0, 1, 12, 16, 20, 28)) for _ in range(n)]
        doff = np.zeros(n + 1, dtype=np.uint64); doff[1:] = np.cumsum(lens)
        aoff = np.zeros(n + 1, dtype=np.uint64); aoff[1:] = np.cumsum(alens)
        if int(doff[-1]) > MAXB:
            continue
        ivs = nprng.integers(0, 256, 12 * n, dtype=np.uint8)
        d_ivs.upload(ivs); d_doff.upload(doff); d_aoff.upload(aoff)
        dbg.force(pkt_lanes=0, pkt_deal=0, batch_lanes=0)
        if kind == "batch":
            pk = nprng.integers(0, 256, kb * n, dtype=np.uint8)
            d_keys.upload(pk)
            dbg.force(batch_lanes=rng.choice((0, 0, 8, 16, 64)))
            lib.batch_crypt_var_dev(False, n, kb, d_keys.ptr, d_ivs.ptr, d_in.ptr, d_doff.ptr, d_out.ptr, d_tags.ptr,
                                    d_aad=d_aad.ptr, d_aad_off=d_aoff.ptr)
        else:
            dbg.force(pkt_lanes={"pkt_w": 64, "pkt_g": 16, "pkt_g8": 8, "pkt_g4": 4, "pkt_l": 1, "pkt_auto": 0}[kind], pkt_deal=rng.choice((0, 1, 3, 16, 64)))
            ctxs[kb].packets_crypt_dev(False, n, d_ivs.ptr, d_in.ptr, d_out.ptr, d_tags.ptr, d_aad=d_aad.ptr,
                                       d_aad_off=d_aoff.ptr, d_data_off=d_doff.ptr)
        launches += 1
        if launches % 40 == 0 or launches == n_launch:                        # verify this launch: a few packets of it
            lib.dev_sync(0)
            tags = bytes(d_tags.download(16 * n))
            for p in {0, n - 1, rng.randrange(n), rng.randrange(n)}:
                a, b = int(doff[p]), int(doff[p + 1])
                ct = bytes(d_out.download(b - a, a)) if b > a else b""
                iv = bytes(ivs[12 * p:12 * p + 12])
                aad = bytes(aadb[int(aoff[p]):int(aoff[p + 1])])
                f = O.Fast(bytes(pk[kb * p:kb * (p + 1)])) if kind == "batch" else orcs[kb]
                want_ct, want_tag = f.encrypt(iv, aad, bytes(data[a:b]))
                verified += 1
                if ct != want_ct or tags[16 * p:16 * p + 16] != want_tag:
                    mismatches += 1
                    print("MISMATCH launch %d kind %s kb %d n %d pkt %d len %d aad %d" % (launches, kind, kb, n, p, b - a, len(aad)), flush=True)
        if launches % 1000 == 0:
            lib.dev_sync(0)
            print("progress %d launches, %d verified, %.0f s" % (launches, verified, time.time() - t0), flush=True)
    lib.dev_sync(0)
    print(json.dumps({"launches": launches, "verified": verified, "mismatches": mismatches, "seconds": round(time.time() - t0, 1)}), flush=True)
    return 0 if mismatches == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
