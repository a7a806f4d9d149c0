"""CPU baseline of bench.py: the reference's CPU path timed on the GPU box's host cores.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (lives under oracle/; the product never imports it).

The reference's software path is `AES.new(key, AES.MODE_GCM, nonce=iv).encrypt(...)` from pycryptodome
(/root/reference/tb/gcm_model.py:18,26).  Probe order: pycryptodome if importable -- that exact call -- else the
system libcrypto `EVP_aes_256_gcm` through ctypes (oracle/libcrypto_ref.py), which is the substitution BASELINE.md
section 2 prescribes when the wheel is absent.  What is timed: the cfg3 plaintext stream (SplitMix64 seed 0xAE5C0003)
in 64 MiB chunks, (i) one core, one stream and (ii) all usable cores as worker PROCESSES, each encrypting its own
slice of the stream under its own IV; wall clock around the encrypt calls only (plaintext generated beforehand),
best and median of >= 3 repetitions.  A bounded sample (default <= 512 MiB resident per worker, passed over 8 times:
about 20-30 core-seconds in all); AES-GCM timing is data independent.

Must be called BEFORE the process initialises the GPU: the workers are forked.
"""
import multiprocessing as mp
import os
import statistics
import time

GiB = 1 << 30
CHUNK = 64 << 20
KEY_SEED, IV_SEED, PT_SEED = 0x4B4559, 0x4956, 0xAE5C0003


def usable_cores():
    """cores this process may really use: affinity mask capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_info():
    model, flags = "unknown", set()
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("flags") and not flags:
                flags = set(line.split(":", 1)[1].split())
                break
    except OSError:
        pass
    keep = [f for f in ("aes", "vaes", "pclmulqdq", "vpclmulqdq", "avx2", "avx512f") if f in flags]
    return model, keep


def probe_library():
    """-> (name, kind): pycryptodome (the reference's own dependency) if importable, else libcrypto"""
    try:
        import Crypto
        from Crypto.Cipher import AES  # noqa: F401
        return "pycryptodome %s (AES.new(key, AES.MODE_GCM, nonce=iv).encrypt -- tb/gcm_model.py:18,26)" % Crypto.__version__, "reference"
    except Exception:
        pass
    from oracle import libcrypto_ref as R
    if R.available():
        return "libcrypto EVP_aes_256_gcm via ctypes (%s); stands in for pycryptodome, which is not installed (BASELINE.md 2)" % R.version(), "library"
    return None, None


def _worker(idx, key, iv, n_chunks, passes, use_pycryptodome, reps, start, done, out_q):
    import numpy as np
    from oracle import oracle as O
    from oracle import libcrypto_ref as R
    # this worker's slice of the cfg3 stream, generated before the clock starts
    pts = [np.frombuffer(O.fill_splitmix64(CHUNK, PT_SEED, (idx * n_chunks + c) * (CHUNK // 8)), dtype=np.uint8) for c in range(n_chunks)]
    ct = np.empty(CHUNK, dtype=np.uint8)
    my_iv = bytes(iv[:8]) + (int.from_bytes(iv[8:], "big") + idx & 0xFFFFFFFF).to_bytes(4, "big")     # own IV per worker
    for _ in range(reps):
        start.wait()
        t0 = time.perf_counter()
        if use_pycryptodome:
            from Crypto.Cipher import AES
            m = AES.new(key, AES.MODE_GCM, nonce=my_iv)
            for _ in range(passes):
                for p in pts:
                    m.encrypt(p.tobytes())
            m.digest()
        else:
            s = R.Stream(key, my_iv)
            for _ in range(passes):                 # one GCM stream: `passes` times over the resident chunks
                for p in pts:
                    s.update(p, ct)
            s.final()
        dt = time.perf_counter() - t0
        done.wait()
        out_q.put((idx, dt))


def _run(n_workers, n_chunks, passes, key, iv, use_pycryptodome, reps):
    ctx = mp.get_context("fork")
    start, done, q = ctx.Barrier(n_workers + 1), ctx.Barrier(n_workers + 1), ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(i, key, iv, n_chunks, passes, use_pycryptodome, reps, start, done, q), daemon=True) for i in range(n_workers)]
    for p in ps:
        p.start()
    rates = []
    for _ in range(reps):
        start.wait(timeout=600)
        t0 = time.perf_counter()
        done.wait(timeout=600)
        wall = time.perf_counter() - t0
        for _ in range(n_workers):
            q.get(timeout=60)
        rates.append(n_workers * n_chunks * passes * CHUNK / wall)
    for p in ps:
        p.join(timeout=30)
    return rates


def measure(reps=3, max_bytes_per_worker=512 << 20, max_total=12 * GiB, passes=8, with_port=True):
    """-> the `cpu_baseline` object of the bench line."""
    from oracle import oracle as O
    name, kind = probe_library()
    cores = usable_cores()
    model, flags = cpu_info()
    if name is None:
        return {"error": "neither pycryptodome nor libcrypto available", "cores": cores, "cpu_model": model}
    key = bytes(O.fill_splitmix64(32, KEY_SEED))
    iv = bytes(O.fill_splitmix64(12, IV_SEED))
    pyc = kind == "reference"
    n_chunks = max(1, min(max_bytes_per_worker, max_total // cores) // CHUNK)
    r1 = _run(1, n_chunks, passes, key, iv, pyc, reps)
    rn = _run(cores, n_chunks, passes, key, iv, pyc, reps)
    out = {
        "value": round(max(rn) / GiB, 3), "unit": "GiB/s", "cores": cores, "kind": kind, "lib": name,
        "value_median": round(statistics.median(rn) / GiB, 3),
        "value_1core": round(max(r1) / GiB, 3), "value_1core_median": round(statistics.median(r1) / GiB, 3),
        "cpu_model": model, "cpu_flags": flags, "reps": reps,
        "sample": "AES-256-GCM, cfg3 plaintext stream (SplitMix64 seed 0xAE5C0003) in 64 MiB chunks: %d worker process(es) x %d MiB resident, "
                  "encrypted %d times over as one message per worker under its own IV (%.1f GiB per repetition in all); wall clock around "
                  "the encrypt calls, best of %d (median beside it)" % (cores, n_chunks * CHUNK >> 20, passes, cores * n_chunks * passes * CHUNK / GiB, reps),
    }
    if with_port:
        try:
            out["port"] = port_rate(cores)
        except Exception as e:           # never break the bench line
            out["port"] = {"error": repr(e)}
    return out


def port_rate(n_threads, per_thread=32 << 20):
    """the oracle's own table-driven C restatement (oracle/aesgcm_oracle.c orc_fast_*), threads; a side note, not the baseline"""
    import threading
    import numpy as np
    from oracle import oracle as O
    key = bytes(O.fill_splitmix64(32, KEY_SEED))
    iv = bytes(O.fill_splitmix64(12, IV_SEED))
    bufs = [np.frombuffer(O.fill_splitmix64(per_thread, PT_SEED, t * per_thread // 8), dtype=np.uint8) for t in range(n_threads)]
    outs = [np.empty_like(b) for b in bufs]

    def run(t):
        O.Fast(key).crypt(False, iv, b"", bufs[t], outs[t])

    ths = [threading.Thread(target=run, args=(t,)) for t in range(n_threads)]
    t0 = time.perf_counter()
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    dt = time.perf_counter() - t0
    return {"value": round(n_threads * per_thread / dt / GiB, 3), "unit": "GiB/s", "threads": n_threads,
            "what": "oracle/aesgcm_oracle.c orc_fast (plain C tables, no AES-NI): the checker's own speed, not the baseline"}


# ---------------------------------------------------------------- BASELINE config 5: per-packet key and IV
_HERE = os.path.dirname(os.path.abspath(__file__))


def evp_batch_lib():
    """oracle/libevpbatch.so (oracle/evp_batch.c: the per-packet EVP loop in C, linked against the system libcrypto)"""
    import ctypes
    import subprocess
    so = os.path.join(_HERE, "libevpbatch.so")
    src = os.path.join(_HERE, "evp_batch.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "evp"])
    L = ctypes.CDLL(so)
    vp, sz = ctypes.c_void_p, ctypes.c_size_t
    L.evp_batch_encrypt.argtypes = [sz, sz, vp, vp, vp, sz, vp, vp]
    L.evp_frames_crypt.argtypes = [sz, sz, vp, vp, vp, vp, ctypes.c_uint64, vp, vp, ctypes.c_uint64, vp, vp]
    return L


def cfg5_inputs(first, n, pkt_len=4096, key_len=16):
    """packets [first, first + n) of the cfg5 streams as numpy arrays: keys (stream 0x4B4559, key_len bytes each), IVs (the
    first 12 bytes of every 16 of stream 0x4956), plaintext (stream 0xAE5C0005)"""
    import numpy as np
    from oracle import oracle as O
    keys = np.frombuffer(O.fill_splitmix64(key_len * n, KEY_SEED, first * key_len // 8), dtype=np.uint8)
    ivw = np.frombuffer(O.fill_splitmix64(16 * n, IV_SEED, first * 2), dtype=np.uint8).reshape(n, 16)
    ivs = np.ascontiguousarray(ivw[:, :12]).reshape(-1)
    pt = np.frombuffer(O.fill_splitmix64(pkt_len * n, 0xAE5C0005, first * pkt_len // 8), dtype=np.uint8)
    return keys, ivs, pt


def _worker5(idx, n, pkt_len, key_len, passes, reps, start, done, out_q):
    import numpy as np
    L = evp_batch_lib()
    keys, ivs, pt = cfg5_inputs(idx * n, n, pkt_len, key_len)
    ct, tags = np.empty(pkt_len * n, dtype=np.uint8), np.empty(16 * n, dtype=np.uint8)
    for _ in range(reps):
        start.wait()
        t0 = time.perf_counter()
        for _ in range(passes):
            rc = L.evp_batch_encrypt(n, key_len, keys.ctypes.data, ivs.ctypes.data, pt.ctypes.data, pkt_len, ct.ctypes.data, tags.ctypes.data)
        dt = time.perf_counter() - t0
        done.wait()
        out_q.put((idx, dt, rc))


def _run5(n_workers, n, pkt_len, key_len, passes, reps):
    ctx = mp.get_context("fork")
    start, done, q = ctx.Barrier(n_workers + 1), ctx.Barrier(n_workers + 1), ctx.Queue()
    ps = [ctx.Process(target=_worker5, args=(i, n, pkt_len, key_len, passes, reps, start, done, q), daemon=True) for i in range(n_workers)]
    for p in ps:
        p.start()
    rates = []
    for _ in range(reps):
        start.wait(timeout=600)
        t0 = time.perf_counter()
        done.wait(timeout=600)
        wall = time.perf_counter() - t0
        for _ in range(n_workers):
            if q.get(timeout=60)[2] != 0:
                raise RuntimeError("evp_batch_encrypt failed")
        rates.append(n_workers * n * passes / wall)            # packets per second
    for p in ps:
        p.join(timeout=30)
    return rates


def measure_cfg5(reps=3, pkts_per_worker=1 << 15, passes=32, pkt_len=4096, key_len=16):
    """-> the `cpu_baseline` object of the cfg5 bench line: libcrypto, one EVP init (key + IV) per packet, the loop in C.
    A bounded sample: every worker encrypts `pkts_per_worker` packets of the cfg5 streams `passes` times per repetition."""
    from oracle import libcrypto_ref as R
    cores = usable_cores()
    model, flags = cpu_info()
    if not R.available():
        return {"error": "libcrypto not available", "cores": cores, "cpu_model": model}
    r1 = _run5(1, pkts_per_worker, pkt_len, key_len, passes, reps)
    rn = _run5(cores, pkts_per_worker, pkt_len, key_len, passes, reps)
    to_gib = pkt_len / GiB
    return {
        "value": round(max(rn) * to_gib, 3), "unit": "GiB/s", "cores": cores, "kind": "library",
        "lib": "libcrypto EVP_aes_%d_gcm (%s), one EVP_EncryptInit_ex(key, iv) per packet, per-packet loop in C (oracle/evp_batch.c); stands in for "
               "pycryptodome, which is not installed (BASELINE.md 2)" % (8 * key_len, R.version()),
        "value_median": round(statistics.median(rn) * to_gib, 3), "mpkt_per_s": round(max(rn) / 1e6, 3),
        "value_1core": round(max(r1) * to_gib, 3), "value_1core_median": round(statistics.median(r1) * to_gib, 3), "mpkt_per_s_1core": round(max(r1) / 1e6, 3),
        "cpu_model": model, "cpu_flags": flags, "reps": reps,
        "sample": "AES-%d-GCM, cfg5 streams (keys 0x4B4559, IVs 0x4956, plaintext 0xAE5C0005): %d worker process(es) x %d packets of %d B, each packet "
                  "under its own key and IV, encrypted %d times over per repetition (%.2f GiB per repetition in all); wall clock around the calls, "
                  "best of %d (median beside it)" % (8 * key_len, cores, pkts_per_worker, pkt_len, passes, cores * pkts_per_worker * passes * pkt_len / GiB, reps),
    }


# ---------------------------------------------------------------- frames under one key (bench.py --config frames)
FRAME_LEN_SEED, FRAME_AAD = 0x4C454E, 28


def frame_lengths(first, n):
    """lengths of frames [first, first + n) of the frames workload: 64 + (word mod 1451) of SplitMix64 stream 0x4C454E -- uniform over 64 .. 1514, MACsec-shaped
    (the reference's README vectors are 802.1AE frames, README.md:251-257)"""
    import numpy as np
    from oracle import oracle as O
    w = np.frombuffer(O.fill_splitmix64(8 * n, FRAME_LEN_SEED, first), dtype="<u8")
    return (64 + (w % np.uint64(1451))).astype(np.int64)


def _worker_frames(idx, n, key_len, passes, reps, start, done, out_q):
    import numpy as np
    from oracle import oracle as O
    L = evp_batch_lib()
    lens = frame_lengths(idx * n, n)
    doff = np.zeros(n + 1, dtype=np.uint64); doff[1:] = np.cumsum(lens)
    aoff = (np.arange(n + 1, dtype=np.uint64) * np.uint64(FRAME_AAD))
    total = int(doff[-1])
    key = np.frombuffer(O.fill_splitmix64(key_len, KEY_SEED, 0), dtype=np.uint8)
    ivw = np.frombuffer(O.fill_splitmix64(16 * n, IV_SEED, idx * n * 2), dtype=np.uint8).reshape(n, 16)
    ivs = np.ascontiguousarray(ivw[:, :12]).reshape(-1)
    pt = np.frombuffer(O.fill_splitmix64((total + 7) // 8 * 8, 0xAE5C0006, 0), dtype=np.uint8)
    aad = np.frombuffer(O.fill_splitmix64(FRAME_AAD * n + 8, 0x414144, 0), dtype=np.uint8)
    ct, tags = np.empty(total + 16, dtype=np.uint8), np.empty(16 * n, dtype=np.uint8)
    for _ in range(reps):
        start.wait()
        t0 = time.perf_counter()
        for _ in range(passes):
            rc = L.evp_frames_crypt(n, key_len, key.ctypes.data, ivs.ctypes.data, aad.ctypes.data, aoff.ctypes.data, 0, pt.ctypes.data, doff.ctypes.data, 0, ct.ctypes.data, tags.ctypes.data)
        dt = time.perf_counter() - t0
        done.wait()
        out_q.put((idx, dt, rc, total))


def _run_frames(n_workers, n, key_len, passes, reps):
    ctx = mp.get_context("fork")
    start, done, q = ctx.Barrier(n_workers + 1), ctx.Barrier(n_workers + 1), ctx.Queue()
    ps = [ctx.Process(target=_worker_frames, args=(i, n, key_len, passes, reps, start, done, q), daemon=True) for i in range(n_workers)]
    for p in ps:
        p.start()
    rates = []
    for _ in range(reps):
        start.wait(timeout=600)
        t0 = time.perf_counter()
        done.wait(timeout=600)
        wall = time.perf_counter() - t0
        total = 0
        for _ in range(n_workers):
            r = q.get(timeout=60)
            if r[2] != 0:
                raise RuntimeError("evp_frames_crypt failed")
            total += r[3]
        rates.append((total * passes / wall, n_workers * n * passes / wall))     # bytes per second, frames per second
    for p in ps:
        p.join(timeout=30)
    return rates


def measure_frames(reps=3, frames_per_worker=1 << 16, passes=16, key_len=32):
    """-> the `cpu_baseline` object of the frames bench line: libcrypto, ONE key, one EVP init (IV) per frame, AAD + data + tag per frame, the loop in C
    (oracle/evp_batch.c evp_frames_crypt).  A bounded sample: every worker encrypts `frames_per_worker` frames of the workload's streams `passes` times per repetition."""
    from oracle import libcrypto_ref as R
    cores = usable_cores()
    model, flags = cpu_info()
    if not R.available():
        return {"error": "libcrypto not available", "cores": cores, "cpu_model": model}
    r1 = _run_frames(1, frames_per_worker, key_len, passes, reps)
    rn = _run_frames(cores, frames_per_worker, key_len, passes, reps)
    best = max(rn)
    return {
        "value": round(best[0] / GiB, 3), "unit": "GiB/s", "cores": cores, "kind": "library",
        "lib": "libcrypto EVP_aes_%d_gcm (%s), one key, one EVP_EncryptInit_ex(iv) + AAD + data + tag per frame, per-frame loop in C (oracle/evp_batch.c); stands in for "
               "pycryptodome, which is not installed (BASELINE.md 2)" % (8 * key_len, R.version()),
        "value_median": round(statistics.median(x[0] for x in rn) / GiB, 3), "mframes_per_s": round(best[1] / 1e6, 3),
        "value_1core": round(max(r1)[0] / GiB, 3), "mframes_per_s_1core": round(max(r1)[1] / 1e6, 3),
        "cpu_model": model, "cpu_flags": flags, "reps": reps,
        "sample": "AES-%d-GCM, frames of 64 .. 1514 bytes (lengths stream 0x4C454E) with %d bytes of AAD each under one key: %d worker process(es) x %d frames, encrypted "
                  "%d times over per repetition; wall clock around the calls, best of %d (median beside it)" % (8 * key_len, FRAME_AAD, cores, frames_per_worker, passes, reps),
    }


if __name__ == "__main__":
    import json
    print(json.dumps(measure(), indent=1))
