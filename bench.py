#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json on MI355X:

    GiB/s of plaintext, AES-256-GCM, 16 GiB stream per GPU, bit-exact tag

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (fused AES-CTR + GHASH kernel, combine kernel, 16-byte tag to
host) over the whole resident workload.

  N = 1   configs[2] of BASELINE.json ("cfg3"): ONE AES-256-GCM message of 16 GiB, SplitMix64
          plaintext seed 0xAE5C0003, already resident in HBM; ciphertext written to a second 16 GiB
          buffer.  The tag of the first step is checked against tests/golden/streams.json.
  N > 1   weak scaling, 16 GiB per GPU: the aggregate N x 16 GiB is configs[3] ("cfg4") cut to N
          ranks: messages of 32 GiB (one GCM message cannot exceed 64 GiB - 32 B, aes_icb.vhd:114)
          from ONE SplitMix64 stream (seed 0xAE5C0004), message m = bytes [m*32 GiB, (m+1)*32 GiB),
          IV last byte + m.  Every message is sharded over ALL ranks (rank r owns the r-th 1/N of
          its blocks); per message each rank produces a 16-byte weighted GHASH partial; ONE RCCL
          all-gather per step moves N x M x 16 bytes; every rank folds and finalises the tags.
          There is no other inter-GPU traffic.  Tags are checked against the cfg4 fixtures.

PyTorch is used only when N > 1, for torch.distributed (backend "nccl" = RCCL) and the 16-byte
partial tensors; the compute path is libaesgcm_hip.so through ctypes.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GiB = 1 << 30
HBM_PEAK_BYTES_PER_S = 8.0e12          # MI355X HBM3E peak (MI355X_MICROARCH.md)
KEY_SEED, IV_SEED = 0x4B4559, 0x4956


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def load_fixture(name):
    try:
        with open(os.path.join(ROOT, "tests", "golden", "streams.json")) as f:
            for c in json.load(f)["cases"]:
                if c["name"] == name:
                    return c
    except OSError:
        pass
    return None


def cpu_baseline(n_threads, budget_s=12.0):
    """The oracle's table-driven port (oracle/aesgcm_oracle.c, orc_fast_*) and, beside it, the strongest
    external AES-GCM on the box (pycryptodome if importable -- the reference's own dependency --
    else system libcrypto), timed on a bounded sample of the same workload (AES-256-GCM, SplitMix64
    plaintext seed 0xAE5C0003): every thread encrypts its own slice as an independent message."""
    import threading
    import numpy as np
    from oracle import oracle as O
    from oracle import libcrypto_ref as R

    key = bytes(O.fill_splitmix64(32, KEY_SEED))
    iv = bytes(O.fill_splitmix64(12, IV_SEED))

    def run(fn, per_thread, threads):
        bufs = []
        for t in range(threads):
            pt = np.frombuffer(O.fill_splitmix64(per_thread, 0xAE5C0003, t * per_thread // 8), dtype=np.uint8)
            bufs.append((pt, np.empty_like(pt)))
        ths = [threading.Thread(target=fn, args=(bufs[t][0], bufs[t][1])) for t in range(threads)]
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        return threads * per_thread / (time.perf_counter() - t0)

    def port(pt, ct):
        f = O.Fast(key)
        f.crypt(False, iv, b"", pt, ct)

    def ext(pt, ct):
        s = R.Stream(key, iv)
        s.update(pt, ct)
        s.final()

    # calibrate on 8 MiB / thread, then size the sample for ~budget_s/2 seconds each
    rate = run(port, 8 << 20, n_threads)
    per = int(min(1 << 30, max(8 << 20, rate * (budget_s * 0.6) / n_threads)) // 16 * 16)
    rate = run(port, per, n_threads)
    out = {"value": round(rate / GiB, 4), "unit": "GiB/s", "cores": n_threads, "kind": "port",
           "sample": "oracle orc_fast (table AES + 8-bit-table GHASH, plain C): %d threads x %d MiB of the cfg3 "
                     "plaintext stream, each slice an independent AES-256-GCM message" % (n_threads, per >> 20)}
    try:
        if R.available():
            name, _ = R.best()
            if name.startswith("pycryptodome"):
                from Crypto.Cipher import AES

                def ext(pt, ct):                      # noqa: F811 -- the exact call tb/gcm_model.py:18,26 makes
                    m = AES.new(key, mode=AES.MODE_GCM, nonce=iv)
                    ct[:] = np.frombuffer(m.encrypt(pt.tobytes()), dtype=np.uint8)
                    m.digest()
            r1 = run(ext, 256 << 20, 1)
            rn = run(ext, 256 << 20, n_threads)
            out["external"] = {"lib": name, "value_1core": round(r1 / GiB, 3), "value_allcores": round(rn / GiB, 3),
                               "cores": n_threads, "unit": "GiB/s",
                               "note": "hardware AES-NI/PCLMUL library, the class of code the reference's model delegates to"}
    except Exception as e:                             # the baseline must never break the bench line
        out["external"] = {"error": repr(e)}
    return out


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


def pmc_summary(tag):
    """committed rocprofv3 PMC summary for this workload (profiles/pmc_<tag>.json), or {}"""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_%s.json" % tag)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def pmc_traffic(tag):
    """HBM bytes per launch of the fused kernel from the committed PMC summary, if one exists; else None."""
    return pmc_summary(tag).get("hbm_bytes_per_launch")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gib-per-gpu", type=float, default=16.0, help="resident plaintext per GPU (default: the metric's 16 GiB)")
    ap.add_argument("--key-bits", type=int, default=256, choices=(128, 192, 256))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="collective backend for N > 1 (gloo: debug on one GPU, partials staged through the host)")
    ap.add_argument("--one-device", action="store_true", help="debug: every rank uses GPU 0 (with --backend gloo)")
    ap.add_argument("--selfcheck", action="store_true",
                    help="rank 0 also encrypts every whole message alone and compares tags (needs the extra memory)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    N = args.gpus
    if world != N and not (N == 1 and world == 1):
        log("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (N, world))
        N = world
    dist = None
    torch = None
    if world > 1 or "RANK" in os.environ:
        # torch FIRST: its bundled HIP runtime must be the one (and only) copy in the process
        import torch
        import torch.distributed as dist
        if args.one_device:
            local = 0
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import aesgcm_amd  # noqa: F401
    from aesgcm_amd import lib, sharding

    dev = local
    per_gpu = int(args.gib_per_gpu * GiB) // (16 * 2 * N) * (16 * 2 * N)
    kbytes = args.key_bits // 8
    key = sharding.splitmix64_bytes(KEY_SEED, kbytes)           # key and IV from the synthetic streams
    iv0 = sharding.splitmix64_bytes(IV_SEED, 12)

    ctx = lib.Context(key, device=dev)
    geo = ctx.geometry()
    d_pt = lib.DeviceBuffer(per_gpu, device=dev)
    d_ct = lib.DeviceBuffer(per_gpu, device=dev)

    standard = (args.gib_per_gpu == 16.0 and args.key_bits == 256)
    plan = sharding.plan_job(N, per_gpu, rank)
    pt_seed = 0xAE5C0003 if N == 1 else 0xAE5C0004
    msgs = []
    for m in plan:
        fixture = None
        if standard:
            fixture = "cfg3_aes256_16GiB" if N == 1 else "cfg4_aes256_msg%d_32GiB" % m["msg"]
        msgs.append(dict(iv=sharding.tweak_iv(iv0, m["iv_tweak"]), total=m["total"], first_block=m["first_block"],
                         off=m["off"], len=m["len"], fixture=fixture))
        d_pt.fill_splitmix64(pt_seed, m["stream_word"], nbytes=m["len"], offset=m["off"])
    if N == 1:
        workload = "cfg3: AES-%d-GCM, one %.3g GiB message, SplitMix64 PT seed 0xAE5C0003, empty AAD" % (args.key_bits, per_gpu / GiB)
    else:
        workload = ("cfg4 cut to %d ranks: %d AES-%d-GCM message(s) of %.3g GiB (SplitMix64 seed 0xAE5C0004), each sharded over "
                    "all %d ranks, one 16 B x %d x %d RCCL all-gather per step" % (N, len(msgs), args.key_bits, msgs[0]["total"] / GiB, N, N, len(msgs)))
    lib.dev_sync(dev)

    stream = None
    if dist is not None:
        # one explicit (non-default) stream for the kernels, the partial tensors and the collective's stream sync
        tstream = torch.cuda.Stream()
        torch.cuda.set_stream(tstream)
        stream = tstream.cuda_stream
        local_parts = torch.zeros((len(msgs), 16), dtype=torch.uint8, device="cuda")
        gathered = torch.zeros((world, len(msgs), 16), dtype=torch.uint8, device="cuda")

    def step():
        """one pass over the resident workload; returns the list of tags (bytes) -- tags sync the stream"""
        if dist is None:
            m = msgs[0]
            return [ctx.encrypt_dev(m["iv"], d_pt.ptr, m["len"], d_ct.ptr)]
        for i, m in enumerate(msgs):
            ctx.shard_crypt_dev(False, m["iv"], d_pt.ptr + m["off"], m["len"], d_ct.ptr + m["off"], m["first_block"], m["total"],
                                local_parts[i].data_ptr(), stream=stream)
        if args.backend == "nccl":
            dist.all_gather_into_tensor(gathered, local_parts)
        else:                                                     # debug path: 16 B x msgs through the host
            lst = [torch.zeros((len(msgs), 16), dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(lst, local_parts.cpu())
            gathered.copy_(torch.stack(lst))
        per_msg = gathered.permute(1, 0, 2).contiguous()          # [msg][rank][16]
        return [ctx.shard_finalize_dev(m["iv"], per_msg[i].data_ptr(), world, 0, m["total"], stream=stream)
                for i, m in enumerate(msgs)]

    def barrier():
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
        else:
            lib.dev_sync(dev)

    # ---- warmup (untimed) + parity check of the tags against the committed fixtures
    tags = None
    for _ in range(max(args.warmup, 1)):
        tags = step()
    tag_ok = None
    checked = []
    for m, t in zip(msgs, tags):
        fx = load_fixture(m["fixture"]) if m["fixture"] else None
        if fx is not None:
            checked.append(t.hex() == fx["tag"])
    if checked:
        tag_ok = all(checked)
    ct_ok = None
    if N == 1 and msgs[0]["fixture"]:
        fx = load_fixture(msgs[0]["fixture"])
        if fx is not None:
            head = bytes(d_ct.download(64, 0))
            tail = bytes(d_ct.download(64, per_gpu - 64))
            ct_ok = (head.hex() == fx["ct_head"] and tail.hex() == fx["ct_tail"])
    selfcheck = None
    if args.selfcheck and dist is not None and rank == 0:
        selfcheck = True
        for m, t in zip(plan, tags):
            whole_pt, whole_ct = lib.DeviceBuffer(m["total"], device=dev), lib.DeviceBuffer(m["total"], device=dev)
            whole_pt.fill_splitmix64(pt_seed, m["msg"] * m["total"] // 8)
            lib.dev_sync(dev)
            t_one = ctx.encrypt_dev(sharding.tweak_iv(iv0, m["iv_tweak"]), whole_pt.ptr, m["total"], whole_ct.ptr)
            mine = bytes(d_ct.download(min(m["len"], 1 << 20), m["off"]))
            ref = bytes(whole_ct.download(min(m["len"], 1 << 20), 16 * m["first_block"]))
            selfcheck = selfcheck and (t_one == t) and (mine == ref)
            whole_pt.free(); whole_ct.free()
        if not selfcheck:
            log("SELFCHECK FAILURE: sharded tags/ciphertext differ from the single-launch result")
            tag_ok = False
    if tag_ok is False or ct_ok is False:
        log("PARITY FAILURE rank %d: tag_ok=%s ct_ok=%s tags=%s" % (rank, tag_ok, ct_ok, [t.hex() for t in tags]))

    # ---- timed region: exactly K steps between barrier + synchronize on both sides
    ctx.timing_enable(True)
    ctx.timing_read(reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    n_launch, kernel_ms = ctx.timing_read(reset=True)
    ctx.timing_enable(False)

    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        ok = torch.tensor([0 if (tag_ok is False) else 1], dtype=torch.int32, device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if tag_ok is not None:
            tag_ok = bool(ok.item())

    total_bytes = per_gpu * N * args.steps
    value = total_bytes / dt / GiB

    # measured HBM read+write rate of a plain copy kernel over the same two buffers (outside the timed region)
    copy_gbps = None
    if rank == 0:
        try:
            ct_keep = bytes(d_ct.download(64, 0))
            best = None
            for _ in range(3):
                lib.dev_sync(dev)
                c0 = time.perf_counter()
                lib.dev_copy(d_ct.ptr, d_pt.ptr, per_gpu, device=dev)
                lib.dev_sync(dev)
                c1 = time.perf_counter() - c0
                best = c1 if best is None or c1 < best else best
            copy_gbps = round(2 * per_gpu / best / 1e9, 1)
            del ct_keep
        except Exception as e:                                 # never break the bench line
            log("copy measurement failed: %r" % (e,))

    if rank == 0:
        # the timed kernel: k_body over the aligned middle of each range when the library splits it, else k_main over all of it
        _, body_blocks = ctx.split(msgs[0]["len"], msgs[0]["first_block"])
        blocks_per_launch = body_blocks if body_blocks else (per_gpu // len(msgs)) // 16
        kname = "k_body" if body_blocks else "k_main"
        alg_bytes = 32 * blocks_per_launch                     # 16 B read + 16 B written per block
        avg_s = kernel_ms / 1e3 / max(n_launch, 1)
        achieved = alg_bytes / avg_s if avg_s > 0 else 0.0
        tag_name = "cfg3_n1" if N == 1 else "cfg4_n%d" % N
        roofline = {"bound": "hbm", "kernel": "%s<%d,ENC> (fused AES-CTR + GHASH)" % (kname, args.key_bits // 32 + 6), "achieved": round(achieved / 1e9, 2),
                    "peak": HBM_PEAK_BYTES_PER_S / 1e9, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_BYTES_PER_S, 4),
                    "traffic": pmc_traffic(tag_name), "alg_bytes_per_launch": alg_bytes, "launches": n_launch,
                    "avg_launch_ms": round(avg_s * 1e3, 4),
                    "lds_busy_frac": (pmc_summary(tag_name).get("lds") or {}).get("lds_busy_frac"),
                    "measured_copy_kernel": {"value": copy_gbps, "unit": "GB/s read+write", "frac_of_copy": (round(achieved / 1e9 / copy_gbps, 4) if copy_gbps else None)}}
        line = {
            "metric": "GiB/s plaintext, AES-%d-GCM %.3g GiB stream, bit-exact tag" % (args.key_bits, args.gib_per_gpu),
            "value": round(value, 3), "unit": "GiB/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload, "bytes_per_gpu": per_gpu, "messages_per_step": len(msgs),
                       "parallelism": "single" if N == 1 else "shard%d" % N, "key_bits": args.key_bits,
                       "workgroups": geo["workgroups"], "wg_lanes": geo["wg_lanes"], "lds_bytes_per_wg": geo["lds_bytes"]},
            "tag_ok": tag_ok, "ct_head_tail_ok": ct_ok, "selfcheck": selfcheck, "tags": [t.hex() for t in tags],
            "roofline": roofline,
        }
        if N == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(usable_cores())
            except Exception as e:
                line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if tag_ok is not False else 1


if __name__ == "__main__":
    sys.exit(main())
