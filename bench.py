#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json on MI355X:

    GiB/s of plaintext, AES-256-GCM, 16 GiB stream per GPU, bit-exact tag

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (fused AES-CTR + GHASH kernel, fold/combine kernels, 16-byte tag to
host) over the whole resident workload.

  N = 1   configs[2] of BASELINE.json ("cfg3"): ONE AES-256-GCM message of 16 GiB, SplitMix64 plaintext seed
          0xAE5C0003, already resident in HBM; ciphertext written to a second 16 GiB buffer.  The tag of the first
          step is checked against tests/golden/streams.json.  (--config cfg2: configs[1], AES-128, 1 GiB, seed
          0xAE5C0002, same checks.  --decrypt: the same message decrypted and authenticated.)
  N > 1   weak scaling, 16 GiB per GPU: the aggregate N x 16 GiB is configs[3] ("cfg4") cut to N ranks: messages of
          32 GiB (one GCM message cannot exceed 64 GiB - 32 B, aes_icb.vhd:114) from ONE SplitMix64 stream (seed
          0xAE5C0004), message m = bytes [m*32 GiB, (m+1)*32 GiB), IV last byte + m.  Every message is sharded over
          ALL ranks (rank r owns the r-th 1/N of its blocks); per message each rank produces a 16-byte weighted
          GHASH partial; ONE RCCL all-gather per step moves N x M x 16 bytes; every rank folds and finalises the
          tags.  There is no other inter-GPU traffic.  Tags are checked against the cfg4 fixtures.

No PyTorch anywhere: torch.distributed.run only LAUNCHES the ranks (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
environment); the collective is RCCL inside libaesgcm_hip.so (aesgcm_comm_*, include/aesgcm.h), the unique id travels
through a file (aesgcm_amd/comm.py), the compute path is the same library through ctypes.
"""
import argparse
import hashlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GiB = 1 << 30
HBM_PEAK_BYTES_PER_S = 8.0e12          # MI355X HBM3E peak (MI355X_MICROARCH.md)
KEY_SEED, IV_SEED = 0x4B4559, 0x4956
CONFIGS = {                            # BASELINE.json configs that fit one GPU
    "cfg3": dict(key_bits=256, gib=16.0, pt_seed=0xAE5C0003, fixture="cfg3_aes256_16GiB"),
    "cfg2": dict(key_bits=128, gib=1.0, pt_seed=0xAE5C0002, fixture="cfg2_aes128_1GiB"),
}


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


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for b in iter(lambda: f.read(1 << 20), b""):
            h.update(b)
    return h.hexdigest()


def git_head():
    """short commit hash without starting a process (this one has the GPU open): $GIT_HEAD, else .git/HEAD by hand"""
    h = os.environ.get("GIT_HEAD")
    if h:
        return h[:7]
    try:
        head = open(os.path.join(ROOT, ".git", "HEAD")).read().strip()
        if head.startswith("ref:"):
            head = open(os.path.join(ROOT, ".git", head.split(None, 1)[1])).read().strip()
        return head[:7] or None
    except OSError:
        return None


def pmc_summary(tag):
    """committed rocprofv3 PMC summary for this workload (profiles/pmc_<tag>.json), or {}"""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_%s.json" % tag)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def cpu_baseline():
    """The reference's CPU path timed on this box's host cores (oracle/cpu_baseline.py: pycryptodome if importable, else
    libcrypto; 1 core and all cores as worker processes).  The ONLY place bench.py touches oracle/."""
    from oracle import cpu_baseline as cb
    return cb.measure()


def sclk_from_trace(trace, waves_per_wg):
    """effective shader clock (MHz) of the last timed launch: per workgroup, the shader cycles its waves were resident
    (s_memtime, summed over the waves, in kilocycles) over its wall duration (s_memrealtime, 100 MHz); median over
    workgroups"""
    v = []
    for (t0, t1, _hw, packed) in trace:
        kc = packed >> 32
        if t1 > t0 and kc:
            v.append((kc * 1024.0 / waves_per_wg) / ((t1 - t0) * 10e-9) / 1e6)
    return round(statistics.median(v), 0) if v else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS), help="N = 1 workload (default: the metric's cfg3)")
    ap.add_argument("--gib-per-gpu", type=float, default=None, help="override: resident plaintext per GPU (no fixture check)")
    ap.add_argument("--key-bits", type=int, default=None, choices=(128, 192, 256), help="override (no fixture check)")
    ap.add_argument("--decrypt", action="store_true", help="time decrypt + authenticate instead of encrypt (N = 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="rccl", choices=("rccl", "nccl", "file", "gloo"),
                    help="exchange for N > 1: rccl (= nccl, the product: RCCL inside the library) or file (= gloo of round 1: "
                         "debug, partials through the host, for several ranks on ONE GPU)")
    ap.add_argument("--one-device", action="store_true", help="debug: every rank uses GPU 0 (with --backend file)")
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

    cfg = dict(CONFIGS[args.config])
    standard = args.gib_per_gpu is None and args.key_bits is None
    if N > 1:
        cfg = dict(key_bits=256, gib=16.0, pt_seed=0xAE5C0004, fixture=None)
    if args.gib_per_gpu is not None:
        cfg["gib"] = args.gib_per_gpu
    if args.key_bits is not None:
        cfg["key_bits"] = args.key_bits

    # ---- CPU baseline FIRST: its worker processes are forked before this process touches the GPU
    cpu_base = None
    if N == 1 and rank == 0 and not args.no_cpu_baseline:
        try:
            cpu_base = cpu_baseline()
        except Exception as e:                                  # the baseline must never break the bench line
            cpu_base = {"error": repr(e)}

    import aesgcm_amd  # noqa: F401
    from aesgcm_amd import comm, lib, sharding
    from aesgcm_amd.build import SO

    dev = 0 if args.one_device else local
    ex = None
    if world > 1:
        ex = comm.make_exchange(rank, world, dev, prefer="rccl" if args.backend in ("rccl", "nccl") else "file")

    per_gpu = int(cfg["gib"] * GiB) // (16 * 2 * N) * (16 * 2 * N)
    key_bits = cfg["key_bits"]
    key = sharding.splitmix64_bytes(KEY_SEED, key_bits // 8)     # key and IV from the synthetic streams
    iv0 = sharding.splitmix64_bytes(IV_SEED, 12)

    ctx = lib.Context(key, device=dev)
    geo = ctx.geometry()
    d_pt = lib.DeviceBuffer(per_gpu, device=dev)
    d_ct = lib.DeviceBuffer(per_gpu, device=dev)

    plan = sharding.plan_job(N, per_gpu, rank)
    msgs = []
    for m in plan:
        fixture = None
        if standard:
            fixture = cfg["fixture"] if N == 1 else ("cfg4_aes256_msg%d_32GiB" % m["msg"] if m["total"] == 32 * GiB else None)
        msgs.append(dict(iv=sharding.tweak_iv(iv0, m["iv_tweak"]), total=m["total"], first_block=m["first_block"],
                         off=m["off"], len=m["len"], fixture=fixture))
        d_pt.fill_splitmix64(cfg["pt_seed"], m["stream_word"], nbytes=m["len"], offset=m["off"])
    M = len(msgs)
    if N == 1:
        workload = "%s: AES-%d-GCM, one %.3g GiB message, SplitMix64 PT seed 0x%X, empty AAD%s" % (
            args.config if standard else "custom", key_bits, per_gpu / GiB, cfg["pt_seed"], ", DECRYPT + authenticate" if args.decrypt else "")
    else:
        workload = ("cfg4 cut to %d ranks: %d AES-%d-GCM message(s) of %.3g GiB (SplitMix64 seed 0xAE5C0004), each sharded over "
                    "all %d ranks, one 16 B x %d x %d all-gather per step (%s)" % (N, M, key_bits, msgs[0]["total"] / GiB, N, N, M, ex.name))
    lib.dev_sync(dev)

    if ex is not None:
        local_parts = lib.DeviceBuffer(16 * M, device=dev)
        gathered = lib.DeviceBuffer(16 * M * world, device=dev)     # [rank][message][16] after the all-gather

    expect_tag = None
    cstream = ctx.stream()

    def step():
        """one pass over the resident workload; returns the list of tags (bytes) -- fetching a tag syncs the stream"""
        if ex is None:
            m = msgs[0]
            if args.decrypt:
                return [ctx.decrypt_dev(m["iv"], d_ct.ptr, m["len"], d_pt.ptr, tag=expect_tag)]
            return [ctx.encrypt_dev(m["iv"], d_pt.ptr, m["len"], d_ct.ptr)]
        for i, m in enumerate(msgs):
            ctx.shard_crypt_dev(False, m["iv"], d_pt.ptr + m["off"], m["len"], d_ct.ptr + m["off"], m["first_block"], m["total"],
                                local_parts.ptr + 16 * i, stream=cstream)
        ex.allgather_dev(local_parts.ptr, gathered.ptr, 16 * M, stream=cstream)    # the context's stream: ordered after the partials
        return [ctx.shard_finalize_dev(m["iv"], gathered.ptr + 16 * i, world, 0, m["total"], stride_bytes=16 * M, stream=cstream)
                for i, m in enumerate(msgs)]

    def barrier():
        lib.dev_sync(dev)
        if ex is not None:
            ex.barrier()
            lib.dev_sync(dev)

    if args.decrypt:                                               # the ciphertext to decrypt, and its tag
        expect_tag = ctx.encrypt_dev(msgs[0]["iv"], d_pt.ptr, msgs[0]["len"], d_ct.ptr)

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
    if args.selfcheck and ex is not None and rank == 0:
        selfcheck = True
        for m, t in zip(plan, tags):
            whole_pt, whole_ct = lib.DeviceBuffer(m["total"], device=dev), lib.DeviceBuffer(m["total"], device=dev)
            whole_pt.fill_splitmix64(cfg["pt_seed"], m["msg"] * m["total"] // 8)
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

    # ---- timed region: exactly K steps between barrier + synchronize on both sides; timing mode OFF (no event
    # records, no trace memset, no in-kernel trace atomics inside it)
    ctx.timing_enable(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0

    if ex is not None:
        dt = ex.allreduce(dt, "max")
        ok = ex.allreduce(0.0 if tag_ok is False else 1.0, "min")
        if tag_ok is not None:
            tag_ok = bool(ok)

    total_bytes = per_gpu * N * args.steps
    value = total_bytes / dt / GiB

    # ---- separate short pass with timing mode ON: HIP events around each launch of the fused kernel on its own
    # stream, the per-workgroup trace (shader clock), then the same instruction stream without HBM traffic
    ctx.timing_enable(True)
    ctx.timing_read(reset=True)
    for _ in range(min(3, max(1, args.steps))):
        step()
    lib.dev_sync(dev)
    n_launch, kernel_ms = ctx.timing_read(reset=True)
    sclk = None
    if ctx.split(msgs[0]["len"], msgs[0]["first_block"])[1]:
        geo = ctx.geometry(body=True)                              # the timed kernel is k_body: report ITS launch geometry
    if rank == 0:
        try:
            sclk = sclk_from_trace(ctx.wg_trace(), geo["wg_lanes"] // 64)
        except Exception as e:
            log("trace failed: %r" % (e,))
    ctx.timing_enable(False)

    if rank == 0:
        # the timed kernel: k_body over the aligned middle of each range when the library splits it, else k_main over all of it
        _, body_blocks = ctx.split(msgs[0]["len"], msgs[0]["first_block"])
        blocks_per_launch = body_blocks if body_blocks else (per_gpu // M) // 16
        kname = "k_body" if body_blocks else "k_main"
        alg_bytes = 32 * blocks_per_launch                     # 16 B read + 16 B written per block
        avg_s = kernel_ms / 1e3 / max(n_launch, 1)
        achieved = alg_bytes / avg_s if avg_s > 0 else 0.0

        ceiling = None
        try:
            best = None
            for _ in range(3):
                ms, nb = ctx.ceiling_probe(msgs[0]["len"])
                best = ms if best is None or ms < best else best
            ceiling = {"value": round(32 * nb / (best / 1e3) / 1e9, 2), "unit": "GB/s", "ms": round(best, 4), "blocks": nb,
                       "what": "k_body<%d,PROBE>: the fused kernel's own instruction stream (LDS T-table lookups, GHASH table "
                               "multiply, scalar loads, chunk dispensers) with its global loads and stores removed, same process, "
                               "same clocks -- the ceiling of the formulation, not of the chip" % (key_bits // 32 + 6)}
            ceiling["achieved_over_ceiling"] = round(achieved / 1e9 / ceiling["value"], 4) if ceiling["value"] else None
            ceiling["sclk_mhz"] = sclk_from_trace(ctx.wg_trace(), geo["wg_lanes"] // 64)      # the clock the no-HBM stream sustains
        except Exception as e:
            ceiling = {"error": repr(e)}

        # measured HBM read+write rate of a plain copy kernel over the same two buffers (outside the timed region)
        copy_gbps = None
        try:
            best = None
            for _ in range(3):
                lib.dev_sync(dev)
                c0 = time.perf_counter()
                lib.dev_copy(d_ct.ptr, d_pt.ptr, per_gpu, device=dev)
                lib.dev_sync(dev)
                c1 = time.perf_counter() - c0
                best = c1 if best is None or c1 < best else best
            copy_gbps = round(2 * per_gpu / best / 1e9, 1)
        except Exception as e:                                 # never break the bench line
            log("copy measurement failed: %r" % (e,))

        # HBM traffic from the committed PMC summary -- only when it was measured on THIS build of the library
        so_sha = sha256_file(SO)
        tag_name = ("%s_n1" % args.config) if N == 1 else "cfg4_n%d" % N
        pm = pmc_summary(tag_name) if standard and not args.decrypt else {}
        same_build = bool(pm) and pm.get("so_sha256") == so_sha
        traffic = pm.get("hbm_bytes_per_launch") if same_build else None
        roofline = {"bound": "hbm", "kernel": "%s<%d,%s> (fused AES-CTR + GHASH)" % (kname, key_bits // 32 + 6, "DEC" if args.decrypt else "ENC"),
                    "achieved": round(achieved / 1e9, 2), "peak": HBM_PEAK_BYTES_PER_S / 1e9, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_BYTES_PER_S, 4), "traffic": traffic,
                    "traffic_source": ("profiles/pmc_%s.json" % tag_name) if pm else None,
                    "traffic_build": {"pmc_so_sha256": pm.get("so_sha256"), "pmc_git": pm.get("git"), "running_so_sha256": so_sha,
                                      "running_git": git_head(), "match": same_build},
                    "alg_bytes_per_launch": alg_bytes, "launches_timed": n_launch, "avg_launch_ms": round(avg_s * 1e3, 4),
                    "timing": "HIP events on the launch stream in a separate %d-step pass after the timed region (timing mode is off inside it)" % min(3, max(1, args.steps)),
                    "sclk_mhz": sclk,
                    "lds_busy_frac": (pm.get("lds") or {}).get("lds_busy_frac") if same_build else None,
                    "formulation_ceiling": ceiling,
                    "measured_copy_kernel": {"value": copy_gbps, "unit": "GB/s read+write", "frac_of_copy": (round(achieved / 1e9 / copy_gbps, 4) if copy_gbps else None)}}
        line = {
            "metric": "GiB/s plaintext, AES-%d-GCM %.3g GiB stream, bit-exact tag" % (key_bits, cfg["gib"]),
            "value": round(value, 3), "unit": "GiB/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload, "bytes_per_gpu": per_gpu, "messages_per_step": M,
                       "parallelism": "single" if N == 1 else "shard%d" % N, "key_bits": key_bits,
                       "workgroups": geo["workgroups"], "wg_lanes": geo["wg_lanes"], "lds_bytes_per_wg": geo["lds_bytes"],
                       "exchange": None if ex is None else {"backend": ex.name, "ranks_seen": ex.world, "torch": "not imported"}},
            "tag_ok": tag_ok, "ct_head_tail_ok": ct_ok, "selfcheck": selfcheck, "tags": [t.hex() for t in tags],
            "roofline": roofline,
        }
        if cpu_base is not None:
            line["cpu_baseline"] = cpu_base
        print(json.dumps(line), flush=True)

    if ex is not None:
        ex.barrier()
        ex.close()
        comm.finish(rank, world)
    return 0 if tag_ok is not False else 1


if __name__ == "__main__":
    sys.exit(main())
