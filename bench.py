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
          tags.  There is no other inter-GPU traffic.  Tags are checked against the cfg4 fixtures.  A rank's messages
          run on separate contexts (own stream + scratch each, --contexts) so that one message's fold / combine tail
          hides behind the next message's fused kernel.
  --config cfg5   configs[4]: 2^20 independent 4 KiB packets, AES-128-GCM, per-packet key and IV (on-GPU aes_kexp),
          keys / IVs / plaintext from the SURVEY 8(d) streams; N > 1 = replicas of 2^20 / N packets, no collective.
          Tags are checked by SHA-256 against tests/golden/batch.json.
  --emulate-rank R --of W   on ONE GPU: exactly rank R's step of the W-GPU job (its shard of every message at the
          real first_block, a device copy in place of the all-gather, the finalizes), with the other ranks' partials
          computed beforehand so that the tags are the real ones.  The scaling prediction while no W-GPU node exists.

`python bench.py --gpus N` WITHOUT a launcher (no RANK in the environment) starts the N ranks itself: N fresh child
processes (RANK / LOCAL_RANK / WORLD_SIZE / rendezvous directory in their environment) created before this process makes
any GPU call; rank 0's JSON line is relayed, the exit code is the worst child's, a hung child is killed and fails the run.

If the exchange that comes up is not RCCL the line still says so (config.exchange.backend) but the run EXITS NON-ZERO
unless --allow-file-exchange is given: a number produced through the debug file exchange is not an RCCL number.

No PyTorch anywhere: torch.distributed.run only LAUNCHES the ranks (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
environment); the collective is RCCL inside libaesgcm_hip.so (aesgcm_comm_*, include/aesgcm.h), the unique id travels
through a file (aesgcm_amd/comm.py), the compute path is the same library through ctypes.
"""
import argparse
import hashlib
import json
import os
import signal
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GiB = 1 << 30
HBM_PEAK_BYTES_PER_S = 8.0e12          # MI355X HBM3E peak (MI355X_MICROARCH.md)
KEY_SEED, IV_SEED = 0x4B4559, 0x4956
CONFIGS = {                            # BASELINE.json configs that fit one GPU
    "cfg3": dict(key_bits=256, gib=16.0, pt_seed=0xAE5C0003, fixture="cfg3_aes256_16GiB"),
    "cfg2": dict(key_bits=128, gib=1.0, pt_seed=0xAE5C0002, fixture="cfg2_aes128_1GiB"),
    "cfg5": dict(key_bits=128, n_pkts=1 << 20, pkt_len=4096, pt_seed=0xAE5C0005),
    # not a BASELINE config: the reference's own deployment (tb/gcm_test.py:76-85: message after message under one key) at message size -- 4096 x 1 MiB as the
    # packets of one aesgcm_packets_crypt_dev call, which goes by rows (round 5)
    "msgs": dict(key_bits=256, n_pkts=4096, pkt_len=1 << 20, pt_seed=0xAE5C0055),
    # ... and at FRAME size (round 6): 2^20 MACsec-shaped frames -- 64 .. 1514 bytes (lengths from SplitMix64 stream 0x4C454E), 28 bytes of SecTAG-like AAD each, packed
    # back to back -- under one key through the offset arrays of one aesgcm_packets_crypt_dev call.  The reference's documented deployment: its README vectors are
    # IEEE 802.1AE frames (README.md:251-257) and its `short` traffic class tops out at 2^12 - 1 bytes (config/gcm_utils.py:144).
    "frames": dict(key_bits=256, n_pkts=1 << 20, aad_len=28, pt_seed=0xAE5C0006, len_seed=0x4C454E),
}
EXIT_NOT_RCCL = 3                      # the exchange that came up is not RCCL and --allow-file-exchange was not given


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# The contract is ONE JSON line on stdout.  Libraries loaded into this process write there too -- RCCL prints a five-line version banner through C stdio at
# its first communicator, and C stdio flushes it at exit, i.e. BEHIND the line.  A process that is going to measure therefore keeps the real stdout for the
# line alone (claim_stdout) and points descriptor 1 at stderr for everything else.
_LINE_FD = None


def claim_stdout():
    global _LINE_FD
    if _LINE_FD is None:
        sys.stdout.flush()
        _LINE_FD = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    data = (json.dumps(line) + "\n").encode()
    if _LINE_FD is None:
        sys.stdout.write(data.decode()); sys.stdout.flush()
    else:
        os.write(_LINE_FD, data)


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


def cpu_baseline(config):
    """The reference's CPU path timed on this box's host cores (oracle/cpu_baseline.py: pycryptodome if importable, else
    libcrypto; 1 core and all cores as worker processes).  The ONLY place bench.py touches oracle/."""
    from oracle import cpu_baseline as cb
    return cb.measure_cfg5() if config == "cfg5" else cb.measure_frames() if config == "frames" else cb.measure()


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


# ------------------------------------------------------------------------------------------------ self-launch
def visible_gpus():
    """GPUs the kernel driver exposes, counted WITHOUT any HIP call (this process must stay GPU-free: its children own
    the devices).  /sys/class/kfd topology nodes with SIMDs are GPUs; honours a *_VISIBLE_DEVICES list.  None = unknown."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            return len([x for x in v.split(",") if x.strip() != ""])
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for d in os.listdir(base):
            try:
                props = dict(line.split(None, 1) for line in open(os.path.join(base, d, "properties")) if " " in line)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        return n or None                                   # nothing readable: unknown, let the ranks find out
    except OSError:
        return None


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _kill_group(p, why):
    """end a child this process started, and whatever it started: the child is its own session / process group leader"""
    if p.poll() is not None:
        return p.returncode
    log("bench.py: killing pid %d: %s" % (p.pid, why))
    try:
        os.killpg(p.pid, signal.SIGKILL)                   # exact process group of a child this process started
    except (OSError, ProcessLookupError):
        try:
            p.kill()
        except OSError:
            pass
    try:
        return p.wait(timeout=30)
    except subprocess.TimeoutExpired:
        return -9


def _run_children(cmds_envs, launch_timeout, grace=20.0):
    """Start the given (argv, env) children, each in a session of its own, and wait for them: child 0's stdout is captured
    (the JSON line), the others' goes to stderr.  A child that fails takes the others down after `grace` seconds (they may
    be blocked in a rendezvous or a collective with the dead one); the deadline kills what is left.  Whatever ends this
    function -- return, exception, SIGTERM / SIGINT / SIGHUP to this process -- no child outlives it: a rank blocked in an
    RCCL rendezvous would otherwise hold its GPU indefinitely and poison the next run on the machine.
    -> (exit codes, child 0's stdout bytes)"""
    import threading
    procs, box = [], {}
    old = {}

    def on_signal(signum, _frame):
        raise KeyboardInterrupt("signal %d" % signum)
    for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        try:
            old[sg] = signal.signal(sg, on_signal)
        except (ValueError, OSError):                      # not the main thread
            pass
    rcs = [None] * len(cmds_envs)
    try:
        for r, (cmd, env) in enumerate(cmds_envs):
            procs.append(subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                          start_new_session=True))

        def drain():                                       # child 0's stdout -- read it so the pipe never fills
            box["out"] = procs[0].stdout.read()
        th = threading.Thread(target=drain, daemon=True)
        th.start()
        deadline = time.monotonic() + launch_timeout
        failed_at = None
        while any(rc is None for rc in rcs):
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    rcs[i] = p.poll()
            now = time.monotonic()
            if failed_at is None and any(rc not in (None, 0) for rc in rcs):
                failed_at = now
            if now > deadline or (failed_at is not None and now - failed_at > grace):
                for i, p in enumerate(procs):
                    if rcs[i] is None:
                        rcs[i] = _kill_group(p, "rank %d: %s" % (i, "launch timeout" if now > deadline else "another rank failed"))
                break
            time.sleep(0.05)
        th.join(timeout=10)
        return rcs, box.get("out", b"") or b""
    finally:
        for i, p in enumerate(procs):
            if p.poll() is None:
                _kill_group(p, "rank %d: the launcher is going away" % i)
        for sg, h in old.items():
            try:
                signal.signal(sg, h)
            except (ValueError, OSError):
                pass


def self_launch(args, argv):
    """--gpus N without a launcher: start the N ranks as fresh child processes and relay rank 0's line.  Nothing here
    touches the GPU (no library load, no HIP call) -- the children initialise their own devices.  If the ranks come back
    saying that no RCCL communicator came up between the processes (exit code EXIT_NOT_RCCL), ONE fresh child drives all N
    devices through the library's single-process path instead (aesgcm_mgpu_*: ncclCommInitAll) and the line says so."""
    N = args.gpus
    have = visible_gpus()
    if have is not None and have < N and not args.one_device:
        log("bench.py: --gpus %d but only %d GPU(s) visible on this node; refusing to measure fewer GPUs than asked for "
            "(--one-device puts every rank on GPU 0 for debugging)" % (N, have))
        return 2
    rdzv = tempfile.mkdtemp(prefix="aesgcm_rdzv_self_", dir="/dev/shm" if os.access("/dev/shm", os.W_OK) else None)
    port = free_port()
    base_env = dict(os.environ)
    base_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this driver
    base_env.setdefault("NCCL_DEBUG", "WARN")                    # if RCCL refuses, its own words reach stderr
    me = [sys.executable, os.path.abspath(__file__)]
    try:
        if args.single_process:
            rcs, out0 = [EXIT_NOT_RCCL], b""
        else:
            rcs, out0 = _run_children([(me + argv, dict(base_env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(N), LOCAL_WORLD_SIZE=str(N),
                                                        MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AESGCM_RDZV_DIR=rdzv, AESGCM_SELF_LAUNCHED="1"))
                                       for r in range(N)], args.launch_timeout)
        worst = max((rc if rc > 0 else 1) if rc else 0 for rc in rcs)      # killed (negative) counts as 1
        if ((worst == EXIT_NOT_RCCL or args.single_process) and not args.allow_file_exchange and not args.one_device and args.config != "cfg5"
                and args.backend in ("rccl", "nccl")):
            if not args.single_process:
                log("bench.py: no RCCL communicator between %d processes (rank exit codes %s); the line above is NOT the result.  "
                    "Falling back to ONE process driving all %d devices (ncclCommInitAll)" % (N, rcs, N))
            rcs, out0 = _run_children([(me + [a for a in argv if a != "--single-process"] + ["--sp-child"], dict(base_env, AESGCM_SELF_LAUNCHED="1"))], args.launch_timeout)
            worst = max((rc if rc > 0 else 1) if rc else 0 for rc in rcs)
        sys.stdout.write(out0.decode(errors="replace"))
        sys.stdout.flush()
    finally:
        try:
            for f in os.listdir(rdzv):
                os.unlink(os.path.join(rdzv, f))
            os.rmdir(rdzv)
        except OSError:
            pass
    if worst:
        log("bench.py: rank exit codes %s" % (rcs,))
    return worst


# ------------------------------------------------------------------------------------------------ exchanges
class EmulatedExchange:
    """--emulate-rank: one GPU plays rank `rank` of `world`; the all-gather is a device copy of this rank's partials into
    their slot of a gathered buffer that already holds the other ranks' (real) partials."""
    def __init__(self, lib, rank, world, device):
        self.lib, self.rank, self.world, self.device = lib, rank, world, device
        self.name = "emulated (device copy; rank %d of %d on one GPU)" % (rank, world)

    def allgather_dev(self, d_send, d_recv, bytes_per_rank, stream=None):
        self.lib.dev_copy(d_recv + self.rank * bytes_per_rank, d_send, bytes_per_rank, device=self.device, stream=stream)

    def allreduce(self, value, op="max"):
        return value

    def barrier(self):
        pass

    def close(self):
        pass


# ------------------------------------------------------------------------------------------------ cfg5
def run_cfg5(args, rank, world, dev, ex, cpu_base):
    """BASELINE config 5: independent packets, per-packet key and IV; replicas for N > 1 (no collective on the data path)"""
    import aesgcm_amd  # noqa: F401
    from aesgcm_amd import comm, lib
    from aesgcm_amd.build import SO
    cfg = CONFIGS["cfg5"]
    N = world
    n_all = args.n_pkts if args.n_pkts else cfg["n_pkts"]
    pkt = args.pkt_len if args.pkt_len else cfg["pkt_len"]
    key_bits = args.key_bits or cfg["key_bits"]
    kb = key_bits // 8
    standard = (n_all, pkt, key_bits) == (cfg["n_pkts"], cfg["pkt_len"], cfg["key_bits"])
    n = n_all // N                                               # this replica's packets [rank * n, (rank + 1) * n)
    first = rank * n
    # keys: stream 0x4B4559, kb bytes per packet; IVs: the first 12 bytes of every 16 of stream 0x4956; plaintext 0xAE5C0005
    d_keys, d_ivw, d_ivs = lib.DeviceBuffer(kb * n, device=dev), lib.DeviceBuffer(16 * n, device=dev), lib.DeviceBuffer(12 * n, device=dev)
    d_keys.fill_splitmix64(KEY_SEED, first * kb // 8)
    d_ivw.fill_splitmix64(IV_SEED, first * 2)
    ivw = bytes(d_ivw.download())
    d_ivs.upload(b"".join(ivw[16 * p:16 * p + 12] for p in range(n)))
    d_ivw.free()
    d_pt, d_ct, d_tags = lib.DeviceBuffer(pkt * n, device=dev), lib.DeviceBuffer(pkt * n, device=dev), lib.DeviceBuffer(16 * n, device=dev)
    d_pt.fill_splitmix64(cfg["pt_seed"], first * pkt // 8)
    lib.dev_sync(dev)

    def step(stream=None):
        lib.batch_crypt_dev(args.decrypt, n, kb, d_keys.ptr, d_ivs.ptr, d_ct.ptr if args.decrypt else d_pt.ptr, pkt,
                            d_pt.ptr if args.decrypt else d_ct.ptr, d_tags.ptr, device=dev, stream=stream)

    def barrier():
        lib.dev_sync(dev)
        if ex is not None:
            ex.barrier()
            lib.dev_sync(dev)

    if args.decrypt:
        lib.batch_crypt_dev(False, n, kb, d_keys.ptr, d_ivs.ptr, d_pt.ptr, pkt, d_ct.ptr, d_tags.ptr, device=dev)
        lib.dev_sync(dev)
    for _ in range(max(args.warmup, 1)):
        step()
    lib.dev_sync(dev)
    # ---- parity: SHA-256 over all tags (and the first 64 tags literally) against the committed libcrypto fixture
    tags = bytes(d_tags.download())
    tag_ok = None
    fx = None
    try:
        with open(os.path.join(ROOT, "tests", "golden", "batch.json")) as f:
            fx = json.load(f)
    except (OSError, ValueError):
        pass
    all_tags = tags
    if ex is not None:                                          # replicas: rank 0 hashes the concatenation of every rank's tags
        d = comm.rendezvous_dir()
        comm._write_atomic(os.path.join(d, "cfg5_tags_%d" % rank), tags)
        if rank == 0:
            all_tags = b"".join(comm._wait_read(os.path.join(d, "cfg5_tags_%d" % r), 16 * n, 300.0) for r in range(world))
    tags_sha = hashlib.sha256(all_tags).hexdigest() if rank == 0 else None
    if fx is not None and standard and rank == 0:
        tag_ok = (tags_sha == fx.get("full_tags_sha256")) and [tags[16 * p:16 * p + 16].hex() for p in range(64)] == fx["first64_tags"]
        if not tag_ok:
            log("PARITY FAILURE cfg5: tags sha %s, fixture %s" % (tags_sha, fx.get("full_tags_sha256")))

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
    value = n * N * pkt * args.steps / dt / GiB

    # ---- the kernel's own duration: HIP events on the stream it is launched on (the NULL stream), separate short pass
    tm = lib.Timer(device=dev)
    k_ms = []
    for _ in range(min(5, max(1, args.steps))):
        tm.start(None)
        step()
        tm.stop(None)
        k_ms.append(tm.ms())
    # ---- the formulation's ceiling: the same kernel without the data's loads and stores (aesgcm_batch_ceiling_probe_dev; 8 lanes per packet only), same process,
    #      same clocks -- how far the launch is from what per-packet aes_kexp + T-table AES + Shoup GHASH cost by themselves on this chip (round-4 verdict, Next 2)
    c_ms = []
    if not args.decrypt and not args.batch_lanes and lib.batch_shape(n, pkt, device=dev) == 8:
        try:
            lib.batch_ceiling_probe_dev(n, kb, d_keys.ptr, d_ivs.ptr, pkt, d_tags.ptr, device=dev)
            lib.dev_sync(dev)
            for _ in range(min(5, max(1, args.steps))):
                tm.start(None)
                lib.batch_ceiling_probe_dev(n, kb, d_keys.ptr, d_ivs.ptr, pkt, d_tags.ptr, device=dev)
                tm.stop(None)
                c_ms.append(tm.ms())
        except lib.AesGcmError as e:
            log("bench.py: cfg5 ceiling probe failed: %r" % (e,))
            c_ms = []
    tm.close()
    if rank == 0:
        avg_s = statistics.mean(k_ms) / 1e3
        alg_bytes = n * (2 * pkt + kb + 12 + 16)                  # per packet: data read + written, key, IV, tag
        achieved = alg_bytes / avg_s
        so_sha = sha256_file(SO)
        pm = pmc_summary("cfg5_n1") if standard and not args.decrypt and N == 1 else {}
        same_build = bool(pm) and pm.get("so_sha256") == so_sha
        nr = key_bits // 32 + 6
        lanes = lib.batch_shape(n, pkt, device=dev)                # the shape the library's own rule gives this call (aesgcm_batch_shape)
        kname = "k_batch3<%d,%d,%d>" % (nr, int(args.decrypt), {8: 3, 16: 4, 64: 6}[lanes])
        roofline = {"bound": "hbm", "kernel": "%s (%d lanes per packet%s: per-packet aes_kexp + AES-CTR + GHASH)" % (kname, lanes, ", FORCED through the debug build" if args.batch_lanes else ""),
                    "achieved": round(achieved / 1e9, 2), "peak": HBM_PEAK_BYTES_PER_S / 1e9, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_BYTES_PER_S, 4), "traffic": pm.get("hbm_bytes_per_launch") if same_build else None,
                    "traffic_source": "profiles/pmc_cfg5_n1.json" if pm else None,
                    "traffic_build": {"pmc_so_sha256": pm.get("so_sha256"), "pmc_git": pm.get("git"), "running_so_sha256": so_sha,
                                      "running_git": git_head(), "match": same_build},
                    "alg_bytes_per_launch": alg_bytes, "launches_timed": len(k_ms), "avg_launch_ms": round(avg_s * 1e3, 4),
                    "timing": "HIP events on the launch stream around each launch in a separate %d-step pass after the timed region" % len(k_ms),
                    "lds_busy_frac": (pm.get("lds") or {}).get("lds_busy_frac") if same_build else None}
        if c_ms:
            c_s = statistics.mean(c_ms) / 1e3
            roofline["formulation_ceiling"] = {"kernel": "k_batch3<%d,PROBE,3> (the same instruction stream without the data's loads and stores; keys, IVs, tags still move)" % nr,
                                               "avg_launch_ms": round(c_s * 1e3, 4), "gib_per_s": round(n * pkt / c_s / GiB, 1), "alg_gb_per_s": round(alg_bytes / c_s / 1e9, 1),
                                               "frac_of_hbm_peak": round(alg_bytes / c_s / HBM_PEAK_BYTES_PER_S, 4)}
            roofline["achieved_over_ceiling"] = round(c_s / avg_s, 4)
        line = {
            "metric": "GiB/s plaintext, AES-%d-GCM, %d independent %d-byte packets, per-packet key/IV, bit-exact tags" % (key_bits, n * N, pkt),
            "value": round(value, 3), "unit": "GiB/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong" if N > 1 else "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "cfg5%s: %d packets x %d B, AES-%d-GCM, per-packet key (stream 0x4B4559) and IV (stream 0x4956), on-GPU aes_kexp, "
                                   "plaintext stream 0xAE5C0005%s%s" % ("" if standard else " (custom size)", n * N, pkt, key_bits,
                                                                       ", DECRYPT + authenticate" if args.decrypt else "",
                                                                       "; %d replicas of %d packets, no collective" % (N, n) if N > 1 else ""),
                       "packets_per_gpu": n, "pkt_len": pkt, "key_bits": key_bits, "mpkt_per_s": round(n * N * args.steps / dt / 1e6, 2),
                       "parallelism": "single" if N == 1 else "replicas%d" % N,
                       "exchange": None if ex is None else {"backend": ex.name, "ranks_seen": ex.world, "use": "barrier and max-over-ranks timing only", "torch": "not imported"}},
            "tag_ok": tag_ok, "tags_sha256": tags_sha, "roofline": roofline,
        }
        if cpu_base is not None:
            line["cpu_baseline"] = cpu_base
        emit(line)
    return tag_ok


# ------------------------------------------------------------------------------------------------ many messages under one key
def run_msgs(args, rank, world, dev, ex, cpu_base):
    """--config msgs: n messages of one size under ONE key as the packets of one aesgcm_packets_crypt_dev call (by rows: k_rows +
    k_rows_close).  A step is one call; the calls of the timed region are queued back to back and waited for once.  N > 1: replicas -- rank r takes messages
    [r n / N, (r + 1) n / N) of the same streams, no collective on the data path (messages are independent objects, as cfg5's packets), one max-over-ranks of the time.
    Parity in the run: the tags of a sample of the messages equal what the single-message
    path of the same library (pinned to the libcrypto fixtures by the test-suite) gives for the same bytes, and so does their ciphertext by SHA-256."""
    import aesgcm_amd  # noqa: F401
    from aesgcm_amd import lib, sharding
    from aesgcm_amd.build import SO
    cfg = CONFIGS["msgs"]
    N = world
    n_all = args.n_pkts if args.n_pkts else cfg["n_pkts"]
    n = n_all // N
    first = rank * n
    size = args.pkt_len if args.pkt_len else cfg["pkt_len"]
    key_bits = args.key_bits or cfg["key_bits"]
    key = sharding.splitmix64_bytes(KEY_SEED, key_bits // 8)
    d_ivw, d_ivs = lib.DeviceBuffer(16 * n, device=dev), lib.DeviceBuffer(12 * n, device=dev)
    d_ivw.fill_splitmix64(IV_SEED, first * 2)
    ivw = bytes(d_ivw.download())
    ivs = b"".join(ivw[16 * p:16 * p + 12] for p in range(n))
    d_ivs.upload(ivs)
    d_ivw.free()
    d_pt, d_ct, d_tags = lib.DeviceBuffer(size * n, device=dev), lib.DeviceBuffer(size * n, device=dev), lib.DeviceBuffer(16 * n, device=dev)
    d_pt.fill_splitmix64(cfg["pt_seed"], first * size // 8)
    al = args.aad_len or 0
    d_aad = lib.DeviceBuffer((al * n + 23) // 8 * 8, device=dev) if al else None
    if al:
        d_aad.fill_splitmix64(0x414144)
    akw = dict(d_aad=d_aad.ptr, aad_len=al) if al else {}
    ctx = lib.Context(key, device=dev)
    shape = lib.SHAPE_ROWS if args.scattered else ctx.packets_shape(n, size)

    if args.scattered:
        import struct
        u64s = lambda base, stride: struct.pack("<%dQ" % n, *[base + stride * i for i in range(n)])
        d_pp, d_cp, d_ln = lib.DeviceBuffer(8 * n, device=dev), lib.DeviceBuffer(8 * n, device=dev), lib.DeviceBuffer(4 * n, device=dev)
        d_pp.upload(u64s(d_pt.ptr, size)); d_cp.upload(u64s(d_ct.ptr, size)); d_ln.upload(struct.pack("<%dI" % n, *([size] * n)))
        skw = {}
        if al:
            d_ap, d_aln = lib.DeviceBuffer(8 * n, device=dev), lib.DeviceBuffer(4 * n, device=dev)
            d_ap.upload(u64s(d_aad.ptr, al)); d_aln.upload(struct.pack("<%dI" % n, *([al] * n)))
            skw = dict(d_aad_ptr=d_ap.ptr, d_aad_len=d_aln.ptr)

    def step():
        if args.scattered:
            return ctx.messages_crypt_dev(args.decrypt, n, d_ivs.ptr, (d_cp if args.decrypt else d_pp).ptr, d_ln.ptr, (d_pp if args.decrypt else d_cp).ptr, d_tags.ptr, **skw)
        ctx.packets_crypt_dev(args.decrypt, n, d_ivs.ptr, d_ct.ptr if args.decrypt else d_pt.ptr, d_pt.ptr if args.decrypt else d_ct.ptr, d_tags.ptr, pkt_len=size, **akw)

    if args.decrypt:
        ctx.packets_crypt_dev(False, n, d_ivs.ptr, d_pt.ptr, d_ct.ptr, d_tags.ptr, pkt_len=size, **akw)
    step()
    lib.dev_sync(dev)
    tags = bytes(d_tags.download())
    one = lib.Context(key, device=dev)
    d_one = lib.DeviceBuffer(size, device=dev)
    d_a1 = lib.DeviceBuffer(max(al, 16), device=dev) if al else None            # (a message's AAD copied to an aligned buffer of its own for the single-message call)
    sample = sorted(p_ for p_ in set([0, n - 1] + [(k * 977) % n for k in range(6)] + [(k * 976) % n for k in range(1, 4)]) if (p_ * size) % 16 == 0)   # the single-message path wants 16-byte aligned buffers
    tag_ok = True
    for p_ in sample:
        if al:
            d_a1.upload(bytes(d_aad.download(al, p_ * al)))
        t = one.encrypt_dev(ivs[12 * p_:12 * p_ + 12], d_pt.ptr + p_ * size, size, d_one.ptr, d_aad=d_a1.ptr if al else None, aad_len=al)
        same_ct = args.decrypt or hashlib.sha256(bytes(d_one.download())).digest() == hashlib.sha256(bytes(d_ct.download(size, p_ * size))).digest()
        tag_ok = tag_ok and t == tags[16 * p_:16 * p_ + 16] and same_ct
    if not tag_ok:
        log("PARITY FAILURE msgs: the packets call and the single-message path disagree")
    for _ in range(max(args.warmup, 1)):                        # the warm-up steps directly in front of the timed ones (the cross-check above leaves the chip idle for a while)
        step()
    lib.dev_sync(dev)
    if ex is not None:
        ex.barrier()
        lib.dev_sync(dev)
    t0 = time.perf_counter()
    done = 0
    while done < args.steps:                                    # the calls are asynchronous and their tags stay on the device: queued back to back (at most 64 deep), waited for once
        for _ in range(min(64, args.steps - done)):
            step()
            done += 1
        lib.dev_sync(dev)
    if ex is not None:
        ex.barrier()
    dt = time.perf_counter() - t0
    if ex is not None:
        dt = ex.allreduce(dt, "max")
        tag_ok = bool(ex.allreduce(1.0 if tag_ok else 0.0, "min"))
    value = n * N * size * args.steps / dt / GiB
    tm = lib.Timer(device=dev)
    k_ms = []
    for _ in range(min(5, max(1, args.steps))):
        tm.start(ctx.stream())
        step()
        tm.stop(ctx.stream())
        k_ms.append(tm.ms())
    tm.close()
    if rank != 0:
        return tag_ok
    avg_s = statistics.mean(k_ms) / 1e3
    alg_bytes = n * (2 * size + 12 + 16 + al)
    achieved = alg_bytes / avg_s
    so_sha = sha256_file(SO)
    pm = pmc_summary("rows_1m") if (n, N, size, key_bits, al) == (cfg["n_pkts"], 1, cfg["pkt_len"], cfg["key_bits"], 0) and not args.decrypt and not args.scattered else {}
    same_build = bool(pm) and pm.get("so_sha256") == so_sha
    nr = key_bits // 32 + 6
    line = {
        "metric": "GiB/s plaintext, AES-%d-GCM, %d messages of %d bytes%s under one key as %s, bit-exact tags" % (key_bits, n * N, size, " with %d bytes of AAD each" % al if al else "",
                                                                                                                   "one call over arrays of their addresses and lengths" if args.scattered else "the packets of one call"),
        "value": round(value, 3), "unit": "GiB/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong" if N > 1 else "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "msgs: %d x %d B AES-%d-GCM messages under ONE key (stream 0x4B4559), per-message IV (stream 0x4956), plaintext stream 0xAE5C0055, one "
                               "aesgcm_packets_crypt_dev call per step%s%s" % (n * N, size, key_bits, ", DECRYPT" if args.decrypt else "", "; %d replicas of %d messages, no collective" % (N, n) if N > 1 else ""),
                   "messages": n * N, "messages_per_gpu": n, "message_bytes": size, "aad_bytes": al, "scattered": bool(args.scattered), "key_bits": key_bits, "shape": "rows" if shape == lib.SHAPE_ROWS else "%d lanes per packet" % shape,
                   "parallelism": "single" if N == 1 else "replicas%d" % N,
                   "exchange": None if ex is None else {"backend": ex.name, "ranks_seen": ex.world, "use": "barrier and max-over-ranks timing only", "torch": "not imported"}},
        "tag_ok": tag_ok, "tags_checked": len(sample),
        "roofline": {"bound": "hbm", "kernel": ("k_rows<%d,%d> + k_rows_close (the rows of all messages through k_body's row loop; one call)" % (nr, int(args.decrypt))) if shape == lib.SHAPE_ROWS else "the packet kernels",
                     "achieved": round(achieved / 1e9, 2), "peak": HBM_PEAK_BYTES_PER_S / 1e9, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_BYTES_PER_S, 4),
                     "traffic": pm.get("hbm_bytes_per_launch") if same_build else None, "traffic_source": "profiles/pmc_rows_1m.json" if pm else None,
                     "traffic_build": {"pmc_so_sha256": pm.get("so_sha256"), "running_so_sha256": so_sha, "running_git": git_head(), "match": same_build},
                     "alg_bytes_per_launch": alg_bytes, "launches_timed": len(k_ms), "avg_launch_ms": round(avg_s * 1e3, 4),
                     "timing": "HIP events on the context's stream around each call (both launches) in a separate %d-step pass after the timed region" % len(k_ms)},
    }
    if cpu_base is not None:
        line["cpu_baseline"] = cpu_base
    emit(line)
    return tag_ok


# ------------------------------------------------------------------------------------------------ frames under one key
def run_frames(args, rank, world, dev, ex, cpu_base):
    """--config frames: n MACsec-shaped frames under ONE key through the offset arrays of one aesgcm_packets_crypt_dev call per step (routed on the device: frames this
    short all take the packet kernels, a lane per frame when they fill the chip).  N > 1: replicas of n / N frames, no collective on the data path.  Parity in the run:
    a sample of the frames through the single-message path of the same library (pinned to the libcrypto fixtures and the 802.1AE vectors by the test-suite).
    The line carries the roofline of the call (algorithmic bytes = 2 len + AAD + 12 + 16 per frame), the ceiling of the formulation (the same packet kernel without the
    data's loads and stores, aesgcm_frames_ceiling_probe_dev) and the CPU baseline (libcrypto, per-frame loop in C)."""
    import struct
    import numpy as np
    import aesgcm_amd  # noqa: F401
    from aesgcm_amd import lib, sharding
    from aesgcm_amd.build import SO
    cfg = CONFIGS["frames"]
    N = world
    n_all = args.n_pkts if args.n_pkts else cfg["n_pkts"]
    n = n_all // N
    first = rank * n
    key_bits = args.key_bits or cfg["key_bits"]
    al = cfg["aad_len"] if args.aad_len is None else args.aad_len
    standard = (n_all, key_bits, al) == (cfg["n_pkts"], cfg["key_bits"], cfg["aad_len"]) and not args.decrypt
    key = sharding.splitmix64_bytes(KEY_SEED, key_bits // 8)
    d_w = lib.DeviceBuffer(8 * n, device=dev)                    # the lengths' stream, made on the device (the host generator is a Python loop)
    d_w.fill_splitmix64(cfg["len_seed"], first)
    w = np.frombuffer(bytes(d_w.download()), dtype="<u8")
    d_w.free()
    lens = (64 + (w % np.uint64(1451))).astype(np.int64)
    doff = np.zeros(n + 1, dtype=np.uint64); doff[1:] = np.cumsum(lens)
    aoff = np.arange(n + 1, dtype=np.uint64) * np.uint64(al)
    total = int(doff[-1])
    d_ivw, d_ivs = lib.DeviceBuffer(16 * n, device=dev), lib.DeviceBuffer(12 * n + 16, device=dev)
    d_ivw.fill_splitmix64(IV_SEED, first * 2)
    ivw = bytes(d_ivw.download())
    ivs = b"".join(ivw[16 * p:16 * p + 12] for p in range(n))
    d_ivs.upload(ivs)
    d_ivw.free()
    d_pt, d_ct, d_tags = lib.DeviceBuffer(total + 64, device=dev), lib.DeviceBuffer(total + 64, device=dev), lib.DeviceBuffer(16 * n, device=dev)
    d_pt.fill_splitmix64(cfg["pt_seed"] + rank, nbytes=(total + 64) // 8 * 8)
    d_aad = lib.DeviceBuffer(al * n + 64, device=dev)
    d_aad.fill_splitmix64(0x414144 + rank, nbytes=(al * n + 64) // 8 * 8)
    d_doff, d_aoff = lib.DeviceBuffer(8 * (n + 1), device=dev), lib.DeviceBuffer(8 * (n + 1), device=dev)
    d_doff.upload(doff.tobytes()); d_aoff.upload(aoff.tobytes())
    akw = dict(d_aad=d_aad.ptr, d_aad_off=d_aoff.ptr) if al else {}
    ctx = lib.Context(key, device=dev)
    for kv in args.opt:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))

    def step():
        ctx.packets_crypt_dev(args.decrypt, n, d_ivs.ptr, d_ct.ptr if args.decrypt else d_pt.ptr, d_pt.ptr if args.decrypt else d_ct.ptr, d_tags.ptr, d_data_off=d_doff.ptr, **akw)

    def barrier():
        lib.dev_sync(dev)
        if ex is not None:
            ex.barrier()
            lib.dev_sync(dev)

    if args.decrypt:
        ctx.packets_crypt_dev(False, n, d_ivs.ptr, d_pt.ptr, d_ct.ptr, d_tags.ptr, d_data_off=d_doff.ptr, **akw)
    step()
    lib.dev_sync(dev)
    route = ctx.last_route()
    status = ctx.status()
    tags = bytes(d_tags.download())
    # ---- parity: a sample of the frames through the single-message path (host buffers) of the same library
    one = lib.Context(key, device=dev)
    sample = sorted(set([0, n - 1] + [(k * 104729) % n for k in range(14)] + [int(np.argmin(lens)), int(np.argmax(lens))]))
    tag_ok = status == (lib.STATUS_OK, 0)
    for p_ in sample:
        lo, ln = int(doff[p_]), int(lens[p_])
        pt = bytes(d_ct.download(ln, lo)) if args.decrypt else bytes(d_pt.download(ln, lo))     # (decrypt steps run in place of the buffers' roles: d_ct holds the ciphertext made above)
        aad = bytes(d_aad.download(al, al * p_)) if al else b""
        if args.decrypt:
            want_pt, want_tag = one.decrypt(ivs[12 * p_:12 * p_ + 12], aad, pt)
            got = bytes(d_pt.download(ln, lo))
            tag_ok = tag_ok and got == want_pt and tags[16 * p_:16 * p_ + 16] == want_tag
        else:
            want_ct, want_tag = one.encrypt(ivs[12 * p_:12 * p_ + 12], aad, pt)
            tag_ok = tag_ok and bytes(d_ct.download(ln, lo)) == want_ct and tags[16 * p_:16 * p_ + 16] == want_tag
    if not tag_ok:
        log("PARITY FAILURE frames: the packets call and the single-message path disagree (status %r)" % (status,))
    for _ in range(max(args.warmup, 1)):
        step()
    barrier()
    t0 = time.perf_counter()
    done = 0
    while done < args.steps:                                    # asynchronous calls, tags stay on the device: queued back to back (at most 64 deep), waited for once
        for _ in range(min(64, args.steps - done)):
            step()
            done += 1
        lib.dev_sync(dev)
    barrier()
    dt = time.perf_counter() - t0
    if ex is not None:
        dt = ex.allreduce(dt, "max")
        tag_ok = bool(ex.allreduce(1.0 if tag_ok else 0.0, "min"))
        total_all = ex.allreduce(float(total), "sum")
    else:
        total_all = float(total)
    value = total_all * args.steps / dt / GiB
    tm = lib.Timer(device=dev)
    k_ms, c_ms = [], []
    for _ in range(min(7, max(1, args.steps))):
        tm.start(ctx.stream()); step(); tm.stop(ctx.stream())
        k_ms.append(tm.ms())
    if not args.decrypt:
        try:
            ctx.frames_ceiling_probe_dev(n, d_ivs.ptr, d_doff.ptr, d_tags.ptr, **akw)
            lib.dev_sync(dev)
            for _ in range(min(7, max(1, args.steps))):
                tm.start(ctx.stream()); ctx.frames_ceiling_probe_dev(n, d_ivs.ptr, d_doff.ptr, d_tags.ptr, **akw); tm.stop(ctx.stream())
                c_ms.append(tm.ms())
            probe_route = ctx.last_route()
        except lib.AesGcmError as e:
            log("bench.py: frames ceiling probe failed: %r" % (e,))
            c_ms = []
    tm.close()
    if rank != 0:
        return tag_ok
    avg_s = statistics.mean(k_ms) / 1e3
    alg_bytes = int(2 * total + n * (al + 12 + 16))
    achieved = alg_bytes / avg_s
    so_sha = sha256_file(SO)
    pm = pmc_summary("frames") if standard and N == 1 else {}
    same_build = bool(pm) and pm.get("so_sha256") == so_sha
    nr = key_bits // 32 + 6
    lanes = int(route["lanes"])
    kname = ("k_pktl<%d,%d,0>" % (nr, int(args.decrypt))) if lanes == 1 else ("k_pktg<%d,%d,%d>" % (nr, int(args.decrypt), {4: 2, 8: 3, 16: 4, 64: 6}.get(lanes, 0))) if lanes else "k_rows (the device sent every frame by rows)"
    roofline = {"bound": "hbm", "kernel": "%s: %d lane(s) per frame, frames taken by falling size class (k_len_hist / _scan / _scatter in front; the device's route: mark %d, %d of %d frames to the packet kernels)"
                                          % (kname, lanes, route["route_min"], route["n_small"], n),
                "achieved": round(achieved / 1e9, 2), "peak": HBM_PEAK_BYTES_PER_S / 1e9, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_BYTES_PER_S, 4),
                "traffic": pm.get("hbm_bytes_per_launch") if same_build else None, "traffic_source": "profiles/pmc_frames.json" if pm else None,
                "traffic_build": {"pmc_so_sha256": pm.get("so_sha256"), "running_so_sha256": so_sha, "running_git": git_head(), "match": same_build},
                "alg_bytes_per_launch": alg_bytes, "alg_bytes": "per frame 2 x length + %d (AAD) + 12 (IV) + 16 (tag)" % al, "launches_timed": len(k_ms), "avg_launch_ms": round(avg_s * 1e3, 4),
                "timing": "HIP events on the context's stream around each call (the sort, the packet kernel, the empty row launches) in a separate %d-step pass after the timed region" % len(k_ms),
                "lds_busy_frac": (pm.get("lds") or {}).get("lds_busy_frac") if same_build else None}
    if c_ms:
        c_s = statistics.mean(c_ms) / 1e3
        roofline["formulation_ceiling"] = {"kernel": "the same call with the packet kernel in its PROBE form (%d lane(s) per frame: the instruction stream without the data's loads and stores; IVs, offsets, AAD and tags still move)" % int(probe_route["lanes"]),
                                           "avg_launch_ms": round(c_s * 1e3, 4), "gib_per_s": round(total / c_s / GiB, 1), "alg_gb_per_s": round(alg_bytes / c_s / 1e9, 1),
                                           "frac_of_hbm_peak": round(alg_bytes / c_s / HBM_PEAK_BYTES_PER_S, 4)}
        roofline["achieved_over_ceiling"] = round(c_s / avg_s, 4)
    line = {
        "metric": "GiB/s plaintext, AES-%d-GCM, %d frames of 64 .. 1514 bytes with %d bytes of AAD each under one key (offset arrays, one call), bit-exact tags" % (key_bits, n * N, al),
        "value": round(value, 3), "unit": "GiB/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong" if N > 1 else "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "frames%s: %d MACsec-shaped frames (64 .. 1514 bytes, lengths stream 0x4C454E; %d bytes of AAD each, stream 0x414144), AES-%d-GCM under ONE key (stream 0x4B4559), "
                               "per-frame IV (stream 0x4956), plaintext stream 0xAE5C0006, packed back to back; one aesgcm_packets_crypt_dev call with offset arrays per step%s%s"
                               % ("" if standard else " (custom)", n * N, al, key_bits, ", DECRYPT + computed tags" if args.decrypt else "", "; %d replicas of %d frames, no collective" % (N, n) if N > 1 else ""),
                   "frames_per_gpu": n, "bytes_per_gpu": total, "mean_frame_bytes": round(total / n, 1), "aad_bytes": al, "key_bits": key_bits, "mframes_per_s": round(n * N * args.steps / dt / 1e6, 2),
                   "parallelism": "single" if N == 1 else "replicas%d" % N,
                   "exchange": None if ex is None else {"backend": ex.name, "ranks_seen": ex.world, "use": "barrier and max-over-ranks timing only", "torch": "not imported"}},
        "tag_ok": tag_ok, "tags_checked": len(sample), "roofline": roofline,
    }
    if cpu_base is not None:
        line["cpu_baseline"] = cpu_base
    emit(line)
    return tag_ok


# ------------------------------------------------------------------------------------------------ messages in flight
def run_inflight(args, dev):
    """Sustained rate of mid-size messages with K of them queued (the reference's back-to-back packets under one key,
    src/gcm_gctr.vhd:142-144).  K contexts of the same key, each with its own stream, scratch and host tag slot, take the
    messages in turn; a call is enqueued with tag = NULL and returns at once; when the context comes round again the tag of
    its previous message is collected (aesgcm_last_tag: a poll of the pinned host slot) and compared with the tag a waited
    call produced for the same buffer before the timed region.  K = 1 is therefore "enqueue, then wait for the tag", the
    waited form the default bench line uses.  The messages rotate over a ring of distinct buffers (plaintext, ciphertext, IV)
    whose total exceeds the Infinity Cache, so no step re-reads what the previous one left in it."""
    import aesgcm_amd  # noqa: F401
    from aesgcm_amd import lib, sharding
    K = max(1, args.inflight)
    key_bits = args.key_bits or 256
    size = int(args.gib_per_gpu * GiB) // 16 * 16
    R = max(K, 2, min(255, int(args.ring_gib * GiB + size - 1) // size))
    key = sharding.splitmix64_bytes(KEY_SEED, key_bits // 8)
    iv0 = sharding.splitmix64_bytes(IV_SEED, 12)
    ivs = [iv0[:8] + r.to_bytes(4, "big") for r in range(R)]
    ctxs = [lib.Context(key, device=dev) for _ in range(K)]
    for c in ctxs:
        if args.half >= 0:
            c.set_option("cyc_half", args.half)                  # 0 / 1: never / always; else the library's own rule (half when another context has a message under way)
        for kv in args.opt:
            k, v = kv.split("=", 1)
            c.set_option(k, int(v, 0))
    d_pt = lib.DeviceBuffer(size * R, device=dev)
    d_ct = lib.DeviceBuffer(size * R, device=dev)
    d_pt.fill_splitmix64(0xAE5C0003, 0)
    lib.dev_sync(dev)
    # the waited form, once per ring slot: the reference tags of the queued calls (and, for the standard 16 GiB / 1 GiB messages, nothing else is needed --
    # the queued and the waited form run the same launches; tests/test_gpu_cyclic.py and tests/test_gpu_inflight.py hold both to the oracle)
    ref = [ctxs[0].encrypt_dev(ivs[r], d_pt.ptr + r * size, size, d_ct.ptr + r * size) for r in range(R)]
    pending = [None] * K
    bad = [0]

    def collect(j):
        if pending[j] is not None:
            if ctxs[j].last_tag() != ref[pending[j]]:
                bad[0] += 1
            pending[j] = None

    def run(n, i0):
        for i in range(i0, i0 + n):
            j, r = i % K, i % R
            collect(j)
            ctxs[j].encrypt_dev(ivs[r], d_pt.ptr + r * size, size, d_ct.ptr + r * size, want_tag=False)
            pending[j] = r
        for j in range(K):
            collect(j)
        return i0 + n

    est_ms = size / GiB + 0.022
    steps = args.steps if args.steps_given else min(20000, max(20, int(300.0 / est_ms)))
    warm = args.warmup if args.warmup_given else min(10000, max(3, int(100.0 / est_ms)))
    i = run(warm, 0)
    lib.dev_sync(dev)
    t0 = time.perf_counter()
    i = run(steps, i)
    lib.dev_sync(dev)
    dt = time.perf_counter() - t0
    value = size * steps / dt / GiB
    # the kernel alone: HIP events around each launch on its stream, ONE context, launches back to back, waited
    c0 = ctxs[0]
    c0.timing_enable(True)
    c0.timing_read(reset=True)
    for r in range(min(R, 8)):
        c0.encrypt_dev(ivs[r], d_pt.ptr + r * size, size, d_ct.ptr + r * size)
    lib.dev_sync(dev)
    n_launch, kernel_ms = c0.timing_read(reset=True)
    c0.timing_enable(False)
    avg_s = kernel_ms / 1e3 / max(n_launch, 1)
    _, body_blocks = c0.split(size, 0)
    blocks = body_blocks if body_blocks else size // 16
    achieved = 32 * blocks / avg_s if avg_s > 0 else 0.0
    line = {
        "metric": "GiB/s plaintext, AES-%d-GCM, %.6g MiB messages, %d in flight, bit-exact tags" % (key_bits, size / (1 << 20), K),
        "value": round(value, 3), "unit": "GiB/s", "n_gpus": 1, "steps": steps, "warmup": warm,
        "ms_per_step": round(dt / steps * 1e3, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "custom: AES-%d-GCM messages of %d bytes under one key, %d queued (contexts rotate, tag = NULL, tags collected one turn late through "
                               "the host slot), ring of %d message buffers = %.3g GiB, SplitMix64 PT seed 0xAE5C0003, empty AAD" % (key_bits, size, K, R, size * R / GiB),
                   "bytes_per_message": size, "inflight": K, "ring": R, "half_shape": {-1: "library rule (when another context has a message under way)", 0: "never", 1: "always"}[args.half], "context_options": args.opt, "parallelism": "single", "key_bits": key_bits, "us_per_message": round(dt / steps * 1e6, 2)},
        "tag_ok": bad[0] == 0, "tags_checked": steps + warm, "tags_wrong": bad[0],
        "roofline": {"bound": "hbm", "kernel": "%s<%d,ENC> (fused AES-CTR + GHASH), one message at a time" % ("k_body" if body_blocks else "k_main", key_bits // 32 + 6),
                     "achieved": round(achieved / 1e9, 2), "peak": HBM_PEAK_BYTES_PER_S / 1e9, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_BYTES_PER_S, 4),
                     "traffic": None, "alg_bytes_per_launch": 32 * blocks, "launches_timed": n_launch, "avg_launch_ms": round(avg_s * 1e3, 5),
                     "sustained_frac": round(2 * size * steps / dt / HBM_PEAK_BYTES_PER_S, 4),
                     "timing": "HIP events on the launch stream, one context, waited calls, separate pass after the timed region"},
    }
    emit(line)
    if bad[0]:
        log("PARITY FAILURE: %d of %d queued tags differ from the waited call's" % (bad[0], steps + warm))
    return bad[0] == 0


# ------------------------------------------------------------------------------------------------ one process, N devices
def run_single_process(args):
    """The N-GPU stream job (cfg4 cut to N devices, as the N-rank launch runs it) from ONE process: aesgcm_mgpu_* --
    ncclCommInitAll, one grouped 16-byte all-gather per message, every device's context derived from the key locally.
    The fallback of self_launch when no RCCL communicator comes up between processes (and --single-process asks for it).
    The messages of a step are queued without a host synchronisation and their tags collected by one finalize launch (aesgcm_mgpu_last_tags), as the N-rank
    path does; the line names the path it is (config.exchange.backend) and says whether its tags equal a single device's ("validated")."""
    import aesgcm_amd  # noqa: F401
    from aesgcm_amd import lib, sharding
    from aesgcm_amd.build import SO
    N = args.gpus
    have = lib.device_count()
    if have < N:
        log("bench.py: --gpus %d but the library sees %d device(s)" % (N, have))
        return 2
    if args.config == "cfg5" or args.decrypt:
        log("bench.py: the single-process path runs the stream configs, encrypt")
        return 2
    cfg = dict(key_bits=256, gib=16.0, pt_seed=0xAE5C0004)
    standard = args.gib_per_gpu is None and args.key_bits is None
    if args.gib_per_gpu is not None:
        cfg["gib"] = args.gib_per_gpu
    if args.key_bits is not None:
        cfg["key_bits"] = args.key_bits
    per_gpu = int(cfg["gib"] * GiB) // (16 * 2 * N) * (16 * 2 * N)
    key_bits = cfg["key_bits"]
    key = sharding.splitmix64_bytes(KEY_SEED, key_bits // 8)
    iv0 = sharding.splitmix64_bytes(IV_SEED, 12)
    plans = [sharding.plan_job(N, per_gpu, r) for r in range(N)]
    M = len(plans[0])
    try:
        mg = lib.MultiGpu(key, list(range(N)))
    except lib.AesGcmError as e:
        log("bench.py: aesgcm_mgpu_create over %d devices failed: %r / %s" % (N, e, lib.load().aesgcm_comm_last_error().decode()))
        return 1
    d_pt = [lib.DeviceBuffer(per_gpu, device=r) for r in range(N)]
    d_ct = [lib.DeviceBuffer(per_gpu, device=r) for r in range(N)]
    for r in range(N):
        for m in plans[r]:
            d_pt[r].fill_splitmix64(cfg["pt_seed"], m["stream_word"], nbytes=m["len"], offset=m["off"])
    for r in range(N):
        lib.dev_sync(r)
    ivs = [sharding.tweak_iv(iv0, m["iv_tweak"]) for m in plans[0]]

    def step():
        # the M messages are only ENQUEUED (tag = NULL: no host synchronisation on any device) and their tags collected with one finalize launch on device 0, as
        # the N-rank path does with aesgcm_shard_finalize_batch_dev (round 5; until then every message waited for its tag)
        for i in range(M):
            mg.crypt_dev(False, ivs[i], [d_pt[r].ptr + plans[r][i]["off"] for r in range(N)], [plans[r][i]["len"] for r in range(N)],
                         [d_ct[r].ptr + plans[r][i]["off"] for r in range(N)], want_tag=False)
        return mg.last_tags(M)

    def sync_all():
        for r in range(N):
            lib.dev_sync(r)

    if M > 8:
        log("bench.py: the single-process path queues at most 8 messages per step (this job has %d)" % M)
        return 2
    tags = None
    for _ in range(max(args.warmup, 1)):
        tags = step()
    sync_all()
    checked = []
    for m, t in zip(plans[0], tags):
        fx = load_fixture("cfg4_aes256_msg%d_32GiB" % m["msg"]) if standard and m["total"] == 32 * GiB else None
        if fx is not None:
            checked.append(t.hex() == fx["tag"])
    tag_ok = all(checked) if checked else None
    # whatever the size: every message once more on device 0 ALONE -- shard by shard from the same SplitMix64 stream, the partials folded there -- and the
    # tags compared: the shard order of plan_job and the cumulative first blocks of aesgcm_mgpu_crypt_dev must describe the same message
    validated = True
    try:
        one = lib.Context(key, device=0)
        biggest = max(plans[r][i]["len"] for r in range(N) for i in range(M))
        d_a, d_b, d_parts = lib.DeviceBuffer(biggest, device=0), lib.DeviceBuffer(biggest, device=0), lib.DeviceBuffer(16 * N, device=0)
        for i in range(M):
            for r in range(N):
                m = plans[r][i]
                d_a.fill_splitmix64(cfg["pt_seed"], m["stream_word"], nbytes=m["len"])
                one.shard_crypt_dev(False, ivs[i], d_a.ptr, m["len"], d_b.ptr, m["first_block"], m["total"], d_parts.ptr + 16 * r)
            alone = one.shard_finalize_dev(ivs[i], d_parts.ptr, N, 0, plans[0][i]["total"])
            validated = validated and alone == tags[i]
        for o in (d_a, d_b, d_parts):
            o.free()
        one.close()
    except lib.AesGcmError as e:
        log("bench.py: the single-device cross-check could not run: %r" % (e,))
        validated = False
    if tag_ok is False or not validated:
        log("PARITY FAILURE (single process): tags=%s fixture %s single-device %s" % ([t.hex() for t in tags], tag_ok, validated))
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    value = per_gpu * N * args.steps / dt / GiB

    ctx0 = mg.context(0)                                          # device 0's fused kernel under HIP events, separate short pass
    ctx0.timing_enable(True)
    ctx0.timing_read(reset=True)
    for _ in range(min(3, max(1, args.steps))):
        step()
    sync_all()
    n_launch, kernel_ms = ctx0.timing_read(reset=True)
    ctx0.timing_enable(False)
    m0 = plans[0][0]
    _, body_blocks = ctx0.split(m0["len"], m0["first_block"])
    blocks_per_launch = body_blocks if body_blocks else m0["len"] // 16
    avg_s = kernel_ms / 1e3 / max(n_launch, 1)
    alg_bytes = 32 * blocks_per_launch
    achieved = alg_bytes / avg_s if avg_s > 0 else 0.0
    geo = ctx0.geometry(body=bool(body_blocks))
    line = {
        "metric": "GiB/s plaintext, AES-%d-GCM %.3g GiB stream, bit-exact tag" % (key_bits, cfg["gib"]),
        "value": round(value, 3), "unit": "GiB/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "cfg4 cut to %d devices: %d AES-%d-GCM message(s) of %.3g GiB (SplitMix64 seed 0xAE5C0004), each sharded over all %d "
                               "devices by ONE process (aesgcm_mgpu_*), one grouped 16 B all-gather per message" % (N, M, key_bits, plans[0][0]["total"] / GiB, N),
                   "bytes_per_gpu": per_gpu, "messages_per_step": M, "parallelism": "shard%d" % N, "key_bits": key_bits,
                   "workgroups": geo["workgroups"], "wg_lanes": geo["wg_lanes"], "lds_bytes_per_wg": geo["lds_bytes"],
                   "exchange": {"backend": "rccl (single process)", "ranks_seen": mg.n_ranks, "init": "ncclCommInitAll", "torch": "not imported"}},
        "tag_ok": tag_ok, "validated": validated, "tags": [t.hex() for t in tags],
        "roofline": {"bound": "hbm", "kernel": "%s<%d,ENC> (fused AES-CTR + GHASH), device 0" % ("k_body" if body_blocks else "k_main", key_bits // 32 + 6),
                     "achieved": round(achieved / 1e9, 2), "peak": HBM_PEAK_BYTES_PER_S / 1e9, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_BYTES_PER_S, 4),
                     "traffic": None, "alg_bytes_per_launch": alg_bytes, "launches_timed": n_launch, "avg_launch_ms": round(avg_s * 1e3, 4),
                     "timing": "HIP events on device 0's launch stream in a separate pass after the timed region",
                     "traffic_build": {"running_so_sha256": sha256_file(SO), "running_git": git_head()}},
    }
    emit(line)
    ok = mg.n_ranks == N and tag_ok is not False and validated
    mg.close()
    return 0 if ok else 1


# ------------------------------------------------------------------------------------------------ main
def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20; cfg2, whose step is 1 ms: 100)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps in front (default 3; cfg2: 50 -- the chip takes 10 - 20 ms under load to reach its clock, "
                    "profiles/archive/r03c/cfg2_warmup.txt: 3 warmup + 20 timed steps of 1 ms measure the ramp, 1030 GiB/s against 1165)")
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS), help="workload (default: the metric's cfg3; N > 1 stream runs are cfg4)")
    ap.add_argument("--gib-per-gpu", type=float, default=None, help="override: resident plaintext per GPU (no fixture check)")
    ap.add_argument("--key-bits", type=int, default=None, choices=(128, 192, 256), help="override (no fixture check)")
    ap.add_argument("--n-pkts", type=int, default=None, help="cfg5 override: packets in all (no fixture check)")
    ap.add_argument("--pkt-len", type=int, default=None, help="cfg5 override: bytes per packet (no fixture check)")
    ap.add_argument("--scattered", action="store_true", help="--config msgs: the same messages through aesgcm_messages_crypt_dev (device arrays of addresses and lengths: messages wherever they live)")
    ap.add_argument("--aad-len", type=int, default=None, help="--config msgs: bytes of AAD per message (a header: 13 for TLS-shaped records; default none); --config frames: default 28")
    ap.add_argument("--decrypt", action="store_true", help="time decrypt + authenticate instead of encrypt (N = 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="rccl", choices=("rccl", "nccl", "file", "gloo"),
                    help="exchange for N > 1: rccl (= nccl, the product: RCCL inside the library) or file (= gloo of round 1: "
                         "debug, partials through the host, for several ranks on ONE GPU; needs --allow-file-exchange)")
    ap.add_argument("--allow-file-exchange", action="store_true",
                    help="accept a run whose exchange is not RCCL (debug); without it such a run prints its line and exits %d" % EXIT_NOT_RCCL)
    ap.add_argument("--one-device", action="store_true", help="debug: every rank uses GPU 0 (RCCL refuses that: the file exchange comes up)")
    ap.add_argument("--selfcheck", action="store_true",
                    help="rank 0 also encrypts every whole message alone and compares tags (needs the extra memory)")
    ap.add_argument("--contexts", type=int, default=0,
                    help="N > 1 / --emulate-rank: contexts (stream + scratch set each) a rank's messages rotate over; 0 = the default, 2 (measured best of 1 / 2 / 4, profiles/archive/r03/emulate_rank.txt)")
    ap.add_argument("--no-batch-finalize", action="store_true", help="debug: one aesgcm_shard_finalize_strided_dev call per message instead of the batched one")
    ap.add_argument("--no-chain", action="store_true", help="debug: do not chain message i's fused kernel behind message i-1's (the contexts start together)")
    ap.add_argument("--emulate-rank", type=int, default=None, help="on ONE GPU, run exactly this rank's step of the --of W job")
    ap.add_argument("--of", type=int, default=8, help="world size emulated by --emulate-rank")
    ap.add_argument("--launch-timeout", type=float, default=3000.0, help="self-launch: seconds before hung ranks are killed")
    ap.add_argument("--inflight", type=int, default=0,
                    help="N = 1, with --gib-per-gpu S: sustained rate of S-sized messages with K of them queued -- K contexts (stream + scratch each) "
                         "rotate, every call is enqueued with tag = NULL and its tag is collected through the host slot when the context comes round "
                         "again; the messages rotate over a ring of buffers larger than the 256 MB Infinity Cache (--ring-gib)")
    ap.add_argument("--ring-gib", type=float, default=1.0, help="--inflight: total plaintext in the ring of message buffers")
    ap.add_argument("--half", type=int, default=-1, choices=(-1, 0, 1),
                    help="--inflight: the cyclic rows in their half shape (context option cyc_half: 256 workgroups of 512 lanes, two per CU, so that one message's "
                         "staging and closing run beside another's rows): 1 always, 0 never, -1 (default) the library's own rule -- half when another context has a message under way")
    ap.add_argument("--batch-lanes", type=int, default=0, choices=(0, 8, 16, 64),
                    help="cfg5, A/B runs only: force the lanes per packet (k_batch3 with 8 / 16 / 64) -- through the DEBUG build of the library "
                         "(libaesgcm_hip_dbg.so, include/aesgcm_debug.h); the line then says so and is not a product measurement")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="--inflight: a context option for every context (aesgcm_ctx_set_option), e.g. --opt cyc_max=0 --opt body_min=16777216; repeatable")
    ap.add_argument("--single-process", action="store_true",
                    help="N > 1: skip the one-process-per-GPU launch and let ONE fresh child drive all N devices (aesgcm_mgpu_*: ncclCommInitAll) -- "
                         "what the self-launch falls back to by itself when no RCCL communicator comes up between processes")
    ap.add_argument("--sp-child", action="store_true", help=argparse.SUPPRESS)      # internal: this process IS that child
    args = ap.parse_args(argv)
    # defaults: 3 warmup + 20 timed steps, except where a step is so short that those would sit inside the chip's clock ramp (10 - 20 ms under load): then about
    # 50 ms of warmup and 100 ms timed (cfg2, 1 ms per step: 50 + 100; --gib-per-gpu 0.0625, 0.09 ms: 556 + 1112)
    if args.gib_per_gpu is not None and not args.gib_per_gpu > 0:
        log("bench.py: --gib-per-gpu must be positive")
        return 2
    est_ms = None
    if args.config == "frames":                                  # ~1.4 ms per 2^20 frames
        est_ms = 1.4 * (args.n_pkts or CONFIGS["frames"]["n_pkts"]) / (1 << 20) / max(args.gpus, 1)
    elif args.config == "msgs":                                    # ~1.1 ms per GiB by rows
        est_ms = 1.1 * (args.n_pkts or CONFIGS["msgs"]["n_pkts"]) * (args.pkt_len or CONFIGS["msgs"]["pkt_len"]) / GiB / max(args.gpus, 1)
    elif args.config != "cfg5":                                    # ~1 ms per GiB; N > 1 stream runs are the 16 GiB-per-GPU cfg4 job whatever --config says
        est_ms = (args.gib_per_gpu if args.gib_per_gpu is not None else 16.0 if (args.gpus > 1 or args.emulate_rank is not None) else
                  {"cfg2": 1.0, "cfg3": 16.0}.get(args.config, 16.0)) * 1.0
    short_steps = est_ms is not None and 0.0 < est_ms < 5.0
    args.steps_given, args.warmup_given = args.steps is not None, args.warmup is not None
    if args.steps is None:
        args.steps = min(5000, max(20, int(100.0 / est_ms + 0.999))) if short_steps else 20
    if args.warmup is None:
        args.warmup = min(2500, max(3, int(50.0 / est_ms + 0.999))) if short_steps else 3

    if "RANK" not in os.environ and args.gpus > 1 and args.emulate_rank is None and not args.sp_child:
        return self_launch(args, argv)                           # before anything touches the GPU
    claim_stdout()                                               # this process measures: stdout is for the line, descriptor 1 for whatever libraries print
    if args.sp_child:
        return run_single_process(args)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    N = args.gpus
    if world != N:
        log("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to print a line for a different GPU count than asked for" % (N, world))
        return 2
    emu = args.emulate_rank
    if emu is not None and (N != 1 or not (0 <= emu < args.of) or args.config == "cfg5"):
        log("bench.py: --emulate-rank R --of W runs on one GPU (--gpus 1), 0 <= R < W, stream configs only")
        return 2

    # ---- CPU baseline FIRST: its worker processes are forked before this process touches the GPU
    cpu_base = None
    if N == 1 and rank == 0 and not args.no_cpu_baseline and emu is None:
        try:
            cpu_base = cpu_baseline(args.config)
        except Exception as e:                                  # the baseline must never break the bench line
            cpu_base = {"error": repr(e)}

    import aesgcm_amd  # noqa: F401
    from aesgcm_amd import comm, lib, sharding
    from aesgcm_amd.build import SO

    dev = 0 if args.one_device else local
    ex = None
    if world > 1:
        ex = comm.make_exchange(rank, world, dev, prefer="rccl" if args.backend in ("rccl", "nccl") else "file")
    not_rccl = ex is not None and ex.name != "rccl"
    if not_rccl and not args.allow_file_exchange and rank == 0:
        log("bench.py: the exchange is '%s', not RCCL -- the line below is NOT a multi-GPU RCCL measurement; exiting %d "
            "(--allow-file-exchange accepts it for debugging)" % (ex.name, EXIT_NOT_RCCL))
    if (not_rccl and not args.allow_file_exchange and args.backend in ("rccl", "nccl") and os.environ.get("AESGCM_SELF_LAUNCHED") == "1"
            and not args.one_device):
        # RCCL was asked for and did not come up: do not spend minutes measuring through the file exchange -- the launching parent
        # (self_launch) will start ONE process over all devices instead.  Every rank leaves the same way.
        log("bench.py rank %d: no RCCL communicator (%s); leaving so that the launcher can fall back to the single-process path" % (rank, ex.name))
        ex.barrier()
        ex.close()
        comm.finish(rank, world)
        return EXIT_NOT_RCCL

    if (not_rccl and not args.allow_file_exchange and args.backend in ("rccl", "nccl") and os.environ.get("AESGCM_SELF_LAUNCHED") != "1"
            and not args.one_device and args.config != "cfg5"):
        # The same under an outside launcher (torchrun starts one process per GPU): the ranks cannot measure an RCCL job, but ONE fresh process over all
        # devices can (aesgcm_mgpu_*: ncclCommInitAll).  Ranks 1 .. N-1 leave (their devices are free again); rank 0 starts that process as a CHILD --
        # a process that has touched the GPU is never replaced by another program -- relays its line and leaves with its exit code.
        log("bench.py rank %d: no RCCL communicator between the %d processes (%s)" % (rank, world, ex.name))
        ex.barrier()
        ex.close()
        comm.finish(rank, world)
        if rank != 0:
            return 0
        log("bench.py: falling back to ONE process driving all %d devices (ncclCommInitAll); the line below says so" % N)
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
        env.update(AESGCM_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("NCCL_DEBUG", "WARN")
        rcs, out0 = _run_children([([sys.executable, os.path.abspath(__file__)] + [a for a in argv if a != "--single-process"] + ["--sp-child"], env)], args.launch_timeout)
        if out0:
            os.write(_LINE_FD, out0)                              # the child's stdout is its line and nothing else (it claims its stdout too)
        return (rcs[0] if rcs[0] > 0 else 1) if rcs[0] else 0

    def finish(ok):
        if ex is not None:
            ex.barrier()
            ex.close()
            comm.finish(rank, world)
        if ok is False:
            return 1
        return EXIT_NOT_RCCL if (not_rccl and not args.allow_file_exchange) else 0

    if args.config == "cfg5":
        if args.batch_lanes:
            with lib.debug_library() as dbg:
                dbg.force(batch_lanes=args.batch_lanes)
                return finish(run_cfg5(args, rank, world, dev, ex, cpu_base))
        return finish(run_cfg5(args, rank, world, dev, ex, cpu_base))
    if args.config == "frames":
        if emu is not None:
            log("bench.py: --config frames has no --emulate-rank")
            return 2
        return finish(run_frames(args, rank, world, dev, ex, cpu_base))
    if args.config == "msgs":
        if emu is not None:
            log("bench.py: --config msgs has no --emulate-rank")
            return 2
        return finish(run_msgs(args, rank, world, dev, ex, cpu_base))
    if args.inflight:
        if N != 1 or args.gib_per_gpu is None or emu is not None or args.decrypt:
            log("bench.py: --inflight K needs --gib-per-gpu S, one GPU, encrypt")
            return 2
        return finish(run_inflight(args, dev))

    cfg = dict(CONFIGS[args.config])
    standard = args.gib_per_gpu is None and args.key_bits is None
    W = args.of if emu is not None else N                       # ranks of the job whose step this process runs
    R = emu if emu is not None else rank
    if W > 1:
        cfg = dict(key_bits=256, gib=16.0, pt_seed=0xAE5C0004, fixture=None)
    if args.gib_per_gpu is not None:
        cfg["gib"] = args.gib_per_gpu
    if args.key_bits is not None:
        cfg["key_bits"] = args.key_bits

    per_gpu = int(cfg["gib"] * GiB) // (16 * 2 * W) * (16 * 2 * W)
    key_bits = cfg["key_bits"]
    key = sharding.splitmix64_bytes(KEY_SEED, key_bits // 8)     # key and IV from the synthetic streams
    iv0 = sharding.splitmix64_bytes(IV_SEED, 12)

    plan = sharding.plan_job(W, per_gpu, R)
    M = len(plan)
    n_ctx = 1 if W == 1 else max(1, min(args.contexts if args.contexts > 0 else 2, M))
    ctxs = [lib.Context(key, device=dev) for _ in range(n_ctx)]
    ctx = ctxs[0]
    geo = ctx.geometry()
    d_pt = lib.DeviceBuffer(per_gpu, device=dev)
    d_ct = lib.DeviceBuffer(per_gpu, device=dev)

    def fixture_name(m):
        if not standard:
            return None
        return cfg["fixture"] if W == 1 else ("cfg4_aes256_msg%d_32GiB" % m["msg"] if m["total"] == 32 * GiB else None)

    msgs = []
    for m in plan:
        msgs.append(dict(iv=sharding.tweak_iv(iv0, m["iv_tweak"]), total=m["total"], first_block=m["first_block"],
                         off=m["off"], len=m["len"], fixture=fixture_name(m)))
        d_pt.fill_splitmix64(cfg["pt_seed"], m["stream_word"], nbytes=m["len"], offset=m["off"])
    if W == 1:
        workload = "%s: AES-%d-GCM, one %.3g GiB message, SplitMix64 PT seed 0x%X, empty AAD%s" % (
            args.config if standard else "custom", key_bits, per_gpu / GiB, cfg["pt_seed"], ", DECRYPT + authenticate" if args.decrypt else "")
    lib.dev_sync(dev)

    if W > 1:
        local_parts = lib.DeviceBuffer(16 * M, device=dev)
        gathered = lib.DeviceBuffer(16 * M * W, device=dev)         # [rank][message][16] after the all-gather
    if emu is not None:
        # the other ranks' REAL partials, computed here one shard at a time (untimed), so that the finalizes of the timed
        # step produce the job's real tags and can be checked against the cfg4 fixtures
        ex = EmulatedExchange(lib, R, W, dev)
        scratch = lib.DeviceBuffer(max(m["len"] for m in plan), device=dev)
        for rr in range(W):
            if rr == R:
                continue
            for i, m in enumerate(sharding.plan_job(W, per_gpu, rr)):
                scratch.fill_splitmix64(cfg["pt_seed"], m["stream_word"], nbytes=m["len"])
                ctx.shard_crypt_dev(False, sharding.tweak_iv(iv0, m["iv_tweak"]), scratch.ptr, m["len"], scratch.ptr, m["first_block"], m["total"],
                                    gathered.ptr + 16 * (rr * M + i))
            lib.dev_sync(dev)
        scratch.free()
    if W > 1:
        workload = ("%scfg4 cut to %d ranks: %d AES-%d-GCM message(s) of %.3g GiB (SplitMix64 seed 0xAE5C0004), each sharded over "
                    "all %d ranks, one 16 B x %d x %d all-gather per step (%s), %d context(s) per rank" % (
                        "rank %d of " % R if emu is not None else "", W, M, key_bits, msgs[0]["total"] / GiB, W, W, M, ex.name, n_ctx))

    expect_tag = None

    def step(cs=None):
        """one pass over the resident workload; returns the list of tags (bytes) -- fetching a tag syncs the stream"""
        cs = ctxs if cs is None else cs
        if W == 1:
            m = msgs[0]
            if args.decrypt:
                return [ctx.decrypt_dev(m["iv"], d_ct.ptr, m["len"], d_pt.ptr, tag=expect_tag)]
            return [ctx.encrypt_dev(m["iv"], d_pt.ptr, m["len"], d_ct.ptr)]
        for i, m in enumerate(msgs):                                 # message i on context i mod K: own stream, own scratch
            if len(cs) > 1 and i > 0 and not args.no_chain:
                cs[i % len(cs)].wait_fused(cs[(i - 1) % len(cs)])   # fused kernels back to back; message i-1's fold / combine run beside message i
            cs[i % len(cs)].shard_crypt_dev(False, m["iv"], d_pt.ptr + m["off"], m["len"], d_ct.ptr + m["off"], m["first_block"], m["total"],
                                            local_parts.ptr + 16 * i)
        for c in cs[1:]:
            cs[0].wait(c)                                            # the all-gather needs every context's partials: stream-ordered, no host sync
        ex.allgather_dev(local_parts.ptr, gathered.ptr, 16 * M, stream=cs[0].stream())
        if M <= 8 and not args.no_batch_finalize:                     # all M tags in one launch and one wait
            return cs[0].shard_finalize_batch_dev([m["iv"] for m in msgs], gathered.ptr, W, [m["total"] for m in msgs])
        return [cs[0].shard_finalize_dev(m["iv"], gathered.ptr + 16 * i, W, 0, m["total"], stride_bytes=16 * M)
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
    if W == 1 and msgs[0]["fixture"]:
        fx = load_fixture(msgs[0]["fixture"])
        if fx is not None:
            head = bytes(d_ct.download(64, 0))
            tail = bytes(d_ct.download(64, per_gpu - 64))
            ct_ok = (head.hex() == fx["ct_head"] and tail.hex() == fx["ct_tail"])
    selfcheck = None
    if args.selfcheck and W > 1 and emu is None and rank == 0:
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
    for c in ctxs:
        c.timing_enable(False)
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
    # stream (ONE context, so the launches do not overlap), the per-workgroup trace (shader clock), then the same
    # instruction stream without HBM traffic
    ctx.timing_enable(True)
    ctx.timing_read(reset=True)
    for _ in range(min(3, max(1, args.steps))):
        step([ctx])
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
        tag_name = ("%s_n1" % args.config) if W == 1 else "cfg4_n%d" % W
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
                    "timing": "HIP events on the launch stream in a separate %d-step pass after the timed region (timing mode is off inside it; one context, launches back to back)" % min(3, max(1, args.steps)),
                    "sclk_mhz": sclk,
                    "lds_busy_frac": (pm.get("lds") or {}).get("lds_busy_frac") if same_build else None,
                    "formulation_ceiling": ceiling,
                    "measured_copy_kernel": {"value": copy_gbps, "unit": "GB/s read+write", "frac_of_copy": (round(achieved / 1e9 / copy_gbps, 4) if copy_gbps else None)}}
        line = {
            "metric": "GiB/s plaintext, AES-%d-GCM %.3g GiB stream, bit-exact tag" % (key_bits, cfg["gib"]),
            "value": round(value, 3), "unit": "GiB/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload, "bytes_per_gpu": per_gpu, "messages_per_step": M, "contexts_per_rank": n_ctx,
                       "parallelism": "single" if W == 1 else "shard%d" % W, "key_bits": key_bits,
                       "workgroups": geo["workgroups"], "wg_lanes": geo["wg_lanes"], "lds_bytes_per_wg": geo["lds_bytes"],
                       "exchange": None if ex is None else {"backend": ex.name, "ranks_seen": ex.world, "torch": "not imported"}},
            "tag_ok": tag_ok, "ct_head_tail_ok": ct_ok, "selfcheck": selfcheck, "tags": [t.hex() for t in tags],
            "roofline": roofline,
        }
        if emu is not None:
            line["emulated"] = {"rank": R, "of": W, "what": "ONE GPU runs rank %d's step of the %d-GPU cfg4 job: its shard of every message at the real "
                                "first_block, a device copy in place of the RCCL all-gather, every finalize; `value` is this one rank's GiB/s "
                                "(the %d-GPU aggregate would be %d x the slowest rank, less the all-gather latency)" % (R, W, W, W)}
        if cpu_base is not None:
            line["cpu_baseline"] = cpu_base
        emit(line)

    if emu is not None:
        ex = None
    return finish(tag_ok)


if __name__ == "__main__":
    sys.exit(main())
