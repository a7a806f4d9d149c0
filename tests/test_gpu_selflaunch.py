"""GPU: `python bench.py --gpus N` WITHOUT a launcher starts its own ranks, and a non-RCCL exchange fails loudly.

The driver's scaling command has the form `python bench.py --gpus N ...`; round 2's bench silently measured ONE GPU in
that case.  Now the parent (which never touches the GPU) spawns N fresh rank processes, relays rank 0's line and exits
with the worst rank's code.  On this one-GPU box both ranks sit on GPU 0 (--one-device): RCCL refuses the duplicate
device, every rank falls back to the debug file exchange, and the run is accepted only with --allow-file-exchange --
without it the line is still printed but the exit code is non-zero (a file-exchange number is not an RCCL number).
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "AESGCM_RDZV_DIR")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--one-device", "--gib-per-gpu", "0.25", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--launch-timeout", str(timeout - 60)] + extra
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


def test_self_launch_two_ranks_one_device_file_exchange_allowed(hip):
    rc, line, err = _run(["--allow-file-exchange", "--selfcheck"])
    assert rc == 0, err[-3000:]
    assert line is not None and line["n_gpus"] == 2
    ex = line["config"]["exchange"]
    assert ex["ranks_seen"] == 2 and ex["backend"].startswith("file")          # RCCL refused two ranks on one device
    assert line["selfcheck"] is True and line["tag_ok"] is not False and len(line["tags"]) == 1


def test_self_launch_without_the_flag_exits_non_zero(hip):
    rc, line, err = _run(["--backend", "file"])
    assert rc != 0, "a run whose exchange is not RCCL must fail without --allow-file-exchange"
    assert line is not None and line["config"]["exchange"]["backend"].startswith("file")       # the line is still printed
    assert "not RCCL" in err


def test_gpus_mismatch_with_world_size_is_refused(hip):
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--no-cpu-baseline"], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode != 0 and "{" not in p.stdout


def test_cfg5_replicas_produce_the_same_tags_as_one_gpu(hip):
    """bench.py --config cfg5: N ranks are replicas of 2^k / N packets with no collective on the data path; the SHA-256 over all
    tags (rank 0 concatenates every rank's) must equal the one-GPU run's over the same packets, and the first 64 tags the fixture's"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "AESGCM_RDZV_DIR")}
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "cfg5", "--n-pkts", "65536", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    one = subprocess.run(base, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run(base + ["--gpus", "2", "--one-device", "--backend", "file", "--allow-file-exchange"], env=env, cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-2000:]
    l1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    l2 = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])
    assert l1["n_gpus"] == 1 and l2["n_gpus"] == 2 and l2["config"]["packets_per_gpu"] == 32768
    assert l1["tags_sha256"] == l2["tags_sha256"]
    assert l1["roofline"]["kernel"].startswith("k_batch3") and l1["roofline"]["frac"] > 0
    import hashlib
    from util import golden, batch_inputs
    # the same 65536 packets start with the fixture's first 64
    fx = golden("batch.json")
    keys, ivs, pt = batch_inputs(0, 64, 4096)
    d_keys, d_ivs, d_in = hip.DeviceBuffer(len(keys)), hip.DeviceBuffer(len(ivs)), hip.DeviceBuffer(len(pt))
    d_keys.upload(keys); d_ivs.upload(ivs); d_in.upload(pt)
    d_out, d_tags = hip.DeviceBuffer(len(pt)), hip.DeviceBuffer(16 * 64)
    hip.batch_crypt_dev(False, 64, 16, d_keys.ptr, d_ivs.ptr, d_in.ptr, 4096, d_out.ptr, d_tags.ptr)
    hip.dev_sync()
    tags = bytes(d_tags.download())
    assert [tags[16 * p:16 * p + 16].hex() for p in range(64)] == fx["first64_tags"]
