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
