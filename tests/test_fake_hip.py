"""CPU: device discipline of the library's host side, on four FAKE devices (round 5; round-4 verdict, Missing 2 / Next 3: no box the builder gets has two GPUs, so
no device index other than 0 had ever run).  tests/fake_hip/ links csrc/aesgcm_host.hip, aesgcm_abi.hip and aesgcm_comm.hip -- which hold no device code since the
library was split -- against a fake HIP runtime, fake kernel launchers and a fake RCCL that record which device is current at every allocation, stream, event,
attribute call, copy, launch and collective; tests/fake_hip/drive.py then runs every family of entry points on every device through the ordinary ctypes binding and
asserts that a context of device k touches device k only, that every pointer a launch carries lives on the launch's device, that the LDS attributes are set on a
device before the first large-LDS launch there, and that the queued single-process multi-GPU path makes no host synchronisation per message."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_host_side_keeps_to_its_device_on_four_fake_devices():
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc (the HIP headers)")
    d = os.path.join(HERE, "fake_hip")
    subprocess.run(["make", "-C", d, "-s"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    env = dict(os.environ, AESGCM_LIB=os.path.join(d, "libaesgcm_fake.so"))
    out = subprocess.run([sys.executable, os.path.join(d, "drive.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert out.returncode == 0 and "FAKE HIP OK" in out.stdout, out.stdout[-3000:]


def test_bench_single_process_queues_its_messages_without_a_host_sync_per_message():
    """bench.py --gpus 4 --single-process (the fallback of the N-rank launch on a box where RCCL comes up inside one process only) over the fake runtime: the line is
    printed, no call leaves its device, and the number of host synchronisations does not grow with the number of timed steps -- the messages of a step are
    queued (aesgcm_mgpu_crypt_dev with tag = NULL) and their tags collected by one finalize launch (aesgcm_mgpu_last_tags)."""
    import json
    import re
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc (the HIP headers)")
    d = os.path.join(HERE, "fake_hip")
    subprocess.run(["make", "-C", d, "-s"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    env = dict(os.environ, AESGCM_LIB=os.path.join(d, "libaesgcm_fake.so"), FAKEHIP_REPORT="1")
    seen = {}
    for steps in (4, 9):
        out = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "4", "--sp-child", "--gib-per-gpu", "0.0625", "--steps", str(steps), "--warmup", "1",
                              "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
        line = json.loads(out.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == 4 and line["config"]["exchange"]["backend"] == "rccl (single process)" and line["validated"] is True
        m = re.search(r"fakehip: syncs=(\d+) launches=(\d+) collectives=(\d+) violations=(\d+) touched=(\d+)", out.stderr)
        assert m, out.stderr[-2000:]
        seen[steps] = tuple(int(x) for x in m.groups())
        assert seen[steps][3] == 0 and seen[steps][4] == 0b1111, seen
    assert seen[4][0] == seen[9][0], "host synchronisations grow with the steps: %r" % (seen,)
    assert seen[9][2] > seen[4][2]                                      # ... while the collectives do
