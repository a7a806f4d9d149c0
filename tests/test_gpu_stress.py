"""GPU: hang tripwire for the packet kernels (two early builds of k_pkt hung the device and the
cause was never isolated).  ~10^4 randomised launches of aesgcm_packets_crypt_dev (both shapes, deal in {1,3,16}) and
aesgcm_batch_crypt_var_dev run in a FRESH CHILD PROCESS under a hard timeout; a sample of the launches is verified
against the oracle inside the child.  On timeout the child is killed and the test fails -- nothing is re-exec'd and the
parent's GPU context is never at stake."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_packet_kernels_do_not_hang_over_many_random_launches():
    n_launches = int(os.environ.get("AESGCM_STRESS_LAUNCHES", "10000"))
    cmd = [sys.executable, os.path.join(ROOT, "tests", "stress_child.py"), str(n_launches), "20241002"]
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT)
    try:
        so, se = p.communicate(timeout=900)
    except subprocess.TimeoutExpired:
        p.kill()
        so, se = p.communicate()
        pytest.fail("packet-kernel stress child did not finish within 900 s (killed); last output: %s" % (so[-500:] + se[-500:]))
    assert p.returncode == 0, (so[-1500:], se[-1500:])
    res = json.loads(so.strip().splitlines()[-1])
    assert res["launches"] >= n_launches and res["mismatches"] == 0 and res["verified"] >= 200, res
