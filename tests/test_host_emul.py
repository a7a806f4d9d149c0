"""CPU: the per-lane code of the HIP kernels (csrc/aesgcm_dev.h, __host__ __device__) executed over
emulated launches and compared with the oracle -- geometry, front padding, Horner with K, tail powers,
workgroup fold, k_combine, shards, streaming carry, keystream, ECB.  See tests/host_emul/emul.cpp."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_lane_code_emulated_launches_match_oracle():
    d = os.path.join(HERE, "host_emul")
    subprocess.run(["make", "-C", d, "-s"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(d, "emul"), "1"], stdout=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0 and "EMUL OK" in out.stdout, out.stdout[-2000:]
