"""CPU: the per-lane code of the HIP kernels (csrc/aesgcm_dev.h, __host__ __device__) executed over
emulated launches and compared with the oracle -- geometry, front padding, Horner with K, chunk items, k_fold
levels, k_combine, the k_body cut, shards, streaming carry, keystream, ECB, packet lanes.
See tests/host_emul/emul.cpp."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_lane_code_emulated_launches_match_oracle():
    d = os.path.join(HERE, "host_emul")
    subprocess.run(["make", "-C", d, "-s"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(d, "emul"), "2"], stdout=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0 and "EMUL OK" in out.stdout, out.stdout[-2000:]


def test_lane_code_under_address_and_ub_sanitizers():
    """The same harness built with the address and undefined-behaviour sanitizers (host side only; GPU sanitizers are
    not available): out-of-bounds reads or writes of the lane code, the planners or the fold levels fail here."""
    import pytest
    d = os.path.join(HERE, "host_emul")
    if not os.path.exists(os.path.join(d, "asan.mk")):
        pytest.skip("sanitizer recipe not shipped to this machine")
    subprocess.run(["make", "-C", d, "-s", "asan"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(d, "emul_asan"), "1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert out.returncode == 0 and "EMUL OK" in out.stdout and "ERROR" not in out.stdout and "runtime error" not in out.stdout, out.stdout[-3000:]
