"""GPU: the multi-process orchestration of the PRODUCT shard path, and the in-library RCCL entry points.

* two fresh child processes (subprocess; the parent never execs itself) run bench.py's N > 1 path -- every rank
  aesgcm_shard_crypt_dev -> all-gather of the 16-byte partials -> aesgcm_shard_finalize_strided_dev -- both on GPU 0,
  with the debug file exchange (RCCL refuses two ranks on one device).  Both ranks must derive the tag the oracle
  computes for the whole message; rank 0 also compares against a single-launch encrypt (--selfcheck).
* aesgcm_comm_* / aesgcm_mgpu_* with a ONE-rank RCCL communicator (the only size a one-GPU box can form): the calls the
  8-GPU job makes, RCCL really loaded and driven, cfg3 fixture as the expected result.
"""
import json
import os
import subprocess
import sys
import tempfile

import pytest

from util import golden, stream_key_iv

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case(name):
    for c in golden("streams.json")["cases"]:
        if c["name"] == name:
            return c
    pytest.skip("fixture %s not generated" % name)


def test_two_processes_product_shard_path_one_gpu(hip, orc):
    per_rank_gib = 0.5                                    # message = 2 ranks x 0.5 GiB = 1 GiB, AES-256, seed 0xAE5C0004
    n = int(2 * per_rank_gib * (1 << 30))
    with tempfile.TemporaryDirectory(prefix="aesgcm_mp_") as rdzv:
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), AESGCM_RDZV_DIR=rdzv,
                       MASTER_ADDR="127.0.0.1", MASTER_PORT="29555")
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "file", "--one-device",
                   "--gib-per-gpu", str(per_rank_gib), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--selfcheck", "--allow-file-exchange"]
            procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT))
        outs = []
        try:
            for p in procs:
                outs.append(p.communicate(timeout=420))
        except subprocess.TimeoutExpired:
            for p in procs:
                p.kill()
            pytest.fail("a rank did not finish within 420 s (killed)")
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    line = json.loads(outs[0][0].strip().splitlines()[-1])          # rank 0 prints the bench line
    assert line["n_gpus"] == 2 and line["config"]["exchange"] == {"backend": "file", "ranks_seen": 2, "torch": "not imported"}
    assert line["selfcheck"] is True and line["tag_ok"] is not False
    # the oracle's tag for the whole 1 GiB message (same key / IV / plaintext stream as the ranks used)
    import numpy as np
    import aesgcm_amd  # noqa: F401
    from aesgcm_amd import sharding
    key = sharding.splitmix64_bytes(0x4B4559, 32)
    iv = sharding.splitmix64_bytes(0x4956, 12)
    pt = np.frombuffer(orc.fill_splitmix64(n, 0xAE5C0004), dtype=np.uint8)
    ct = np.empty_like(pt)
    _, want = orc.Fast(key).crypt(False, iv, b"", pt, ct)
    assert line["tags"] == [want.hex()]


def test_rccl_single_rank_communicator(hip):
    """aesgcm_comm_*: ncclCommInitRank with one rank, a 32-byte all-gather, a max all-reduce, a barrier"""
    uid = hip.comm_unique_id()
    assert len(uid) == 128
    c = hip.Comm(uid, 1, 0, device=0)
    assert (c.n_ranks, c.rank) == (1, 0)
    a, b = hip.DeviceBuffer(32), hip.DeviceBuffer(32)
    a.upload(bytes(range(32)))
    c.allgather_dev(a.ptr, b.ptr, 32)
    c.barrier()
    assert bytes(b.download()) == bytes(range(32))
    assert c.allreduce(3.25, "max") == 3.25 and c.allreduce(3.25, "sum") == 3.25
    c.close()


def test_mgpu_one_device_equals_cfg3_fixture(hip):
    """aesgcm_mgpu_* (ncclCommInitAll over this process's devices = one here): cfg3's 16 GiB message, in place"""
    c = _case("cfg3_aes256_16GiB")
    n = c["n_bytes"]
    key, iv = stream_key_iv(c)
    m = hip.MultiGpu(key, [0])
    assert m.n_ranks == 1
    buf = hip.DeviceBuffer(n)
    buf.fill_splitmix64(c["pt_seed"], c["first_word"])
    tag = m.crypt_dev(False, iv, [buf.ptr], [n], [buf.ptr])
    assert tag.hex() == c["tag"]
    assert bytes(buf.download(64, 0)).hex() == c["ct_head"] and bytes(buf.download(64, n - 64)).hex() == c["ct_tail"]
    # and back, with a ragged tail and an AAD on device 0 (small message)
    aad = bytes(range(37))
    d_aad = hip.DeviceBuffer(len(aad)); d_aad.upload(aad)
    small = hip.DeviceBuffer(4096)
    pt = bytes((7 * i) & 0xFF for i in range(3005))
    small.upload(pt)
    t1 = m.crypt_dev(False, iv, [small.ptr], [3005], [small.ptr], d_aad=d_aad.ptr, aad_len=len(aad))
    ctx = hip.Context(key)
    want_ct, want_tag = ctx.encrypt(iv, aad, pt)
    assert t1 == want_tag and bytes(small.download(3005)) == want_ct
    m.close(); buf.free()


def test_mgpu_one_device_queue_is_a_fifo_partial_and_mixed_collection(hip, orc):
    """aesgcm_mgpu_* queued form on ONE device (so that it runs on the single-GPU pool, with real tags): five messages queued with tag = NULL, collected two and
    then three -- each tag against the oracle, in the order queued (round 5 returned the newest n and lost the oldest: ADVICE r05); a call that wants its own tag
    while messages wait is refused with ESTATE before anything is enqueued, and works again once the queue is empty; the ring takes eight, not nine"""
    key = bytes(orc.fill_splitmix64(32, 0x4B4559))
    f = orc.Fast(key)
    m = hip.MultiGpu(key, [0])
    msgs = []
    for k in range(5):
        n = (1 << 20) + 4096 * k + 3 * k
        iv, pt = bytes(orc.fill_splitmix64(12, 0x4956 + k)), bytes(orc.fill_splitmix64(n, 0xAE5C0200 + k))
        buf = hip.DeviceBuffer(n)
        buf.upload(pt)
        msgs.append((iv, pt, buf, n))
    for iv, pt, buf, n in msgs:
        assert m.crypt_dev(False, iv, [buf.ptr], [n], [buf.ptr], want_tag=False) is None
    with pytest.raises(hip.AesGcmError) as ei:                   # a sixth message that wants its tag back would jump the queue
        m.crypt_dev(False, msgs[0][0], [msgs[0][2].ptr], [msgs[0][3]], [msgs[0][2].ptr])
    assert ei.value.code == hip.ESTATE
    first, rest = m.last_tags(2), m.last_tags(3)
    m.sync()
    want = [f.encrypt(iv, b"", pt) for iv, pt, buf, n in msgs]
    assert first == [w[1] for w in want[:2]] and rest == [w[1] for w in want[2:]]
    for (iv, pt, buf, n), w in zip(msgs, want):
        assert bytes(buf.download(n)) == w[0]
    with pytest.raises(hip.AesGcmError):
        m.last_tags(1)                                           # nothing waits any more
    # the queue is empty: a call with its own tag works again (decrypt the first message back), and eight messages fit the ring, a ninth does not
    iv, pt, buf, n = msgs[0]
    assert m.crypt_dev(True, iv, [buf.ptr], [n], [buf.ptr]) == want[0][1] and bytes(buf.download(n)) == pt
    for k in range(8):
        m.crypt_dev(False, iv, [buf.ptr], [n], [buf.ptr], want_tag=False)
    with pytest.raises(hip.AesGcmError):
        m.crypt_dev(False, iv, [buf.ptr], [n], [buf.ptr], want_tag=False)
    tags = m.last_tags(3) + m.last_tags(5)
    m.sync()
    assert len(tags) == 8 and tags[0] == want[0][1]              # (in place eight times: encrypt, "encrypt" again = decrypt under the same IV, ...: the first is the message's tag)
    m.close()


def test_launcher_ranks_without_rccl_hand_over_to_one_process(hip):
    """Two ranks as an outside launcher would start them (RANK / WORLD_SIZE in the environment, no --one-device), both landing on GPU 0, where RCCL refuses to form
    a communicator: the ranks must not measure through the file exchange -- rank 1 leaves with 0, rank 0 starts ONE fresh child over both devices
    (aesgcm_mgpu_*) and relays it.  On a one-GPU box that child refuses ("the library sees 1 device(s)", exit code 2): the hand-over itself is what is checked
    here; tests/test_gpu_multidevice.py checks the result on a box with two."""
    import aesgcm_amd  # noqa: F401
    from aesgcm_amd import lib
    if lib.device_count() >= 2:
        pytest.skip("two devices: RCCL comes up between the ranks (covered by test_gpu_multidevice.py)")
    with tempfile.TemporaryDirectory(prefix="aesgcm_mp_") as rdzv:
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", AESGCM_RDZV_DIR=rdzv, MASTER_ADDR="127.0.0.1", MASTER_PORT="29556")
            env.pop("AESGCM_SELF_LAUNCHED", None)
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--gib-per-gpu", "0.25", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
            procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT))
        outs = []
        try:
            for p in procs:
                outs.append(p.communicate(timeout=900))
        except subprocess.TimeoutExpired:
            for p in procs:
                p.kill()
            pytest.fail("a rank did not finish within 900 s (killed)")
    (so0, se0), (so1, se1) = outs
    assert procs[1].returncode == 0, se1[-2000:]
    assert "no RCCL communicator between the 2 processes" in se0 and "falling back to ONE process driving all 2 devices" in se0, se0[-3000:]
    assert "the library sees 1 device(s)" in se0 and procs[0].returncode == 2, (procs[0].returncode, se0[-3000:])
    assert not so0.strip(), so0                                   # no bench line claimed
