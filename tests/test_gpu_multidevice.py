"""GPU, two or more devices: everything here SKIPS on a one-GPU box and runs by itself on any box with two.

Up to round 3 no device index other than 0 had ever executed and RCCL had never met a second rank (the builder's boxes have one GPU; the
driver's 8-GPU run is the first contact).  These tests make any multi-GPU box validate that contact, each against the oracle:

* a plain Context on device 1 (known-answer vector, a ragged message with AAD, the packet and batch entry points);
* aesgcm_mgpu_* over devices [0, 1] (ncclCommInitAll, the grouped 16-byte all-gather) against a single-launch encrypt and the oracle;
* `bench.py --gpus 2` self-launched: two processes, one device each, ncclCommInitRank over a file-passed unique id -- the line must
  say `rccl`, `ranks_seen` 2, and carry the oracle's tag;
* `bench.py --gpus 2 --single-process`: the fallback the self-launch takes when no communicator comes up between processes.

Reference counterpart of the split: src/gcm_ghash.vhd:317-333 (one multiplication as two halves XORed); the per-message limit that forces
several messages per job: src/aes_icb.vhd:114.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from util import golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def two(hip):
    n = hip.device_count()
    if n < 2:
        pytest.skip("needs two GPUs (this box has %d)" % n)
    return hip


def test_context_on_device_1_known_answers(two, orc):
    hip = two
    for v in golden("kat.json")["vectors"]:
        key, iv, aad, pt = (bytes.fromhex(v[k]) for k in ("key", "iv", "aad", "pt"))
        with hip.Context(key, device=1) as c:
            assert c.device == 1
            ct, tag = c.encrypt(iv, aad, pt)
            assert ct.hex() == v["ct"] and tag.hex() == v["tag"], v.get("name")
            back, _ = c.decrypt(iv, aad, ct, tag=tag)
            assert back == pt
    # a ragged multi-MiB message with AAD, device-resident on device 1, every key size (cyclic rows + in-launch closing on a device that is not 0)
    for kb in (16, 24, 32):
        key, iv, aad = bytes(orc.fill_splitmix64(kb, 0x4B4559)), bytes(orc.fill_splitmix64(12, 0x4956)), bytes(orc.fill_splitmix64(20, 0x414144))
        n = (5 << 20) + 5
        pt = np.frombuffer(orc.fill_splitmix64(n, 0xD1 + kb), dtype=np.uint8)
        want_ct = np.empty_like(pt)
        _, want_tag = orc.Fast(key).crypt(False, iv, aad, pt, want_ct)
        with hip.Context(key, device=1) as c:
            d_in, d_out, d_aad = hip.DeviceBuffer(n, device=1), hip.DeviceBuffer(n, device=1), hip.DeviceBuffer(len(aad), device=1)
            d_in.upload(pt); d_aad.upload(aad)
            tag = c.encrypt_dev(iv, d_in.ptr, n, d_out.ptr, d_aad=d_aad.ptr, aad_len=len(aad))
            assert tag == want_tag
            assert bytes(d_out.download()) == want_ct.tobytes()
            for b in (d_in, d_out, d_aad):
                b.free()


def test_packet_and_batch_entry_points_on_device_1(two, orc):
    hip = two
    from util import batch_inputs
    fx = golden("batch.json")
    keys, ivs, pt = batch_inputs(0, 64, 4096)
    d_keys, d_ivs, d_in = hip.DeviceBuffer(len(keys), device=1), hip.DeviceBuffer(len(ivs), device=1), hip.DeviceBuffer(len(pt), device=1)
    d_keys.upload(keys); d_ivs.upload(ivs); d_in.upload(pt)
    d_out, d_tags = hip.DeviceBuffer(len(pt), device=1), hip.DeviceBuffer(16 * 64, device=1)
    hip.batch_crypt_dev(False, 64, 16, d_keys.ptr, d_ivs.ptr, d_in.ptr, 4096, d_out.ptr, d_tags.ptr, device=1)
    hip.dev_sync(1)
    tags = bytes(d_tags.download())
    assert [tags[16 * p:16 * p + 16].hex() for p in range(64)] == fx["first64_tags"]
    # packets under one key on device 1: 300 frames of 0 .. 1514 bytes, packed back to back, against the oracle
    key = bytes(orc.fill_splitmix64(32, 77))
    rng = np.random.default_rng(5)
    lens = [int(x) for x in rng.integers(0, 1515, size=300)]
    packed_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    total = int(packed_off[-1])
    packed = bytes(orc.fill_splitmix64(total, 78))
    pivs = bytes(orc.fill_splitmix64(12 * 300, 79))
    with hip.Context(key, device=1) as c:
        d_data, d_o = hip.DeviceBuffer(total, device=1), hip.DeviceBuffer(total, device=1)
        d_piv, d_t, d_off = hip.DeviceBuffer(len(pivs), device=1), hip.DeviceBuffer(16 * 300, device=1), hip.DeviceBuffer(8 * 301, device=1)
        d_data.upload(packed); d_off.upload(packed_off.tobytes()); d_piv.upload(pivs)
        c.packets_crypt_dev(False, 300, d_piv.ptr, d_data.ptr, d_o.ptr, d_t.ptr, d_data_off=d_off.ptr)
        hip.dev_sync(1)
        got_ct, got_tags = bytes(d_o.download()), bytes(d_t.download())
    f = orc.Fast(key)
    for p in range(300):
        a, b = int(packed_off[p]), int(packed_off[p + 1])
        want_ct, want_tag = f.encrypt(pivs[12 * p:12 * p + 12], b"", packed[a:b])
        assert got_ct[a:b] == want_ct and got_tags[16 * p:16 * p + 16] == want_tag, p


def test_mgpu_two_devices_equals_single_launch_and_oracle(two, orc):
    hip = two
    key, iv, aad = bytes(orc.fill_splitmix64(32, 0x4B4559)), bytes(orc.fill_splitmix64(12, 0x4956)), bytes(range(37))
    for n in ((24 << 20) + 16 * 3 + 5, 4096 + 7, 2 * 16):
        n0 = (n // 2) // 16 * 16                               # shards are cut at block boundaries; only the last may be ragged
        pt = np.frombuffer(orc.fill_splitmix64(n, 0xAE5C0004), dtype=np.uint8)
        want_ct = np.empty_like(pt)
        _, want_tag = orc.Fast(key).crypt(False, iv, aad, pt, want_ct)
        m = hip.MultiGpu(key, [0, 1])
        assert m.n_ranks == 2                                   # what ncclCommCount reports
        a, b = hip.DeviceBuffer(max(n0, 16), device=0), hip.DeviceBuffer(max(n - n0, 16), device=1)
        a.upload(pt[:n0].tobytes()); b.upload(pt[n0:].tobytes())
        d_aad = hip.DeviceBuffer(len(aad), device=0); d_aad.upload(aad)
        tag = m.crypt_dev(False, iv, [a.ptr, b.ptr], [n0, n - n0], [a.ptr, b.ptr], d_aad=d_aad.ptr, aad_len=len(aad))
        assert tag == want_tag, n
        assert bytes(a.download(n0)) + bytes(b.download(n - n0)) == want_ct.tobytes()
        with hip.Context(key, device=0) as c:                   # and the single launch on one device
            ct1, tag1 = c.encrypt(iv, aad, pt.tobytes())
            assert tag1 == tag and ct1 == want_ct.tobytes()
        # decrypt in place over the two devices
        back = m.crypt_dev(True, iv, [a.ptr, b.ptr], [n0, n - n0], [a.ptr, b.ptr], d_aad=d_aad.ptr, aad_len=len(aad))
        assert back == want_tag and bytes(a.download(n0)) + bytes(b.download(n - n0)) == pt.tobytes()
        m.close()
        for x in (a, b, d_aad):
            x.free()


def test_mgpu_queued_messages_and_their_tags(two, orc):
    """the queued form (round 5): four messages enqueued with tag = NULL -- no host synchronisation -- and their tags collected by one finalize launch, against the
    oracle; the ring refuses a ninth message; rows on device 1 (many messages under one key through k_rows) ride along"""
    hip = two
    key = bytes(orc.fill_splitmix64(32, 0x4B4559))
    m = hip.MultiGpu(key, [0, 1])
    n, n0 = (3 << 20) + 21, (1 << 20) + 16 * 7
    msgs = []
    for k in range(4):
        iv = bytes(orc.fill_splitmix64(12, 0x4956 + k))
        pt = bytes(orc.fill_splitmix64(n, 0xAE5C0100 + k))
        a, b = hip.DeviceBuffer(n0, device=0), hip.DeviceBuffer(n - n0, device=1)
        a.upload(pt[:n0]); b.upload(pt[n0:])
        msgs.append((iv, pt, a, b))
    for iv, pt, a, b in msgs:
        assert m.crypt_dev(False, iv, [a.ptr, b.ptr], [n0, n - n0], [a.ptr, b.ptr], want_tag=False) is None
    tags = m.last_tags(4)
    m.sync()
    f = orc.Fast(key)
    for (iv, pt, a, b), t in zip(msgs, tags):
        want_ct, want_tag = f.encrypt(iv, b"", pt)
        assert t == want_tag and bytes(a.download()) + bytes(b.download()) == want_ct
    for k in range(8):
        m.crypt_dev(False, msgs[0][0], [msgs[0][2].ptr, msgs[0][3].ptr], [n0, n - n0], [msgs[0][2].ptr, msgs[0][3].ptr], want_tag=False)
    with pytest.raises(hip.AesGcmError):
        m.crypt_dev(False, msgs[0][0], [msgs[0][2].ptr, msgs[0][3].ptr], [n0, n - n0], [msgs[0][2].ptr, msgs[0][3].ptr], want_tag=False)
    assert len(m.last_tags(8)) == 8
    m.close()
    # many messages under one key by rows, on device 1
    cnt, size = 40, 65536 + 1024 + 5
    ivs, pt = bytes(orc.fill_splitmix64(12 * cnt, 91)), bytes(orc.fill_splitmix64(cnt * size, 92))
    with hip.Context(key, device=1) as c:
        d_ivs, d_buf, d_tags = hip.DeviceBuffer(len(ivs), device=1), hip.DeviceBuffer(len(pt), device=1), hip.DeviceBuffer(16 * cnt, device=1)
        d_ivs.upload(ivs); d_buf.upload(pt)
        assert c.packets_shape(cnt, size) == hip.SHAPE_ROWS
        c.packets_crypt_dev(False, cnt, d_ivs.ptr, d_buf.ptr, d_buf.ptr, d_tags.ptr, pkt_len=size)
        hip.dev_sync(1)
        ct, tags = bytes(d_buf.download()), bytes(d_tags.download())
    for p_ in range(cnt):
        assert (ct[p_ * size:(p_ + 1) * size], tags[16 * p_:16 * p_ + 16]) == f.encrypt(ivs[12 * p_:12 * p_ + 12], b"", pt[p_ * size:(p_ + 1) * size]), p_


def test_routed_calls_and_scattered_messages_on_device_1(two, orc):
    """device 1 (skips on a single-GPU box, runs by itself on any box with two): one offset-array call whose messages are of both kinds -- the long ones by rows on the
    context's side stream, the short ones through the packet kernels --, the same messages in buffers of their own (aesgcm_messages_crypt_dev), the route the device
    chose, a refused length, and the state of a stream moved from a context of device 0 to one of device 1"""
    import random
    import struct
    hip = two
    rng = random.Random(4141)
    key = bytes(orc.fill_splitmix64(32, 0x4B4559))
    f = orc.Fast(key)
    n = 3000
    lens = [int(rng.betavariate(0.1, 0.1) * 40000) for _ in range(n)]
    doff = [0]
    for x in lens:
        doff.append(doff[-1] + x)
    ivs, pt = bytes(orc.fill_splitmix64(12 * n, 93)), bytes(orc.fill_splitmix64(doff[-1], 94))

    def up(b):
        d = hip.DeviceBuffer(max(len(b), 16), device=1); d.upload(b); return d
    with hip.Context(key, device=1) as c:
        c.set_option("route_blocks_min", 0)
        d_ivs, d_in, d_out, d_tags = up(ivs), up(pt), hip.DeviceBuffer(doff[-1] + 16, device=1), hip.DeviceBuffer(16 * n, device=1)
        d_off = up(struct.pack("<%dQ" % (n + 1), *doff))
        c.packets_crypt_dev(False, n, d_ivs.ptr, d_in.ptr, d_out.ptr, d_tags.ptr, d_data_off=d_off.ptr)
        hip.dev_sync(1)
        r = c.last_route()
        assert c.status() == (hip.STATUS_OK, 0) and r["route_min"] == 2048 and 0 < r["n_small"] < n and r["row_units"] > 0, r
        ct, tags = bytes(d_out.download(doff[-1])), bytes(d_tags.download())
        for p_ in range(0, n, 3):
            assert (ct[doff[p_]:doff[p_ + 1]], tags[16 * p_:16 * p_ + 16]) == f.encrypt(ivs[12 * p_:12 * p_ + 12], b"", pt[doff[p_]:doff[p_ + 1]]), p_
        d_ptr_in, d_ptr_out = up(struct.pack("<%dQ" % n, *[d_in.ptr + x for x in doff[:-1]])), up(struct.pack("<%dQ" % n, *[d_out.ptr + x for x in doff[:-1]]))
        d_len = up(struct.pack("<%dI" % n, *lens))
        d_out.upload(bytes(doff[-1]))
        c.messages_crypt_dev(False, n, d_ivs.ptr, d_ptr_in.ptr, d_len.ptr, d_ptr_out.ptr, d_tags.ptr)
        hip.dev_sync(1)
        assert bytes(d_out.download(doff[-1])) == ct and bytes(d_tags.download()) == tags
        bad = list(lens); bad[7] = 1 << 28
        c.messages_crypt_dev(False, n, d_ivs.ptr, d_ptr_in.ptr, up(struct.pack("<%dI" % n, *bad)).ptr, d_ptr_out.ptr, d_tags.ptr)
        hip.dev_sync(1)
        assert c.status() == (hip.STATUS_LENGTH, 7)
        with hip.Context(key, device=0) as c0:
            c0.stream_begin(ivs[:12]); c0.stream_update(pt[:4096])
            c.stream_import(c0.stream_export())
            c0.stream_final()
        rest = c.stream_update(pt[4096:70001])
        want = f.encrypt(ivs[:12], b"", pt[:70001])
        assert c.stream_final() == want[1] and rest == want[0][4096:]


def _bench(extra, timeout=1500):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "AESGCM_RDZV_DIR")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--gib-per-gpu", "0.5", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--launch-timeout", str(timeout - 60)] + extra
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


def _oracle_tag_of_the_job(orc, n):
    import aesgcm_amd  # noqa: F401
    from aesgcm_amd import sharding
    key, iv = sharding.splitmix64_bytes(0x4B4559, 32), sharding.splitmix64_bytes(0x4956, 12)
    pt = np.frombuffer(orc.fill_splitmix64(n, 0xAE5C0004), dtype=np.uint8)
    ct = np.empty_like(pt)
    _, want = orc.Fast(key).crypt(False, iv, b"", pt, ct)
    return want.hex()


def test_two_rank_bench_over_rccl(two, orc):
    rc, line, err = _bench(["--selfcheck"])
    assert rc == 0, err[-4000:]
    assert line is not None and line["n_gpus"] == 2
    ex = line["config"]["exchange"]
    assert ex["backend"] in ("rccl", "rccl (single process)"), ex       # never the file exchange on a box that has two devices
    assert ex["ranks_seen"] == 2
    assert line["tags"] == [_oracle_tag_of_the_job(orc, 1 << 30)]
    if ex["backend"] == "rccl":
        assert line["selfcheck"] is True
    else:                                                                 # the fallback ran: say why in the test log
        print("per-process RCCL did not come up; stderr tail:\n" + err[-3000:])


def test_single_process_fallback_path(two, orc):
    rc, line, err = _bench(["--single-process"])
    assert rc == 0, err[-4000:]
    ex = line["config"]["exchange"]
    assert ex["backend"] == "rccl (single process)" and ex["ranks_seen"] == 2 and ex["init"] == "ncclCommInitAll"
    assert line["n_gpus"] == 2 and line["tags"] == [_oracle_tag_of_the_job(orc, 1 << 30)]
    assert line["roofline"]["frac"] > 0 and line["value"] > 0
