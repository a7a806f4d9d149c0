"""GPU: parity of the HIP path (through the C ABI) against the oracle and the golden fixtures.
Bar: bit-exact ciphertext and tag (integer/byte work)."""
import hashlib

import pytest

from util import golden, matrix_inputs, splitmix_bytes, stream_key_iv

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------- unit level
def test_key_expand_matches_reference_key_exp(hip):
    for name, v in golden("key_schedule.json")["vectors"].items():
        rk, nr = hip.key_expand(bytes.fromhex(v["key"]))
        assert rk.hex() == v["expanded"], name
        assert nr == {"128": 10, "192": 12, "256": 14}[v["size"]]


def test_preexpanded_key_context_equals_on_gpu_expansion(hip, orc):
    for klen in (16, 24, 32):
        key, iv = splitmix_bytes(70 + klen, klen), splitmix_bytes(71, 12)
        rk, _ = orc.key_expand(key)
        a = hip.Context(key)
        b = hip.Context(expanded_key=rk)
        pt = splitmix_bytes(72, 1000)
        assert a.encrypt(iv, b"hdr", pt) == b.encrypt(iv, b"hdr", pt)
        assert a.h() == b.h() == orc.Fast(key).h


def test_ecb_blocks_fips197_and_random(hip, orc):
    pt = bytes.fromhex("00112233445566778899aabbccddeeff")
    for klen, exp in ((16, "69c4e0d86a7b0430d8cdb78070b4c55a"), (24, "dda97ca4864cdfe06eaf70a0ec0d7191"),
                      (32, "8ea2b7ca516745bfeafc49904b496089")):
        c = hip.Context(bytes(range(klen)))
        assert c.ecb_encrypt(pt).hex() == exp
        blocks = splitmix_bytes(5 + klen, 16 * 3001)
        f = orc.Fast(bytes(range(klen)))
        want = b"".join(f.encrypt_block(blocks[i:i + 16]) for i in range(0, len(blocks), 16))
        assert c.ecb_encrypt(blocks) == want


def test_gfmul_vectors(hip):
    g = golden("gfmul.json")["mul"]
    h = b"".join(bytes.fromhex(v["h"]) for v in g)
    x = b"".join(bytes.fromhex(v["x"]) for v in g)
    z = hip.gfmul(h, x)
    assert z == b"".join(bytes.fromhex(v["z"]) for v in g)


def test_ghash_chaining_value(hip, orc):
    key = splitmix_bytes(33, 32)
    c = hip.Context(key)
    h = orc.Fast(key).h
    assert c.h() == h
    f = orc.Fast(key)
    for n in (0, 1, 15, 16, 17, 8191, 8192, 8193, 100000, 5 << 20):
        data = splitmix_bytes(34 + n, n)
        got = c.ghash(data)
        assert got == orc.gfmul(h, f.ghash_poly(data)) if n else got == bytes(16)
        if n <= 100000:
            assert got == orc.ghash_update(h, bytes(16), data)      # literal serial Horner


def test_keystream_blocks(hip, orc):
    key, iv = splitmix_bytes(40, 24), splitmix_bytes(41, 12)
    c, f = hip.Context(key), orc.Fast(key)
    for first, n in ((0, 1), (0, 700), (12345, 513), (0x01FFFF00 - 2, 600), ((1 << 32) - 2 - 5, 5)):
        assert c.keystream(iv, first, n) == f.keystream(iv, first, n)
    with pytest.raises(hip.AesGcmError) as e:
        c.keystream(iv, (1 << 32) - 3, 2)
    assert e.value.code == hip.ETOOLONG


# ---------------------------------------------------------------- whole messages
def test_kat_vectors(hip):
    for v in golden("kat.json")["vectors"]:
        key, iv, aad, pt = (bytes.fromhex(v[k]) for k in ("key", "iv", "aad", "pt"))
        c = hip.Context(key)
        ct, tag = c.encrypt(iv, aad, pt)
        assert (ct.hex(), tag.hex()) == (v["ct"], v["tag"]), v["name"]
        back, t2 = c.decrypt(iv, aad, ct, tag=tag)
        assert back == pt and t2 == tag


def test_length_matrix_encrypt_decrypt(hip):
    ctxs = {}
    for cell in golden("length_matrix.json")["cells"]:
        key, iv, aad, pt = matrix_inputs(cell["kbits"], cell["aad_len"], cell["pt_len"])
        c = ctxs.setdefault(key, hip.Context(key))
        ct, tag = c.encrypt(iv, aad, pt)
        assert tag.hex() == cell["tag"], cell
        assert hashlib.sha256(ct).hexdigest() == cell["ct_sha256"], cell
        back, t2 = c.decrypt(iv, aad, ct, tag=bytes.fromhex(cell["tag"]))
        assert back == pt and t2 == tag


def test_u_shaped_random_lengths_vs_oracle(hip, orc):
    """The reference's own length distribution: int(betavariate(.1,.1) * max) for AAD and data,
    independently (tb/gcm_gctr.py:279-281), all three key sizes, both directions."""
    import random
    rng = random.Random(0xC0C07B)
    for it in range(60):
        klen = rng.choice((16, 24, 32))
        mx = rng.choice(((1 << 12) - 1, (1 << 16) - 1, (1 << 20) - 1))
        al = int(rng.betavariate(.1, .1) * min(mx, (1 << 16) - 1))
        pl = int(rng.betavariate(.1, .1) * mx)
        key, iv = splitmix_bytes(1000 + it, klen), splitmix_bytes(2000 + it, 12)
        aad, pt = splitmix_bytes(3000 + it, al), splitmix_bytes(4000 + it, pl)
        want = orc.Fast(key).encrypt(iv, aad, pt)
        c = hip.Context(key)
        assert c.encrypt(iv, aad, pt) == want, (klen, al, pl)
        assert c.decrypt(iv, aad, want[0], tag=want[1]) == (pt, want[1])


def test_block_boundary_lengths_around_workgroup_and_grid_edges(hip, orc):
    """Lengths that straddle the launch geometry: 512-lane workgroups, the full grid (T = 1 -> 2)."""
    key, iv = splitmix_bytes(50, 32), splitmix_bytes(51, 12)
    c, f = hip.Context(key), orc.Fast(key)
    g = c.geometry()
    full = g["workgroups"] * g["wg_lanes"] * 16
    sizes = [511 * 16, 512 * 16, 513 * 16, 512 * 16 + 1, 1024 * 16 - 1,
             full - 16, full - 1, full, full + 1, full + 16, 2 * full + 5, 3 * full - 7]
    for n in sizes:
        for al in (0, 20, 16 * 600 + 3):
            aad, pt = splitmix_bytes(52 + al, al), splitmix_bytes(53 + n, n)
            assert c.encrypt(iv, aad, pt) == f.encrypt(iv, aad, pt), (n, al)


def test_tamper_detection(hip):
    key, iv, aad, pt = splitmix_bytes(60, 16), splitmix_bytes(61, 12), splitmix_bytes(62, 28), splitmix_bytes(63, 48)
    c = hip.Context(key)
    ct, tag = c.encrypt(iv, aad, pt)
    bad_tag = bytes([tag[0] ^ 1]) + tag[1:]
    with pytest.raises(hip.AuthenticationError):
        c.decrypt(iv, aad, ct, tag=bad_tag)
    assert c.last_plaintext == pt                       # plaintext is still produced (tb/gcm_model.py:30 then :44)
    bad_ct = ct[:7] + bytes([ct[7] ^ 0x80]) + ct[8:]
    with pytest.raises(ValueError):                     # AuthenticationError is a ValueError, as pycryptodome's
        c.decrypt(iv, aad, bad_ct, tag=tag)
    with pytest.raises(hip.AuthenticationError):
        c.decrypt(iv, aad + b"x", ct, tag=tag)
    assert c.decrypt(iv, aad, ct)[1] == tag             # no expected tag: returns the computed one


def test_argument_errors(hip):
    with pytest.raises(hip.AesGcmError) as e:
        hip.Context(b"x" * 20)
    assert e.value.code == hip.EKEYLEN
    c = hip.Context(b"x" * 16)
    with pytest.raises(hip.AesGcmError) as e:
        c.encrypt(b"short iv", b"", b"data")
    assert e.value.code == hip.EIVLEN


# ---------------------------------------------------------------- device-pointer path, in place, streams
def test_device_pointer_path_in_place_and_stream_fixtures(hip):
    for case in golden("streams.json")["cases"]:
        if case["n_bytes"] > (64 << 20):
            continue
        key, iv = stream_key_iv(case)
        aad = bytes.fromhex(case["aad"])
        n = case["n_bytes"]
        c = hip.Context(key)
        buf = hip.DeviceBuffer(n + 16)
        buf.fill_splitmix64(case["pt_seed"], case["first_word"], nbytes=n)
        d_aad = None
        if aad:
            d_aad = hip.DeviceBuffer(len(aad))
            d_aad.upload(aad)
        tag = c.encrypt_dev(iv, buf.ptr, n, buf.ptr, d_aad=d_aad.ptr if aad else None, aad_len=len(aad))   # in place
        assert tag.hex() == case["tag"], case["name"]
        ct = buf.download(n)
        assert hashlib.sha256(ct).hexdigest() == case["ct_sha256"], case["name"]
        assert bytes(ct[:64]).hex() == case["ct_head"] and bytes(ct[-64:]).hex() == case["ct_tail"]
        # decrypt in place back to the generator's plaintext
        t2 = c.decrypt_dev(iv, buf.ptr, n, buf.ptr, d_aad=d_aad.ptr if aad else None, aad_len=len(aad), tag=tag)
        assert t2 == tag
        ref = hip.DeviceBuffer(n + 16)
        ref.fill_splitmix64(case["pt_seed"], case["first_word"], nbytes=n)
        assert hashlib.sha256(buf.download(n)).digest() == hashlib.sha256(ref.download(n)).digest()
        buf.free(); ref.free()


def test_device_fill_matches_host_generator(hip, orc):
    for n, fw in ((0, 0), (1, 0), (7, 3), (8, 0), (4099, 17), (1 << 20, 123456789)):
        b = hip.DeviceBuffer(max(n, 8))
        b.fill_splitmix64(0xAE5C0003, fw, nbytes=n)
        assert bytes(b.download(n)) == bytes(orc.fill_splitmix64(n, 0xAE5C0003, fw))


def test_misaligned_device_pointer_is_rejected(hip):
    c = hip.Context(b"k" * 32)
    b = hip.DeviceBuffer(4096)
    with pytest.raises(hip.AesGcmError) as e:
        c.encrypt_dev(b"i" * 12, b.ptr + 4, 64, b.ptr + 4)
    assert e.value.code == hip.EALIGN


# ---------------------------------------------------------------- shards (multi-GPU algebra on one GPU)
def test_shard_partials_match_fixture_and_fold_to_tag(hip):
    s = golden("shards.json")
    key, iv, aad, pt = (bytes.fromhex(s[k]) for k in ("key", "iv", "aad", "pt"))
    c = hip.Context(key)
    n = len(pt)
    din, dout = hip.DeviceBuffer(n + 16), hip.DeviceBuffer(n + 16)
    din.upload(pt)
    d_aad = hip.DeviceBuffer(len(aad)); d_aad.upload(aad)
    parts = hip.DeviceBuffer(16 * len(s["shards"]))
    for g, sh in enumerate(s["shards"]):
        first, end = sh["first_block"], sh["end_block"]
        ln = min(n, 16 * end) - 16 * first
        c.shard_crypt_dev(False, iv, din.ptr + 16 * first, ln, dout.ptr + 16 * first, first, n, parts.ptr + 16 * g,
                          d_aad=d_aad.ptr if g == 0 else None, aad_len=len(aad) if g == 0 else 0)
    got = bytes(parts.download())
    # fixture weights exclude the AAD; shard 0 on the GPU also carries the AAD polynomial: check shards 1..7 directly
    for g, sh in enumerate(s["shards"]):
        if g:
            assert got[16 * g:16 * g + 16].hex() == sh["weighted"], g
    tag = c.shard_finalize_dev(iv, parts.ptr, len(s["shards"]), len(aad), n)
    assert tag.hex() == s["tag"]
    assert bytes(dout.download(n)).hex() == s["ct"]


def test_sharded_message_equals_single_launch(hip, orc):
    key, iv = splitmix_bytes(80, 32), splitmix_bytes(81, 12)
    c = hip.Context(key)
    for n, al, ranks in ((16 * 100000 + 9, 33, 8), (5 << 20, 0, 4), (16 * 7, 20, 8), (1 << 20, 0, 2)):
        aad = splitmix_bytes(82, al)
        din, dout = hip.DeviceBuffer(n + 16), hip.DeviceBuffer(n + 16)
        din.fill_splitmix64(83, 0, nbytes=n)
        pt = bytes(din.download(n))
        want_ct, want_tag = orc.Fast(key).encrypt(iv, aad, pt)
        d_aad = hip.DeviceBuffer(max(al, 1)); d_aad.upload(aad)
        parts = hip.DeviceBuffer(16 * ranks)
        total_blocks = (n + 15) // 16
        first = 0
        for r in range(ranks):
            blocks = total_blocks // ranks + (1 if r < total_blocks % ranks else 0)
            end = first + blocks
            ln = (n if end == total_blocks else 16 * end) - 16 * first
            c.shard_crypt_dev(False, iv, din.ptr + 16 * first, ln, dout.ptr + 16 * first, first, n, parts.ptr + 16 * r,
                              d_aad=d_aad.ptr if r == 0 else None, aad_len=al if r == 0 else 0)
            first = end
        assert c.shard_finalize_dev(iv, parts.ptr, ranks, al, n) == want_tag
        assert bytes(dout.download(n)) == want_ct


def test_batched_finalize_equals_one_call_per_message(hip, orc):
    """aesgcm_shard_finalize_batch_dev: the tags of M messages from ONE [rank][message][16] gather in one launch -- three messages
    of different length / AAD, four shards each, the shards spread over two contexts whose fused kernels are chained
    (aesgcm_ctx_wait_fused) and joined (aesgcm_ctx_wait) as in bench.py's N > 1 step; pure-body shards included (the context option
    "body_min" = 4096 forces the k_body cut so that the direct weighted-partial path runs at this size)."""
    key = splitmix_bytes(90, 32)
    cs = [hip.Context(key).set_option("body_min", 4096), hip.Context(key).set_option("body_min", 4096)]
    ranks = 4
    msgs = [dict(n=4 * 256 * 16 * 8, al=0, iv=splitmix_bytes(91, 12)),          # four shards of 8 whole 256-block super-rows: pure bodies
            dict(n=(3 << 20) + 5, al=20, iv=splitmix_bytes(92, 12)),
            dict(n=16 * 11, al=0, iv=splitmix_bytes(93, 12))]
    M = len(msgs)
    gathered = hip.DeviceBuffer(16 * M * ranks)                                     # [rank][message][16]
    bufs, want = [], []
    for i, m in enumerate(msgs):
        aad = splitmix_bytes(94 + i, m["al"])
        din, dout = hip.DeviceBuffer(m["n"] + 16), hip.DeviceBuffer(m["n"] + 16)
        din.fill_splitmix64(97 + i, 0, nbytes=m["n"])
        want.append(orc.Fast(key).encrypt(m["iv"], aad, bytes(din.download(m["n"]))))
        d_aad = hip.DeviceBuffer(max(m["al"], 1)); d_aad.upload(aad)
        bufs.append((din, dout, d_aad))
        total_blocks = (m["n"] + 15) // 16
        first = 0
        c = cs[i % 2]
        if i:
            c.wait_fused(cs[(i - 1) % 2])
        for r in range(ranks):
            blocks = total_blocks // ranks + (1 if r < total_blocks % ranks else 0)
            end = first + blocks
            ln = (m["n"] if end == total_blocks else 16 * end) - 16 * first
            c.shard_crypt_dev(False, m["iv"], din.ptr + 16 * first, ln, dout.ptr + 16 * first, first, m["n"], gathered.ptr + 16 * (r * M + i),
                              d_aad=d_aad.ptr if r == 0 else None, aad_len=m["al"] if r == 0 else 0)
            first = end
    cs[0].wait(cs[1])
    tags = cs[0].shard_finalize_batch_dev([m["iv"] for m in msgs], gathered.ptr, ranks, [m["n"] for m in msgs], aad_lens=[m["al"] for m in msgs])
    one_by_one = [cs[0].shard_finalize_dev(m["iv"], gathered.ptr + 16 * i, ranks, m["al"], m["n"], stride_bytes=16 * M) for i, m in enumerate(msgs)]
    assert tags == one_by_one == [w[1] for w in want]
    for (din, dout, _), m, w in zip(bufs, msgs, want):
        assert bytes(dout.download(m["n"])) == w[0]
    with pytest.raises(hip.AesGcmError):
        cs[0].shard_finalize_batch_dev([msgs[0]["iv"]] * 9, gathered.ptr, ranks, [16] * 9)       # more than 8 messages per call


def test_timer_and_context_ordering_entry_points(hip):
    """aesgcm_timer_* brackets work on the stream it is recorded on; aesgcm_ctx_wait / _wait_fused accept two contexts of one device,
    are no-ops on one context and reject NULL"""
    key = splitmix_bytes(70, 16)
    a, b = hip.Context(key), hip.Context(key)
    n = 64 << 20
    din, dout = hip.DeviceBuffer(n), hip.DeviceBuffer(n)
    din.fill_splitmix64(71)
    hip.dev_sync()
    t = hip.Timer()
    t.start(a.stream())
    a.encrypt_dev(splitmix_bytes(72, 12), din.ptr, n, dout.ptr, want_tag=False)
    t.stop(a.stream())
    ms = t.ms()
    assert 0.02 < ms < 50.0, ms                       # 64 MiB at ~0.2 - 1 TB/s
    t.close()
    a.wait(a); a.wait_fused(a)                        # same context: nothing to order
    b.wait_fused(a); b.wait(a)                        # message on b ordered behind a's work: same tag as a alone
    tag_b = b.encrypt_dev(splitmix_bytes(72, 12), din.ptr, n, dout.ptr)
    tag_a = a.encrypt_dev(splitmix_bytes(72, 12), din.ptr, n, dout.ptr)
    assert tag_a == tag_b
    L = hip.load()
    assert L.aesgcm_ctx_wait(None, a._c) == hip.EARG and L.aesgcm_ctx_wait_fused(a._c, None) == hip.EARG
    assert L.aesgcm_timer_ms(None, None) == hip.EARG


def test_distinct_contexts_from_distinct_threads(hip, orc):
    """The ABI's threading contract: a context is not thread-safe, distinct contexts may run concurrently."""
    import threading
    errs = []

    def worker(t):
        try:
            key, iv = splitmix_bytes(900 + t, (16, 24, 32)[t % 3]), splitmix_bytes(950 + t, 12)
            c = hip.Context(key)
            f = orc.Fast(key)
            for it in range(6):
                n = (1 << 20) * (1 + (t + it) % 3) + 17 * t + it
                aad, pt = splitmix_bytes(1000 + 10 * t + it, 13 * it), splitmix_bytes(2000 + 10 * t + it, n)
                want = f.encrypt(iv, aad, pt)
                if c.encrypt(iv, aad, pt) != want:
                    errs.append((t, it, "enc"))
                if c.decrypt(iv, aad, want[0], tag=want[1])[0] != pt:
                    errs.append((t, it, "dec"))
        except Exception as e:          # noqa: BLE001
            errs.append((t, repr(e)))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errs, errs


@pytest.mark.gpu
def test_dev_copy(hip):
    """aesgcm_dev_copy: the plain copy kernel bench.py uses for its measured-bandwidth figure"""
    n = (1 << 20) + 48
    src, dst = hip.DeviceBuffer(n), hip.DeviceBuffer(n)
    src.fill_splitmix64(99)
    hip.dev_copy(dst.ptr, src.ptr, n)
    hip.dev_sync()
    assert bytes(dst.download()) == bytes(src.download())
    with pytest.raises(hip.AesGcmError):
        hip.dev_copy(dst.ptr, src.ptr + 1, 16)


@pytest.mark.gpu
def test_plain_c_caller(hip):
    """examples/kat.c: the reference's README vector through aesgcm_encrypt / aesgcm_decrypt / aesgcm_stream_* from C"""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "examples"), "-s"], check=True)
    r = subprocess.run([os.path.join(root, "examples", "kat")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0 and "KAT OK" in r.stdout, (r.stdout, r.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("frames", [300, 200000])
def test_plain_c_caller_frames(hip, frames):
    """examples/frames.c: MACsec-shaped frames packed back to back under one key through aesgcm_packets_crypt_dev from C -- frame 0 is the reference's README
    vector, one frame is checked against aesgcm_encrypt, all are decrypted in place and authenticated, a forged tag is reported.  300 frames take a lane
    group each in array order, 200 000 the lane kernel in the order of the length classes."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "examples"), "-s", "frames"], check=True)
    r = subprocess.run([os.path.join(root, "examples", "frames"), str(frames)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0 and "FRAMES OK" in r.stdout, (r.stdout, r.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("n,size", [(200, 65536 + 4096 + 16), (37, (1 << 20) + 48), (300, 20000)])
def test_plain_c_caller_messages(hip, n, size):
    """examples/messages.c: many messages under one key as ONE aesgcm_packets_crypt_dev call from C (by rows from 8 KiB per message) -- a sample
    against aesgcm_encrypt_dev, all decrypted in place and authenticated, a forged tag reported and its message wiped (wipe_on_auth_fail)"""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "examples"), "-s", "messages"], check=True)
    r = subprocess.run([os.path.join(root, "examples", "messages"), str(n), str(size)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0 and "MESSAGES OK" in r.stdout and "by rows" in r.stdout, (r.stdout, r.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("tw", ["", "1", "3", "cyc"])
def test_body_split_forced_on_small_messages(hip, orc, tw):
    """The head / k_body / tail cut (aesgcm_ctx_split) normally starts at 256 MiB and the cyclic rows at 64 KiB; the context options
    "body_min" / "cyc_min" / "cyc_max" / "tw" (aesgcm_ctx_set_option) bring them down so that whole messages, decrypts and shards with odd
    first blocks run through k_body -- dealt chunks, or cyclic rows -- at sizes the oracle checks in full."""
    def options(c):
        c.set_option("body_min", 4096)
        c.set_option("cyc_min", 4096 if tw == "cyc" else 0)
        c.set_option("cyc_max", 1 << 50 if tw == "cyc" else 0)
        if tw and tw != "cyc":
            c.set_option("tw", int(tw))
        return c
    for klen in (16, 24, 32):
        key, iv = splitmix_bytes(700 + klen, klen), splitmix_bytes(701, 12)
        c, f = options(hip.Context(key)), orc.Fast(key)
        for n, al in ((16 * 3000 + 5, 0), (16 * (254 + 2048 * 3 + 777) + 11, 20), (1 << 20, 37), (16 * 254 + 16 * 1024 * 2, 16), (3 << 20 | 7, 0)):
            head, body = c.split(n)
            assert body > 0 and head == 0, (n, head, body)          # a whole message starts at block 0: no head
            aad, pt = splitmix_bytes(702 + al, al), splitmix_bytes(703 + n % 97, n)
            want_ct, want_tag = f.encrypt(iv, aad, pt)
            assert c.encrypt(iv, aad, pt) == (want_ct, want_tag), (klen, n, al, "enc")
            assert c.decrypt(iv, aad, want_ct, tag=want_tag) == (pt, want_tag), (klen, n, al, "dec")
        # shards: every rank's range has its own head (first block not a multiple of 256)
        n, al, ranks = (5 << 20) + 9, 33, 3
        aad, pt = splitmix_bytes(710, al), splitmix_bytes(711, n)
        want_ct, want_tag = f.encrypt(iv, aad, pt)
        din, dout = hip.DeviceBuffer(n + 16), hip.DeviceBuffer(n + 16)
        din.upload(pt)
        d_aad = hip.DeviceBuffer(al); d_aad.upload(aad)
        parts = hip.DeviceBuffer(16 * ranks)
        total_blocks, first = (n + 15) // 16, 0
        for r in range(ranks):
            blocks = total_blocks // ranks + (1 if r < total_blocks % ranks else 0)
            end = first + blocks
            ln = (n if end == total_blocks else 16 * end) - 16 * first
            sh, sb = c.split(ln, first)
            assert sb > 0 and sh == (-first) % 256, (first, sh, sb)
            c.shard_crypt_dev(False, iv, din.ptr + 16 * first, ln, dout.ptr + 16 * first, first, n, parts.ptr + 16 * r,
                              d_aad=d_aad.ptr if r == 0 else None, aad_len=al if r == 0 else 0)
            first = end
        assert c.shard_finalize_dev(iv, parts.ptr, ranks, al, n) == want_tag
        assert bytes(dout.download(n)) == want_ct


@pytest.mark.gpu
def test_rekey_equals_a_fresh_context(hip, orc):
    """aesgcm_ctx_rekey (the reference core's key load between frames, tb/gcm_gctr.py:144-175): one context through keys of all three sizes in turn -- every result
    equals the oracle's under that key (message sizes of every launch structure, packets under the context's key too), options survive, a rekey inside an open
    streaming session is refused, and it is cheaper than a new context."""
    import struct, time
    ctx = hip.Context(splitmix_bytes(1, 16)).set_option("cyc_half", 1)
    sizes = (0, 17, 4096, 70000, (1 << 20) + 5, 24 << 20)
    d_in, d_out = hip.DeviceBuffer(max(sizes) + 64), hip.DeviceBuffer(max(sizes) + 64)
    pt = splitmix_bytes(77, max(sizes))
    d_in.upload(pt)
    for it, klen in enumerate((32, 16, 24, 32, 16)):
        key, iv, aad = splitmix_bytes(100 + it, klen), splitmix_bytes(200 + it, 12), splitmix_bytes(300 + it, 20)
        ctx.rekey(key)
        f = orc.Fast(key)
        d_aad = hip.DeviceBuffer(64); d_aad.upload(aad)
        for n in sizes:
            tag = ctx.encrypt_dev(iv, d_in.ptr, n, d_out.ptr, d_aad=d_aad.ptr, aad_len=20)
            want_ct, want_tag = f.encrypt(iv, aad, pt[:n])
            assert tag == want_tag and bytes(d_out.download(n) if n else b"") == want_ct, (it, klen, n)
        # packets under the context's (new) key
        m, plen = 200, 1000
        ivs = splitmix_bytes(400 + it, 12 * m)
        d_ivs, d_tags = hip.DeviceBuffer(12 * m), hip.DeviceBuffer(16 * m)
        d_ivs.upload(ivs)
        ctx.packets_crypt_dev(False, m, d_ivs.ptr, d_in.ptr, d_out.ptr, d_tags.ptr, pkt_len=plen)
        hip.dev_sync()
        ct, tags = bytes(d_out.download(m * plen)), bytes(d_tags.download())
        for p in (0, 57, 199):
            assert (ct[p * plen:(p + 1) * plen], tags[16 * p:16 * p + 16]) == f.encrypt(ivs[12 * p:12 * p + 12], b"", pt[p * plen:(p + 1) * plen]), (it, p)
    ctx.stream_begin(bytes(12))
    with pytest.raises(hip.AesGcmError):
        ctx.rekey(bytes(16))
    ctx.stream_aad(b"")
    ctx.stream_final()
    # a message queued on ANOTHER stream with tag = NULL is still reading the key material when rekey is called: rekey waits for every stream of the device
    # (round 4 waited for the context's own stream only), so the queued message is encrypted under the key it was enqueued with
    key_a, key_b, iv = splitmix_bytes(501, 32), splitmix_bytes(502, 32), splitmix_bytes(503, 12)
    other = hip.Context(key_a)                                      # its stream carries the queued message
    ctx.rekey(key_a)
    n = 24 << 20
    ctx.encrypt_dev(iv, d_in.ptr, n, d_out.ptr, stream=other.stream(), want_tag=False)
    ctx.rekey(key_b)
    want_ct, want_tag = orc.Fast(key_a).encrypt(iv, b"", pt[:n])
    assert bytes(d_out.download(n)) == want_ct
    # (the cost of a rekey against destroy + create is a profiling matter: profiles/ctx_time.py; a wall-clock comparison does not belong in a parity test)
