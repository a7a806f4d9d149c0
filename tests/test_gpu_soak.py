"""GPU, slow: standing soak of the in-launch closings (round 3 ran it once by hand, profiles/cyc_soak.py; the round-3 verdict asked for it in the suite).

The tag of a whole message of 64 KiB .. 1 GiB is closed INSIDE the one launch that encrypts it (cyc_close: a tree per workgroup, memory-side atomics across
workgroups, the last arrival publishes to the pinned host slot), and behind the dealt kernel the first k_fold level closes it the same way (FoldClose,
acc_arrive).  What orders the data before the tag there is not a release fence but write-through stores plus s_waitcnt (aesgcm_kernels.hip, fetch_tag), i.e.
the argument is partly empirical -- so it gets volume:

* 3 000 random messages (64 KiB .. 24 MiB, the U-shaped lengths of tb/gcm_gctr.py:279-281: many small, many near the top, random AAD up to 8 KB, odd
  lengths, random source offsets, encrypt and decrypt, in place and not) through a context with the in-launch closing; EVERY tag against the oracle
  (libcrypto over the same bytes; every 50th message also against the C restatement and, with the ciphertext, against a context that keeps k_fold + k_combine
  behind the launch);
* 20 dealt whole messages of 1 GiB + k x 128 KiB (one aligned body: FoldClose closes the tag) or + k x 4 KiB (a tail behind the body: carried state),
  AES-128 / 192 / 256, with and without FoldClose: tag against the oracle, the two forms against each other.
"""
import os
import random

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.slow]
MiB = 1 << 20
SEED = int(os.environ.get("AESGCM_SOAK_SEED", "0"))      # re-draws the two message soaks (lengths, offsets, IVs, keys, source bytes); the suite runs 0, profiles/archive/runs/r04_run74.sh / 79 1 .. 60


def _u_len(rng, lo, hi):
    """U-shaped: mass at both ends (random.betavariate(.3, .3) as in the reference's length draws)"""
    return lo + int(rng.betavariate(.3, .3) * (hi - lo))


def test_three_thousand_cyclic_messages_every_tag_against_the_oracle(hip, orc):
    from oracle import libcrypto_ref as R
    rng = random.Random(424242 + SEED)
    nmax = 24 * MiB + 1024
    span = 64 * MiB
    d_src = hip.DeviceBuffer(span + nmax + 64)
    d_src.fill_splitmix64(7 + 1000 * SEED)
    src = np.frombuffer(bytes(d_src.download()), dtype=np.uint8)
    d_out = hip.DeviceBuffer(nmax + 64)
    d_work = hip.DeviceBuffer(nmax + 64)
    aad_all = bytes(orc.fill_splitmix64(8192, 8))
    d_aad = hip.DeviceBuffer(8192 + 64); d_aad.upload(aad_all)
    ctxs = {}
    for kb in (16, 24, 32):
        key = bytes(orc.fill_splitmix64(kb, 0x50AC + kb + 4096 * SEED))
        ctxs[kb] = (key, hip.Context(key), hip.Context(key).set_option("cyc_close", 0), orc.Fast(key))
    bad = []
    for it in range(3000):
        kb = rng.choice((16, 24, 32))
        key, fused, three, fast = ctxs[kb]
        n = rng.choice((_u_len(rng, 64 << 10, 24 * MiB), _u_len(rng, 64 << 10, 2 * MiB), 1024 * rng.randint(64, 24576) + rng.choice((0, 1, 16, 1008, 1023))))
        al = rng.choice((0, 0, 0, 1, 16, 20, 1000, rng.randint(1, 8000)))
        aoff = rng.choice((0, 4, 8, 16, 1))
        off = rng.randrange(0, span, 16)
        iv = bytes(rng.randrange(256) for _ in range(12))
        aad = aad_all[aoff:aoff + al]
        pt = src[off:off + n]
        want_ct, want_tag = R.encrypt(key, iv, aad, pt)
        mode = it % 4                                                         # 0: encrypt out of place, 1: encrypt in place, 2: decrypt out of place, 3: decrypt in place
        kw = dict(d_aad=d_aad.ptr + aoff if al else None, aad_len=al)
        if mode == 0:
            tag = fused.encrypt_dev(iv, d_src.ptr + off, n, d_out.ptr, **kw)
            res = d_out
        elif mode == 1:
            hip.dev_copy(d_work.ptr, d_src.ptr + off, (n + 15) // 16 * 16)
            tag = fused.encrypt_dev(iv, d_work.ptr, n, d_work.ptr, **kw)
            res = d_work
        else:
            d_work.upload(want_ct)
            res = d_out if mode == 2 else d_work
            tag = fused.decrypt_dev(iv, d_work.ptr, n, res.ptr, tag=want_tag, **kw)      # raises if the computed tag differs
        if tag != want_tag:
            bad.append((it, kb, n, al, off, mode, tag.hex(), want_tag.hex()))
        if it % 50 == 0:                                                      # the bytes, the C restatement, and the three-launch form
            got = bytes(res.download(n))
            assert got == (bytes(want_ct) if mode < 2 else pt.tobytes()), (it, "data", n, al, mode)
            assert fast.encrypt(iv, aad, pt.tobytes())[1] == want_tag, (it, "the two oracles disagree")
            t3 = three.encrypt_dev(iv, d_src.ptr + off, n, d_out.ptr, **kw)
            assert t3 == want_tag and bytes(d_out.download(n)) == bytes(want_ct), (it, "k_fold + k_combine form")
        assert len(bad) <= 5, bad
    assert not bad, bad


def test_half_shape_two_contexts_in_flight_every_tag_against_the_oracle(hip, orc):
    """800 random messages of 64 KiB .. 40 MiB through TWO contexts in the half shape (k_bodyh), queued (tag = NULL) so that workgroups of two messages share
    the CUs, tags collected one turn late through the host slot; every third message is decrypted; every tag against libcrypto, every 40th output too.  One
    context asks for the half shape, the other leaves it to the library's rule (another context has a message under way)"""
    from oracle import libcrypto_ref as R
    rng = random.Random(515151 + SEED)
    nmax, span = 40 * MiB + 1024, 32 * MiB
    d_src = hip.DeviceBuffer(span + nmax + 64)
    d_src.fill_splitmix64(17 + 1000 * SEED)
    src = np.frombuffer(bytes(d_src.download()), dtype=np.uint8)
    d_out = [hip.DeviceBuffer(nmax + 64), hip.DeviceBuffer(nmax + 64)]
    d_in = [hip.DeviceBuffer(nmax + 64), hip.DeviceBuffer(nmax + 64)]
    aad_all = bytes(orc.fill_splitmix64(4096, 18))
    d_aad = hip.DeviceBuffer(4096 + 64); d_aad.upload(aad_all)
    key = bytes(orc.fill_splitmix64(32, 0x4A1F + 4096 * SEED))
    ctxs = [hip.Context(key).set_option("cyc_half", 1), hip.Context(key)]
    pending = [None, None]
    checked = 0

    def collect(j):
        nonlocal checked
        if pending[j] is None:
            return
        it, n, want_ct, want_tag = pending[j]
        assert ctxs[j].last_tag() == want_tag, (it, n)
        if it % 40 == 0:
            assert bytes(d_out[j].download(n)) == bytes(want_ct), (it, n, "output")
        checked += 1
        pending[j] = None

    for it in range(800):
        j = it & 1
        collect(j)
        n = rng.choice((_u_len(rng, 64 << 10, 40 * MiB), _u_len(rng, 64 << 10, 4 * MiB), 1024 * rng.randint(64, 40960) + rng.choice((0, 1, 16, 1008, 1023))))
        al = rng.choice((0, 0, 16, 20, 1000, rng.randint(1, 4000)))
        off = rng.randrange(0, span, 16)
        iv = bytes(rng.randrange(256) for _ in range(12))
        want_ct, want_tag = R.encrypt(key, iv, aad_all[:al], src[off:off + n])
        if it % 3 == 2:                                                      # decrypt (k_bodyh<.., 1>): the ciphertext goes up first, the plaintext must come back
            d_in[j].upload(want_ct)
            ctxs[j].decrypt_dev(iv, d_in[j].ptr, n, d_out[j].ptr, d_aad=d_aad.ptr if al else None, aad_len=al, want_tag=False)
            pending[j] = (it, n, src[off:off + n].tobytes(), want_tag)
        else:
            ctxs[j].encrypt_dev(iv, d_src.ptr + off, n, d_out[j].ptr, d_aad=d_aad.ptr if al else None, aad_len=al, want_tag=False)
            pending[j] = (it, n, want_ct, want_tag)
    collect(0); collect(1)
    assert checked == 800


@pytest.mark.parametrize("kb", [16, 24, 32])
def test_dealt_gib_messages_with_and_without_foldclose(hip, orc, kb):
    from oracle import libcrypto_ref as R
    rng = random.Random(777 + kb)
    GiB = 1 << 30
    nmax = GiB + 4096 * 4096
    d_src, d_o1, d_o2 = hip.DeviceBuffer(nmax), hip.DeviceBuffer(nmax), hip.DeviceBuffer(nmax)
    d_src.fill_splitmix64(0xF01D + kb)
    src = np.frombuffer(bytes(d_src.download()), dtype=np.uint8)
    key = bytes(orc.fill_splitmix64(kb, 0xF0 + kb))
    fc, nofc = hip.Context(key), hip.Context(key).set_option("fold_close", 0)
    for it in range(7 if kb != 32 else 6):                                   # 20 messages over the three key sizes
        # even turns: a whole number of 128 KiB super-chunks, so the message is ONE aligned body and the first k_fold level closes the tag (FoldClose);
        # odd turns: 4 KiB granularity, i.e. a tail behind the body -- the general path with a carried state, where nothing closes in k_fold
        n = GiB + (128 << 10) * rng.randint(0, 128) if it % 2 == 0 else GiB + 4096 * rng.randint(1, 4096)
        iv = bytes(rng.randrange(256) for _ in range(12))
        head, body = fc.split(n)
        assert head == 0 and body > 0 and (body == n // 16 or it % 2 == 1), (n, head, body)      # the dealt kernel by the library's own rule
        t1 = fc.encrypt_dev(iv, d_src.ptr, n, d_o1.ptr)
        t2 = nofc.encrypt_dev(iv, d_src.ptr, n, d_o2.ptr)
        want_ct, want_tag = R.encrypt(key, iv, b"", src[:n])
        assert t1 == want_tag and t2 == want_tag, (kb, it, n)
        for off in (0, n - MiB, rng.randrange(0, n - MiB, 16)):
            a, b = bytes(d_o1.download(MiB, off)), bytes(d_o2.download(MiB, off))
            assert a == bytes(want_ct[off:off + MiB]) and b == a, (kb, it, n, off)
    for b in (d_src, d_o1, d_o2):
        b.free()
