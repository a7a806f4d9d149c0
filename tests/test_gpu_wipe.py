"""GPU: the context option "wipe_on_auth_fail" (round 5) -- a decrypt call that verifies a tag leaves zeros, not unauthenticated plaintext, where verification
fails; with the option off (the default) the bytes stay, as in the reference model (tb/gcm_model.py:29-30,47-51: decrypt returns the plaintext, then raises).
Also here: aesgcm_ctx_last_launch, and aesgcm_last_tag's refusal of a stream the message was not enqueued on."""
import struct

import pytest

from util import splitmix_bytes

pytestmark = pytest.mark.gpu


def _up(hip, b):
    d = hip.DeviceBuffer(max(len(b), 16))
    d.upload(b)
    return d


@pytest.mark.parametrize("n", [1000, 70000, (3 << 20) + 5])
def test_whole_message_paths(hip, orc, n):
    key, iv, aad = splitmix_bytes(9001, 32), splitmix_bytes(9002, 12), splitmix_bytes(9003, 20)
    pt = splitmix_bytes(9004, n)
    ct, tag = orc.Fast(key).encrypt(iv, aad, pt)
    bad = bytes([tag[0] ^ 1]) + tag[1:]
    for wipe in (0, 1):
        ctx = hip.Context(key).set_option("wipe_on_auth_fail", wipe)
        want = bytes(n) if wipe else pt
        # host buffers
        out = bytearray(b"\xAA" * n)
        with pytest.raises(hip.AuthenticationError):
            ctx.decrypt(iv, aad, ct, tag=bad, out=out)
        assert bytes(out) == (bytes(n) if wipe else pt), (wipe, "decrypt")
        assert ctx.decrypt(iv, aad, ct, tag=tag)[0] == pt                      # the right tag: plaintext, with the option on as well
        # device buffers
        d_ct, d_pt, d_aad = _up(hip, ct), hip.DeviceBuffer(n + 16), _up(hip, aad)
        with pytest.raises(hip.AuthenticationError):
            ctx.decrypt_dev(iv, d_ct.ptr, n, d_pt.ptr, d_aad=d_aad.ptr, aad_len=len(aad), tag=bad)
        assert bytes(d_pt.download(n)) == want, (wipe, "decrypt_dev")
        # pipelined host buffers
        out = bytearray(b"\xAA" * n)
        with pytest.raises(hip.AuthenticationError):
            ctx.decrypt_pipelined(iv, aad, ct, tag=bad, out=out, chunk_bytes=1 << 20)
        assert bytes(out) == want, (wipe, "decrypt_pipelined")


@pytest.mark.parametrize("rows", [False, True])
def test_packet_paths(hip, orc, rows):
    """packets under one key, through the packet kernels and by rows: with the option on exactly the packets whose tags were forged come back as zeros"""
    key = splitmix_bytes(9010, 16)
    f = orc.Fast(key)
    lens = [70000, 65536, 1000, 0, 131072 + 5, 66000] if rows else [1000, 64, 0, 1500, 4096, 17]
    m = len(lens)
    doff = [0]
    for a in lens:
        doff.append(doff[-1] + a)
    ivs, pt = splitmix_bytes(9011, 12 * m), splitmix_bytes(9012, doff[-1])
    cts = [f.encrypt(ivs[12 * p:12 * p + 12], b"", pt[doff[p]:doff[p + 1]]) for p in range(m)]
    ct = b"".join(c for c, _ in cts)
    tags = bytearray(b"".join(t for _, t in cts))
    forged = (1, 4)
    for p in forged:
        tags[16 * p + 3] ^= 0x10
    for wipe in (0, 1):
        ctx = hip.Context(key).set_option("wipe_on_auth_fail", wipe)
        d_ivs, d_buf, d_off, d_exp = _up(hip, ivs), _up(hip, ct), _up(hip, struct.pack("<%dQ" % (m + 1), *doff)), _up(hip, bytes(tags))
        d_tags, d_auth = hip.DeviceBuffer(16 * m), hip.DeviceBuffer(4 * m)
        ctx.packets_crypt_dev(True, m, d_ivs.ptr, d_buf.ptr, d_buf.ptr, d_tags.ptr, pkt_len=65536 if rows else 0, d_data_off=d_off.ptr,
                              d_expect_tags=d_exp.ptr, d_auth=d_auth.ptr)
        hip.dev_sync()
        assert ctx.packets_shape(m, 65536 if rows else 0, True) == hip.SHAPE_MIXED
        auth = struct.unpack("<%di" % m, bytes(d_auth.download()))
        assert [i for i, a in enumerate(auth) if not a] == list(forged)
        back = bytes(d_buf.download(doff[-1]))
        for p in range(m):
            want = bytes(lens[p]) if (wipe and p in forged) else pt[doff[p]:doff[p + 1]]
            assert back[doff[p]:doff[p + 1]] == want, (rows, wipe, p)


def test_wipe_without_verdicts_is_refused(hip):
    """the option asks for "no unauthenticated plaintext"; a decrypt call that names expected tags but has no d_auth array would compare nothing and wipe nothing
    (round 5 returned OK with the plaintext in place: ADVICE r05) -- AESGCM_EARG, for packets (fixed records, offset arrays) and for messages wherever they live"""
    ctx = hip.Context(bytes(16)).set_option("wipe_on_auth_fail", 1)
    n, pkt = 4, 1024
    d_ivs, d_buf, d_tags, d_exp = hip.DeviceBuffer(12 * n), hip.DeviceBuffer(pkt * n), hip.DeviceBuffer(16 * n), hip.DeviceBuffer(16 * n)
    d_off = _up(hip, struct.pack("<%dQ" % (n + 1), *[pkt * k for k in range(n + 1)]))
    d_ptr, d_len = _up(hip, struct.pack("<%dQ" % n, *[d_buf.ptr + pkt * k for k in range(n)])), _up(hip, struct.pack("<%dI" % n, *[pkt] * n))
    for call in (lambda: ctx.packets_crypt_dev(True, n, d_ivs.ptr, d_buf.ptr, d_buf.ptr, d_tags.ptr, pkt_len=pkt, d_expect_tags=d_exp.ptr),
                 lambda: ctx.packets_crypt_dev(True, n, d_ivs.ptr, d_buf.ptr, d_buf.ptr, d_tags.ptr, d_data_off=d_off.ptr, d_expect_tags=d_exp.ptr),
                 lambda: ctx.messages_crypt_dev(True, n, d_ivs.ptr, d_ptr.ptr, d_len.ptr, d_ptr.ptr, d_tags.ptr, d_expect_tags=d_exp.ptr)):
        with pytest.raises(hip.AesGcmError) as ei:
            call()
        assert ei.value.code == hip.EARG
    ctx.set_option("wipe_on_auth_fail", 0)                    # without the option the same call is the reference model's: plaintext and computed tags, nothing compared
    ctx.packets_crypt_dev(True, n, d_ivs.ptr, d_buf.ptr, d_buf.ptr, d_tags.ptr, pkt_len=pkt, d_expect_tags=d_exp.ptr)
    hip.dev_sync()


def test_wipe_helper_behind_a_batch_call(hip, orc):
    """aesgcm_wipe_failed_dev: the same for callers of the context-free batch entry points (a key per packet)"""
    n, pkt, klen = 40, 1024 + 7, 16
    keys, ivs, pt = splitmix_bytes(9020, klen * n), splitmix_bytes(9021, 12 * n), splitmix_bytes(9022, pkt * n)
    want = [orc.Fast(keys[klen * p:klen * (p + 1)]).encrypt(ivs[12 * p:12 * p + 12], b"", pt[pkt * p:pkt * (p + 1)]) for p in range(n)]
    tags = bytearray(b"".join(t for _, t in want))
    tags[16 * 7] ^= 1
    tags[16 * 39 + 15] ^= 0x80
    d_keys, d_ivs, d_buf, d_exp = _up(hip, keys), _up(hip, ivs), _up(hip, b"".join(c for c, _ in want)), _up(hip, bytes(tags))
    d_tags, d_auth = hip.DeviceBuffer(16 * n), hip.DeviceBuffer(4 * n)
    hip.batch_crypt_dev(True, n, klen, d_keys.ptr, d_ivs.ptr, d_buf.ptr, pkt, d_buf.ptr, d_tags.ptr, d_expect_tags=d_exp.ptr, d_auth=d_auth.ptr)
    hip.wipe_failed_dev(n, d_buf.ptr, d_auth.ptr, pkt_len=pkt)
    hip.dev_sync()
    back = bytes(d_buf.download(pkt * n))
    for p in range(n):
        assert back[pkt * p:pkt * (p + 1)] == (bytes(pkt) if p in (7, 39) else pt[pkt * p:pkt * (p + 1)]), p


def test_last_launch_and_last_tag_on_the_wrong_stream(hip, orc):
    key, iv = splitmix_bytes(9030, 32), splitmix_bytes(9031, 12)
    ctx, other = hip.Context(key), hip.Context(key)
    d_in, d_out = hip.DeviceBuffer(4 << 20), hip.DeviceBuffer(4 << 20)
    d_in.fill_splitmix64(9032)
    assert ctx.last_launch() == hip.LAUNCH_NONE
    ctx.encrypt_dev(iv, d_in.ptr, 1000, d_out.ptr)
    assert ctx.last_launch() == hip.LAUNCH_MAIN
    ctx.set_option("cyc_half", 0)
    ctx.encrypt_dev(iv, d_in.ptr, 4 << 20, d_out.ptr)
    assert ctx.last_launch() == hip.LAUNCH_CYCLIC
    ctx.set_option("cyc_half", 1)
    t = ctx.encrypt_dev(iv, d_in.ptr, 4 << 20, d_out.ptr)
    assert ctx.last_launch() == hip.LAUNCH_CYCLIC_HALF
    assert t == orc.Fast(key).encrypt(iv, b"", bytes(d_in.download()))[1]
    # a message enqueued on `other`'s stream, its tag asked for with the context's own stream and no time to poll: the wait falls back to a synchronisation of the
    # stream it was GIVEN, which says nothing about the message -- refused (or, if the message happens to be through already, answered correctly), never a stale tag
    ctx.set_option("poll_us", 0)
    hip.dev_sync()
    big = hip.DeviceBuffer(512 << 20)
    big.fill_splitmix64(9033)
    hip.dev_sync()
    ctx.encrypt_dev(iv, big.ptr, 512 << 20, big.ptr, stream=other.stream(), want_tag=False)
    try:
        got = ctx.last_tag()
    except hip.AesGcmError as e:
        assert e.code == hip.ESTATE
    else:
        hip.dev_sync()
        assert got == ctx.last_tag(stream=other.stream())
    hip.dev_sync()
    assert len(ctx.last_tag(stream=other.stream())) == 16
