"""GPU: many messages under one key BY ROWS (k_rows / k_rows_close, csrc/aesgcm_rows.h; round 5) -- aesgcm_packets_crypt_dev from 64 KiB per packet.
The reference's deployment is frame after frame under one key (tb/gcm_test.py:76-85, src/gcm_gctr.vhd:142-144); every message here is compared with the
oracle byte for byte, tags included, and decrypted in place with per-message authentication."""
import hashlib
import os
import random
import struct

import pytest

from util import splitmix_bytes

pytestmark = pytest.mark.gpu


def _up(hip, b):
    d = hip.DeviceBuffer(max(len(b), 16))
    d.upload(b)
    return d


def _check_var(hip, orc, ctx, key, lens, aads, seed, hint, misalign=0, forged=(), aad_array=True):
    """encrypt the messages (offset arrays) through ctx, compare with the oracle, decrypt in place with the tags of `forged` spoiled (aad_array = False: the call
    has no AAD at all -- an empty message then has no unit of work in the row launch, its tag is the closing launch's alone)"""
    m = len(lens)
    f = orc.Fast(key)
    doff, aoff = [misalign], [0]
    for a, b in zip(lens, aads):
        doff.append(doff[-1] + a)
        aoff.append(aoff[-1] + b)
    ivs, aad, pt = splitmix_bytes(seed, 12 * m), splitmix_bytes(seed + 1, max(aoff[-1], 16)), splitmix_bytes(seed + 2, doff[-1])
    d_ivs, d_aad, d_buf = _up(hip, ivs), _up(hip, aad), _up(hip, pt)
    d_doff, d_aoff = _up(hip, struct.pack("<%dQ" % (m + 1), *doff)), _up(hip, struct.pack("<%dQ" % (m + 1), *aoff))
    d_tags, d_auth = hip.DeviceBuffer(16 * m), hip.DeviceBuffer(4 * m)
    akw = dict(d_aad=d_aad.ptr, d_aad_off=d_aoff.ptr) if aad_array else {}
    assert aad_array or not any(aads)
    ctx.packets_crypt_dev(False, m, d_ivs.ptr, d_buf.ptr, d_buf.ptr, d_tags.ptr, pkt_len=hint, d_data_off=d_doff.ptr, **akw)
    hip.dev_sync()
    ct, tags = bytes(d_buf.download(doff[-1])), bytes(d_tags.download())
    assert ct[:misalign] == pt[:misalign]
    for p in range(m):
        want = f.encrypt(ivs[12 * p:12 * p + 12], aad[aoff[p]:aoff[p + 1]], pt[doff[p]:doff[p + 1]])
        assert tags[16 * p:16 * p + 16] == want[1], (p, lens[p], aads[p])
        assert ct[doff[p]:doff[p + 1]] == want[0], (p, lens[p], aads[p])
    bad = bytearray(tags)
    for p in forged:
        bad[16 * p + (p % 16)] ^= 1 << (p % 8)
    d_exp, d_t2 = _up(hip, bytes(bad)), hip.DeviceBuffer(16 * m)
    ctx.packets_crypt_dev(True, m, d_ivs.ptr, d_buf.ptr, d_buf.ptr, d_t2.ptr, pkt_len=hint, d_data_off=d_doff.ptr, d_expect_tags=d_exp.ptr, d_auth=d_auth.ptr, **akw)
    hip.dev_sync()
    assert bytes(d_buf.download(doff[-1])) == pt
    assert bytes(d_t2.download()) == tags
    auth = struct.unpack("<%di" % m, bytes(d_auth.download()))
    assert [i for i, a in enumerate(auth) if not a] == sorted(forged)


@pytest.mark.parametrize("klen", [16, 24, 32])
def test_mixed_message_sizes_with_aad_and_ragged_ends(hip, orc, klen):
    """64 KiB .. 16 MiB, AAD of nothing / a header / more than a row, ends ragged in every way (whole rows, 1 .. 3 rows behind the last super-row, a tail of 63
    blocks + 15 bytes), one empty and one tiny message in between; the call goes by rows because the caller says its packets are message-sized (pkt_len hint)."""
    rng = random.Random(500 + klen)
    key = splitmix_bytes(7000 + klen, klen)
    ctx = hip.Context(key)
    assert ctx.packets_shape(10, 1 << 20, True) == hip.SHAPE_MIXED and ctx.packets_shape(10, 0, True) == hip.SHAPE_MIXED      # offset arrays: routed per message on the device, whatever pkt_len says
    lens = [65536, 65536 + 1023, (1 << 20) + 17, 3 * 4096 + 2048 + 5, 0, 16 << 20, 200000, 131072 + 1008 + 15, 7, (4 << 20) - 16, 65536 + 4096 * 3 + 1024 * 3,
            rng.randrange(65536, 1 << 20), rng.randrange(65536, 1 << 20), (2 << 20) + 1]
    aads = [0, 20, 28, 0, 13, 16, 1100, 0, 8, 0, 33, 2048, 0, 1]
    _check_var(hip, orc, ctx, key, lens, aads, 810 + klen, hint=1 << 20, forged=(2, 5, 13))


@pytest.mark.parametrize("d", [1, 2, 3, 7, 64])
def test_dealt_blocks_of_forced_sizes(hip, orc, d):
    """blocks of 1 / 2 / 3 / 7 / 64 units dealt from the dispensers (the library's own cut of a call this small is one block per wave): cuts in the middle of
    strands, pieces of a single row, messages that are a tail and nothing else"""
    key = splitmix_bytes(7100 + d, 32)
    ctx = hip.Context(key).set_option("rows_block", d)
    lens = [0, 1, 15, 16, 1023, 1024, 1040, 4096, 4097, 5 * 1024 + 1008 + 15, 3 * 4096 + 2 * 1024 + 17, 9 * 4096, 7 * 4096 + 3 * 1024 + 1023, 29 * 4096 + 100, 64 * 4096]
    aads = [0, 20, 0, 16, 1, 0, 33, 0, 13, 1024 + 7, 0, 8, 2048, 0, 5]
    _check_var(hip, orc, ctx, key, lens, aads, 830 + d, hint=1 << 16, forged=(0, 14))


@pytest.mark.parametrize("klen", [16, 32])
def test_messages_without_a_unit_of_work(hip, orc, klen):
    """no AAD in the call and empty messages among the others -- first, several in a row, last: the row launch has nothing for them (rows_units = 0), the closing
    launch's message lanes make their tags, E_K(J0) alone; messages that end on a row have no tail unit; a tail of 64 blocks"""
    key = splitmix_bytes(7150 + klen, klen)
    ctx = hip.Context(key)
    lens = [0, 0, 0, 65536, 0, 1024, 70000, 0, 0, 66560 + 1009, 3 * 1024, 0]
    _check_var(hip, orc, ctx, key, lens, [0] * len(lens), 840 + klen, hint=65536, forged=(1, 3, 11), aad_array=False)
    with hip.debug_library() as dbg:
        dbg.force(pkt_rows=1)
        _check_var(hip, orc, hip.Context(key), key, [0] * 300, [0] * 300, 845 + klen, hint=65536, forged=(7,), aad_array=False)      # nothing but empty messages: no row launch work at all


@pytest.mark.parametrize("klen", [16, 24, 32])
def test_headers_and_ragged_ends_of_every_kind(hip, orc, klen):
    """what is not a whole row -- AAD and tail blocks -- is walked block by block by the lanes of the closing launch, whatever message a block belongs to: AAD of
    1 byte .. 64 blocks (and 65: rows of its own in the row launch), tails of 1 .. 64 blocks with ragged last blocks, messages shorter than a row (no unit in the
    row launch at all), messages that are all AAD; TLS-shaped records (16 KiB + 13 bytes of header) by the hundred"""
    rng = random.Random(900 + klen)
    key = splitmix_bytes(7400 + klen, klen)
    ctx = hip.Context(key)
    lens = [16400, 16384, 100, 16384 + 1023, 0, 5000, 16, 40 * 16 + 3, 2048 + 700, 0, 65536 + 1, 1024 * 9 + 1009, 33] + [rng.randrange(0, 70000) for _ in range(40)]
    aads = [13, 13, 1024, 1025, 20, 0, 1023, 600, 16, 2000, 1, 1040, 0] + [rng.choice((0, 5, 13, 16, 17, 64, 333, 1024, 1025, 3000)) for _ in range(40)]
    _check_var(hip, orc, ctx, key, lens, aads, 860 + klen, hint=65536, forged=(0, 9, 30))
    with hip.debug_library() as dbg:
        dbg.force(pkt_rows=1)
        c2 = hip.Context(key)
        _check_var(hip, orc, c2, key, [16384 + 16 * (i % 3) + (i % 5) for i in range(700)], [13] * 700, 865 + klen, hint=16384, forged=(699,))


@pytest.mark.parametrize("seed", range(int(os.environ.get("AESGCM_ROWS_SEEDS", "6"))))       # (a soak: AESGCM_ROWS_SEEDS=400, profiles/r05/rows_soak.txt)
def test_random_calls_by_rows(hip, orc, seed):
    """random calls forced by rows: 1 .. 250 messages, lengths drawn from empty / shorter than a block / shorter than a row / whole rows / ragged / a few
    hundred KiB, AAD from none / a header / exactly 64 blocks / 65 and more, with or without an AAD array at all, random units per dealt block (or the
    library's cut), a random byte address to pack from, random forged tags -- every message against the oracle, decrypt in place"""
    rng = random.Random(4200 + seed)
    klen = rng.choice((16, 24, 32))
    key = splitmix_bytes(7500 + seed, klen)
    n = rng.randrange(1, 251)

    def length():
        k = rng.randrange(7)
        return (0, rng.randrange(1, 16), rng.randrange(16, 1024), 1024 * rng.randrange(1, 40), rng.randrange(1024, 70000), 1024 * rng.randrange(1, 9) + 1008 + rng.randrange(1, 16),
                rng.randrange(100000, 400000))[k]
    lens = [length() for _ in range(n)]
    with_aad = rng.random() < 0.7
    aads = [rng.choice((0, 0, 13, 20, rng.randrange(1, 1025), 1024, 1025, rng.randrange(1025, 5000))) if with_aad else 0 for _ in range(n)]
    forged = tuple(sorted(rng.sample(range(n), min(n, rng.randrange(0, 4)))))
    with hip.debug_library() as dbg:
        dbg.force(pkt_rows=1)
        ctx = hip.Context(key)
        d = rng.choice((0, 0, 1, 2, 5, 64, 300))
        if d:
            ctx.set_option("rows_block", d)
        _check_var(hip, orc, ctx, key, lens, aads, 9000 + seed, hint=rng.choice((0, 4096, 1 << 20)), misalign=rng.choice((0, 0, 3, 16, 21)), forged=forged, aad_array=with_aad or rng.random() < 0.5)


def test_packed_from_an_odd_byte_address(hip, orc):
    key = splitmix_bytes(7200, 16)
    ctx = hip.Context(key)
    _check_var(hip, orc, ctx, key, [70001, 65536, 99999, 131073], [0, 5, 0, 20], 850, hint=65536, misalign=5, forged=(1,))


@pytest.mark.parametrize("klen,pkt,al,n", [(32, 65536, 0, 300), (16, 65536 + 48, 20, 70), (24, 1 << 20, 16, 9), (32, 3 * 4096 + 1024 + 1, 0, 40), (16, 700, 12, 50),
                                           (32, 1024, 0, 77), (16, 2033, 0, 30), (24, 0, 0, 20), (24, 0, 7, 20), (32, 4096, 20, 1000),
                                           (32, 16400, 13, 500), (16, 16384, 1024, 40), (24, 17000, 1040, 40), (32, 500, 3000, 33)])
def test_fixed_size_records(hip, orc, klen, pkt, al, n):
    """fixed-size records: the library's own rule (by rows from 8 KiB per packet, from 2 KiB when few) and, below that, rows forced through the debug
    library -- including records shorter than a row, which then are tails only, and records of no bytes, which are the closing launch's alone"""
    key = splitmix_bytes(7300 + pkt % 1000, klen)
    f = orc.Fast(key)
    ivs, aad, pt = splitmix_bytes(871, 12 * n), splitmix_bytes(872, max(al * n, 16)), splitmix_bytes(873, pkt * n)

    def run(lib_ctx):
        d_ivs, d_aad, d_buf = _up(hip, ivs), _up(hip, aad), _up(hip, pt)
        d_tags, d_auth = hip.DeviceBuffer(16 * n), hip.DeviceBuffer(4 * n)
        lib_ctx.packets_crypt_dev(False, n, d_ivs.ptr, d_buf.ptr, d_buf.ptr, d_tags.ptr, pkt_len=pkt, d_aad=d_aad.ptr if al else None, aad_len=al)
        hip.dev_sync()
        ct, tags = bytes(d_buf.download(pkt * n)), bytes(d_tags.download())
        for p in range(n):
            want = f.encrypt(ivs[12 * p:12 * p + 12], aad[al * p:al * (p + 1)], pt[pkt * p:pkt * (p + 1)])
            assert (ct[pkt * p:pkt * (p + 1)], tags[16 * p:16 * p + 16]) == want, (pkt, al, p)
        d_exp = _up(hip, tags)
        lib_ctx.packets_crypt_dev(True, n, d_ivs.ptr, d_buf.ptr, d_buf.ptr, d_tags.ptr, pkt_len=pkt, d_aad=d_aad.ptr if al else None, aad_len=al,
                                  d_expect_tags=d_exp.ptr, d_auth=d_auth.ptr)
        hip.dev_sync()
        assert bytes(d_buf.download(pkt * n)) == pt
        assert set(struct.unpack("<%di" % n, bytes(d_auth.download()))) == {1}

    if pkt >= 2048:
        ctx = hip.Context(key)
        assert ctx.packets_shape(n, pkt) == hip.SHAPE_ROWS
        return run(ctx)
    with hip.debug_library() as dbg:
        dbg.force(pkt_rows=1)
        run(hip.Context(key))


def test_the_rule_that_sends_a_call_by_rows(hip):
    """aesgcm_packets_shape: by rows from 8 KiB per packet, from 2 KiB while the packets are at most 16384 (the packet kernels want a packet per lane to fill the
    chip); "rows_min" moves the mark, 0 = never"""
    ctx = hip.Context(bytes(32))
    rows = lambda n, pkt, var=False: ctx.packets_shape(n, pkt, var) == hip.SHAPE_ROWS
    assert rows(1 << 20, 8192) and rows(1 << 19, 8208) and rows(1000, 16400) and rows(5, 1 << 20) and rows(16384, 2048) and rows(100, 4100)
    assert not rows(16385, 8191) and not rows(1 << 20, 4096) and not rows(1000, 2047) and not rows(10, 1514)
    assert ctx.packets_shape(100000, 65536, True) == ctx.packets_shape(1000, 0, True) == hip.SHAPE_MIXED      # with offset arrays the device routes every message; pkt_len is ignored
    # many fixed-size packets of 8 .. 16 KiB whose last, partial row has more than four blocks stay with the packet kernels
    assert not rows(262144, 9000) and rows(262144, 8192 + 64) and not rows(262144, 8192 + 80) and rows(262144, 16384 + 1000) and rows(16384, 9000)
    ctx.set_option("rows_min", 65536)
    assert rows(10, 65536) and rows(10, 16384) and not rows(16385, 32768) and not rows(10, 16383)
    ctx.set_option("rows_min", 0)
    assert not rows(10, 1 << 20)


def test_rows_and_packet_kernels_agree_and_calls_queue_back_to_back(hip, orc):
    """the same 64 messages of 96 KiB through the row kernel (the library's rule) and through the wave-per-packet kernel (rows_min = 0): identical bytes; then
    six calls over different inputs queued on the context's stream without a wait in between (the scratch and the dispensers of one call are the next call's)"""
    key = splitmix_bytes(7400, 32)
    n, pkt = 64, 96 * 1024
    ivs, pt = splitmix_bytes(881, 12 * n), splitmix_bytes(882, pkt * n)
    outs = []
    for rows_min in (65536, 0):
        ctx = hip.Context(key).set_option("rows_min", rows_min)
        d_ivs, d_in, d_out, d_tags = _up(hip, ivs), _up(hip, pt), hip.DeviceBuffer(pkt * n), hip.DeviceBuffer(16 * n)
        ctx.packets_crypt_dev(False, n, d_ivs.ptr, d_in.ptr, d_out.ptr, d_tags.ptr, pkt_len=pkt)
        hip.dev_sync()
        outs.append((hashlib.sha256(bytes(d_out.download())).hexdigest(), bytes(d_tags.download())))
    assert outs[0] == outs[1]
    f = orc.Fast(key)
    assert outs[0][1][:16] == f.encrypt(ivs[:12], b"", pt[:pkt])[1]
    ctx = hip.Context(key)
    runs = []
    for r in range(6):
        m = 20 + 7 * r
        lens = [65536 + 1000 * r + 17 * k for k in range(m)]
        doff = [0]
        for a in lens:
            doff.append(doff[-1] + a)
        ivr, ptr = splitmix_bytes(890 + r, 12 * m), splitmix_bytes(900 + r, doff[-1])
        runs.append((m, doff, ivr, ptr, _up(hip, ivr), _up(hip, ptr), _up(hip, struct.pack("<%dQ" % (m + 1), *doff)), hip.DeviceBuffer(doff[-1] + 16), hip.DeviceBuffer(16 * m)))
    hip.dev_sync()
    for m, doff, ivr, ptr, d_iv, d_in, d_off, d_out, d_tags in runs:
        ctx.packets_crypt_dev(False, m, d_iv.ptr, d_in.ptr, d_out.ptr, d_tags.ptr, pkt_len=65536, d_data_off=d_off.ptr)
    hip.dev_sync()
    for m, doff, ivr, ptr, d_iv, d_in, d_off, d_out, d_tags in runs:
        ct, tags = bytes(d_out.download(doff[-1])), bytes(d_tags.download())
        for p in range(0, m, 3):
            assert (ct[doff[p]:doff[p + 1]], tags[16 * p:16 * p + 16]) == f.encrypt(ivr[12 * p:12 * p + 12], b"", ptr[doff[p]:doff[p + 1]]), p


@pytest.mark.parametrize("klen", [16, 32])
def test_messages_wherever_they_live(hip, orc, klen):
    """aesgcm_messages_crypt_dev: every message in buffers of its own (separate allocations, some at odd addresses inside them, output elsewhere or in place),
    addresses and lengths in device arrays, AAD likewise or none -- by rows, against the oracle; decrypt to a third set of buffers with forged tags, and with the
    context option wipe_on_auth_fail the forged messages come back as zeros"""
    rng = random.Random(1300 + klen)
    key = splitmix_bytes(7700 + klen, klen)
    f = orc.Fast(key)
    lens = [65536, 0, 100, 16400, 1 << 20, 3 * 1024, 70001, 5, 16384 + 1023, 200000, 1024, 33] + [rng.randrange(0, 50000) for _ in range(30)]
    aads = [13, 0, 1024, 0, 20, 1025, 0, 16, 7, 0, 3000, 0] + [rng.choice((0, 13, 64, 1024)) for _ in range(30)]
    n = len(lens)
    ivs = splitmix_bytes(1310 + klen, 12 * n)
    skew = [rng.choice((0, 0, 16, 3, 21)) for _ in range(n)]
    pts = [splitmix_bytes(1400 + k, lens[k]) for k in range(n)]
    aad = [splitmix_bytes(1500 + k, aads[k]) for k in range(n)]
    b_in = [hip.DeviceBuffer(lens[k] + 48) for k in range(n)]
    b_out = [hip.DeviceBuffer(lens[k] + 48) for k in range(n)]
    b_back = [hip.DeviceBuffer(lens[k] + 48) for k in range(n)]
    b_aad = [hip.DeviceBuffer(aads[k] + 16) for k in range(n)]
    for k in range(n):
        if lens[k]:
            b_in[k].upload(bytes(skew[k]) + pts[k])
        if aads[k]:
            b_aad[k].upload(aad[k])
    u64s = lambda v: _up(hip, struct.pack("<%dQ" % n, *v))
    u32s = lambda v: _up(hip, struct.pack("<%dI" % n, *v))
    d_ivs, d_len, d_alen = _up(hip, ivs), u32s(lens), u32s(aads)
    d_inp, d_outp = u64s([b_in[k].ptr + skew[k] for k in range(n)]), u64s([b_out[k].ptr + (skew[k] ^ 16) % 24 for k in range(n)])
    out_skew = [(skew[k] ^ 16) % 24 for k in range(n)]
    d_backp, d_aadp = u64s([b_back[k].ptr + skew[k] for k in range(n)]), u64s([b_aad[k].ptr for k in range(n)])
    d_tags, d_auth = hip.DeviceBuffer(16 * n), hip.DeviceBuffer(4 * n)
    ctx = hip.Context(key)
    ctx.messages_crypt_dev(False, n, d_ivs.ptr, d_inp.ptr, d_len.ptr, d_outp.ptr, d_tags.ptr, d_aad_ptr=d_aadp.ptr, d_aad_len=d_alen.ptr)
    hip.dev_sync()
    tags = bytes(d_tags.download())
    for k in range(n):
        want = f.encrypt(ivs[12 * k:12 * k + 12], aad[k], pts[k])
        got = bytes(b_out[k].download(lens[k], out_skew[k])) if lens[k] else b""
        assert (got, tags[16 * k:16 * k + 16]) == want, (k, lens[k], aads[k])
    forged = (0, 4, n - 1)
    bad = bytearray(tags)
    for k in forged:
        bad[16 * k + 5] ^= 2
    d_exp, d_t2 = _up(hip, bytes(bad)), hip.DeviceBuffer(16 * n)
    ctx.set_option("wipe_on_auth_fail", 1)
    ctx.messages_crypt_dev(True, n, d_ivs.ptr, d_outp.ptr, d_len.ptr, d_backp.ptr, d_t2.ptr, d_aad_ptr=d_aadp.ptr, d_aad_len=d_alen.ptr, d_expect_tags=d_exp.ptr, d_auth=d_auth.ptr)
    hip.dev_sync()
    assert bytes(d_t2.download()) == tags
    auth = struct.unpack("<%di" % n, bytes(d_auth.download()))
    assert [k for k in range(n) if not auth[k]] == sorted(forged)
    for k in range(n):
        if lens[k]:
            assert bytes(b_back[k].download(lens[k], skew[k])) == (bytes(lens[k]) if k in forged else pts[k]), k
    # no AAD arrays at all, in place
    ctx2 = hip.Context(key)
    ctx2.messages_crypt_dev(False, n, d_ivs.ptr, d_inp.ptr, d_len.ptr, d_inp.ptr, d_tags.ptr)
    hip.dev_sync()
    tags = bytes(d_tags.download())
    for k in (0, 1, 4, 9, n - 1):
        want = f.encrypt(ivs[12 * k:12 * k + 12], b"", pts[k])
        assert ((bytes(b_in[k].download(lens[k], skew[k])) if lens[k] else b""), tags[16 * k:16 * k + 16]) == want, k
    with pytest.raises(hip.AesGcmError):
        ctx2.messages_crypt_dev(False, n, d_ivs.ptr, d_inp.ptr, d_len.ptr, d_inp.ptr, d_tags.ptr, d_aad_ptr=d_aadp.ptr)      # an AAD address array without its lengths


@pytest.mark.parametrize("n", [4097, 300000, 1200000])
def test_three_hundred_thousand_small_messages_rows_and_packet_kernels_agree(hip, orc, n):
    """300 000 messages of 0 .. 3000 bytes with headers of 0 .. 40 bytes in one call (offset arrays): nearly all of the work is the closing launch's -- smalls
    blocks by the million, three prefix sums over 300 000 messages, tails that straddle the waves -- and a third of the messages has a row for the row launch.
    Forced by rows and through the packet kernels (themselves held to the oracle by test_gpu_batch.py): the same ciphertext (SHA-256) and the same tags; the
    first and last hundred messages against the oracle; decrypt by rows restores the plaintext and finds the forged tags"""
    rng = random.Random(77 + n)                                      # (4097: the first count whose plan is made by the five launches with a thread per message, not by one workgroup)
    key = splitmix_bytes(7600, 32)
    lens = [rng.randrange(0, 3001 if n < 1000000 else 1300) for _ in range(n)]      # (1 200 000: more than 1024 workgroups of the plan's launches -- its scans of the workgroups' sums go in two tiles)
    aads = [rng.choice((0, 13, 16, 40)) for _ in range(n)]
    doff, aoff = [0], [0]
    for a, b in zip(lens, aads):
        doff.append(doff[-1] + a)
        aoff.append(aoff[-1] + b)
    d_in, d_aad, d_ivs = hip.DeviceBuffer(doff[-1] + 16), hip.DeviceBuffer(aoff[-1] + 16), hip.DeviceBuffer(12 * n)
    d_in.fill_splitmix64(0x51, nbytes=(doff[-1] + 16) // 8 * 8)
    d_aad.fill_splitmix64(0x52, nbytes=(aoff[-1] + 16) // 8 * 8)
    d_ivs.fill_splitmix64(0x53, nbytes=12 * n)
    d_doff, d_aoff = _up(hip, struct.pack("<%dQ" % (n + 1), *doff)), _up(hip, struct.pack("<%dQ" % (n + 1), *aoff))
    res = []
    with hip.debug_library() as dbg:
        for rows in (1, 2):
            dbg.force(pkt_rows=rows)
            ctx = hip.Context(key)
            d_out, d_tags = hip.DeviceBuffer(doff[-1] + 16), hip.DeviceBuffer(16 * n)
            ctx.packets_crypt_dev(False, n, d_ivs.ptr, d_in.ptr, d_out.ptr, d_tags.ptr, pkt_len=2048, d_data_off=d_doff.ptr, d_aad=d_aad.ptr, d_aad_off=d_aoff.ptr)
            hip.dev_sync()
            res.append((hashlib.sha256(bytes(d_out.download(doff[-1]))).hexdigest(), bytes(d_tags.download())))
            if rows == 1:
                assert ctx.packets_shape(n, 2048, True) == hip.SHAPE_MIXED
                ct_head, ct_tail = bytes(d_out.download(doff[100])), bytes(d_out.download(doff[n] - doff[n - 100], doff[n - 100]))
                forged = (0, 1234, n - 1)
                bad = bytearray(res[0][1])
                for p in forged:
                    bad[16 * p] ^= 0x40
                d_exp, d_auth, d_t2 = _up(hip, bytes(bad)), hip.DeviceBuffer(4 * n), hip.DeviceBuffer(16 * n)
                ctx.packets_crypt_dev(True, n, d_ivs.ptr, d_out.ptr, d_out.ptr, d_t2.ptr, pkt_len=2048, d_data_off=d_doff.ptr, d_aad=d_aad.ptr, d_aad_off=d_aoff.ptr,
                                      d_expect_tags=d_exp.ptr, d_auth=d_auth.ptr)
                hip.dev_sync()
                assert bytes(d_out.download(doff[-1])) == bytes(d_in.download(doff[-1]))
                assert bytes(d_t2.download()) == res[0][1]
                auth = struct.unpack("<%di" % n, bytes(d_auth.download()))
                assert sum(auth) == n - len(forged) and all(auth[p] == 0 for p in forged)
    assert res[0] == res[1]
    f = orc.Fast(key)
    ivs, pt_head, aad_all = bytes(d_ivs.download()), bytes(d_in.download(doff[100])), bytes(d_aad.download(aoff[-1]))
    pt_tail = bytes(d_in.download(doff[n] - doff[n - 100], doff[n - 100]))
    for p in list(range(100)) + list(range(n - 100, n)):
        base, pt, ct = (0, pt_head, ct_head) if p < 100 else (doff[n - 100], pt_tail, ct_tail)
        want = f.encrypt(ivs[12 * p:12 * p + 12], aad_all[aoff[p]:aoff[p + 1]], pt[doff[p] - base:doff[p + 1] - base])
        assert (ct[doff[p] - base:doff[p + 1] - base], res[0][1][16 * p:16 * p + 16]) == want, p


@pytest.mark.slow
def test_4096_messages_of_one_mib_by_checksum(hip, orc):
    """the benchmark's shape at full size -- 4096 x 1 MiB, 4 GiB -- against the single-message path of the same library (itself pinned to the libcrypto fixtures at
    1 GiB and 16 GiB by tests/test_gpu_large.py): 64 of the messages re-encrypted one at a time, ciphertext by SHA-256 and tags compared; decrypt restores the
    SplitMix64 plaintext (checked on the device by re-encrypting: the round trip's tags are the same)"""
    key = splitmix_bytes(7500, 32)
    n, pkt = 4096, 1 << 20
    ctx = hip.Context(key)
    d_ivw, d_ivs = hip.DeviceBuffer(16 * n), hip.DeviceBuffer(12 * n)
    d_ivw.fill_splitmix64(0x4956)
    ivw = bytes(d_ivw.download())
    ivs = b"".join(ivw[16 * p:16 * p + 12] for p in range(n))
    d_ivs.upload(ivs)
    d_pt, d_ct, d_tags, d_auth = hip.DeviceBuffer(pkt * n), hip.DeviceBuffer(pkt * n), hip.DeviceBuffer(16 * n), hip.DeviceBuffer(4 * n)
    d_pt.fill_splitmix64(0xAE5C0055)
    assert ctx.packets_shape(n, pkt) == hip.SHAPE_ROWS
    ctx.packets_crypt_dev(False, n, d_ivs.ptr, d_pt.ptr, d_ct.ptr, d_tags.ptr, pkt_len=pkt)
    hip.dev_sync()
    tags = bytes(d_tags.download())
    one = hip.Context(key)
    d_one = hip.DeviceBuffer(pkt)
    for p in list(range(0, n, 67)) + [n - 1]:
        t = one.encrypt_dev(ivs[12 * p:12 * p + 12], d_pt.ptr + p * pkt, pkt, d_one.ptr)
        assert t == tags[16 * p:16 * p + 16], p
        assert hashlib.sha256(bytes(d_one.download())).digest() == hashlib.sha256(bytes(d_ct.download(pkt, p * pkt))).digest(), p
    f = orc.Fast(key)
    assert f.encrypt(ivs[:12], b"", bytes(d_pt.download(pkt)))[1] == tags[:16]
    d_t2 = hip.DeviceBuffer(16 * n)
    ctx.packets_crypt_dev(True, n, d_ivs.ptr, d_ct.ptr, d_ct.ptr, d_t2.ptr, pkt_len=pkt, d_expect_tags=d_tags.ptr, d_auth=d_auth.ptr)
    hip.dev_sync()
    assert bytes(d_t2.download()) == tags and set(struct.unpack("<%di" % n, bytes(d_auth.download()))) == {1}
    assert bytes(d_ct.download(pkt, 1234 * pkt)) == bytes(d_pt.download(pkt, 1234 * pkt))


@pytest.mark.parametrize("aad_len,scattered", [(0, False), (13, False), (13, True)])
def test_bench_line_of_the_messages_config(aad_len, scattered):
    """bench.py --config msgs at a small size (and with a header per message: --aad-len): one JSON line, the call goes by rows, the tags and ciphertext of the
    sampled messages equal the single-message path's"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "msgs", "--n-pkts", "300", "--pkt-len", str(65536 + 1024 * 3 + 17), "--steps", "3", "--warmup", "1",
                          "--aad-len", str(aad_len), "--no-cpu-baseline"] + (["--scattered"] if scattered else []), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["tag_ok"] is True and line["config"]["shape"] == "rows" and line["unit"] == "GiB/s" and line["n_gpus"] == 1 and line["config"]["aad_bytes"] == aad_len and line["config"]["scattered"] == scattered
    assert line["roofline"]["kernel"].startswith("k_rows<14,0>") and 0 < line["roofline"]["frac"] < 1
