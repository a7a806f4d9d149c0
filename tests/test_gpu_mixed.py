"""GPU: ONE call that holds messages of both kinds (round 6) -- aesgcm_packets_crypt_dev with offset arrays and aesgcm_messages_crypt_dev route every message by its
own size ON THE DEVICE: the long ones go by rows (k_rows), the short ones to the packet kernels (k_pktl / k_pktg), inside the same call (csrc/aesgcm_rows.h RowsHdr,
k_len_scan).  The reference's own traffic is of both kinds at once: tb/gcm_gctr.py:279-281 draws n_bytes = int(betavariate(0.1, 0.1) * max) -- a U-shaped
distribution, lengths near 0 and near max in the same stream.  Until round 5 the whole call went one way (by a caller's hint) and took a 2 - 14 x cliff for
the other half.

Every message of every call here is compared with libcrypto (the survey's primary external oracle O1, through the per-frame EVP loop in C of oracle/evp_batch.c --
2^18 messages / 8 GiB in seconds) and a sample of them, chosen around the routing mark, with the C restatement of the reference's RTL (oracle/aesgcm_oracle.c)."""
import ctypes
import os
import random
import struct

import numpy as np
import pytest

from util import splitmix_bytes

pytestmark = pytest.mark.gpu


def _up(hip, b):
    d = hip.DeviceBuffer(max(len(b), 16))
    d.upload(b)
    return d


def u_shaped(rng, n, max_len):
    """the reference harness's draw (tb/gcm_gctr.py:279-281)"""
    return [int(rng.betavariate(0.1, 0.1) * max_len) for _ in range(n)]


def _evp():
    from oracle import cpu_baseline
    return cpu_baseline.evp_batch_lib()


def check_against_libcrypto(hip, key, ivs, aad, aoff, doff, d_in, d_out, tags, chunk_bytes=256 << 20):
    """every message of the call: ciphertext (device buffer d_out) and tag against libcrypto, chunk by chunk so that the host never holds more than a chunk"""
    L = _evp()
    n = len(doff) - 1
    doff_a, aoff_a = np.array(doff, dtype=np.uint64), np.array(aoff, dtype=np.uint64)
    ivs_a, aad_a = np.frombuffer(ivs, dtype=np.uint8), np.frombuffer(aad if len(aad) else b"\0", dtype=np.uint8)
    a = 0
    while a < n:
        b = a + 1
        while b < n and doff[b + 1] - doff[a] <= chunk_bytes:
            b += 1
        lo, hi = doff[a], doff[b]
        pt = np.frombuffer(bytes(d_in.download(hi - lo, lo)) if hi > lo else b"", dtype=np.uint8)
        got = np.frombuffer(bytes(d_out.download(hi - lo, lo)) if hi > lo else b"", dtype=np.uint8)
        want, wtags = np.empty(max(hi - lo, 1), dtype=np.uint8), np.empty(16 * (b - a), dtype=np.uint8)
        pt_p = pt.ctypes.data if hi > lo else want.ctypes.data
        rc = L.evp_frames_crypt(b - a, len(key), key, ivs_a[12 * a:].ctypes.data, aad_a.ctypes.data, aoff_a[a:].ctypes.data, 0, pt_p, doff_a[a:].ctypes.data, lo, want.ctypes.data, wtags.ctypes.data)
        assert rc == 0
        if not np.array_equal(want[:hi - lo], got):
            for p in range(a, b):
                assert bytes(got[doff[p] - lo:doff[p + 1] - lo]) == bytes(want[doff[p] - lo:doff[p + 1] - lo]), ("ciphertext", p, doff[p + 1] - doff[p], aoff[p + 1] - aoff[p])
        if wtags.tobytes() != tags[16 * a:16 * b]:
            for p in range(a, b):
                assert tags[16 * p:16 * p + 16] == wtags[16 * (p - a):16 * (p - a) + 16].tobytes(), ("tag", p, doff[p + 1] - doff[p], aoff[p + 1] - aoff[p])
        a = b


SPLIT = dict(route_mid_min=0, route_blocks_min=0)      # context options: always the high mark (8 KiB), always a packet launch for what lies below it -- the library's own rule sends small calls by rows altogether


def _mixed_call(hip, orc, klen, lens, aads, seed, misalign=0, sample=200, opts=None):
    n = len(lens)
    key = splitmix_bytes(seed, klen)
    doff, aoff = [misalign], [0]
    for a, b in zip(lens, aads):
        doff.append(doff[-1] + a)
        aoff.append(aoff[-1] + b)
    total = doff[-1]
    ivs, aad = splitmix_bytes(seed + 1, 12 * n), splitmix_bytes(seed + 2, max(aoff[-1], 16))
    d_in, d_out = hip.DeviceBuffer(total + 64), hip.DeviceBuffer(total + 64)
    d_in.fill_splitmix64(seed + 3, nbytes=(total + 64) // 8 * 8)
    d_ivs, d_aad = _up(hip, ivs), _up(hip, aad)
    d_doff, d_aoff = _up(hip, struct.pack("<%dQ" % (n + 1), *doff)), _up(hip, struct.pack("<%dQ" % (n + 1), *aoff))
    d_tags, d_auth = hip.DeviceBuffer(16 * n), hip.DeviceBuffer(4 * n)
    ctx = hip.Context(key)
    for k, v in (opts or {}).items():
        ctx.set_option(k, v)
    assert ctx.packets_shape(n, 0, True) == hip.SHAPE_MIXED
    ctx.packets_crypt_dev(False, n, d_ivs.ptr, d_in.ptr, d_out.ptr, d_tags.ptr, d_data_off=d_doff.ptr, d_aad=d_aad.ptr, d_aad_off=d_aoff.ptr)
    hip.dev_sync()
    assert ctx.status() == (hip.STATUS_OK, 0)
    tags = bytes(d_tags.download())
    check_against_libcrypto(hip, key, ivs, aad, aoff, doff, d_in, d_out, tags)
    # a sample against the restatement of the RTL: the first and last messages, the shortest and longest, and those closest to the routing marks (2 KiB, 8 KiB)
    f = orc.Fast(key)
    size = [lens[p] + aads[p] for p in range(n)]
    by_mark = sorted(range(n), key=lambda p: min(abs(size[p] - 2048), abs(size[p] - 8192)))[:sample // 2]
    rng = random.Random(seed)
    pick = set(by_mark) | {0, n - 1, min(range(n), key=lambda p: lens[p]), max(range(n), key=lambda p: lens[p])} | set(rng.sample(range(n), min(n, sample // 2)))
    budget = 64 << 20                                            # bytes of oracle work (the restatement runs at 90 MB/s)
    for p in sorted(pick, key=lambda p: lens[p]):
        if lens[p] > budget:
            break
        budget -= lens[p]
        pt = bytes(d_in.download(lens[p], doff[p])) if lens[p] else b""
        want = f.encrypt(ivs[12 * p:12 * p + 12], aad[aoff[p]:aoff[p + 1]], pt)
        got = bytes(d_out.download(lens[p], doff[p])) if lens[p] else b""
        assert (got, tags[16 * p:16 * p + 16]) == want, (p, lens[p], aads[p])
    # decrypt in place with forged tags: the plaintext comes back, the computed tags are the same, exactly the forged ones fail
    forged = sorted(set([0, n - 1, n // 2] + rng.sample(range(n), min(n, 20)) + by_mark[:10]))
    bad = bytearray(tags)
    for p in forged:
        bad[16 * p + (p % 16)] ^= 1 << (p % 8)
    d_exp, d_t2 = _up(hip, bytes(bad)), hip.DeviceBuffer(16 * n)
    ctx.packets_crypt_dev(True, n, d_ivs.ptr, d_out.ptr, d_out.ptr, d_t2.ptr, d_data_off=d_doff.ptr, d_aad=d_aad.ptr, d_aad_off=d_aoff.ptr, d_expect_tags=d_exp.ptr, d_auth=d_auth.ptr)
    hip.dev_sync()
    assert bytes(d_t2.download()) == tags
    auth = np.frombuffer(bytes(d_auth.download()), dtype=np.int32)
    assert np.flatnonzero(auth == 0).tolist() == forged and set(np.unique(auth).tolist()) <= {0, 1}
    step = 256 << 20
    for lo in range(misalign, total, step):
        m = min(step, total - lo)
        assert d_out.download(m, lo) == d_in.download(m, lo), lo
    for d in (d_in, d_out):
        d.free()
    return ctx


@pytest.mark.slow
@pytest.mark.parametrize("klen", [32, 16])
def test_u_shaped_lengths_up_to_64k_in_one_call(hip, orc, klen):
    """2^18 lengths from betavariate(.1, .1) x 65535 (the reference's `short` traffic scaled to its counter's 16-bit length field, config/gcm_utils.py:144), MACsec-
    sized headers as AAD, packed back to back from an odd byte address: 8 GiB in one call, about 40 % of the messages below the 8 KiB mark"""
    rng = random.Random(6000 + klen)
    n = 1 << 18
    lens = u_shaped(rng, n, 65535)
    aads = [rng.choice((0, 13, 20, 28, 68)) for _ in range(n)]
    _mixed_call(hip, orc, klen, lens, aads, 61000 + klen, misalign=5)


@pytest.mark.slow
def test_u_shaped_lengths_up_to_1m_in_one_call(hip, orc):
    """lengths from betavariate(.1, .1) x 2^20: 2^14 of them (8 GiB again -- 2^18 would be 128 GiB, more than the host side of the check can hold), no AAD array entries
    for most, some AADs longer than a row"""
    rng = random.Random(6100)
    n = 1 << 14
    lens = u_shaped(rng, n, 1 << 20)
    aads = [rng.choice((0, 0, 0, 20, 1040, 5000)) for _ in range(n)]
    _mixed_call(hip, orc, 32, lens, aads, 62000)


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("klen,n,top", [(16, 3000, 20000), (24, 40000, 12000), (32, 300, 300000), (32, 20000, 2100), (16, 150000, 3000)])
def test_small_mixed_calls(hip, orc, klen, n, top, split):
    """fewer messages -- by the library's own rule (the low mark, 2 KiB, unless 65536 messages lie between the marks; everything by rows while the short messages hold
    fewer than 2^21 blocks between them) and with the split forced (context options): calls whose short messages are a handful, or all of them"""
    rng = random.Random(6200 + n)
    lens = u_shaped(rng, n, top)
    aads = [rng.choice((0, 0, 13, 20, 28, 1024)) for _ in range(n)]
    _mixed_call(hip, orc, klen, lens, aads, 63000 + n, misalign=rng.choice((0, 3, 16)), opts=SPLIT if split else None)


@pytest.mark.parametrize("lanes", [1, 4, 8, 16, 64])
def test_every_packet_kernel_shape_takes_the_short_half(hip, orc, lanes):
    """the debug library forces the packet kernel shape (which also sends everything to the packet kernels), or everything by rows: the same bytes every way"""
    rng = random.Random(6300 + lanes)
    n = 5000
    lens = u_shaped(rng, n, 9000)
    aads = [rng.choice((0, 8, 20)) for _ in range(n)]
    with hip.debug_library() as dbg:
        dbg.force(pkt_lanes=lanes)
        _mixed_call(hip, orc, 32, lens, aads, 64000 + lanes)
    if lanes == 1:
        with hip.debug_library() as dbg:
            dbg.force(pkt_rows=1)
            _mixed_call(hip, orc, 32, lens, aads, 64100)


@pytest.mark.parametrize("klen", [16, 32])
def test_messages_wherever_they_live_are_routed_too(hip, orc, klen):
    """aesgcm_messages_crypt_dev: 20 000 messages of U-shaped length in buffers of their own (one arena with gaps, outputs elsewhere at other alignments), AAD
    likewise; the short ones take the packet kernels, which read the same address and length arrays -- every message against the restatement of the RTL for the
    short ones and a sample of the long ones; decrypt to a third place with forged tags and wipe_on_auth_fail"""
    rng = random.Random(6400 + klen)
    n = 20000
    lens = u_shaped(rng, n, 40000)
    aads = [rng.choice((0, 0, 13, 20, 200)) for _ in range(n)]
    key = splitmix_bytes(6500 + klen, klen)
    f = orc.Fast(key)
    gap_in = [rng.choice((0, 1, 16, 29)) for _ in range(n)]
    gap_out = [rng.choice((0, 7, 16, 48)) for _ in range(n)]
    pos_in, pos_out, pos_aad, a, b, c = [], [], [], 0, 0, 0
    for k in range(n):
        a += gap_in[k]; b += gap_out[k]
        pos_in.append(a); pos_out.append(b); pos_aad.append(c)
        a += lens[k]; b += lens[k]; c += aads[k]
    d_in, d_out, d_back, d_aad = hip.DeviceBuffer(a + 64), hip.DeviceBuffer(b + 64), hip.DeviceBuffer(a + 64), hip.DeviceBuffer(c + 64)
    d_in.fill_splitmix64(6600, nbytes=(a + 64) // 8 * 8)
    d_aad.fill_splitmix64(6601, nbytes=(c + 64) // 8 * 8)
    ivs = splitmix_bytes(6602 + klen, 12 * n)
    u64s = lambda v: _up(hip, struct.pack("<%dQ" % n, *v))
    u32s = lambda v: _up(hip, struct.pack("<%dI" % n, *v))
    d_ivs, d_len, d_alen = _up(hip, ivs), u32s(lens), u32s(aads)
    d_inp, d_outp, d_backp, d_aadp = u64s([d_in.ptr + x for x in pos_in]), u64s([d_out.ptr + x for x in pos_out]), u64s([d_back.ptr + x for x in pos_in]), u64s([d_aad.ptr + x for x in pos_aad])
    d_tags, d_auth = hip.DeviceBuffer(16 * n), hip.DeviceBuffer(4 * n)
    ctx = hip.Context(key).set_option("route_blocks_min", 0).set_option("route_mid_min", 0)      # (the library's own rule would send a call this small by rows altogether)
    ctx.messages_crypt_dev(False, n, d_ivs.ptr, d_inp.ptr, d_len.ptr, d_outp.ptr, d_tags.ptr, d_aad_ptr=d_aadp.ptr, d_aad_len=d_alen.ptr)
    hip.dev_sync()
    assert ctx.status() == (hip.STATUS_OK, 0)
    tags = bytes(d_tags.download())
    pt_all, ct_all, aad_all = bytes(d_in.download()), bytes(d_out.download()), bytes(d_aad.download())
    budget = 96 << 20
    for k in sorted(range(n), key=lambda k: lens[k]):
        if lens[k] > budget:
            break
        budget -= lens[k]
        want = f.encrypt(ivs[12 * k:12 * k + 12], aad_all[pos_aad[k]:pos_aad[k] + aads[k]], pt_all[pos_in[k]:pos_in[k] + lens[k]])
        assert (ct_all[pos_out[k]:pos_out[k] + lens[k]], tags[16 * k:16 * k + 16]) == want, (k, lens[k], aads[k])
    forged = sorted(set([0, n - 1] + rng.sample(range(n), 12)))
    bad = bytearray(tags)
    for k in forged:
        bad[16 * k + 9] ^= 4
    d_exp, d_t2 = _up(hip, bytes(bad)), hip.DeviceBuffer(16 * n)
    ctx.set_option("wipe_on_auth_fail", 1)
    ctx.messages_crypt_dev(True, n, d_ivs.ptr, d_outp.ptr, d_len.ptr, d_backp.ptr, d_t2.ptr, d_aad_ptr=d_aadp.ptr, d_aad_len=d_alen.ptr, d_expect_tags=d_exp.ptr, d_auth=d_auth.ptr)
    hip.dev_sync()
    assert bytes(d_t2.download()) == tags                        # every tag, the long messages' too: decrypt computes them over the same ciphertext
    auth = struct.unpack("<%di" % n, bytes(d_auth.download()))
    assert [k for k in range(n) if not auth[k]] == forged
    back = bytes(d_back.download())
    for k in range(n):
        assert back[pos_in[k]:pos_in[k] + lens[k]] == (bytes(lens[k]) if k in forged else pt_all[pos_in[k]:pos_in[k] + lens[k]]), k


def test_scattered_messages_a_lane_each_read_the_launchs_records(hip, orc):
    """a lane per packet (forced: the shape of calls of 2^18 messages and more) takes its packets from the 48-byte records the sort writes in the launch's order (round 6,
    PktDesc) -- here for messages wherever they live: 4000 of them at odd addresses, outputs elsewhere, AAD of 0 .. 40 bytes, every one against the restatement of the RTL"""
    rng = random.Random(7400)
    n = 4000
    lens = [rng.choice((0, 1, 15, 16, 17, 63, 64, 65, rng.randrange(0, 1600), rng.randrange(0, 1600), rng.randrange(1500, 5000))) for _ in range(n)]
    aads = [rng.choice((0, 0, 1, 16, 28, 40)) for _ in range(n)]
    key = splitmix_bytes(7500, 32)
    f = orc.Fast(key)
    pos_in, pos_out, pos_aad, a, b, c = [], [], [], 0, 0, 0
    for k in range(n):
        a += rng.choice((0, 1, 3, 16)); b += rng.choice((0, 5, 16, 64))
        pos_in.append(a); pos_out.append(b); pos_aad.append(c)
        a += lens[k]; b += lens[k]; c += aads[k]
    d_in, d_out, d_aad = hip.DeviceBuffer(a + 64), hip.DeviceBuffer(b + 64), hip.DeviceBuffer(c + 64)
    d_in.fill_splitmix64(7600, nbytes=(a + 64) // 8 * 8)
    d_aad.fill_splitmix64(7601, nbytes=(c + 64) // 8 * 8)
    ivs = splitmix_bytes(7602, 12 * n)
    u64s = lambda v: _up(hip, struct.pack("<%dQ" % n, *v))
    u32s = lambda v: _up(hip, struct.pack("<%dI" % n, *v))
    d_ivs, d_len, d_alen = _up(hip, ivs), u32s(lens), u32s(aads)
    d_inp, d_outp, d_aadp = u64s([d_in.ptr + x for x in pos_in]), u64s([d_out.ptr + x for x in pos_out]), u64s([d_aad.ptr + x for x in pos_aad])
    d_tags, d_t2, d_auth = hip.DeviceBuffer(16 * n), hip.DeviceBuffer(16 * n), hip.DeviceBuffer(4 * n)
    pt_all, aad_all = bytes(d_in.download()), bytes(d_aad.download())
    with hip.debug_library() as dbg:
        dbg.force(pkt_lanes=1)
        ctx = hip.Context(key)
        ctx.messages_crypt_dev(False, n, d_ivs.ptr, d_inp.ptr, d_len.ptr, d_outp.ptr, d_tags.ptr, d_aad_ptr=d_aadp.ptr, d_aad_len=d_alen.ptr)
        hip.dev_sync()
        assert ctx.status() == (hip.STATUS_OK, 0) and ctx.last_route()["lanes"] == 1
        tags, ct_all = bytes(d_tags.download()), bytes(d_out.download())
        for k in range(n):
            want = f.encrypt(ivs[12 * k:12 * k + 12], aad_all[pos_aad[k]:pos_aad[k] + aads[k]], pt_all[pos_in[k]:pos_in[k] + lens[k]])
            assert (ct_all[pos_out[k]:pos_out[k] + lens[k]], tags[16 * k:16 * k + 16]) == want, (k, lens[k], aads[k])
        forged = sorted(set([0, n - 1] + rng.sample(range(n), 9)))
        bad = bytearray(tags)
        for k in forged:
            bad[16 * k + 3] ^= 0x40
        d_exp = _up(hip, bytes(bad))
        ctx.messages_crypt_dev(True, n, d_ivs.ptr, d_outp.ptr, d_len.ptr, d_outp.ptr, d_t2.ptr, d_aad_ptr=d_aadp.ptr, d_aad_len=d_alen.ptr, d_expect_tags=d_exp.ptr, d_auth=d_auth.ptr)
        hip.dev_sync()
        assert bytes(d_t2.download()) == tags
        auth = struct.unpack("<%di" % n, bytes(d_auth.download()))
        assert [k for k in range(n) if not auth[k]] == forged
        back = bytes(d_out.download())
        for k in range(n):
            assert back[pos_out[k]:pos_out[k] + lens[k]] == pt_all[pos_in[k]:pos_in[k] + lens[k]], k
        ctx.close()


def test_a_length_the_call_cannot_take_is_reported_not_truncated(hip, orc):
    """round 5 cast device-side lengths to 32 bits and ran on garbage.  Now: a length of 2^28 in d_len, a 2^29 gap in d_data_off, offsets that fall -- the plan
    kernel refuses the whole call: outputs, tags and verdicts stay as they were, aesgcm_ctx_status names the first such message (AESGCM_STATUS_LENGTH), and
    aesgcm_last_tag / aesgcm_ctx_wait answer AESGCM_ETOOLONG until the status has been read (the RTL raises a flag when its counter cannot go on,
    src/aes_icb.vhd:65,98,114,119).  The context works as before afterwards."""
    key = splitmix_bytes(6700, 32)
    f = orc.Fast(key)
    rng = random.Random(6701)
    for n in (100, 6000):                                        # the plan of one workgroup, and the plan of five launches
        lens = [rng.choice((0, 100, 1500, 3000, 20000)) for _ in range(n)]
        doff = [0]
        for x in lens:
            doff.append(doff[-1] + x)
        ivs, pt = splitmix_bytes(6702, 12 * n), splitmix_bytes(6703, doff[-1])
        want = [f.encrypt(ivs[12 * p:12 * p + 12], b"", pt[doff[p]:doff[p + 1]]) for p in range(0, n, 7)]
        d_ivs, d_in = _up(hip, ivs), _up(hip, pt)
        ctx, other = hip.Context(key), hip.Context(key)
        at = n // 2 + 3
        cases = []
        gap = list(doff)
        for k in range(at + 1, n + 1):
            gap[k] += 1 << 29                                    # message `at` seems to be 2^29 bytes longer
        cases.append(("gap", dict(d_data_off=gap), at))
        fall = list(doff)
        fall[at + 1] = fall[at] - 1 if fall[at] else 0
        if fall[at + 1] < fall[at]:
            cases.append(("falling", dict(d_data_off=fall), at))
        big_len = list(lens)
        big_len[at] = 1 << 28
        cases.append(("len", dict(lens=big_len), at))
        for name, kw, first_bad in cases:
            d_out, d_tags, d_auth = hip.DeviceBuffer(doff[-1] + 16), hip.DeviceBuffer(16 * n), hip.DeviceBuffer(4 * n)
            mark = bytes([0xA5]) * (doff[-1] + 16)
            d_out.upload(mark); d_tags.upload(bytes([0x5A]) * (16 * n)); d_auth.upload(bytes([0x77]) * (4 * n))
            if "lens" in kw:
                d_len = _up(hip, struct.pack("<%dI" % n, *kw["lens"]))
                d_inp, d_outp = _up(hip, struct.pack("<%dQ" % n, *[d_in.ptr + x for x in doff[:-1]])), _up(hip, struct.pack("<%dQ" % n, *[d_out.ptr + x for x in doff[:-1]]))
                ctx.messages_crypt_dev(True, n, d_ivs.ptr, d_inp.ptr, d_len.ptr, d_outp.ptr, d_tags.ptr, d_expect_tags=d_tags.ptr, d_auth=d_auth.ptr)
            else:
                d_off = _up(hip, struct.pack("<%dQ" % (n + 1), *kw["d_data_off"]))
                ctx.packets_crypt_dev(True, n, d_ivs.ptr, d_in.ptr, d_out.ptr, d_tags.ptr, d_data_off=d_off.ptr, d_expect_tags=d_tags.ptr, d_auth=d_auth.ptr)
            hip.dev_sync()
            assert bytes(d_out.download()) == mark and bytes(d_tags.download()) == bytes([0x5A]) * (16 * n) and bytes(d_auth.download()) == bytes([0x77]) * (4 * n), (n, name)
            for call in (ctx.last_tag, lambda: other.wait(ctx), lambda: ctx.wait(other)):
                with pytest.raises(hip.AesGcmError) as ei:
                    call()
                assert ei.value.code == hip.ETOOLONG, (n, name)
            assert ctx.status() == (hip.STATUS_LENGTH, first_bad), (n, name)
            assert ctx.status() == (hip.STATUS_OK, 0)
            other.wait(ctx)
            # the same context, the true offsets: everything as it should be
            d_off = _up(hip, struct.pack("<%dQ" % (n + 1), *doff))
            ctx.packets_crypt_dev(False, n, d_ivs.ptr, d_in.ptr, d_out.ptr, d_tags.ptr, d_data_off=d_off.ptr)
            hip.dev_sync()
            assert ctx.status() == (hip.STATUS_OK, 0)
            ct, tags = bytes(d_out.download(doff[-1])), bytes(d_tags.download())
            for i, p in enumerate(range(0, n, 7)):
                assert (ct[doff[p]:doff[p + 1]], tags[16 * p:16 * p + 16]) == want[i], (n, name, p)


@pytest.mark.gpu
@pytest.mark.parametrize("klen", [16, 32])
def test_a_full_band_above_the_high_mark_moves_the_mark_to_the_last_class(hip, orc, klen):
    """option route_top_min (458 752 in the library: profiles/r06/route_band.txt) brought down to 1000: a call with that many messages of 8 .. 16 KiB sends them to the
    packet kernels as well -- the mark becomes 16 320 bytes, the last size the sort resolves -- and only what is longer goes by rows; bytes and tags as ever"""
    rng = random.Random(777 + klen)
    n = 3000
    lens = [rng.choice((rng.randrange(8192, 16320), rng.randrange(8192, 16320), rng.randrange(0, 3000), rng.randrange(16320, 40000))) for _ in range(n)]
    aads = [rng.choice((0, 13, 28)) for _ in range(n)]
    ctx = _mixed_call(hip, orc, klen, lens, aads, 88000 + klen, misalign=5, opts=dict(route_top_min=1000, route_blocks_min=0))
    r = ctx.last_route()
    assert r["route_min"] == 16320 and r["n_small"] == sum(1 for a, b in zip(lens, aads) if a + b < 16320), r
    ctx.close()
    ctx = _mixed_call(hip, orc, klen, lens[:900], aads[:900], 88100 + klen, opts=dict(route_top_min=1000, route_blocks_min=0))      # fewer than that in the band: the usual marks
    assert ctx.last_route()["route_min"] in (2048, 8192)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("n", [1, 2, 63, 65, 257, 2049, 4097, 16384, 16385])
def test_counts_at_the_edges_of_the_sort_and_lengths_at_the_marks(hip, orc, n, split):
    """the counting sort -- one launch of one workgroup up to 16 384 messages, three launches with 256 slices beyond -- at counts that leave most slices empty or one over, with
    lengths on and beside everything that decides: a block, the two routing marks (2 KiB, 8 KiB), the last size class (16 320 bytes), and messages of no bytes --
    with and without AAD, so that data + AAD straddles a mark where data alone does not"""
    rng = random.Random(9000 + n)
    pool = [0, 0, 1, 15, 16, 17, 63, 64, 65, 2019, 2020, 2047, 2048, 2049, 8163, 8164, 8191, 8192, 8193, 16319, 16320, 16383, 16384, 16385, 70001]
    lens = [pool[(p * 7 + rng.randrange(3)) % len(pool)] for p in range(n)]
    aads = [rng.choice([0, 0, 1, 16, 28, 29]) for _ in range(n)]
    if n > 2:
        lens[1], aads[1] = 0, 0                                   # a message of nothing at all
    _mixed_call(hip, orc, 32 if n % 2 else 16, lens, aads, 9100 + n, misalign=n % 16, opts=SPLIT if split else None).close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,max_len", [(3000, 1514), (2000, 40000)])
def test_a_routed_call_is_capture_safe(n, max_len):
    """examples/graph_replay: after one ordinary call, aesgcm_packets_crypt_dev with offset arrays is captured into a hipGraph on the caller's stream (the context's
    side stream joins the capture through its fork / join events), and the graph -- replayed over OTHER lengths than at capture time, the route being the device's --
    gives the bytes and tags of a direct call: the enqueue path allocates nothing, waits for nothing and reads nothing back (include/aesgcm.h "capture")"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "examples"), "-s", "graph_replay"], check=True)
    r = subprocess.run([os.path.join(root, "examples", "graph_replay"), str(n), "5", str(max_len)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0 and "GRAPH REPLAY OK" in r.stdout, (r.stdout, r.stderr)
    assert '"equal": true' in r.stdout
