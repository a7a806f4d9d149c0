"""GPU: batch path (BASELINE config 5) -- independent packets with per-packet key and IV, on-GPU aes_kexp."""
import hashlib
import random

import pytest

from util import golden, batch_inputs, splitmix_bytes

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["default", "lanes16", "lanes8", "lanes64"])
def batch_shape(request, hip):
    """the batch tests run with the host's own choice of kernel shape (the product library) and with each shape forced: 16 lanes per packet (k_batch3: four
    packets per wave, GHASH fused into the CTR loop), 8 lanes per packet (eight packets per wave, one table slot each, every multiply split over a lane pair) and
    64 lanes per packet (k_batch3<.., 6>; the two-pass k_batch of rounds 2 - 3 is gone).  Forcing a shape is a function of the debug build only (libaesgcm_hip_dbg.so, include/aesgcm_debug.h)."""
    lanes = {"lanes16": 16, "lanes8": 8, "lanes64": 64}.get(request.param)
    if lanes is None:
        yield request.param
        return
    with hip.debug_library() as dbg:
        dbg.force(batch_lanes=lanes)
        yield request.param


PKT_LANES = {"wave": 64, "group16": 16, "g8": 8, "g4": 4, "lane": 1}


def _run(hip, decrypt, keys, ivs, data, pkt_len, key_len, aad=b"", aad_len=0, expect=None, inplace=False):
    n = len(ivs) // 12
    d_keys, d_ivs = hip.DeviceBuffer(max(len(keys), 16)), hip.DeviceBuffer(max(len(ivs), 16))
    d_keys.upload(keys); d_ivs.upload(ivs)
    d_in = hip.DeviceBuffer(max(len(data), 16)); d_in.upload(data)
    d_out = d_in if inplace else hip.DeviceBuffer(max(len(data), 16))
    d_tags = hip.DeviceBuffer(16 * n)
    d_aad = None
    if aad_len:
        d_aad = hip.DeviceBuffer(len(aad)); d_aad.upload(aad)
    d_exp = d_auth = None
    if expect is not None:
        d_exp = hip.DeviceBuffer(16 * n); d_exp.upload(expect)
        d_auth = hip.DeviceBuffer(4 * n)
    hip.batch_crypt_dev(decrypt, n, key_len, d_keys.ptr, d_ivs.ptr, d_in.ptr, pkt_len, d_out.ptr, d_tags.ptr,
                        d_aad=d_aad.ptr if d_aad else None, aad_len=aad_len,
                        d_expect_tags=d_exp.ptr if d_exp else None, d_auth=d_auth.ptr if d_auth else None)
    hip.dev_sync()
    out = bytes(d_out.download(len(data))) if data else b""
    tags = bytes(d_tags.download(16 * n))
    auth = None
    if d_auth is not None:
        raw = bytes(d_auth.download(4 * n))
        auth = [int.from_bytes(raw[4 * i:4 * i + 4], "little") for i in range(n)]
    return out, tags, auth


def test_cfg5_first_64_packets_match_fixture(hip, batch_shape):
    fx = golden("batch.json")
    keys, ivs, pt = batch_inputs(0, 64, 4096)
    ct, tags, _ = _run(hip, False, keys, ivs, pt, 4096, 16)
    assert [tags[16 * p:16 * p + 16].hex() for p in range(64)] == fx["first64_tags"]
    assert hashlib.sha256(ct).hexdigest() == fx["first64_ct_sha256"]
    # decrypt in place, authenticated
    back, tags2, auth = _run(hip, True, keys, ivs, ct, 4096, 16, expect=tags, inplace=True)
    assert back == pt and tags2 == tags and auth == [1] * 64


def test_batch_shapes_vs_oracle(hip, orc, batch_shape):
    rng = random.Random(5)
    for klen, pkt_len, aad_len, n in ((16, 0, 0, 3), (16, 1, 0, 5), (24, 15, 7, 9), (32, 16, 16, 17), (16, 48, 28, 33), (32, 1000, 20, 40),
                                      (24, 1024, 0, 70), (16, 4096, 13, 130), (32, 9001, 68, 21), (16, 65536, 0, 6)):
        keys = splitmix_bytes(100 + pkt_len, klen * n)
        ivs = splitmix_bytes(200 + pkt_len, 12 * n)
        aad = splitmix_bytes(300 + pkt_len, aad_len * n)
        pt = splitmix_bytes(400 + pkt_len, pkt_len * n)
        ct, tags, _ = _run(hip, False, keys, ivs, pt, pkt_len, klen, aad=aad, aad_len=aad_len)
        for p in range(n):
            want_ct, want_tag = orc.Fast(keys[klen * p:klen * (p + 1)]).encrypt(ivs[12 * p:12 * p + 12], aad[aad_len * p:aad_len * (p + 1)], pt[pkt_len * p:pkt_len * (p + 1)])
            assert ct[pkt_len * p:pkt_len * (p + 1)] == want_ct, (klen, pkt_len, aad_len, p)
            assert tags[16 * p:16 * p + 16] == want_tag, (klen, pkt_len, aad_len, p)
        # tamper one packet's tag and one packet's ciphertext: exactly those two fail authentication
        if n >= 3 and pkt_len:
            bad = bytearray(tags); bad[16 * 1] ^= 1
            ctb = bytearray(ct); ctb[pkt_len * 2] ^= 0x80
            back, _, auth = _run(hip, True, keys, ivs, bytes(ctb), pkt_len, klen, aad=aad, aad_len=aad_len, expect=bytes(bad))
            assert auth == [0 if p in (1, 2) else 1 for p in range(n)]
            assert back[:pkt_len] == pt[:pkt_len]


@pytest.mark.slow
def test_cfg5_full_million_packets(hip, batch_shape):
    fx = golden("batch.json")
    if "full_tags_sha256" not in fx:
        pytest.skip("full batch fixture not generated")
    n, pkt = fx["full_n_pkts"], 4096
    d_keys, d_ivw, d_ivs = hip.DeviceBuffer(16 * n), hip.DeviceBuffer(16 * n), hip.DeviceBuffer(12 * n)
    d_keys.fill_splitmix64(0x4B4559)
    d_ivw.fill_splitmix64(0x4956)
    ivw = bytes(d_ivw.download())
    d_ivs.upload(b"".join(ivw[16 * p:16 * p + 12] for p in range(n)))
    d_pt, d_ct, d_tags = hip.DeviceBuffer(pkt * n), hip.DeviceBuffer(pkt * n), hip.DeviceBuffer(16 * n)
    d_pt.fill_splitmix64(0xAE5C0005)
    hip.batch_crypt_dev(False, n, 16, d_keys.ptr, d_ivs.ptr, d_pt.ptr, pkt, d_ct.ptr, d_tags.ptr)
    hip.dev_sync()
    assert hashlib.sha256(bytes(d_tags.download())).hexdigest() == fx["full_tags_sha256"]
    sha = hashlib.sha256()
    step = 256 << 20
    for off in range(0, pkt * n, step):
        sha.update(d_ct.download(min(step, pkt * n - off), off))
    assert sha.hexdigest() == fx["full_ct_sha256"]


def test_variable_length_packets_macsec_shaped(hip, orc, batch_shape):
    """Per-packet lengths and AAD through offset arrays: frames of 0..1600 bytes with 0..40-byte headers,
    including the reference's two README vectors as packets 0 and 1."""
    import struct
    rng = random.Random(99)
    kat = {v["name"]: v for v in golden("kat.json")["vectors"]}
    n = 200
    klen = 16
    pkts = []
    v0 = kat["readme_251_aes128"]
    pkts.append((bytes.fromhex(v0["key"]), bytes.fromhex(v0["iv"]), bytes.fromhex(v0["aad"]), bytes.fromhex(v0["pt"])))
    for i in range(1, n):
        al, pl = rng.choice((0, 8, 20, 28, 37, 40)), rng.choice((0, 1, 15, 16, 46, 60, 64, 128, 333, 1024, 1500, 1600))
        pkts.append((splitmix_bytes(5000 + i, klen), splitmix_bytes(6000 + i, 12), splitmix_bytes(7000 + i, al), splitmix_bytes(8000 + i, pl)))
    keys = b"".join(p[0] for p in pkts); ivs = b"".join(p[1] for p in pkts)
    aad = b"".join(p[2] for p in pkts); data = b"".join(p[3] for p in pkts)
    aoff, doff = [0], [0]
    for p in pkts:
        aoff.append(aoff[-1] + len(p[2])); doff.append(doff[-1] + len(p[3]))
    def up(b):
        d = hip.DeviceBuffer(max(len(b), 16)); d.upload(b); return d
    d_keys, d_ivs, d_aad, d_in = up(keys), up(ivs), up(aad), up(data)
    d_aoff, d_doff = up(struct.pack("<%dQ" % (n + 1), *aoff)), up(struct.pack("<%dQ" % (n + 1), *doff))
    d_out, d_tags = hip.DeviceBuffer(max(len(data), 16)), hip.DeviceBuffer(16 * n)
    hip.batch_crypt_var_dev(False, n, klen, d_keys.ptr, d_ivs.ptr, d_in.ptr, d_doff.ptr, d_out.ptr, d_tags.ptr, d_aad=d_aad.ptr, d_aad_off=d_aoff.ptr)
    hip.dev_sync()
    ct, tags = bytes(d_out.download(len(data))), bytes(d_tags.download())
    assert tags[:16].hex().upper() == "4F8D55E7D3F06FD5A13C0C29B9D5B880" and ct[:48].hex() == v0["ct"]
    for i, p in enumerate(pkts):
        want_ct, want_tag = orc.Fast(p[0]).encrypt(p[1], p[2], p[3])
        assert ct[doff[i]:doff[i + 1]] == want_ct and tags[16 * i:16 * i + 16] == want_tag, i
    # decrypt in place with verification
    d_auth = hip.DeviceBuffer(4 * n)
    d_t2 = hip.DeviceBuffer(16 * n)
    hip.batch_crypt_var_dev(True, n, klen, d_keys.ptr, d_ivs.ptr, d_out.ptr, d_doff.ptr, d_out.ptr, d_t2.ptr, d_aad=d_aad.ptr, d_aad_off=d_aoff.ptr,
                            d_expect_tags=d_tags.ptr, d_auth=d_auth.ptr)
    hip.dev_sync()
    assert bytes(d_out.download(len(data))) == data
    assert set(struct.unpack("<%di" % n, bytes(d_auth.download()))) == {1}


@pytest.mark.parametrize("shape", ["wave", "group16", "g8", "g4", "lane", "lane_ilp"])
def test_packets_under_one_key(hip, orc, shape):
    with hip.debug_library() as dbg:
        # one lane per packet has two forms: 768-lane workgroups with one keystream chain per lane, and (for batches that do not fill the chip) 512-lane
        # workgroups with four chains side by side; a batch of this size would take the second by the library's own rule
        dbg.force(pkt_lanes=PKT_LANES[shape.split("_")[0]], pkt_ilp={"lane": 2, "lane_ilp": 1}.get(shape, 0))
        _packets_under_one_key(hip, orc)


def _packets_under_one_key(hip, orc):
    """aesgcm_packets_crypt_dev: one key (context), per-packet IV, AAD and length; fixed-size records and offset
    arrays; decrypt in place with per-packet authentication.  The three kernel shapes (one wave per packet and 16 lanes per
    packet: k_pktg<.., 6> / k_pktg<.., 4>; one lane per packet: k_pktl) are forced in turn; the library picks between them by
    packet count and size otherwise."""
    import struct
    rng = random.Random(4242)
    for klen in (16, 24, 32):
        key = splitmix_bytes(300 + klen, klen)
        ctx = hip.Context(key)
        f = orc.Fast(key)
        # fixed-size records (cfg5-shaped, but one key)
        n, pkt, al = 300, 4096, 20
        ivs, aad, pt = splitmix_bytes(31, 12 * n), splitmix_bytes(32, al * n), splitmix_bytes(33, pkt * n)
        def up(b):
            d = hip.DeviceBuffer(max(len(b), 16)); d.upload(b); return d
        d_ivs, d_aad, d_in = up(ivs), up(aad), up(pt)
        d_out, d_tags = hip.DeviceBuffer(pkt * n), hip.DeviceBuffer(16 * n)
        ctx.packets_crypt_dev(False, n, d_ivs.ptr, d_in.ptr, d_out.ptr, d_tags.ptr, pkt_len=pkt, d_aad=d_aad.ptr, aad_len=al)
        hip.dev_sync()
        ct, tags = bytes(d_out.download()), bytes(d_tags.download())
        for p in range(n):
            want = f.encrypt(ivs[12 * p:12 * p + 12], aad[al * p:al * (p + 1)], pt[pkt * p:pkt * (p + 1)])
            assert (ct[pkt * p:pkt * (p + 1)], tags[16 * p:16 * p + 16]) == want, (klen, p)
        # variable lengths through offset arrays, incl. empty packets and empty AAD
        lens = [rng.choice((0, 1, 15, 16, 17, 46, 64, 1000, 1024, 1500, 4096, 9000, 70000)) for _ in range(120)]
        aads = [rng.choice((0, 0, 8, 20, 28, 40, 100)) for _ in range(120)]
        m = len(lens)
        doff, aoff = [0], [0]
        for a, b in zip(lens, aads):
            doff.append(doff[-1] + a); aoff.append(aoff[-1] + b)
        ivs, aad, pt = splitmix_bytes(41, 12 * m), splitmix_bytes(42, aoff[-1]), splitmix_bytes(43, doff[-1])
        d_ivs, d_aad, d_buf = up(ivs), up(aad), up(pt)
        d_doff, d_aoff = up(struct.pack("<%dQ" % (m + 1), *doff)), up(struct.pack("<%dQ" % (m + 1), *aoff))
        d_tags, d_auth = hip.DeviceBuffer(16 * m), hip.DeviceBuffer(4 * m)
        ctx.packets_crypt_dev(False, m, d_ivs.ptr, d_buf.ptr, d_buf.ptr, d_tags.ptr, d_data_off=d_doff.ptr, d_aad=d_aad.ptr, d_aad_off=d_aoff.ptr)
        hip.dev_sync()
        ct, tags = bytes(d_buf.download(doff[-1])), bytes(d_tags.download())
        for p in range(m):
            want = f.encrypt(ivs[12 * p:12 * p + 12], aad[aoff[p]:aoff[p + 1]], pt[doff[p]:doff[p + 1]])
            assert (ct[doff[p]:doff[p + 1]], tags[16 * p:16 * p + 16]) == want, (klen, p, lens[p], aads[p])
        bad = bytearray(tags); bad[16 * 5 + 3] ^= 4
        d_exp, d_t2 = up(bytes(bad)), hip.DeviceBuffer(16 * m)
        ctx.packets_crypt_dev(True, m, d_ivs.ptr, d_buf.ptr, d_buf.ptr, d_t2.ptr, d_data_off=d_doff.ptr, d_aad=d_aad.ptr, d_aad_off=d_aoff.ptr,
                              d_expect_tags=d_exp.ptr, d_auth=d_auth.ptr)
        hip.dev_sync()
        assert bytes(d_buf.download(doff[-1])) == pt
        auth = struct.unpack("<%di" % m, bytes(d_auth.download()))
        assert [i for i, a in enumerate(auth) if not a] == [5]


@pytest.mark.parametrize("shape", [None, "wave", "group16", "g8", "g4", "lane"])
def test_fixed_size_records_of_odd_lengths(hip, orc, shape):
    if shape is None:
        return _fixed_size_records_of_odd_lengths(hip, orc)
    with hip.debug_library() as dbg:
        dbg.force(pkt_lanes=PKT_LANES[shape])
        _fixed_size_records_of_odd_lengths(hip, orc)


def _fixed_size_records_of_odd_lengths(hip, orc):
    """aesgcm_packets_crypt_dev with fixed-size records that are empty, shorter than a block, ragged, or not a multiple of 4 bytes apart (the
    byte-wise load / store paths), with and without AAD, in every kernel shape and in the host's own choice; decrypt in place"""
    import struct
    key = splitmix_bytes(610, 24)
    ctx, f = hip.Context(key), orc.Fast(key)
    def up(b):
        d = hip.DeviceBuffer(max(len(b), 16)); d.upload(b); return d
    for n, pkt, al in ((5, 0, 0), (70, 0, 12), (130, 1, 0), (67, 17, 20), (200, 1000, 0), (90, 1500, 28), (33, 4099, 1), (300, 64, 16)):
        ivs, aad, pt = splitmix_bytes(611 + pkt, 12 * n), splitmix_bytes(612 + pkt, al * n), splitmix_bytes(613 + pkt, pkt * n)
        d_ivs, d_aad, d_buf = up(ivs), up(aad), up(pt)
        d_tags, d_auth = hip.DeviceBuffer(16 * n), hip.DeviceBuffer(4 * n)
        ctx.packets_crypt_dev(False, n, d_ivs.ptr, d_buf.ptr, d_buf.ptr, d_tags.ptr, pkt_len=pkt, d_aad=d_aad.ptr if al else None, aad_len=al)
        hip.dev_sync()
        ct, tags = bytes(d_buf.download(pkt * n)) if pkt else b"", bytes(d_tags.download())
        for p in range(n):
            want = f.encrypt(ivs[12 * p:12 * p + 12], aad[al * p:al * (p + 1)], pt[pkt * p:pkt * (p + 1)])
            assert (ct[pkt * p:pkt * (p + 1)], tags[16 * p:16 * p + 16]) == want, (shape, n, pkt, al, p)
        d_exp = up(tags)
        ctx.packets_crypt_dev(True, n, d_ivs.ptr, d_buf.ptr, d_buf.ptr, d_tags.ptr, pkt_len=pkt, d_aad=d_aad.ptr if al else None, aad_len=al,
                              d_expect_tags=d_exp.ptr, d_auth=d_auth.ptr)
        hip.dev_sync()
        assert (bytes(d_buf.download(pkt * n)) if pkt else b"") == pt
        assert set(struct.unpack("<%di" % n, bytes(d_auth.download()))) == {1}


@pytest.mark.gpu
def test_many_small_packets_default_shape(hip, orc):
    """50 000 MACsec-sized frames (0..1514 B, AAD 0..32 B) under one key: the count makes the library choose the
    lane-per-packet kernel by itself; every ciphertext and tag is compared with the oracle, then decrypted in place."""
    import struct
    rng = random.Random(777)
    m = 50000
    lens = [rng.choice((0, 46, 64, 128, 256, 512, 1000, 1500, 1514, rng.randrange(0, 1515))) for _ in range(m)]
    aads = [rng.choice((0, 8, 16, 20, 28, 32)) for _ in range(m)]
    doff, aoff = [0], [0]
    for a, b in zip(lens, aads):
        doff.append(doff[-1] + a); aoff.append(aoff[-1] + b)
    key = splitmix_bytes(901, 32)
    ctx, f = hip.Context(key), orc.Fast(key)
    ivs, aad, pt = splitmix_bytes(51, 12 * m), splitmix_bytes(52, aoff[-1]), splitmix_bytes(53, doff[-1])
    def up(b):
        d = hip.DeviceBuffer(max(len(b), 16)); d.upload(b); return d
    d_ivs, d_aad, d_buf = up(ivs), up(aad), up(pt)
    d_doff, d_aoff = up(struct.pack("<%dQ" % (m + 1), *doff)), up(struct.pack("<%dQ" % (m + 1), *aoff))
    d_tags, d_auth = hip.DeviceBuffer(16 * m), hip.DeviceBuffer(4 * m)
    ctx.packets_crypt_dev(False, m, d_ivs.ptr, d_buf.ptr, d_buf.ptr, d_tags.ptr, d_data_off=d_doff.ptr, d_aad=d_aad.ptr, d_aad_off=d_aoff.ptr)
    hip.dev_sync()
    ct, tags = bytes(d_buf.download(doff[-1])), bytes(d_tags.download())
    for p in range(m):
        want = f.encrypt(ivs[12 * p:12 * p + 12], aad[aoff[p]:aoff[p + 1]], pt[doff[p]:doff[p + 1]])
        assert (ct[doff[p]:doff[p + 1]], tags[16 * p:16 * p + 16]) == want, (p, lens[p], aads[p])
    bad = bytearray(tags); bad[16 * 4321] ^= 0x80; bad[16 * 49999 + 15] ^= 1
    d_exp = up(bytes(bad))
    ctx.packets_crypt_dev(True, m, d_ivs.ptr, d_buf.ptr, d_buf.ptr, d_tags.ptr, d_data_off=d_doff.ptr, d_aad=d_aad.ptr, d_aad_off=d_aoff.ptr,
                          d_expect_tags=d_exp.ptr, d_auth=d_auth.ptr)
    hip.dev_sync()
    assert bytes(d_buf.download(doff[-1])) == pt
    auth = struct.unpack("<%di" % m, bytes(d_auth.download()))
    assert [i for i, a in enumerate(auth) if not a] == [4321, 49999]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [None, "group16", "g8", "g4", "lane"])
def test_mixed_length_packets_taken_by_length_class(hip, orc, shape):
    """6 000 frames of 0 .. 9 KiB through offset arrays, taken in the order of a counting sort by length class (k_len_hist / k_len_scan / k_len_scatter;
    the context option pkt_order is the packet count from which the library does that by itself: here 1) and in array order (0).  Every kernel shape, both orders: the same ciphertext and tags, all against the
    oracle; then decrypt in place with two forged tags."""
    import struct
    rng = random.Random(20261)
    m = 6000
    lens = [rng.choice((0, 16, 46, 64, 64, 128, 500, 1000, 1500, 1514, 4096, 9000, rng.randrange(0, 1515), rng.randrange(0, 1515))) for _ in range(m)]
    aads = [rng.choice((0, 0, 8, 20, 28)) for _ in range(m)]
    doff, aoff = [0], [0]
    for a, b in zip(lens, aads):
        doff.append(doff[-1] + a); aoff.append(aoff[-1] + b)
    key = splitmix_bytes(902, 32)
    f = orc.Fast(key)
    ivs, aad, pt = splitmix_bytes(61, 12 * m), splitmix_bytes(62, max(aoff[-1], 16)), splitmix_bytes(63, doff[-1])
    want = [f.encrypt(ivs[12 * p:12 * p + 12], aad[aoff[p]:aoff[p + 1]], pt[doff[p]:doff[p + 1]]) for p in range(m)]

    def up(b):
        d = hip.DeviceBuffer(max(len(b), 16)); d.upload(b); return d

    def run():
        for order in (1, 0):
            ctx = hip.Context(key).set_option("pkt_order", order)
            d_ivs, d_aad, d_in, d_out = up(ivs), up(aad), up(pt), hip.DeviceBuffer(doff[-1] + 16)
            d_doff, d_aoff = up(struct.pack("<%dQ" % (m + 1), *doff)), up(struct.pack("<%dQ" % (m + 1), *aoff))
            d_tags, d_auth = hip.DeviceBuffer(16 * m), hip.DeviceBuffer(4 * m)
            ctx.packets_crypt_dev(False, m, d_ivs.ptr, d_in.ptr, d_out.ptr, d_tags.ptr, d_data_off=d_doff.ptr, d_aad=d_aad.ptr, d_aad_off=d_aoff.ptr)
            hip.dev_sync()
            ct, tags = bytes(d_out.download(doff[-1])), bytes(d_tags.download())
            for p in range(m):
                assert (ct[doff[p]:doff[p + 1]], tags[16 * p:16 * p + 16]) == want[p], (shape, order, p, lens[p], aads[p])
            bad = bytearray(tags); bad[16 * 17] ^= 1; bad[16 * 5999 + 15] ^= 0x80
            d_exp = up(bytes(bad))
            ctx.packets_crypt_dev(True, m, d_ivs.ptr, d_out.ptr, d_out.ptr, d_tags.ptr, d_data_off=d_doff.ptr, d_aad=d_aad.ptr, d_aad_off=d_aoff.ptr,
                                  d_expect_tags=d_exp.ptr, d_auth=d_auth.ptr)
            hip.dev_sync()
            assert bytes(d_out.download(doff[-1])) == pt, (shape, order)
            auth = struct.unpack("<%di" % m, bytes(d_auth.download()))
            assert [i for i, a in enumerate(auth) if not a] == [17, 5999], (shape, order)

    if shape is None:
        return run()
    with hip.debug_library() as dbg:
        dbg.force(pkt_lanes=PKT_LANES[shape])
        run()


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [8, 16])
def test_variable_length_batch_taken_by_length_class(hip, orc, lanes):
    """3 000 packets with a key each and lengths of 0 .. 5 KiB through offset arrays: k_batch3 takes them in the order of the counting sort by length class
    (forced: the library does it by itself from 262144 / 98304 packets) and in array order -- the same bytes, against the oracle; decrypt in place with a forged tag."""
    import struct
    rng = random.Random(31337)
    n, klen = 3000, 16
    lens = [rng.choice((0, 16, 64, 64, 100, 500, 1000, 1500, 1514, 4096, 5000, rng.randrange(0, 1515))) for _ in range(n)]
    aads = [rng.choice((0, 0, 8, 20, 28)) for _ in range(n)]
    doff, aoff = [0], [0]
    for a, b in zip(lens, aads):
        doff.append(doff[-1] + a); aoff.append(aoff[-1] + b)
    keys, ivs, aad, pt = splitmix_bytes(71, klen * n), splitmix_bytes(72, 12 * n), splitmix_bytes(73, max(aoff[-1], 16)), splitmix_bytes(74, doff[-1])
    want = [orc.Fast(keys[klen * p:klen * (p + 1)]).encrypt(ivs[12 * p:12 * p + 12], aad[aoff[p]:aoff[p + 1]], pt[doff[p]:doff[p + 1]]) for p in range(n)]

    def up(b):
        d = hip.DeviceBuffer(max(len(b), 16)); d.upload(b); return d
    with hip.debug_library() as dbg:
        for order in (1, 2):                                     # always / never
            dbg.force(batch_lanes=lanes, batch_order=order)
            d_keys, d_ivs, d_aad, d_in, d_out = up(keys), up(ivs), up(aad), up(pt), hip.DeviceBuffer(doff[-1] + 16)
            d_doff, d_aoff = up(struct.pack("<%dQ" % (n + 1), *doff)), up(struct.pack("<%dQ" % (n + 1), *aoff))
            d_tags, d_auth = hip.DeviceBuffer(16 * n), hip.DeviceBuffer(4 * n)
            hip.batch_crypt_var_dev(False, n, klen, d_keys.ptr, d_ivs.ptr, d_in.ptr, d_doff.ptr, d_out.ptr, d_tags.ptr, d_aad=d_aad.ptr, d_aad_off=d_aoff.ptr)
            hip.dev_sync()
            ct, tags = bytes(d_out.download(doff[-1])), bytes(d_tags.download())
            for p in range(n):
                assert (ct[doff[p]:doff[p + 1]], tags[16 * p:16 * p + 16]) == want[p], (lanes, order, p, lens[p], aads[p])
            bad = bytearray(tags); bad[16 * 2999 + 7] ^= 2
            d_exp = up(bytes(bad))
            hip.batch_crypt_var_dev(True, n, klen, d_keys.ptr, d_ivs.ptr, d_out.ptr, d_doff.ptr, d_out.ptr, d_tags.ptr, d_aad=d_aad.ptr, d_aad_off=d_aoff.ptr,
                                    d_expect_tags=d_exp.ptr, d_auth=d_auth.ptr)
            hip.dev_sync()
            assert bytes(d_out.download(doff[-1])) == pt, (lanes, order)
            auth = struct.unpack("<%di" % n, bytes(d_auth.download()))
            assert [i for i, a in enumerate(auth) if not a] == [2999], (lanes, order)


@pytest.mark.gpu
def test_ordered_launches_back_to_back_reuse_their_scratch(hip, orc):
    """The four scratch slots of the launch order are reused while earlier launches may still be running: ten launches over packets of mixed length queued
    without a wait in between, each with its own offsets -- through a context on its stream (launches of ONE context share its dispenser and must stay on one
    stream), and through the context-free batch entry point on two streams in turn (a dispenser per launch: those may overlap).  Tags against the oracle."""
    import struct
    rng = random.Random(8086)
    key = splitmix_bytes(903, 16)
    ctx = hip.Context(key).set_option("pkt_order", 1)
    s1, s2 = hip.Context(key), hip.Context(key)            # kept alive: their streams carry the batch launches
    f = orc.Fast(key)
    m = 3000
    runs = []
    for r in range(20):
        lens = [rng.choice((0, 16, 64, 200, 700, 1500, 1514, 3000, rng.randrange(0, 1515))) for _ in range(m)]
        doff = [0]
        for a in lens:
            doff.append(doff[-1] + a)
        ivs, pt = splitmix_bytes(700 + r, 12 * m), splitmix_bytes(800 + r, doff[-1])
        d = {"doff": doff, "ivs": ivs, "pt": pt}
        for k, b in (("d_ivs", ivs), ("d_in", pt), ("d_doff", struct.pack("<%dQ" % (m + 1), *doff)), ("d_keys", key * m)):
            d[k] = hip.DeviceBuffer(max(len(b), 16)); d[k].upload(b)
        d["d_out"], d["d_tags"] = hip.DeviceBuffer(doff[-1] + 16), hip.DeviceBuffer(16 * m)
        runs.append(d)
    hip.dev_sync()
    with hip.debug_library() as dbg:
        dbg.force(batch_order=1)
        for r, d in enumerate(runs):
            if r < 10:
                ctx.packets_crypt_dev(False, m, d["d_ivs"].ptr, d["d_in"].ptr, d["d_out"].ptr, d["d_tags"].ptr, d_data_off=d["d_doff"].ptr)
            else:
                hip.batch_crypt_var_dev(False, m, 16, d["d_keys"].ptr, d["d_ivs"].ptr, d["d_in"].ptr, d["d_doff"].ptr, d["d_out"].ptr, d["d_tags"].ptr,
                                        stream=(s1 if r % 2 else s2).stream())
        hip.dev_sync()
    for r, d in enumerate(runs):
        ct, tags, doff = bytes(d["d_out"].download(d["doff"][-1])), bytes(d["d_tags"].download()), d["doff"]
        for p in range(0, m, 7):
            want = f.encrypt(d["ivs"][12 * p:12 * p + 12], b"", d["pt"][doff[p]:doff[p + 1]])
            assert (ct[doff[p]:doff[p + 1]], tags[16 * p:16 * p + 16]) == want, (r, p)
