"""GPU, GiB scale: BASELINE.json's full sizes.  Ciphertext is checked bit-for-bit through SHA-256 of the
whole stream (downloaded in 256 MiB slices) against fixtures produced by libcrypto over the same SplitMix64
plaintext (tests/golden/gen_golden.py --large), the tag against the same fixture; decryption is checked by
its tag (GHASH over the input) plus plaintext windows against the generator, and by size-independent
properties (in-place == out-of-place, sharded == whole)."""
import hashlib
import random

import pytest

from util import golden, stream_key_iv

pytestmark = [pytest.mark.gpu, pytest.mark.slow]
SLICE = 256 << 20


def _case(name):
    for c in golden("streams.json")["cases"]:
        if c["name"] == name:
            return c
    pytest.skip("fixture %s not generated" % name)


def _sha_device(buf, n):
    sha = hashlib.sha256()
    host = bytearray(SLICE)
    for off in range(0, n, SLICE):
        m = min(SLICE, n - off)
        view = memoryview(host)[:m]
        buf.download(m, off, out=view)
        sha.update(view)
    return sha.hexdigest()


@pytest.mark.parametrize("name", ["cfg2_aes128_1GiB", "aes192_1GiB", "aes256_1GiB", "cfg3_aes256_16GiB"])
def test_full_size_stream_bit_exact(hip, orc, name):
    """AES-128 / 192 / 256 at sizes where the production split engages by itself (the dealt k_body<10 | 12 | 14, *> with FoldClose from 1 GiB:
    src/aes_pkg.vhd:31-33 treats the three modes alike, so does this)"""
    c = _case(name)
    n = c["n_bytes"]
    key, iv = stream_key_iv(c)
    ctx = hip.Context(key)
    head, body = ctx.split(n)
    assert head == 0 and body == n // 16, (head, body)                      # nobody forced it: the library's own cut, whole message = one aligned body
    pt, ct = hip.DeviceBuffer(n), hip.DeviceBuffer(n)
    pt.fill_splitmix64(c["pt_seed"], c["first_word"])
    tag = ctx.encrypt_dev(iv, pt.ptr, n, ct.ptr)
    assert tag.hex() == c["tag"]
    assert bytes(ct.download(64, 0)).hex() == c["ct_head"] and bytes(ct.download(64, n - 64)).hex() == c["ct_tail"]
    assert _sha_device(ct, n) == c["ct_sha256"]
    # decrypt into the plaintext buffer (overwriting it), authenticated against the fixture tag
    t2 = ctx.decrypt_dev(iv, ct.ptr, n, pt.ptr, tag=bytes.fromhex(c["tag"]))
    assert t2 == tag
    rng = random.Random(n)
    win = 1 << 20
    for off in [0, n - win] + [rng.randrange(0, n - win) // 16 * 16 for _ in range(8)]:
        assert bytes(pt.download(win, off)) == bytes(orc.fill_splitmix64(win, c["pt_seed"], c["first_word"] + off // 8)), off
    # in place == out of place
    t3 = ctx.encrypt_dev(iv, pt.ptr, n, pt.ptr)
    assert t3 == tag
    for off in [0, n - win, (n // 2) // 16 * 16]:
        assert bytes(pt.download(win, off)) == bytes(ct.download(win, off))
    pt.free(); ct.free()


def test_general_path_4GiB_with_aad_and_ragged_end(hip, orc):
    """4 GiB - 5 bytes with 20 bytes of AAD: the general path of a large message -- head (AAD) through k_main, the aligned middle through the dealt
    k_body, the ragged tail through k_main, the GHASH state carried between the pieces on the device -- at a size nobody forces; then decrypt in
    place, authenticated, and the plaintext back against the generator"""
    c = _case("aes256_4GiB_aad20_minus5")
    n = c["n_bytes"]
    assert n == (4 << 30) - 5
    key, iv = stream_key_iv(c)
    aad = bytes.fromhex(c["aad"])
    ctx = hip.Context(key)
    head, body = ctx.split(n)
    assert body > 0 and 16 * body < n                                        # a middle and a tail
    buf = hip.DeviceBuffer(n + 16)
    buf.fill_splitmix64(c["pt_seed"], c["first_word"], nbytes=(n + 7) // 8 * 8)
    d_aad = hip.DeviceBuffer(len(aad)); d_aad.upload(aad)
    out = hip.DeviceBuffer(n + 16)
    tag = ctx.encrypt_dev(iv, buf.ptr, n, out.ptr, d_aad=d_aad.ptr, aad_len=len(aad))
    assert tag.hex() == c["tag"]
    assert bytes(out.download(64, 0)).hex() == c["ct_head"] and bytes(out.download(64, n - 64)).hex() == c["ct_tail"]
    assert _sha_device(out, n) == c["ct_sha256"]
    t2 = ctx.decrypt_dev(iv, out.ptr, n, out.ptr, d_aad=d_aad.ptr, aad_len=len(aad), tag=bytes.fromhex(c["tag"]))     # in place
    assert t2 == tag
    win = 1 << 20
    for off in (0, (n - win) // 16 * 16, (n // 3) // 16 * 16):
        m = min(win, n - off)
        assert bytes(out.download(m, off)) == bytes(orc.fill_splitmix64(m, c["pt_seed"], c["first_word"] + off // 8)), off
    with pytest.raises(hip.AuthenticationError):
        ctx.decrypt_dev(iv, buf.ptr, n, out.ptr, d_aad=d_aad.ptr, aad_len=len(aad), tag=bytes.fromhex(c["tag"]))      # the plaintext is not the ciphertext
    buf.free(); out.free()


def test_sharded_16GiB_equals_fixture(hip):
    """cfg3 stream cut into 8 shards on one GPU (the multi-GPU algebra at full size): same tag, same CT."""
    c = _case("cfg3_aes256_16GiB")
    n = c["n_bytes"]
    key, iv = stream_key_iv(c)
    ctx = hip.Context(key)
    pt, ct = hip.DeviceBuffer(n), hip.DeviceBuffer(n)
    pt.fill_splitmix64(c["pt_seed"], c["first_word"])
    import aesgcm_amd  # noqa: F401
    from aesgcm_amd import sharding
    ranks = 8
    parts = hip.DeviceBuffer(16 * ranks)
    for r in range(ranks):
        first, end, ln = sharding.shard_bounds(n, ranks, r)
        ctx.shard_crypt_dev(False, iv, pt.ptr + 16 * first, ln, ct.ptr + 16 * first, first, n, parts.ptr + 16 * r)
    assert ctx.shard_finalize_dev(iv, parts.ptr, ranks, 0, n).hex() == c["tag"]
    assert _sha_device(ct, n) == c["ct_sha256"]
    pt.free(); ct.free()


@pytest.mark.parametrize("msg", [0, 1, 2, 3])
def test_cfg4_messages_sharded_8_equal_fixtures(hip, msg):
    """BASELINE config 4 at its own size on ONE GPU: the 128 GiB job is 4 messages x 32 GiB (one GCM message cannot
    exceed 2^36 - 32 bytes, src/aes_icb.vhd:114), each cut into 8 shards of 4 GiB exactly as the 8 ranks would own
    them (sharding.shard_bounds).  Here the 8 shards run one after the other, IN PLACE (32 GiB resident), through the
    same aesgcm_shard_crypt_dev / aesgcm_shard_finalize_dev calls; tag, CT head/tail and SHA-256 of all 32 GiB of
    ciphertext must equal the libcrypto fixture."""
    c = _case("cfg4_aes256_msg%d_32GiB" % msg)
    n = c["n_bytes"]
    assert n == 32 << 30
    key, iv = stream_key_iv(c)
    import aesgcm_amd  # noqa: F401
    from aesgcm_amd import sharding
    assert iv == sharding.tweak_iv(sharding.splitmix64_bytes(c["iv_seed"], 12), msg)     # the IV rule bench.py uses for N > 1
    ctx = hip.Context(key)
    buf = hip.DeviceBuffer(n)
    buf.fill_splitmix64(c["pt_seed"], c["first_word"])
    ranks = 8
    parts = hip.DeviceBuffer(16 * ranks)
    for r in range(ranks):
        first, end, ln = sharding.shard_bounds(n, ranks, r)
        assert ln == 4 << 30
        ctx.shard_crypt_dev(False, iv, buf.ptr + 16 * first, ln, buf.ptr + 16 * first, first, n, parts.ptr + 16 * r)
    assert ctx.shard_finalize_dev(iv, parts.ptr, ranks, 0, n).hex() == c["tag"]
    assert bytes(buf.download(64, 0)).hex() == c["ct_head"] and bytes(buf.download(64, n - 64)).hex() == c["ct_tail"]
    assert _sha_device(buf, n) == c["ct_sha256"]
    buf.free(); parts.free()


def test_maximum_message_and_length_limit(hip, orc):
    """2^36 - 32 bytes: the largest message the 32-bit block counter allows (src/aes_icb.vhd:114), in place,
    with a 20-byte AAD; one byte more must be refused."""
    c = _case("aes256_max_message")
    n = c["n_bytes"]
    assert n == (1 << 36) - 32
    key, iv = stream_key_iv(c)
    aad = bytes.fromhex(c["aad"])
    ctx = hip.Context(key)
    buf = hip.DeviceBuffer(n + 64)
    buf.fill_splitmix64(c["pt_seed"], c["first_word"], nbytes=n)
    d_aad = hip.DeviceBuffer(len(aad)); d_aad.upload(aad)
    tag = ctx.encrypt_dev(iv, buf.ptr, n, buf.ptr, d_aad=d_aad.ptr, aad_len=len(aad))
    assert tag.hex() == c["tag"]
    assert bytes(buf.download(64, 0)).hex() == c["ct_head"] and bytes(buf.download(64, n - 64)).hex() == c["ct_tail"]
    # ciphertext windows against keystream from the oracle (counter values near 2^32 included)
    f = orc.Fast(key)
    win = 1 << 16
    for off in (0, (1 << 35) - win, n - win + 0):
        off = off // 16 * 16
        m = min(win, n - off)
        pt = bytes(orc.fill_splitmix64(m, c["pt_seed"], c["first_word"] + off // 8))
        ks = f.keystream(iv, off // 16, (m + 15) // 16)[:m]
        assert bytes(buf.download(m, off)) == bytes(a ^ b for a, b in zip(pt, ks)), off
    with pytest.raises(hip.AesGcmError) as e:
        ctx.encrypt_dev(iv, buf.ptr, n + 1, buf.ptr)
    assert e.value.code == hip.ETOOLONG
    buf.free()


def test_large_unaligned_aad(hip, orc):
    """256 MiB of AAD from a misaligned device pointer + a short ragged message: the AAD rides the same chunked
    polynomial evaluation as the data."""
    key, iv = bytes(range(32)), bytes(range(12))
    ctx = hip.Context(key)
    n_aad = (256 << 20) + 5
    d = hip.DeviceBuffer(n_aad + 64)
    d.fill_splitmix64(0xA11)
    import numpy as np
    host = np.frombuffer(d.download(), dtype=np.uint8)
    aad = host[3:3 + n_aad]
    pt = bytes(orc.fill_splitmix64(1000, 0xB22))
    d_pt, d_ct = hip.DeviceBuffer(1024), hip.DeviceBuffer(1024)
    d_pt.upload(pt)
    tag = ctx.encrypt_dev(iv, d_pt.ptr, len(pt), d_ct.ptr, d_aad=d.ptr + 3, aad_len=n_aad)
    f = orc.Fast(key)
    f.begin(iv); f.aad(aad); want_ct = bytes(f.update(pt)); want_tag = f.final()
    assert tag == want_tag and bytes(d_ct.download(len(pt))) == want_ct
