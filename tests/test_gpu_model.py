"""GPU: the gcm_model drop-in driven the way the reference harness drives tb/gcm_model.py --
16-byte beats through monitor callbacks, scoreboard lists, tamper convention."""
import pytest

from util import golden, splitmix_bytes

pytestmark = pytest.mark.gpu


def _hexdict(b):
    return {'data': b.hex().upper(), 'n_bytes': len(b)}


def _beats(b):
    return [b[i:i + 16] for i in range(0, len(b), 16)]


def drive(gcm_model, key, iv, aad, data, ed, dut_tag=None):
    """What tb/gcm_test.py:45,76-94 does: construct, feed AAD beats, feed data beats, then the tag."""
    m = gcm_model.gcm(_hexdict(key), _hexdict(iv), ed)
    for beat in _beats(aad):
        m.load_aad(beat)
    for beat in _beats(data):
        (m.load_plain_text if ed == 'enc' else m.load_cipher_text)(beat)
    m.get_tag(dut_tag if dut_tag is not None else bytes(16))
    return m


def test_readme_vectors_beat_by_beat(hip):
    from aesgcm_amd import gcm_model
    for v in golden("kat.json")["vectors"]:
        if not v["name"].startswith("readme"):
            continue
        key, iv, aad, pt, ct, tag = (bytes.fromhex(v[k]) for k in ("key", "iv", "aad", "pt", "ct", "tag"))
        m = drive(gcm_model, key, iv, aad, pt, 'enc', dut_tag=tag)
        assert b"".join(m.data_out) == ct and m.tag == [tag]
        assert [len(x) for x in m.data_out] == [len(x) for x in _beats(pt)]     # one entry per call, in order
        d = drive(gcm_model, key, iv, aad, ct, 'dec', dut_tag=tag)
        assert b"".join(d.data_out) == pt and d.tag == [tag]


def test_streaming_equals_one_shot_and_oracle(hip, orc):
    from aesgcm_amd import gcm_model
    for it, (klen, al, pl) in enumerate(((16, 0, 0), (16, 5, 0), (24, 0, 7), (32, 33, 100), (32, 16, 16), (16, 68, 333), (24, 200, 1000))):
        key, iv = splitmix_bytes(500 + it, klen), splitmix_bytes(600 + it, 12)
        aad, pt = splitmix_bytes(700 + it, al), splitmix_bytes(800 + it, pl)
        want_ct, want_tag = orc.Fast(key).encrypt(iv, aad, pt)
        m = drive(gcm_model, key, iv, aad, pt, 'enc', dut_tag=want_tag)
        assert b"".join(m.data_out) == want_ct and m.tag == [want_tag]
        assert gcm_model.encrypt(key, iv, aad, pt) == (want_ct, want_tag)
        assert gcm_model.decrypt(key, iv, aad, want_ct, tag=want_tag) == (pt, want_tag)


def test_tamper_convention_inverted_tag(hip):
    """dec + wrong tag: the model appends the bit-inverted received tag (tb/gcm_model.py:49-51)."""
    from aesgcm_amd import gcm_model
    key, iv, aad, pt = splitmix_bytes(1, 16), splitmix_bytes(2, 12), splitmix_bytes(3, 20), splitmix_bytes(4, 64)
    ct, tag = gcm_model.encrypt(key, iv, aad, pt)
    bad = bytes([tag[0] ^ 0x01]) + tag[1:]
    d = drive(gcm_model, key, iv, aad, ct, 'dec', dut_tag=bad)
    assert b"".join(d.data_out) == pt
    assert d.tag == [bytes(b ^ 0xFF for b in bad)]
    ok = drive(gcm_model, key, iv, aad, ct, 'dec', dut_tag=tag)
    assert ok.tag == [tag]


def test_out_of_order_calls_raise_like_pycryptodome(hip):
    from aesgcm_amd import gcm_model
    m = gcm_model.gcm(_hexdict(bytes(16)), _hexdict(bytes(12)), 'enc')
    m.load_plain_text(bytes(16))
    with pytest.raises(TypeError):
        m.load_aad(b"late aad")
    # any chunking is legal, as with pycryptodome's encrypt(): ragged chunks in the middle included
    key, iv, aad = splitmix_bytes(21, 24), splitmix_bytes(22, 12), splitmix_bytes(23, 37)
    pt = splitmix_bytes(24, 200001)
    m2 = gcm_model.gcm(_hexdict(key), _hexdict(iv), 'enc')
    m2.load_aad(aad[:5]); m2.load_aad(aad[5:])
    o, sizes = 0, (5, 16, 1, 31, 70000, 64, 3)
    k = 0
    while o < len(pt):
        n = sizes[k % len(sizes)]; k += 1
        m2.load_plain_text(pt[o:o + n]); o += n
    m2.get_tag(bytes(16))
    want = gcm_model.encrypt(key, iv, aad, pt)
    assert (b"".join(m2.data_out), m2.tag[0]) == want
    with pytest.raises(TypeError):
        m2.load_plain_text(b"after the tag")


def test_chunked_stream_large_chunks(hip, orc):
    """State carry across calls with big chunks: 3 x 1 MiB + ragged tail."""
    key, iv, aad = splitmix_bytes(11, 32), splitmix_bytes(12, 12), splitmix_bytes(13, 4095)
    pt = splitmix_bytes(14, (3 << 20) + 77)
    c = hip.Context(key)
    c.stream_begin(iv)
    c.stream_aad(aad[:2048]); c.stream_aad(aad[2048:])
    out = b"".join(c.stream_update(pt[o:o + (1 << 20)]) for o in range(0, len(pt), 1 << 20))
    tag = c.stream_final()
    assert (out, tag) == orc.Fast(key).encrypt(iv, aad, pt)
