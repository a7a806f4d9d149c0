"""GPU: pipelined host-buffer path (H2D || kernel || D2H, GHASH state carried across chunks) == one-shot."""
import pytest

from util import splitmix_bytes

pytestmark = pytest.mark.gpu


def test_pipelined_equals_oracle(hip, orc):
    key, iv = splitmix_bytes(21, 32), splitmix_bytes(22, 12)
    c = hip.Context(key)
    f = orc.Fast(key)
    for n, aad_len, chunk in ((0, 0, 0), (5, 20, 0), (1 << 20, 0, 1 << 18), ((3 << 20) + 77, 4095, 1 << 20),
                              ((8 << 20) - 16, 16, 3 << 20), (1000, 0, 1024), ((1 << 20) + 1, 33, 64 << 10)):
        aad, pt = splitmix_bytes(23 + n, aad_len), splitmix_bytes(24 + n, n)
        want = f.encrypt(iv, aad, pt)
        assert c.encrypt_pipelined(iv, aad, pt, chunk_bytes=chunk) == want, (n, aad_len, chunk)
        back, tag = c.decrypt_pipelined(iv, aad, want[0], tag=want[1], chunk_bytes=chunk)
        assert back == pt and tag == want[1]
    with pytest.raises(hip.AuthenticationError):
        c.decrypt_pipelined(iv, b"", want[0], tag=want[1])


def test_pipelined_pinned_buffers_large(hip, orc):
    import numpy as np
    n = 192 << 20
    key, iv = splitmix_bytes(31, 16), splitmix_bytes(32, 12)
    src, dst = hip.PinnedBuffer(n), hip.PinnedBuffer(n)
    np.frombuffer(src.view, dtype=np.uint8)[:] = np.frombuffer(orc.fill_splitmix64(n, 33), dtype=np.uint8)
    c = hip.Context(key)
    _, tag = c.encrypt_pipelined(iv, b"hdr", src.view, out=dst.view, chunk_bytes=32 << 20)
    want_ct = np.empty(n, dtype=np.uint8)
    f = orc.Fast(key)
    _, want_tag = f.crypt(False, iv, b"hdr", np.frombuffer(src.view, dtype=np.uint8), want_ct)
    assert tag == want_tag
    assert np.array_equal(np.frombuffer(dst.view, dtype=np.uint8), want_ct)
    src.free(); dst.free()


def test_pipelined_refused_inside_a_streaming_session(hip, orc):
    """the pipelined path carries its GHASH value in the streaming slot: inside stream_begin .. stream_final it must
    return AESGCM_ESTATE and leave the session's running GHASH intact"""
    key, iv = splitmix_bytes(41, 32), splitmix_bytes(42, 12)
    c = hip.Context(key)
    f = orc.Fast(key)
    aad, pt = splitmix_bytes(43, 20), splitmix_bytes(44, 4096 + 16)
    c.stream_begin(iv)
    c.stream_aad(aad)
    out = c.stream_update(pt[:4096])
    with pytest.raises(hip.AesGcmError) as e:
        c.encrypt_pipelined(iv, b"", pt)
    assert e.value.code == hip.ESTATE
    out += c.stream_update(pt[4096:])
    tag = c.stream_final()
    assert (out, tag) == f.encrypt(iv, aad, pt)
    assert c.encrypt_pipelined(iv, aad, pt) == f.encrypt(iv, aad, pt)        # and works again once the session is closed


def test_device_buffer_bounds(hip):
    import numpy as np
    b = hip.DeviceBuffer(64)
    b.upload(bytes(range(64)))
    assert bytes(b.download(16, 48)) == bytes(range(48, 64))
    for args in ((32, 48), (65, 0), (1, 64)):
        with pytest.raises(hip.AesGcmError) as e:
            b.download(*args)
        assert e.value.code == hip.EARG
    with pytest.raises(hip.AesGcmError):
        b.download(32, 0, out=bytearray(16))                                  # output smaller than requested
    ro = np.zeros(32, dtype=np.uint8); ro.flags.writeable = False
    with pytest.raises(TypeError):
        b.download(32, 0, out=ro)
    with pytest.raises(hip.AesGcmError):
        b.fill_splitmix64(1, nbytes=65)
    b.free()
