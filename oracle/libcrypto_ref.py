"""External AES-GCM cross-check: system libcrypto (OpenSSL) EVP_aes_{128,192,256}_gcm through ctypes.

TEST INFRASTRUCTURE ONLY.  The reference's software model (tb/gcm_model.py:18-44) delegates to
pycryptodome's AES.MODE_GCM; pycryptodome is not installable in this image, so the golden vectors
(tests/golden/gen_golden.py) and the large-stream checks use an independent conforming build of the
same published algorithm instead.  If pycryptodome IS importable at run time, `best()` prefers it,
because that is the exact call the reference makes.
"""
import ctypes
import ctypes.util

_lc = None


def _load():
    global _lc
    if _lc is not None:
        return _lc
    name = ctypes.util.find_library("crypto") or "libcrypto.so.3"
    L = ctypes.CDLL(name)
    vp = ctypes.c_void_p
    L.EVP_CIPHER_CTX_new.restype = vp
    L.EVP_CIPHER_CTX_free.argtypes = [vp]
    for nm in ("EVP_aes_128_gcm", "EVP_aes_192_gcm", "EVP_aes_256_gcm"):
        getattr(L, nm).restype = vp
    L.EVP_CipherInit_ex.argtypes = [vp, vp, vp, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
    L.EVP_CIPHER_CTX_ctrl.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp]
    L.EVP_CipherUpdate.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_int), vp, ctypes.c_int]
    L.EVP_CipherFinal_ex.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_int)]
    L.OpenSSL_version.restype = ctypes.c_char_p
    L.OpenSSL_version.argtypes = [ctypes.c_int]
    _lc = L
    return L


def available():
    try:
        _load()
        return True
    except OSError:
        return False


def version():
    return _load().OpenSSL_version(0).decode()


_EVP_CTRL_GCM_SET_IVLEN = 0x9
_EVP_CTRL_GCM_GET_TAG = 0x10
_EVP_CTRL_GCM_SET_TAG = 0x11


def _addr(buf):
    if buf is None:
        return None
    if isinstance(buf, bytes):
        return ctypes.cast(ctypes.c_char_p(buf), ctypes.c_void_p).value if buf else None
    if isinstance(buf, bytearray):
        return ctypes.addressof((ctypes.c_char * len(buf)).from_buffer(buf)) if buf else None
    if hasattr(buf, "ctypes"):
        return buf.ctypes.data
    return ctypes.addressof(buf)


class Stream:
    """Streaming AES-GCM (encrypt or decrypt) over libcrypto; chunks of any size < 2 GiB."""

    def __init__(self, key: bytes, iv: bytes, dec: bool = False):
        L = _load()
        self.L = L
        self.dec = dec
        ciph = {16: L.EVP_aes_128_gcm, 24: L.EVP_aes_192_gcm, 32: L.EVP_aes_256_gcm}[len(key)]()
        self.ctx = L.EVP_CIPHER_CTX_new()
        enc = 0 if dec else 1
        assert L.EVP_CipherInit_ex(self.ctx, ciph, None, None, None, enc) == 1
        assert L.EVP_CIPHER_CTX_ctrl(self.ctx, _EVP_CTRL_GCM_SET_IVLEN, len(iv), None) == 1
        assert L.EVP_CipherInit_ex(self.ctx, None, None, key, iv, enc) == 1

    def aad(self, aad):
        if len(aad):
            n = ctypes.c_int(0)
            assert self.L.EVP_CipherUpdate(self.ctx, None, ctypes.byref(n), _addr(aad), len(aad)) == 1

    def update(self, data, out=None):
        n_in = len(data)
        if out is None:
            out = bytearray(n_in)
        off = 0
        step = 1 << 30
        n = ctypes.c_int(0)
        base_in, base_out = _addr(data), _addr(out)
        while off < n_in:
            m = min(step, n_in - off)
            assert self.L.EVP_CipherUpdate(self.ctx, base_out + off, ctypes.byref(n), base_in + off, m) == 1
            off += m
        return out

    def final(self, expect_tag: bytes = None):
        """enc: -> tag.  dec: sets expect_tag and returns True/False for verification."""
        n = ctypes.c_int(0)
        scratch = ctypes.create_string_buffer(16)
        if self.dec:
            ok = True
            if expect_tag is not None:
                tb = ctypes.create_string_buffer(bytes(expect_tag), 16)
                assert self.L.EVP_CIPHER_CTX_ctrl(self.ctx, _EVP_CTRL_GCM_SET_TAG, 16, ctypes.addressof(tb)) == 1
                ok = self.L.EVP_CipherFinal_ex(self.ctx, scratch, ctypes.byref(n)) == 1
            self.close()
            return ok
        assert self.L.EVP_CipherFinal_ex(self.ctx, scratch, ctypes.byref(n)) == 1
        tag = ctypes.create_string_buffer(16)
        assert self.L.EVP_CIPHER_CTX_ctrl(self.ctx, _EVP_CTRL_GCM_GET_TAG, 16, ctypes.addressof(tag)) == 1
        self.close()
        return tag.raw

    def close(self):
        if self.ctx:
            self.L.EVP_CIPHER_CTX_free(self.ctx)
            self.ctx = None

    __del__ = close


def encrypt(key, iv, aad, pt):
    s = Stream(key, iv, False)
    s.aad(aad)
    ct = s.update(pt)
    return bytes(ct), s.final()


def decrypt(key, iv, aad, ct, tag):
    """-> (pt, ok)"""
    s = Stream(key, iv, True)
    s.aad(aad)
    pt = s.update(ct)
    return bytes(pt), s.final(tag)


def best():
    """Name + one-shot encrypt callable of the strongest available external AES-GCM:
    pycryptodome (the reference's own dependency, tb/gcm_model.py:1) if importable, else libcrypto."""
    try:
        from Crypto.Cipher import AES  # noqa
        import Crypto

        def enc(key, iv, aad, pt):
            m = AES.new(key, mode=AES.MODE_GCM, nonce=iv)
            if len(aad):
                m.update(aad)
            ct = m.encrypt(bytes(pt))
            return ct, m.digest()

        return "pycryptodome " + Crypto.__version__, enc
    except Exception:
        return version(), encrypt
