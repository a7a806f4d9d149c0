"""ctypes binding of oracle/liboracle.so -- the CPU restatement of the reference's arithmetic.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by the product package.  See the header of aesgcm_oracle.c for the parity status and
the reference file:line each routine follows.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    """Compile liboracle.so with gcc (plain C99)."""
    src = os.path.join(_HERE, "aesgcm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.orc_sbox.restype = ctypes.c_uint8
        L.orc_sbox.argtypes = [ctypes.c_uint8]
        L.orc_key_expand.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
        L.orc_aes_encrypt_block.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p]
        L.orc_gfmul.argtypes = [ctypes.c_char_p] * 3
        L.orc_gfmul.restype = None
        L.orc_ghash_update.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
        L.orc_ghash_update.restype = None
        L.orc_gcm_crypt.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p,
                                    ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                                    ctypes.c_char_p, ctypes.c_char_p]
        L.orc_gfpow.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_char_p]
        L.orc_gfpow.restype = None
        L.orc_ghash_poly.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
        L.orc_ghash_poly.restype = None
        L.orc_fill_splitmix64.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint64]
        L.orc_fill_splitmix64.restype = None
        L.orc_fast_new.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        L.orc_fast_new.restype = ctypes.c_void_p
        L.orc_fast_free.argtypes = [ctypes.c_void_p]
        L.orc_fast_free.restype = None
        L.orc_fast_encrypt_block.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p]
        L.orc_fast_encrypt_block.restype = None
        L.orc_fast_get_h.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        L.orc_fast_get_h.restype = None
        L.orc_fast_begin.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
        L.orc_fast_aad.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        L.orc_fast_update.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.orc_fast_final.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        L.orc_fast_crypt.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t,
                                     ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_char_p]
        L.orc_fast_keystream.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
        L.orc_fast_keystream.restype = None
        L.orc_fast_ghash_poly.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p]
        L.orc_fast_ghash_poly.restype = None
        _lib = L
    return _lib


def _addr(buf):
    """Address of a bytes / bytearray / numpy array / ctypes buffer without copying."""
    if buf is None:
        return None
    if isinstance(buf, (bytes, bytearray)):
        if len(buf) == 0:
            return None
        if isinstance(buf, bytes):
            return ctypes.cast(ctypes.c_char_p(buf), ctypes.c_void_p).value
        return ctypes.addressof((ctypes.c_char * len(buf)).from_buffer(buf))
    if hasattr(buf, "ctypes"):          # numpy
        return buf.ctypes.data
    return ctypes.addressof(buf)


# ---------------------------------------------------------------- literal layer
def sbox_table():
    L = lib()
    return [L.orc_sbox(x) for x in range(256)]


def key_expand(key: bytes):
    """-> (round_keys: bytes of 16*(nr+1), nr).  tb/key_exp.py:79-114."""
    rk = ctypes.create_string_buffer(240)
    nr = ctypes.c_int(0)
    rc = lib().orc_key_expand(key, len(key), rk, ctypes.byref(nr))
    if rc:
        raise ValueError("bad key length %d" % len(key))
    return rk.raw[: 16 * (nr.value + 1)], nr.value


def aes_encrypt_block(key: bytes, block: bytes) -> bytes:
    rk, nr = key_expand(key)
    out = ctypes.create_string_buffer(16)
    lib().orc_aes_encrypt_block(rk, nr, block, out)
    return out.raw


def gfmul(h: bytes, x: bytes) -> bytes:
    z = ctypes.create_string_buffer(16)
    lib().orc_gfmul(h, x, z)
    return z.raw


def gfpow(h: bytes, e: int) -> bytes:
    z = ctypes.create_string_buffer(16)
    lib().orc_gfpow(h, e, z)
    return z.raw


def ghash_update(h: bytes, y: bytes, data: bytes) -> bytes:
    yb = ctypes.create_string_buffer(bytes(y), 16)
    lib().orc_ghash_update(h, yb, data, len(data))
    return yb.raw


def ghash_poly(h: bytes, data: bytes) -> bytes:
    p = ctypes.create_string_buffer(16)
    lib().orc_ghash_poly(h, data, len(data), p)
    return p.raw


def gcm_encrypt(key, iv, aad, pt):
    """Literal (slow) whole-message path -> (ct, tag)."""
    out = ctypes.create_string_buffer(max(len(pt), 1))
    tag = ctypes.create_string_buffer(16)
    rc = lib().orc_gcm_crypt(0, key, len(key), iv, aad, len(aad), pt, len(pt), out, tag)
    if rc:
        raise ValueError("orc_gcm_crypt rc=%d" % rc)
    return out.raw[: len(pt)], tag.raw


def gcm_decrypt(key, iv, aad, ct):
    """-> (pt, computed_tag)."""
    out = ctypes.create_string_buffer(max(len(ct), 1))
    tag = ctypes.create_string_buffer(16)
    rc = lib().orc_gcm_crypt(1, key, len(key), iv, aad, len(aad), ct, len(ct), out, tag)
    if rc:
        raise ValueError("orc_gcm_crypt rc=%d" % rc)
    return out.raw[: len(ct)], tag.raw


def fill_splitmix64(n_bytes: int, seed: int, first_word: int = 0) -> bytearray:
    buf = bytearray(n_bytes)
    if n_bytes:
        lib().orc_fill_splitmix64(_addr(buf), n_bytes, seed, first_word)
    return buf


# ---------------------------------------------------------------- fast layer
class Fast:
    """Table-driven restatement (tables generated from the literal layer).  Streaming capable."""

    def __init__(self, key: bytes):
        self._c = lib().orc_fast_new(key, len(key))
        if not self._c:
            raise ValueError("bad key length %d" % len(key))

    def close(self):
        if self._c:
            lib().orc_fast_free(self._c)
            self._c = None

    __del__ = close

    @property
    def h(self) -> bytes:
        b = ctypes.create_string_buffer(16)
        lib().orc_fast_get_h(self._c, b)
        return b.raw

    def encrypt_block(self, block: bytes) -> bytes:
        out = ctypes.create_string_buffer(16)
        lib().orc_fast_encrypt_block(self._c, block, out)
        return out.raw

    def crypt(self, dec: bool, iv: bytes, aad, data, out=None):
        """One-shot.  data/out may be bytes, bytearray or numpy uint8 arrays.  -> (out, tag)."""
        n = len(data)
        if out is None:
            out = bytearray(n)
        tag = ctypes.create_string_buffer(16)
        rc = lib().orc_fast_crypt(self._c, int(dec), iv, _addr(aad), len(aad) if aad is not None else 0,
                                  _addr(data), n, _addr(out), tag)
        if rc:
            raise ValueError("orc_fast_crypt rc=%d" % rc)
        return out, tag.raw

    def encrypt(self, iv, aad, pt):
        out, tag = self.crypt(False, iv, aad, pt)
        return bytes(out), tag

    def decrypt(self, iv, aad, ct):
        out, tag = self.crypt(True, iv, aad, ct)
        return bytes(out), tag

    # streaming
    def begin(self, iv: bytes, dec: bool = False):
        lib().orc_fast_begin(self._c, iv, int(dec))

    def aad(self, aad):
        if lib().orc_fast_aad(self._c, _addr(aad), len(aad)):
            raise ValueError("aad after data / ragged aad chunk")

    def update(self, data, out=None):
        n = len(data)
        if out is None:
            out = bytearray(n)
        rc = lib().orc_fast_update(self._c, _addr(data), n, _addr(out))
        if rc:
            raise ValueError("orc_fast_update rc=%d" % rc)
        return out

    def final(self) -> bytes:
        tag = ctypes.create_string_buffer(16)
        lib().orc_fast_final(self._c, tag)
        return tag.raw

    def keystream(self, iv: bytes, first: int, n: int) -> bytes:
        out = bytearray(16 * n)
        if n:
            lib().orc_fast_keystream(self._c, iv, first, n, _addr(out))
        return bytes(out)

    def ghash_poly(self, data) -> bytes:
        p = ctypes.create_string_buffer(16)
        lib().orc_fast_ghash_poly(self._c, _addr(data), len(data), p)
        return p.raw
