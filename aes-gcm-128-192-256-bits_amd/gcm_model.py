"""Drop-in for the reference's software model tb/gcm_model.py, backed by the MI355X HIP path.

The reference harness does `import gcm_model` and `gcm_model.gcm(key, iv, enc_dec)`
(tb/gcm_test.py:4,45), then wires load_aad / load_plain_text / load_cipher_text / get_tag as monitor
callbacks (tb/gcm_test.py:76-85) and hands .data_out / .tag to the scoreboard (:93-94).  This module
keeps exactly that surface -- same class name, constructor arguments, methods, list attributes and
tamper convention (tb/gcm_model.py:47-51) -- and replaces `from Crypto.Cipher import AES` with the
GPU library.  `from cocotb import log` becomes a standard `logging` logger when cocotb is absent.

Additionally it exports the one-shot bulk entry points BASELINE.json's north_star names:
    encrypt(key, iv, aad, data) -> (ct, tag)        decrypt(key, iv, aad, data, tag=None) -> (pt, tag)
"""
import hashlib
import logging
import os

from . import lib, sharding
from .lib import AesGcmError, AuthenticationError   # noqa: F401

try:                                    # tb/gcm_model.py:2
    from cocotb import log
except Exception:                       # cocotb is not installed outside the simulator
    log = logging.getLogger("gcm_model")


class gcm:
    """Same contract as the reference model class (tb/gcm_model.py:5-51), backed by the HIP library.

    gcm(key, icb, ed) with key = {'data': HEX, 'n_bytes': 16|24|32}, icb = {'data': HEX, 'n_bytes': 12},
    ed = 'enc' | 'dec'.  The harness feeds all AAD first (<= 16 bytes per call), then data (16 bytes per call,
    only the last may be shorter); like pycryptodome, any chunking is accepted.  Every load_plain_text / load_cipher_text call appends its output chunk to
    `data_out` before returning; get_tag(dut_tag) appends exactly one 16-byte entry to `tag`:
      enc : the tag this model computed;
      dec : the received tag if it authenticates, otherwise its bitwise complement, which makes the
            scoreboard comparison fail (the reference's convention, tb/gcm_model.py:47-51).
    """

    _FLUSH = 1 << 16            # bytes of data handed to the GPU per GHASH absorption
    _KS_MIN, _KS_MAX = 1 << 16, 1 << 22

    def __init__(self, key, icb, ed, device=0):
        self.ed = ed
        self.data_out = []
        self.tag = []
        key_bytes = int(key['data'], 16).to_bytes(key['n_bytes'], byteorder='big')
        iv_bytes = int(icb['data'], 16).to_bytes(icb['n_bytes'], byteorder='big')
        if len(iv_bytes) != 12:
            raise ValueError("the IP core and this model support 96-bit IVs only (src/gcm_pkg.vhd:15-17)")
        self.model = lib.Context(key_bytes, device=device)      # key load: on-GPU key expansion + H tables
        self.model.stream_begin(iv_bytes, decrypt=(ed != 'enc'))
        self._iv = iv_bytes
        self._final_tag = None
        # The harness calls once per 16-byte beat and needs the output chunk before the call returns.  A GPU round
        # trip per beat would cost ~60 us each, so the beats are answered from a keystream the GPU produced ahead
        # (aesgcm_keystream: E_K(IV || 2+i) is data-independent, src/gcm_gctr.vhd:141-150) and the inputs are handed
        # to the GPU for GHASH (and a re-check of the outputs) in blocks of _FLUSH bytes and at get_tag.
        self._aad = bytearray()
        self._aad_sent = False
        self._pend_in = bytearray()
        self._pend_out = bytearray()
        self._pos = 0                       # bytes of data seen so far
        self._ks = b""
        self._ks_off = 0                    # byte position of self._ks[0] in the keystream
        self._ks_next = self._KS_MIN

    # -- monitor callbacks (tb/gcm_test.py:76-85) ---------------------------------------------------
    def load_aad(self, aad):
        if self._pos or self._final_tag is not None:
            # pycryptodome raises TypeError when update() follows encrypt()/decrypt()
            raise TypeError("AAD must precede data")
        self._aad += bytes(aad)

    def load_plain_text(self, pt):
        self.data_out.append(self._crypt(bytes(pt)))

    def load_cipher_text(self, ct):
        self.data_out.append(self._crypt(bytes(ct)))

    def get_tag(self, tag):
        if self._final_tag is None:
            self._flush(final=True)
            self._final_tag = self.model.stream_final()
        mine = self._final_tag
        if self.ed == 'enc':
            self.tag.append(mine)
            log.info('model tag %032X', int.from_bytes(mine, 'big'))
            if bytes(tag) == mine:
                log.info('tags match')
            else:
                log.error('tag mismatch: DUT %s, model %s', bytes(tag).hex(), mine.hex())
            return
        if _ct_equal(bytes(tag), mine):
            self.tag.append(tag)
            log.info('tag verified: the message is authentic')
        else:
            log.error('authentication failed: key or IV incorrect, or message corrupted')
            flipped = ~int.from_bytes(tag, 'big') & ((1 << 128) - 1)
            self.tag.append(flipped.to_bytes(16, 'big'))

    # -- helpers -----------------------------------------------------------------------------------
    def _crypt(self, chunk):
        if self._final_tag is not None:
            raise TypeError("the message is already finalised")
        n = len(chunk)
        if self._pos + n > sharding.MAX_MESSAGE:
            raise ValueError("message exceeds the GCM length limit")           # aes_icb.vhd:114
        if not n:
            return b""
        lo = self._pos - self._ks_off
        if lo < 0 or lo + n > len(self._ks):
            first = self._pos // 16
            nblocks = max(self._ks_next // 16, (self._pos + n + 15) // 16 - first)
            nblocks = min(nblocks, (sharding.MAX_MESSAGE + 31) // 16 - first)
            self._ks = self._call(self.model.keystream, self._iv, first, nblocks)
            self._ks_off = 16 * first
            self._ks_next = min(2 * self._ks_next, self._KS_MAX)
            lo = self._pos - self._ks_off
        out = (int.from_bytes(chunk, 'big') ^ int.from_bytes(self._ks[lo:lo + n], 'big')).to_bytes(n, 'big')
        self._pend_in += chunk
        self._pend_out += out
        self._pos += n
        if len(self._pend_in) >= self._FLUSH:
            self._flush(final=False)
        return out

    def _flush(self, final):
        """GHASH absorption on the GPU: the AAD once, then the buffered inputs in whole blocks (everything at the end).
        The GPU recomputes the outputs as well; they must equal what the keystream produced."""
        if not self._aad_sent:
            if self._aad:
                self._call(self.model.stream_aad, bytes(self._aad))
            self._aad_sent = True
        whole = len(self._pend_in) if final else len(self._pend_in) // 16 * 16
        if whole:
            out = self._call(self.model.stream_update, bytes(self._pend_in[:whole]))
            if out != bytes(self._pend_out[:whole]):
                raise RuntimeError("keystream path and fused path disagree")
            del self._pend_in[:whole]
            del self._pend_out[:whole]

    def _call(self, fn, *args):
        try:
            return fn(*args)
        except AesGcmError as e:
            if e.code == lib.ESTATE:
                raise TypeError("calls out of order") from e
            if e.code == lib.ETOOLONG:
                raise ValueError("message exceeds the GCM length limit") from e
            raise


def _ct_equal(a, b):
    if len(a) != len(b):
        return False
    d = 0
    for x, y in zip(a, b):
        d |= x ^ y
    return d == 0


# ======================================================================================
# one-shot bulk surface (north_star: encrypt/decrypt(key, iv, aad, data) -> (ct, tag))
_ctx_cache = {}          # at most 8 contexts, keyed by a keyed digest of the key (raw keys are not kept on the host)
_ctx_salt = os.urandom(16)


def _ctx(key, device):
    k = (hashlib.blake2b(bytes(key), key=_ctx_salt, digest_size=16).digest(), len(key), device)
    c = _ctx_cache.get(k)
    if c is None:
        if len(_ctx_cache) >= 8:
            _ctx_cache.pop(next(iter(_ctx_cache))).close()
        c = _ctx_cache[k] = lib.Context(bytes(key), device=device)
    return c


def encrypt(key, iv, aad, data, device=0):
    """AES-GCM encrypt on the GPU -> (ciphertext, tag).  key 16/24/32 bytes, iv 12 bytes."""
    return _ctx(key, device).encrypt(iv, aad or b"", data)


def decrypt(key, iv, aad, data, tag=None, device=0):
    """AES-GCM decrypt on the GPU -> (plaintext, computed_tag).  With `tag` given, raises
    AuthenticationError (a ValueError, like pycryptodome's verify) on mismatch."""
    return _ctx(key, device).decrypt(iv, aad or b"", data, tag=tag)
