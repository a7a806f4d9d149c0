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
import logging

from . import lib
from .lib import AesGcmError, AuthenticationError   # noqa: F401

try:                                    # tb/gcm_model.py:2
    from cocotb import log
except Exception:                       # cocotb is not installed outside the simulator
    log = logging.getLogger("gcm_model")


class gcm:
    """Same contract as the reference model class (tb/gcm_model.py:5-51), backed by the HIP library.

    gcm(key, icb, ed) with key = {'data': HEX, 'n_bytes': 16|24|32}, icb = {'data': HEX, 'n_bytes': 12},
    ed = 'enc' | 'dec'.  The harness feeds all AAD first (<= 16 bytes per call), then data (16 bytes per call,
    only the last may be shorter).  Every load_plain_text / load_cipher_text call appends its output chunk to
    `data_out` before returning; get_tag(dut_tag) appends exactly one 16-byte entry to `tag`:
      enc : the tag this model computed;
      dec : the received tag if it authenticates, otherwise its bitwise complement, which makes the
            scoreboard comparison fail (the reference's convention, tb/gcm_model.py:47-51).
    """

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
        self._final_tag = None

    # -- monitor callbacks (tb/gcm_test.py:76-85) ---------------------------------------------------
    def load_aad(self, aad):
        self._call(self.model.stream_aad, aad)

    def load_plain_text(self, pt):
        self.data_out.append(self._call(self.model.stream_update, pt))

    def load_cipher_text(self, ct):
        self.data_out.append(self._call(self.model.stream_update, ct))

    def get_tag(self, tag):
        if self._final_tag is None:
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
    def _call(self, fn, data):
        try:
            return fn(bytes(data))
        except AesGcmError as e:
            if e.code == lib.ESTATE:
                # pycryptodome raises TypeError for out-of-order calls (update() after encrypt())
                raise TypeError("AAD must precede data and only the last chunk may be ragged") from e
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
_ctx_cache = {}


def _ctx(key, device):
    k = (bytes(key), device)
    c = _ctx_cache.get(k)
    if c is None:
        if len(_ctx_cache) >= 8:
            _ctx_cache.pop(next(iter(_ctx_cache))).close()
        c = _ctx_cache[k] = lib.Context(k[0], device=device)
    return c


def encrypt(key, iv, aad, data, device=0):
    """AES-GCM encrypt on the GPU -> (ciphertext, tag).  key 16/24/32 bytes, iv 12 bytes."""
    return _ctx(key, device).encrypt(iv, aad or b"", data)


def decrypt(key, iv, aad, data, tag=None, device=0):
    """AES-GCM decrypt on the GPU -> (plaintext, computed_tag).  With `tag` given, raises
    AuthenticationError (a ValueError, like pycryptodome's verify) on mismatch."""
    return _ctx(key, device).decrypt(iv, aad or b"", data, tag=tag)
