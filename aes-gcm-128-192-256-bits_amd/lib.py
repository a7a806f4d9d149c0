"""ctypes binding of libaesgcm_hip.so -- one Python function per entry point of include/aesgcm.h.

Mirrors the C ABI one to one (same names without the aesgcm_ prefix, same argument meaning, error
codes turned into exceptions).  No torch, no numpy requirement (numpy arrays are accepted as
buffers).  Loading fails loudly when the library is missing; compute calls fail loudly
(AesGcmError: AESGCM_EHIP) when no HIP device is usable -- there is no CPU path to fall back to.
"""
import ctypes
import os

from .build import SO, SO_DEBUG, build

OK, EARG, EKEYLEN, EIVLEN, ETOOLONG, EAUTH, EHIP, ENOMEM, ESTATE, EALIGN, ERCCL = 0, -1, -2, -3, -4, -5, -6, -7, -8, -9, -10

LAUNCH_NONE, LAUNCH_MAIN, LAUNCH_CYCLIC, LAUNCH_CYCLIC_HALF, LAUNCH_DEALT = range(5)      # aesgcm_ctx_last_launch
SHAPE_ROWS = 1 << 20    # AESGCM_SHAPE_ROWS: what packets_shape says for calls that go by rows
SHAPE_MIXED = 1 << 21   # AESGCM_SHAPE_MIXED: ... for calls with offset arrays: every message is routed by its own size on the device
STATUS_OK, STATUS_PLAN, STATUS_LENGTH, STATUS_UNITS = range(4)      # aesgcm_ctx_status
ABI_VERSION = 5         # AESGCM_ABI_VERSION of include/aesgcm.h this binding was written against

# every symbol include/aesgcm.h declares (tests check the .so exports exactly these)
SYMBOLS = [
    "aesgcm_abi_version", "aesgcm_strerror", "aesgcm_last_error", "aesgcm_device_count", "aesgcm_device_name",
    "aesgcm_key_expand", "aesgcm_ecb_encrypt", "aesgcm_gfmul", "aesgcm_ghash", "aesgcm_get_h",
    "aesgcm_ctx_create", "aesgcm_ctx_create_preexpanded", "aesgcm_ctx_rekey", "aesgcm_ctx_destroy", "aesgcm_ctx_device", "aesgcm_ctx_set_option", "aesgcm_ctx_stream", "aesgcm_ctx_wait", "aesgcm_ctx_wait_fused",
    "aesgcm_encrypt_pipelined", "aesgcm_decrypt_pipelined", "aesgcm_host_alloc", "aesgcm_host_free",
    "aesgcm_encrypt", "aesgcm_decrypt", "aesgcm_encrypt_dev", "aesgcm_decrypt_dev", "aesgcm_last_tag",
    "aesgcm_keystream", "aesgcm_keystream_dev",
    "aesgcm_shard_crypt_dev", "aesgcm_shard_finalize_dev", "aesgcm_shard_finalize_strided_dev", "aesgcm_shard_finalize_batch_dev", "aesgcm_batch_crypt_dev", "aesgcm_batch_crypt_var_dev", "aesgcm_packets_crypt_dev", "aesgcm_messages_crypt_dev", "aesgcm_batch_shape", "aesgcm_packets_shape",
    "aesgcm_stream_begin", "aesgcm_stream_aad", "aesgcm_stream_update", "aesgcm_stream_final",
    "aesgcm_dev_alloc", "aesgcm_dev_free", "aesgcm_dev_upload", "aesgcm_dev_download", "aesgcm_dev_sync", "aesgcm_dev_copy",
    "aesgcm_fill_splitmix64_dev",
    "aesgcm_ctx_timing_enable", "aesgcm_ctx_timing_read", "aesgcm_ctx_geometry", "aesgcm_ctx_body_geometry", "aesgcm_ctx_split", "aesgcm_ctx_wg_trace",
    "aesgcm_ctx_ceiling_probe",
    "aesgcm_timer_create", "aesgcm_timer_start", "aesgcm_timer_stop", "aesgcm_timer_ms", "aesgcm_timer_destroy",
    "aesgcm_comm_last_error", "aesgcm_comm_unique_id", "aesgcm_comm_create", "aesgcm_comm_ranks", "aesgcm_comm_allgather_dev",
    "aesgcm_comm_allreduce_f64", "aesgcm_comm_barrier", "aesgcm_comm_destroy",
    "aesgcm_mgpu_create", "aesgcm_mgpu_ranks", "aesgcm_mgpu_ctx", "aesgcm_mgpu_crypt_dev", "aesgcm_mgpu_destroy",
    "aesgcm_ctx_last_launch", "aesgcm_wipe_failed_dev", "aesgcm_mgpu_last_tags", "aesgcm_mgpu_sync", "aesgcm_batch_ceiling_probe_dev",
    "aesgcm_ctx_status", "aesgcm_stream_update_dev", "aesgcm_stream_export", "aesgcm_stream_import", "aesgcm_frames_ceiling_probe_dev", "aesgcm_ctx_last_route",
]


class AesGcmError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        msg = _strerror(code)
        if detail:
            msg += ": " + detail
        super().__init__("aesgcm error %d: %s" % (code, msg))


class AuthenticationError(AesGcmError, ValueError):
    """Tag mismatch on decrypt (the ValueError pycryptodome's verify() raises, tb/gcm_model.py:47)."""


_L = None
vp, sz, u64, cint = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_int
cp = ctypes.c_char_p


def load():
    """Load (building first if sources are newer) and type the library."""
    global _L
    if _L is not None:
        return _L
    build()                       # no-op when the .so is newer than csrc/ and include/aesgcm.h; tolerates a missing hipcc if a prebuilt .so exists
    _L = _typed(ctypes.CDLL(SO))
    return _L


_DBG = None


class debug_library:
    """`with lib.debug_library() as dbg:` -- inside the block every function of this module goes to libaesgcm_hip_dbg.so, the
    -DAESGCM_DEBUG_KNOBS build of the same sources, whose one extra export forces kernel shapes (include/aesgcm_debug.h):
    `dbg.force(pkt_lanes=8)`, `dbg.force(batch_lanes=16)`, ...; leaving the block clears every force and switches back to the
    product library.  Device memory is the process's (either library's allocations serve both); contexts belong to the library
    that made them, so create them inside the block.  Tests and profiling scripts only."""

    def __enter__(self):
        global _L, _DBG
        load()
        if _DBG is None:
            _DBG = _typed(ctypes.CDLL(SO_DEBUG))
            _DBG.aesgcm_debug_force_shape.argtypes = [cp, cint]
        self._prev, _L = _L, _DBG
        return self

    def force(self, **kw):
        for k, v in kw.items():
            _chk(_DBG.aesgcm_debug_force_shape(k.encode(), int(v)))

    def __exit__(self, *a):
        global _L
        for k in ("pkt_lanes", "pkt_deal", "batch_lanes", "batch_deal", "batch_order", "pkt_ilp", "pkt_rows"):
            _DBG.aesgcm_debug_force_shape(k.encode(), 0)
        _L = self._prev


def _typed(L):
    L.aesgcm_strerror.restype = cp
    L.aesgcm_strerror.argtypes = [cint]
    L.aesgcm_last_error.restype = cp
    L.aesgcm_device_count.argtypes = [ctypes.POINTER(cint)]
    L.aesgcm_device_name.argtypes = [cint, cp, sz]
    L.aesgcm_key_expand.argtypes = [cint, vp, sz, vp, ctypes.POINTER(cint)]
    L.aesgcm_ecb_encrypt.argtypes = [vp, vp, sz, vp]
    L.aesgcm_gfmul.argtypes = [cint, vp, vp, vp, sz]
    L.aesgcm_ghash.argtypes = [vp, vp, sz, vp]
    L.aesgcm_get_h.argtypes = [vp, vp]
    L.aesgcm_ctx_create.argtypes = [ctypes.POINTER(vp), cint, vp, sz]
    L.aesgcm_ctx_create_preexpanded.argtypes = [ctypes.POINTER(vp), cint, vp, cint]
    L.aesgcm_ctx_rekey.argtypes = [vp, vp, sz]
    L.aesgcm_ctx_destroy.argtypes = [vp]
    L.aesgcm_ctx_device.argtypes = [vp]
    L.aesgcm_ctx_set_option.argtypes = [vp, cp, ctypes.c_int64]
    L.aesgcm_ctx_stream.argtypes = [vp, ctypes.POINTER(vp)]
    L.aesgcm_ctx_wait.argtypes = [vp, vp]
    L.aesgcm_ctx_wait_fused.argtypes = [vp, vp]
    L.aesgcm_encrypt.argtypes = [vp, vp, vp, sz, vp, sz, vp, vp]
    L.aesgcm_decrypt.argtypes = [vp, vp, vp, sz, vp, sz, vp, vp, vp]
    L.aesgcm_encrypt_pipelined.argtypes = [vp, vp, vp, sz, vp, sz, vp, vp, sz]
    L.aesgcm_decrypt_pipelined.argtypes = [vp, vp, vp, sz, vp, sz, vp, vp, vp, sz]
    L.aesgcm_host_alloc.argtypes = [ctypes.POINTER(vp), sz]
    L.aesgcm_host_free.argtypes = [vp]
    L.aesgcm_encrypt_dev.argtypes = [vp, vp, vp, sz, vp, sz, vp, vp, vp]
    L.aesgcm_decrypt_dev.argtypes = [vp, vp, vp, sz, vp, sz, vp, vp, vp, vp]
    L.aesgcm_last_tag.argtypes = [vp, vp, vp]
    L.aesgcm_keystream.argtypes = [vp, vp, u64, u64, vp]
    L.aesgcm_keystream_dev.argtypes = [vp, vp, u64, u64, vp, vp]
    L.aesgcm_shard_crypt_dev.argtypes = [vp, cint, vp, vp, sz, vp, sz, vp, u64, u64, vp, vp]
    L.aesgcm_shard_finalize_dev.argtypes = [vp, vp, vp, sz, sz, u64, vp, vp]
    L.aesgcm_shard_finalize_strided_dev.argtypes = [vp, vp, vp, sz, sz, sz, u64, vp, vp]
    L.aesgcm_shard_finalize_batch_dev.argtypes = [vp, sz, vp, vp, sz, sz, sz, ctypes.POINTER(sz), ctypes.POINTER(u64), vp, vp]
    L.aesgcm_batch_crypt_dev.argtypes = [cint, cint, sz, sz, vp, vp, vp, sz, vp, sz, vp, vp, vp, vp, vp]
    L.aesgcm_batch_crypt_var_dev.argtypes = [cint, cint, sz, sz, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.aesgcm_packets_crypt_dev.argtypes = [vp, cint, sz, vp, vp, sz, vp, vp, sz, vp, vp, vp, vp, vp, vp]
    L.aesgcm_messages_crypt_dev.argtypes = [vp, cint, sz, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.aesgcm_batch_shape.argtypes = [cint, sz, sz, cint, ctypes.POINTER(cint)]
    L.aesgcm_packets_shape.argtypes = [vp, sz, sz, cint, ctypes.POINTER(cint)]
    L.aesgcm_stream_begin.argtypes = [vp, vp, cint]
    L.aesgcm_stream_aad.argtypes = [vp, vp, sz]
    L.aesgcm_stream_update.argtypes = [vp, vp, sz, vp]
    L.aesgcm_stream_final.argtypes = [vp, vp]
    L.aesgcm_dev_alloc.argtypes = [cint, ctypes.POINTER(vp), sz]
    L.aesgcm_dev_free.argtypes = [cint, vp]
    L.aesgcm_dev_upload.argtypes = [cint, vp, vp, sz]
    L.aesgcm_dev_download.argtypes = [cint, vp, vp, sz]
    L.aesgcm_dev_sync.argtypes = [cint]
    L.aesgcm_dev_copy.argtypes = [cint, vp, vp, sz, vp]
    L.aesgcm_fill_splitmix64_dev.argtypes = [cint, vp, sz, u64, u64, vp]
    L.aesgcm_ctx_timing_enable.argtypes = [vp, cint]
    L.aesgcm_ctx_timing_read.argtypes = [vp, ctypes.POINTER(u64), ctypes.POINTER(ctypes.c_double), cint]
    L.aesgcm_ctx_wg_trace.argtypes = [vp, vp, sz, ctypes.POINTER(sz)]
    L.aesgcm_ctx_geometry.argtypes = [vp, ctypes.POINTER(cint), ctypes.POINTER(cint), ctypes.POINTER(cint)]
    L.aesgcm_ctx_body_geometry.argtypes = [vp, ctypes.POINTER(cint), ctypes.POINTER(cint), ctypes.POINTER(cint)]
    L.aesgcm_ctx_split.argtypes = [vp, sz, u64, ctypes.POINTER(u64), ctypes.POINTER(u64)]
    L.aesgcm_ctx_ceiling_probe.argtypes = [vp, sz, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(u64)]
    L.aesgcm_timer_create.argtypes = [ctypes.POINTER(vp), cint]
    L.aesgcm_timer_start.argtypes = [vp, vp]
    L.aesgcm_timer_stop.argtypes = [vp, vp]
    L.aesgcm_timer_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
    L.aesgcm_timer_destroy.argtypes = [vp]
    L.aesgcm_comm_last_error.restype = cp
    L.aesgcm_comm_unique_id.argtypes = [vp]
    L.aesgcm_comm_create.argtypes = [ctypes.POINTER(vp), cint, vp, cint, cint]
    L.aesgcm_comm_ranks.argtypes = [vp, ctypes.POINTER(cint), ctypes.POINTER(cint)]
    L.aesgcm_comm_allgather_dev.argtypes = [vp, vp, vp, sz, vp]
    L.aesgcm_comm_allreduce_f64.argtypes = [vp, ctypes.POINTER(ctypes.c_double), cint]
    L.aesgcm_comm_barrier.argtypes = [vp]
    L.aesgcm_comm_destroy.argtypes = [vp]
    L.aesgcm_mgpu_create.argtypes = [ctypes.POINTER(vp), cint, ctypes.POINTER(cint), vp, sz]
    L.aesgcm_mgpu_ranks.argtypes = [vp, ctypes.POINTER(cint)]
    L.aesgcm_mgpu_ctx.argtypes = [vp, cint, ctypes.POINTER(vp)]
    L.aesgcm_mgpu_crypt_dev.argtypes = [vp, cint, vp, vp, sz, ctypes.POINTER(vp), ctypes.POINTER(sz), ctypes.POINTER(vp), vp]
    L.aesgcm_mgpu_destroy.argtypes = [vp]
    L.aesgcm_mgpu_last_tags.argtypes = [vp, sz, vp]
    L.aesgcm_mgpu_sync.argtypes = [vp]
    L.aesgcm_ctx_last_launch.argtypes = [vp, ctypes.POINTER(cint)]
    L.aesgcm_wipe_failed_dev.argtypes = [cint, sz, vp, sz, vp, vp, vp]
    L.aesgcm_batch_ceiling_probe_dev.argtypes = [cint, sz, sz, vp, vp, sz, vp, vp]
    L.aesgcm_ctx_status.argtypes = [vp, ctypes.POINTER(cint), ctypes.POINTER(u64)]
    L.aesgcm_stream_update_dev.argtypes = [vp, vp, sz, vp, vp]
    L.aesgcm_stream_export.argtypes = [vp, vp]
    L.aesgcm_stream_import.argtypes = [vp, vp]
    L.aesgcm_frames_ceiling_probe_dev.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp]
    L.aesgcm_ctx_last_route.argtypes = [vp, ctypes.POINTER(u64)]
    if L.aesgcm_abi_version() != ABI_VERSION:
        raise ImportError("libaesgcm_hip.so ABI %d, expected %d (stale build? rebuild with `make -C csrc`)" % (L.aesgcm_abi_version(), ABI_VERSION))
    return L


def _strerror(code):
    try:
        return load().aesgcm_strerror(code).decode()
    except Exception:
        return "code %d" % code


def _chk(rc):
    if rc == OK:
        return
    detail = load().aesgcm_last_error().decode() if rc == EHIP else ""
    if rc == ERCCL or (rc == EHIP and not detail):
        detail = load().aesgcm_comm_last_error().decode()
    if rc == EAUTH:
        raise AuthenticationError(rc, "MAC check failed")
    raise AesGcmError(rc, detail)


class _Buf:
    """(address, length, keepalive) view of bytes / bytearray / memoryview / numpy array / None."""

    def __init__(self, obj, writable=False):
        self.keep = obj
        if obj is None:
            self.addr, self.n = None, 0
        elif isinstance(obj, bytes):
            if writable:
                raise TypeError("output buffer must be writable")
            self.n = len(obj)
            self.addr = ctypes.cast(ctypes.c_char_p(obj), vp).value if self.n else None
        elif hasattr(obj, "ctypes") and hasattr(obj, "nbytes"):          # numpy
            if writable and not obj.flags.writeable:
                raise TypeError("output buffer must be writable")
            if not obj.flags.c_contiguous:
                raise TypeError("buffer must be C-contiguous")
            self.n = obj.nbytes
            self.addr = obj.ctypes.data if self.n else None
        else:
            mv = memoryview(obj).cast("B")
            self.n = mv.nbytes
            if self.n == 0:
                self.addr = None
            elif mv.readonly:
                if writable:
                    raise TypeError("output buffer must be writable")
                self.keep = bytes(mv)
                self.addr = ctypes.cast(ctypes.c_char_p(self.keep), vp).value
            else:
                self.keep = (ctypes.c_char * self.n).from_buffer(mv)
                self.addr = ctypes.addressof(self.keep)


def _fixed(b, n, what):
    b = bytes(b)
    if len(b) != n:
        raise AesGcmError(EIVLEN if what == "iv" else EARG, "%s must be %d bytes, got %d" % (what, n, len(b)))
    return b


# ---------------------------------------------------------------- module-level (no context)
def device_count():
    n = cint(0)
    _chk(load().aesgcm_device_count(ctypes.byref(n)))
    return n.value


def device_name(device=0):
    b = ctypes.create_string_buffer(256)
    _chk(load().aesgcm_device_name(device, b, 256))
    return b.value.decode()


def key_expand(key, device=0):
    """-> (expanded key bytes (16*(nr+1)), nr).  GPU twin of tb/key_exp.py aes_expand_key."""
    key = bytes(key)
    rk = ctypes.create_string_buffer(240)
    nr = cint(0)
    _chk(load().aesgcm_key_expand(device, key, len(key), rk, ctypes.byref(nr)))
    return rk.raw[:16 * (nr.value + 1)], nr.value


def gfmul(h, x, device=0):
    """Element-wise GF(2^128) products of equal-length sequences of 16-byte blocks."""
    h, x = bytes(h), bytes(x)
    if len(h) != len(x) or len(h) % 16:
        raise AesGcmError(EARG, "h and x must be equal-length multiples of 16 bytes")
    z = ctypes.create_string_buffer(max(len(h), 1))
    _chk(load().aesgcm_gfmul(device, h, x, z, len(h) // 16))
    return z.raw[:len(h)]


class DeviceBuffer:
    """Device memory owned by the library (hipMalloc); upload/download/fill helpers."""

    def __init__(self, nbytes, device=0):
        self.device, self.nbytes = device, nbytes
        p = vp()
        _chk(load().aesgcm_dev_alloc(device, ctypes.byref(p), nbytes))
        self.ptr = p.value

    def upload(self, data, offset=0):
        b = _Buf(data)
        if offset < 0 or offset + b.n > self.nbytes:
            raise AesGcmError(EARG, "upload past end of buffer")
        _chk(load().aesgcm_dev_upload(self.device, self.ptr + offset, b.addr, b.n))

    def download(self, nbytes=None, offset=0, out=None):
        n = self.nbytes - offset if nbytes is None else nbytes
        if offset < 0 or n < 0 or offset + n > self.nbytes:
            raise AesGcmError(EARG, "download past end of buffer")
        if out is None:
            out = bytearray(n)
        b = _Buf(out, writable=True)
        if b.n < n:
            raise AesGcmError(EARG, "output buffer smaller than the %d bytes requested" % n)
        _chk(load().aesgcm_dev_download(self.device, b.addr, self.ptr + offset, n))
        return out

    def fill_splitmix64(self, seed, first_word=0, nbytes=None, offset=0, stream=None):
        n = self.nbytes - offset if nbytes is None else nbytes
        if offset < 0 or n < 0 or offset + n > self.nbytes:
            raise AesGcmError(EARG, "fill past end of buffer")
        _chk(load().aesgcm_fill_splitmix64_dev(self.device, self.ptr + offset, n, seed, first_word, stream))

    def free(self):
        if self.ptr:
            load().aesgcm_dev_free(self.device, self.ptr)
            self.ptr = None

    __del__ = free


def batch_crypt_dev(decrypt, n_pkts, key_len, d_keys, d_ivs, d_in, pkt_len, d_out, d_tags, d_aad=None, aad_len=0,
                    d_expect_tags=None, d_auth=None, device=0, stream=None):
    """n independent packets with per-packet key and IV, all arrays contiguous device memory (aesgcm.h)."""
    _chk(load().aesgcm_batch_crypt_dev(device, int(bool(decrypt)), n_pkts, key_len, d_keys, d_ivs, d_aad, aad_len,
                                       d_in, pkt_len, d_out, d_tags, d_expect_tags, d_auth, stream))


def batch_crypt_var_dev(decrypt, n_pkts, key_len, d_keys, d_ivs, d_in, d_data_off, d_out, d_tags, d_aad=None, d_aad_off=None,
                        d_expect_tags=None, d_auth=None, device=0, stream=None):
    """Variable-length packets: uint64 offset arrays (n_pkts + 1 entries, device memory) delimit data and AAD."""
    _chk(load().aesgcm_batch_crypt_var_dev(device, int(bool(decrypt)), n_pkts, key_len, d_keys, d_ivs, d_aad, d_aad_off,
                                           d_in, d_data_off, d_out, d_tags, d_expect_tags, d_auth, stream))


def wipe_failed_dev(n_pkts, d_out, d_auth, pkt_len=0, d_data_off=None, device=0, stream=None):
    """aesgcm_wipe_failed_dev: zero the output of every packet whose d_auth entry is 0 (what the context option wipe_on_auth_fail does for the packet calls of a context)"""
    _chk(load().aesgcm_wipe_failed_dev(device, n_pkts, d_out, pkt_len, d_data_off, d_auth, stream))


def batch_ceiling_probe_dev(n_pkts, key_len, d_keys, d_ivs, pkt_len, d_tags, device=0, stream=None):
    """aesgcm_batch_ceiling_probe_dev: one launch of the batch kernel without the data's loads and stores (8 lanes per packet only); time it with a Timer"""
    _chk(load().aesgcm_batch_ceiling_probe_dev(device, n_pkts, key_len, d_keys, d_ivs, pkt_len, d_tags, stream))


def batch_shape(n_pkts, pkt_len=0, var_len=False, device=0):
    """lanes per packet the batch entry points take for such a call: 8 / 16 (k_batch3) or 64 (k_batch)"""
    v = cint(0)
    _chk(load().aesgcm_batch_shape(device, n_pkts, pkt_len, int(bool(var_len)), ctypes.byref(v)))
    return v.value


class PinnedBuffer:
    """Page-locked host memory (hipHostMalloc) exposed as a writable memoryview / numpy-compatible buffer."""

    def __init__(self, nbytes):
        p = vp()
        _chk(load().aesgcm_host_alloc(ctypes.byref(p), nbytes))
        self.ptr, self.nbytes = p.value, nbytes
        self._arr = (ctypes.c_ubyte * max(nbytes, 1)).from_address(self.ptr)
        self.view = memoryview(self._arr).cast("B")[:nbytes]

    def free(self):
        if self.ptr:
            self.view = None
            self._arr = None
            load().aesgcm_host_free(self.ptr)
            self.ptr = None

    __del__ = free


def dev_sync(device=0):
    _chk(load().aesgcm_dev_sync(device))


def dev_copy(d_dst, d_src, nbytes, device=0, stream=None):
    """asynchronous device-to-device copy by the library's plain copy kernel"""
    _chk(load().aesgcm_dev_copy(device, d_dst, d_src, nbytes, stream))


class Timer:
    """aesgcm_timer: two HIP events recorded on the stream the timed launches run on."""

    def __init__(self, device=0):
        self._t = None
        t = vp()
        _chk(load().aesgcm_timer_create(ctypes.byref(t), device))
        self._t = t.value

    def start(self, stream=None):
        _chk(load().aesgcm_timer_start(self._t, stream))

    def stop(self, stream=None):
        _chk(load().aesgcm_timer_stop(self._t, stream))

    def ms(self):
        v = ctypes.c_double(0)
        _chk(load().aesgcm_timer_ms(self._t, ctypes.byref(v)))
        return v.value

    def close(self):
        if self._t:
            load().aesgcm_timer_destroy(self._t)
            self._t = None

    __del__ = close


# ---------------------------------------------------------------- context
class Context:
    """aesgcm_ctx: (device, expanded key, H, H-power tables).  One per key."""

    def __init__(self, key=None, device=0, expanded_key=None):
        self._c = None
        L = load()
        c = vp()
        if expanded_key is not None:
            ek = bytes(expanded_key)
            nr = len(ek) // 16 - 1
            if len(ek) % 16 or nr not in (10, 12, 14):
                raise AesGcmError(EKEYLEN, "expanded key must be 176/208/240 bytes")
            _chk(L.aesgcm_ctx_create_preexpanded(ctypes.byref(c), device, ek, nr))
        else:
            key = bytes(key)
            _chk(L.aesgcm_ctx_create(ctypes.byref(c), device, key, len(key)))
        self._c = c.value
        self.device = device
        self._lib = L               # the library that made it destroys it (debug_library switches the module's library for a block)

    _borrowed = False           # True for a view of a context another object owns (MultiGpu.context)
    _lib = None

    def close(self):
        if self._c:
            if not self._borrowed:
                self._lib.aesgcm_ctx_destroy(self._c)
            self._c = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def rekey(self, key):
        """aesgcm_ctx_rekey: a new key (16 / 24 / 32 bytes) for this context -- the reference core's key load between frames; everything else stays"""
        key = bytes(key)
        _chk(self._lib.aesgcm_ctx_rekey(self._c, key, len(key)))
        return self

    def set_option(self, key, value):
        """aesgcm_ctx_set_option: "tw", "body_min", "cyc_min", "cyc_max", "cyc_close", "fold_close", "cyc_prio", "pkt_order", "rows_min", "route_top_min", "route_mid_min", "route_blocks_min", "rows_block", "wipe_on_auth_fail", "poll_us" (include/aesgcm.h)"""
        _chk(self._lib.aesgcm_ctx_set_option(self._c, key.encode(), int(value)))
        return self

    def last_launch(self):
        """aesgcm_ctx_last_launch: which launch structure the last whole-message call took (LAUNCH_MAIN / CYCLIC / CYCLIC_HALF / DEALT)"""
        v = cint(0)
        _chk(self._lib.aesgcm_ctx_last_launch(self._c, ctypes.byref(v)))
        return v.value

    def status(self):
        """aesgcm_ctx_status -> (code, detail): what an asynchronous call on this context (packets with offset arrays, messages) was refused for on the device --
        STATUS_LENGTH / STATUS_PLAN / STATUS_UNITS, detail = the first offending message for LENGTH -- or (STATUS_OK, 0).  Reading clears it.  Synchronise first."""
        code, detail = cint(0), u64(0)
        _chk(self._lib.aesgcm_ctx_status(self._c, ctypes.byref(code), ctypes.byref(detail)))
        return code.value, detail.value

    def last_route(self):
        """aesgcm_ctx_last_route -> dict(route_min, n_small, lanes, row_units): what the device decided for the last call with offset arrays / scattered messages"""
        v = (u64 * 4)()
        _chk(self._lib.aesgcm_ctx_last_route(self._c, v))
        return {"route_min": v[0], "n_small": v[1], "lanes": v[2], "row_units": v[3]}

    def packets_shape(self, n_pkts, pkt_len=0, var_len=False):
        """lanes per packet packets_crypt_dev takes for such a call: 1 (k_pktl), 4 / 8 / 16 or 64 (k_pktg), or SHAPE_ROWS: by rows (k_rows); with offset arrays
        (var_len) SHAPE_MIXED: every message is routed by its own size on the device, pkt_len is ignored"""
        v = cint(0)
        _chk(self._lib.aesgcm_packets_shape(self._c, n_pkts, pkt_len, int(bool(var_len)), ctypes.byref(v)))
        return v.value

    def stream(self):
        """the context's own HIP stream as an integer handle (what stream=None means)"""
        s = vp()
        _chk(self._lib.aesgcm_ctx_stream(self._c, ctypes.byref(s)))
        return s.value

    def wait(self, other):
        """what is enqueued on this context's stream from now on starts after everything enqueued so far on `other`'s"""
        _chk(self._lib.aesgcm_ctx_wait(self._c, other._c))

    def wait_fused(self, other):
        """... starts after `other`'s most recently enqueued fused kernel (not its fold / combine tail)"""
        _chk(self._lib.aesgcm_ctx_wait_fused(self._c, other._c))

    # unit level
    def h(self):
        b = ctypes.create_string_buffer(16)
        _chk(self._lib.aesgcm_get_h(self._c, b))
        return b.raw

    def ecb_encrypt(self, blocks):
        b = _Buf(blocks)
        if b.n % 16:
            raise AesGcmError(EARG, "ECB input must be a multiple of 16 bytes")
        out = bytearray(b.n)
        o = _Buf(out, writable=True)
        _chk(self._lib.aesgcm_ecb_encrypt(self._c, b.addr, b.n // 16, o.addr))
        return bytes(out)

    def ghash(self, data):
        b = _Buf(data)
        y = ctypes.create_string_buffer(16)
        _chk(self._lib.aesgcm_ghash(self._c, b.addr, b.n, y))
        return y.raw

    def keystream(self, iv, first_block, nblocks):
        out = bytearray(16 * nblocks)
        o = _Buf(out, writable=True)
        _chk(self._lib.aesgcm_keystream(self._c, _fixed(iv, 12, "iv"), first_block, nblocks, o.addr))
        return bytes(out)

    # whole messages, host buffers
    def encrypt(self, iv, aad, pt, out=None):
        """-> (ct, tag)"""
        a, p = _Buf(aad), _Buf(pt)
        ret = out if out is not None else bytearray(p.n)
        o = _Buf(ret, writable=True)
        tag = ctypes.create_string_buffer(16)
        _chk(self._lib.aesgcm_encrypt(self._c, _fixed(iv, 12, "iv"), a.addr, a.n, p.addr, p.n, o.addr, tag))
        return (bytes(ret) if out is None else ret), tag.raw

    def decrypt(self, iv, aad, ct, tag=None, out=None):
        """-> (pt, computed_tag); raises AuthenticationError when `tag` is given and does not match
        (the plaintext has been produced regardless, as in the reference model)."""
        a, c = _Buf(aad), _Buf(ct)
        ret = out if out is not None else bytearray(c.n)
        o = _Buf(ret, writable=True)
        tout = ctypes.create_string_buffer(16)
        exp = _fixed(tag, 16, "tag") if tag is not None else None
        rc = self._lib.aesgcm_decrypt(self._c, _fixed(iv, 12, "iv"), a.addr, a.n, c.addr, c.n, o.addr, exp, tout)
        self.last_plaintext = bytes(ret) if out is None else ret
        _chk(rc)
        return self.last_plaintext, tout.raw

    def encrypt_pipelined(self, iv, aad, pt, out=None, chunk_bytes=0):
        """Host buffers, H2D / kernel / D2H overlapped in chunks -> (ct, tag)."""
        a, p = _Buf(aad), _Buf(pt)
        ret = out if out is not None else bytearray(p.n)
        o = _Buf(ret, writable=True)
        tag = ctypes.create_string_buffer(16)
        _chk(self._lib.aesgcm_encrypt_pipelined(self._c, _fixed(iv, 12, "iv"), a.addr, a.n, p.addr, p.n, o.addr, tag, chunk_bytes))
        return (bytes(ret) if out is None else ret), tag.raw

    def decrypt_pipelined(self, iv, aad, ct, tag=None, out=None, chunk_bytes=0):
        a, c = _Buf(aad), _Buf(ct)
        ret = out if out is not None else bytearray(c.n)
        o = _Buf(ret, writable=True)
        tout = ctypes.create_string_buffer(16)
        exp = _fixed(tag, 16, "tag") if tag is not None else None
        rc = self._lib.aesgcm_decrypt_pipelined(self._c, _fixed(iv, 12, "iv"), a.addr, a.n, c.addr, c.n, o.addr, exp, tout, chunk_bytes)
        self.last_plaintext = bytes(ret) if out is None else ret
        _chk(rc)
        return self.last_plaintext, tout.raw

    # whole messages, device buffers
    def encrypt_dev(self, iv, d_pt, nbytes, d_ct, d_aad=None, aad_len=0, stream=None, want_tag=True):
        tag = ctypes.create_string_buffer(16) if want_tag else None
        _chk(self._lib.aesgcm_encrypt_dev(self._c, _fixed(iv, 12, "iv"), d_aad, aad_len, d_pt, nbytes, d_ct, tag, stream))
        return tag.raw if want_tag else None

    def decrypt_dev(self, iv, d_ct, nbytes, d_pt, d_aad=None, aad_len=0, tag=None, stream=None, want_tag=True):
        tout = ctypes.create_string_buffer(16) if want_tag else None
        exp = _fixed(tag, 16, "tag") if tag is not None else None
        _chk(self._lib.aesgcm_decrypt_dev(self._c, _fixed(iv, 12, "iv"), d_aad, aad_len, d_ct, nbytes, d_pt, exp, tout, stream))
        return tout.raw if want_tag else None

    def last_tag(self, stream=None):
        t = ctypes.create_string_buffer(16)
        _chk(self._lib.aesgcm_last_tag(self._c, t, stream))
        return t.raw

    def keystream_dev(self, iv, first_block, nblocks, d_out, stream=None):
        _chk(self._lib.aesgcm_keystream_dev(self._c, _fixed(iv, 12, "iv"), first_block, nblocks, d_out, stream))

    # many packets under this context's key
    def packets_crypt_dev(self, decrypt, n_pkts, d_ivs, d_in, d_out, d_tags, pkt_len=0, d_data_off=None,
                          d_aad=None, aad_len=0, d_aad_off=None, d_expect_tags=None, d_auth=None, stream=None):
        _chk(self._lib.aesgcm_packets_crypt_dev(self._c, int(bool(decrypt)), n_pkts, d_ivs, d_aad, aad_len, d_aad_off,
                                             d_in, pkt_len, d_data_off, d_out, d_tags, d_expect_tags, d_auth, stream))

    def frames_ceiling_probe_dev(self, n_pkts, d_ivs, d_data_off, d_tags, d_aad=None, d_aad_off=None, stream=None):
        """aesgcm_frames_ceiling_probe_dev: the packet kernels' instruction stream over these frames without the data's loads and stores (measurement support)"""
        _chk(self._lib.aesgcm_frames_ceiling_probe_dev(self._c, n_pkts, d_ivs, d_aad, d_aad_off, d_data_off, d_tags, stream))

    def messages_crypt_dev(self, decrypt, n_msgs, d_ivs, d_in_ptr, d_len, d_out_ptr, d_tags, d_aad_ptr=None, d_aad_len=None, d_expect_tags=None, d_auth=None, stream=None):
        """aesgcm_messages_crypt_dev: n_msgs messages wherever they live -- device arrays of addresses (uint64) and lengths (uint32) -- under the context's key, by rows"""
        _chk(self._lib.aesgcm_messages_crypt_dev(self._c, int(bool(decrypt)), n_msgs, d_ivs, d_aad_ptr, d_aad_len, d_in_ptr, d_len, d_out_ptr, d_tags, d_expect_tags, d_auth, stream))

    # shards
    def shard_crypt_dev(self, decrypt, iv, d_in, nbytes, d_out, first_block, total_len, d_partial,
                        d_aad=None, aad_len=0, stream=None):
        _chk(self._lib.aesgcm_shard_crypt_dev(self._c, int(bool(decrypt)), _fixed(iv, 12, "iv"), d_aad, aad_len,
                                           d_in, nbytes, d_out, first_block, total_len, d_partial, stream))

    def shard_finalize_dev(self, iv, d_partials, n_partials, aad_len, total_len, stream=None, want_tag=True, stride_bytes=16):
        tag = ctypes.create_string_buffer(16) if want_tag else None
        _chk(self._lib.aesgcm_shard_finalize_strided_dev(self._c, _fixed(iv, 12, "iv"), d_partials, n_partials, stride_bytes, aad_len, total_len, tag, stream))
        return tag.raw if want_tag else None

    def shard_finalize_batch_dev(self, ivs, d_partials, n_partials, total_lens, aad_lens=None, stride_bytes=None, msg_stride_bytes=16, stream=None):
        """the tags of len(ivs) messages in one launch and one wait (aesgcm_shard_finalize_batch_dev); default layout [rank][message][16]"""
        n = len(ivs)
        ivb = b"".join(_fixed(iv, 12, "iv") for iv in ivs)
        tl = (u64 * n)(*total_lens)
        al = (sz * n)(*aad_lens) if aad_lens is not None else None
        tags = ctypes.create_string_buffer(16 * n)
        _chk(self._lib.aesgcm_shard_finalize_batch_dev(self._c, n, ivb, d_partials, n_partials, 16 * n if stride_bytes is None else stride_bytes,
                                                    msg_stride_bytes, al, tl, tags, stream))
        return [tags.raw[16 * m:16 * m + 16] for m in range(n)]

    # streaming
    def stream_begin(self, iv, decrypt=False):
        _chk(self._lib.aesgcm_stream_begin(self._c, _fixed(iv, 12, "iv"), int(bool(decrypt))))

    def stream_aad(self, aad):
        b = _Buf(aad)
        _chk(self._lib.aesgcm_stream_aad(self._c, b.addr, b.n))

    def stream_update(self, data):
        b = _Buf(data)
        out = bytearray(b.n)
        o = _Buf(out, writable=True)
        _chk(self._lib.aesgcm_stream_update(self._c, b.addr, b.n, o.addr))
        return bytes(out)

    def stream_final(self):
        t = ctypes.create_string_buffer(16)
        _chk(self._lib.aesgcm_stream_final(self._c, t))
        return t.raw

    def stream_update_dev(self, d_in, nbytes, d_out, stream=None):
        """aesgcm_stream_update_dev: the next chunk of the open session on device pointers, asynchronous on `stream` (None = the context's own)"""
        _chk(self._lib.aesgcm_stream_update_dev(self._c, d_in, nbytes, d_out, stream))

    def stream_export(self):
        """aesgcm_stream_export -> the 64-byte state of the open session (no key in it); the session stays open"""
        b = ctypes.create_string_buffer(64)
        _chk(self._lib.aesgcm_stream_export(self._c, b))
        return b.raw

    def stream_import(self, blob):
        """aesgcm_stream_import: open a session in this context at the point `blob` (another context's stream_export under the same key) was taken"""
        _chk(self._lib.aesgcm_stream_import(self._c, _fixed(blob, 64, "blob")))

    # measurement
    def timing_enable(self, on=True):
        _chk(self._lib.aesgcm_ctx_timing_enable(self._c, int(on)))

    def timing_read(self, reset=True):
        n, ms = u64(0), ctypes.c_double(0)
        _chk(self._lib.aesgcm_ctx_timing_read(self._c, ctypes.byref(n), ctypes.byref(ms), int(reset)))
        return n.value, ms.value

    def wg_trace(self, max_wgs=512):
        """[(start, end, hw_id, xcc_id)] per workgroup of the last timed launch (100 MHz wall clock)."""
        buf = (u64 * (4 * max_wgs))()
        n = sz(0)
        _chk(self._lib.aesgcm_ctx_wg_trace(self._c, buf, max_wgs, ctypes.byref(n)))
        return [tuple(buf[4 * i:4 * i + 4]) for i in range(n.value)]

    def geometry(self, body=False):
        """launch geometry of k_main, or (body=True) of k_body, the kernel of the aligned middle of ranges >= 256 MiB"""
        a, b, c = cint(0), cint(0), cint(0)
        fn = self._lib.aesgcm_ctx_body_geometry if body else self._lib.aesgcm_ctx_geometry
        _chk(fn(self._c, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return dict(workgroups=a.value, wg_lanes=b.value, lds_bytes=c.value)

    def ceiling_probe(self, nbytes):
        """(ms, blocks) of the fused kernel's instruction stream without its HBM traffic (aesgcm_ctx_ceiling_probe)"""
        ms, nb = ctypes.c_double(0), u64(0)
        _chk(self._lib.aesgcm_ctx_ceiling_probe(self._c, nbytes, ctypes.byref(ms), ctypes.byref(nb)))
        return ms.value, nb.value

    def split(self, nbytes, first_block=0):
        """(head_blocks, body_blocks) of the head / k_body / tail cut of a data range; body_blocks = 0: one k_main launch"""
        h, b = u64(0), u64(0)
        _chk(self._lib.aesgcm_ctx_split(self._c, nbytes, first_block, ctypes.byref(h), ctypes.byref(b)))
        return h.value, b.value


# ---------------------------------------------------------------- the exchange step (RCCL inside the library)
def comm_unique_id():
    """128-byte RCCL unique id (rank 0 makes it, every rank passes it to Comm)."""
    b = ctypes.create_string_buffer(128)
    _chk(load().aesgcm_comm_unique_id(b))
    return b.raw


class Comm:
    """aesgcm_comm: one rank of an RCCL communicator (ncclCommInitRank), one process per GPU."""

    def __init__(self, unique_id, n_ranks, rank, device=0):
        self._c = None
        c = vp()
        _chk(load().aesgcm_comm_create(ctypes.byref(c), device, _fixed(unique_id, 128, "unique id"), n_ranks, rank))
        self._c = c.value
        n, r = cint(0), cint(0)
        _chk(load().aesgcm_comm_ranks(self._c, ctypes.byref(n), ctypes.byref(r)))
        self.n_ranks, self.rank = n.value, r.value        # what RCCL reports

    def allgather_dev(self, d_send, d_recv, bytes_per_rank, stream=None):
        _chk(load().aesgcm_comm_allgather_dev(self._c, d_send, d_recv, bytes_per_rank, stream))

    def allreduce(self, value, op="max"):
        v = ctypes.c_double(value)
        _chk(load().aesgcm_comm_allreduce_f64(self._c, ctypes.byref(v), {"max": 0, "min": 1, "sum": 2}[op]))
        return v.value

    def barrier(self):
        _chk(load().aesgcm_comm_barrier(self._c))

    def close(self):
        if self._c:
            load().aesgcm_comm_destroy(self._c)
            self._c = None

    __del__ = close


class MultiGpu:
    """aesgcm_mgpu: one process, ndev GPUs (ncclCommInitAll); one message sharded over them."""

    def __init__(self, key, devices):
        self._m = None
        devices = list(devices)
        arr = (cint * len(devices))(*devices)
        m = vp()
        key = bytes(key)
        _chk(load().aesgcm_mgpu_create(ctypes.byref(m), len(devices), arr, key, len(key)))
        self._m, self.devices = m.value, devices
        n = cint(0)
        _chk(load().aesgcm_mgpu_ranks(self._m, ctypes.byref(n)))
        self.n_ranks = n.value                             # communicator size RCCL reports

    def crypt_dev(self, decrypt, iv, d_in, shard_len, d_out, d_aad=None, aad_len=0, want_tag=True):
        """d_in / d_out / shard_len: one entry per device -> tag.  want_tag=False: the message is only enqueued (up to 8 may wait); last_tags() collects"""
        g = len(self.devices)
        if not (len(d_in) == len(d_out) == len(shard_len) == g):
            raise AesGcmError(EARG, "one shard per device")
        pin, pout, ln = (vp * g)(*d_in), (vp * g)(*d_out), (sz * g)(*shard_len)
        tag = ctypes.create_string_buffer(16) if want_tag else None
        _chk(load().aesgcm_mgpu_crypt_dev(self._m, int(bool(decrypt)), _fixed(iv, 12, "iv"), d_aad, aad_len, pin, ln, pout, tag))
        return tag.raw if want_tag else None

    def last_tags(self, n):
        """the tags of the OLDEST n messages still queued (want_tag=False), in the order queued; they leave the queue, the rest stays (one finalize launch on the first device)"""
        t = ctypes.create_string_buffer(16 * n)
        _chk(load().aesgcm_mgpu_last_tags(self._m, n, t))
        return [t.raw[16 * k:16 * k + 16] for k in range(n)]

    def sync(self):
        _chk(load().aesgcm_mgpu_sync(self._m))

    def context(self, g):
        """device g's Context, borrowed from the mgpu object (closing it is a no-op)"""
        c = vp()
        _chk(load().aesgcm_mgpu_ctx(self._m, g, ctypes.byref(c)))
        ctx = Context.__new__(Context)
        ctx._c, ctx.device, ctx._borrowed, ctx._owner, ctx._lib = c.value, self.devices[g], True, self, load()
        return ctx

    def close(self):
        if self._m:
            load().aesgcm_mgpu_destroy(self._m)
            self._m = None

    __del__ = close
