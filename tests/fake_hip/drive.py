"""Drives the HOST side of the library against the fake HIP runtime of this directory (libaesgcm_fake.so: tests/fake_hip/fakehip.cpp) through the ordinary ctypes
binding: every family of entry points on every one of four fake devices, and the single-process multi-GPU object over all four.  Asserts device discipline --
only the device asked for is ever touched, every stream / event / pointer is used on its own device, the LDS attributes are set on every device used -- and that
the queued multi-GPU path has no host synchronisation per message.  Run by tests/test_fake_hip.py with AESGCM_LIB pointing at the fake library."""
import ctypes
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import aesgcm_amd  # noqa: E402,F401
from aesgcm_amd import lib  # noqa: E402

assert os.environ.get("AESGCM_LIB", "").endswith("libaesgcm_fake.so"), "run with AESGCM_LIB = the fake library"
L = lib.load()
F = ctypes.CDLL(os.environ["AESGCM_LIB"])
F.fake_syncs.restype = F.fake_launches.restype = F.fake_collectives.restype = ctypes.c_long
F.fake_live_allocations.restype = ctypes.c_size_t


def violations():
    b = ctypes.create_string_buffer(1 << 16)
    n = F.fake_violations(b, len(b))
    return n, b.value.decode()


def check(what, only=None):
    n, txt = violations()
    assert n == 0, "%s: %d violations\n%s" % (what, n, txt)
    if only is not None:
        assert F.fake_touched() == 1 << only, "%s: devices touched %s, wanted only %d" % (what, bin(F.fake_touched()), only)
    F.fake_reset()


assert lib.device_count() == 4
key = bytes(range(32))
MB = 1 << 20
for k in range(4):
    F.fake_reset()
    ctx = lib.Context(key, device=k)
    assert F.fake_attrs() == (2 << k) - 1, "LDS attributes set on %s after the first context of device %d" % (bin(F.fake_attrs()), k)      # once per device, when its first context is made
    check("ctx_create", k)
    d_in, d_out, d_aad = lib.DeviceBuffer(40 * MB, k), lib.DeviceBuffer(40 * MB, k), lib.DeviceBuffer(4096, k)
    iv = bytes(12)
    # whole messages on device buffers: k_main (+ k_combine), the cyclic launch in both shapes, dealt chunks with a fold level, the general shape with pieces
    for n, opts in ((0, {}), (1000, {}), (50000, {}), (4 * MB, {"cyc_half": 0}), (4 * MB, {"cyc_half": 1}), (4 * MB + 5, {"cyc_close": 0}),
                    (32 * MB, {"cyc_min": 0, "cyc_max": 0, "body_min": MB}), (32 * MB + 777, {"cyc_min": 0, "cyc_max": 0, "body_min": MB}), (8 * MB, {"cyc_min": 0, "cyc_max": 0, "body_min": MB, "fold_close": 0})):
        c2 = lib.Context(key, device=k)
        for o, v in opts.items():
            c2.set_option(o, v)
        c2.encrypt_dev(iv, d_in.ptr, n, d_out.ptr, d_aad=d_aad.ptr, aad_len=20 if n % 2 else 0)
        try:
            c2.decrypt_dev(iv, d_out.ptr, n, d_in.ptr, tag=bytes(16))
        except lib.AuthenticationError:
            pass
        c2.encrypt_dev(iv, d_in.ptr, n, d_out.ptr, want_tag=False)
        c2.last_tag()
        c2.close()
        check("encrypt_dev %d %r" % (n, opts), k)
    # host buffers, pipelined, streaming, unit level, rekey
    ctx.encrypt(iv, b"a" * 20, bytes(70000)); ctx.decrypt(iv, b"", bytes(3 * MB))
    ctx.encrypt_pipelined(iv, b"hdr", bytes(5 * MB + 3), chunk_bytes=MB); ctx.decrypt_pipelined(iv, b"", bytes(2 * MB), chunk_bytes=MB)
    ctx.stream_begin(iv); ctx.stream_aad(b"x" * 16); ctx.stream_update(bytes(4096)); ctx.stream_update(bytes(100)); ctx.stream_final()
    # the state of a message under way leaves its context (round 6): export, import into another context of the device, device-pointer updates of every launch structure
    other = lib.Context(key, device=k)
    ctx.stream_begin(iv); ctx.stream_aad(b"y" * 32); ctx.stream_update_dev(d_in.ptr, 4096, d_out.ptr)
    blob = ctx.stream_export()
    other.stream_import(blob)
    for nb in (1024, 3 * MB, 33 * MB + 16):
        other.stream_update_dev(d_in.ptr, nb, d_out.ptr)
    other.stream_update_dev(d_in.ptr, 100, d_out.ptr, stream=ctx.stream()); other.stream_final(); ctx.stream_final()
    assert ctx.status() == (lib.STATUS_OK, 0)
    other.close()
    ctx.keystream(iv, 0, 100); ctx.ecb_encrypt(bytes(64)); ctx.ghash(bytes(1000)); ctx.h(); ctx.rekey(bytes(16)); ctx.rekey(key)
    lib.key_expand(key, device=k); lib.gfmul(bytes(32), bytes(32), device=k)
    check("host-buffer paths", k)
    # shards
    d_part = lib.DeviceBuffer(64, k)
    ctx.shard_crypt_dev(False, iv, d_in.ptr, 8 * MB, d_out.ptr, 256, 40 * MB, d_part.ptr)
    ctx.shard_crypt_dev(False, iv, d_in.ptr, 1008, d_out.ptr, 0, 40 * MB, d_part.ptr, d_aad=d_aad.ptr, aad_len=16)
    ctx.shard_finalize_dev(iv, d_part.ptr, 1, 0, 40 * MB)
    ctx.shard_finalize_batch_dev([iv, iv], d_part.ptr, 1, [MB, MB])
    check("shards", k)
    # packets under the context's key: every shape the host can pick, offset arrays with the launch order, by rows (fixed and offset arrays), with the wipe
    n = 200000
    d_ivs, d_tags, d_auth = lib.DeviceBuffer(12 * n, k), lib.DeviceBuffer(16 * n, k), lib.DeviceBuffer(4 * n, k)
    d_off = lib.DeviceBuffer(8 * (n + 1), k)
    d_off.upload(struct.pack("<%dQ" % (n + 1), *[64 * i for i in range(n + 1)]))
    ctx.set_option("wipe_on_auth_fail", 1)
    for np_, plen in ((n, 64), (100000, 256), (5000, 1024), (300, 4096), (40, 65536), (9, MB)):
        ctx.packets_crypt_dev(False, np_, d_ivs.ptr, d_in.ptr, d_out.ptr, d_tags.ptr, pkt_len=plen)
        ctx.packets_crypt_dev(True, np_, d_ivs.ptr, d_out.ptr, d_out.ptr, d_tags.ptr, pkt_len=plen, d_expect_tags=d_tags.ptr, d_auth=d_auth.ptr)
    ctx.packets_crypt_dev(False, n, d_ivs.ptr, d_in.ptr, d_out.ptr, d_tags.ptr, d_data_off=d_off.ptr)                      # taken by length class (n >= 98304)
    ctx.packets_crypt_dev(False, 1000, d_ivs.ptr, d_in.ptr, d_out.ptr, d_tags.ptr, pkt_len=65536, d_data_off=d_off.ptr)   # by rows, planned on the device
    ctx.packets_crypt_dev(True, 1000, d_ivs.ptr, d_in.ptr, d_out.ptr, d_tags.ptr, pkt_len=65536, d_data_off=d_off.ptr, d_aad=d_aad.ptr, d_aad_off=d_off.ptr,
                          d_expect_tags=d_tags.ptr, d_auth=d_auth.ptr)                                                   # ... with an AAD array (the smalls of the closing launch), verified, wiped
    ctx.packets_crypt_dev(False, 300, d_ivs.ptr, d_in.ptr, d_out.ptr, d_tags.ptr, pkt_len=16400, d_aad=d_aad.ptr, aad_len=13)   # by rows, fixed-size records with a header and a ragged end
    d_ptrs, d_lens = lib.DeviceBuffer(8 * 1000, k), lib.DeviceBuffer(4 * 1000, k)
    d_ptrs.upload(struct.pack("<1000Q", *[d_in.ptr + 4096 * i for i in range(1000)]))
    d_lens.upload(struct.pack("<1000I", *([4000] * 1000)))
    ctx.messages_crypt_dev(False, 1000, d_ivs.ptr, d_ptrs.ptr, d_lens.ptr, d_ptrs.ptr, d_tags.ptr)                          # messages wherever they live: arrays of addresses and lengths
    ctx.messages_crypt_dev(True, 1000, d_ivs.ptr, d_ptrs.ptr, d_lens.ptr, d_ptrs.ptr, d_tags.ptr, d_aad_ptr=d_ptrs.ptr, d_aad_len=d_lens.ptr, d_expect_tags=d_tags.ptr, d_auth=d_auth.ptr)
    # routed calls fork: the row launches on the context's side stream, the packet kernels on the caller's -- with the caller's own stream too; the probe of the frame path
    ctx.packets_crypt_dev(False, n, d_ivs.ptr, d_in.ptr, d_out.ptr, d_tags.ptr, d_data_off=d_off.ptr, d_aad=d_aad.ptr, d_aad_off=d_off.ptr, stream=lib.Context(key, device=k).stream())
    ctx.frames_ceiling_probe_dev(n, d_ivs.ptr, d_off.ptr, d_tags.ptr, d_aad=d_aad.ptr, d_aad_off=d_off.ptr)
    assert set(ctx.last_route()) == {"route_min", "n_small", "lanes", "row_units"} and ctx.status() == (lib.STATUS_OK, 0)
    check("packets", k)
    # a key per packet (context-free: the device is an argument)
    d_keys = lib.DeviceBuffer(32 * n, k)
    for np_, plen in ((n, 64), (70000, 1024), (20000, 4096), (50, 65536)):
        lib.batch_crypt_dev(False, np_, 16, d_keys.ptr, d_ivs.ptr, d_in.ptr, plen, d_out.ptr, d_tags.ptr, device=k, stream=ctx.stream())
        lib.batch_crypt_dev(True, np_, 32, d_keys.ptr, d_ivs.ptr, d_out.ptr, plen, d_out.ptr, d_tags.ptr, d_expect_tags=d_tags.ptr, d_auth=d_auth.ptr, device=k, stream=ctx.stream())
    lib.batch_crypt_var_dev(False, n, 16, d_keys.ptr, d_ivs.ptr, d_in.ptr, d_off.ptr, d_out.ptr, d_tags.ptr, device=k, stream=ctx.stream())
    lib.wipe_failed_dev(50, d_out.ptr, d_auth.ptr, pkt_len=65536, device=k, stream=ctx.stream())
    check("batch", k)
    # utilities, timing
    lib.dev_copy(d_out.ptr, d_in.ptr, MB, device=k); d_in.fill_splitmix64(1); lib.dev_sync(k)
    t = lib.Timer(k); t.start(ctx.stream()); t.stop(ctx.stream()); t.ms(); t.close()
    ctx.timing_enable(True); ctx.encrypt_dev(iv, d_in.ptr, 32 * MB, d_out.ptr); ctx.timing_read(); ctx.ceiling_probe(64 * MB)
    a, b = lib.Context(key, device=k), lib.Context(key, device=k)
    a.wait(b); a.wait_fused(b); a.wait_fused(b)
    a.close(); b.close()
    check("utilities", k)
    for o in (d_in, d_out, d_aad, d_part, d_ivs, d_tags, d_auth, d_off, d_keys):
        o.free()
    ctx.close()
    check("teardown", k)

# negative control: the checks have teeth -- a buffer of device 0 handed to a context of device 2 is seen
F.fake_reset()
c2, wrong = lib.Context(key, device=2), lib.DeviceBuffer(4 * MB, 0)
c2.encrypt_dev(bytes(12), wrong.ptr, 4 * MB, wrong.ptr)
n_bad, txt = violations()
assert n_bad >= 2 and "lives on device 0, current device is 2" in txt, txt
c2.close(); wrong.free()
F.fake_reset()

# one process, four GPUs: a message sharded over all of them; four messages queued and collected with ONE finalize
F.fake_reset()
m = lib.MultiGpu(key, [0, 1, 2, 3])
assert m.n_ranks == 4 and F.fake_attrs() == 0b1111 and F.fake_touched() == 0b1111
n, _ = violations()
assert n == 0, violations()[1]
bufs = [(lib.DeviceBuffer(2 * MB, g), lib.DeviceBuffer(2 * MB, g)) for g in range(4)]
F.fake_reset()
for msg in range(4):
    m.crypt_dev(False, bytes([msg]) * 12, [b[0].ptr for b in bufs], [2 * MB, 2 * MB, 2 * MB, 2 * MB - 5], [b[1].ptr for b in bufs], want_tag=False)
assert F.fake_syncs() == 0, "queued multi-GPU messages made %d host synchronisations" % F.fake_syncs()
assert F.fake_collectives() == 16 and F.fake_touched() == 0b1111
try:
    m.crypt_dev(False, bytes(12), [b[0].ptr for b in bufs], [MB] * 4, [b[1].ptr for b in bufs])     # a call that wants its own tag while four wait: refused, nothing enqueued
    raise SystemExit("a call with its own tag jumped the queue")
except lib.AesGcmError as e:
    assert e.code == lib.ESTATE and F.fake_collectives() == 16
tags = m.last_tags(1) + m.last_tags(3)                      # the queue is a FIFO: a partial collect, then the rest
assert len(tags) == 4
try:
    m.last_tags(1)
    raise SystemExit("a tag was collected from an empty queue")
except lib.AesGcmError as e:
    assert e.code == lib.EARG
m.sync()
check("mgpu queued")
t = m.crypt_dev(True, bytes(12), [b[0].ptr for b in bufs], [MB] * 4, [b[1].ptr for b in bufs])
assert len(t) == 16
try:
    for msg in range(9):
        m.crypt_dev(False, bytes(12), [b[0].ptr for b in bufs], [MB] * 4, [b[1].ptr for b in bufs], want_tag=False)
    raise SystemExit("a ninth queued message was accepted")
except lib.AesGcmError as e:
    assert e.code == lib.ESTATE
m.last_tags(8)
check("mgpu")
m.close()
for a, b in bufs:
    a.free(); b.free()
check("mgpu teardown")
assert F.fake_live_allocations() <= 16, "%d allocations left" % F.fake_live_allocations()       # the per-device tables and dispensers stay for the life of the process
print("FAKE HIP OK")
