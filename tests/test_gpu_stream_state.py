"""GPU: the state of a message under way can leave its context (round 6) -- aesgcm_stream_export / aesgcm_stream_import / aesgcm_stream_update_dev.
The RTL holds that state in its Y register (src/gcm_ghash.vhd:174-186) and its block counter (src/aes_icb.vhd:97-100) and cannot hand it out, nor can the
pycryptodome model (tb/gcm_model.py); SURVEY.md 5 (checkpoint / resume) and 8(f2) name it as what a pipeline around the core has to carry.  Here a message of
unknown total length moves between contexts, between the host-buffer and the device-pointer form of the call, and between PROCESSES through a file; ciphertext and
tag always equal the oracle's one-shot result."""
import json
import os
import random
import subprocess
import sys

import pytest

from util import splitmix_bytes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cuts(rng, n, pieces):
    """pieces - 1 random 16-byte boundaries inside [0, n)"""
    pts = sorted(set(16 * rng.randrange(0, n // 16 + 1) for _ in range(pieces - 1)))
    return [0] + pts + [n]


@pytest.mark.parametrize("klen,n,aad_len,pieces", [(16, 1000, 0, 3), (24, 70000, 20, 4), (32, (3 << 20) + 5, 37, 5), (32, 16 * 4096, 16, 2), (16, 0, 9, 2), (32, (24 << 20) + 16 * 3 + 1, 0, 3)])
def test_split_at_random_block_boundaries_across_two_contexts(hip, orc, klen, n, aad_len, pieces):
    """every piece on the other context: export from the one that did the last piece, import into the other -- host-buffer updates for small pieces, device-pointer
    updates (any size: cyclic rows and dealt chunks for the multi-MiB pieces) for the rest; encrypt, then decrypt the same way"""
    rng = random.Random(8000 + n % 1000 + klen)
    key, iv, aad, pt = splitmix_bytes(8001 + klen, klen), splitmix_bytes(8002, 12), splitmix_bytes(8003, aad_len), splitmix_bytes(8004 + n % 97, n)
    want_ct, want_tag = orc.Fast(key).encrypt(iv, aad, pt)
    ctxs = [hip.Context(key), hip.Context(key)]
    d_in, d_out = hip.DeviceBuffer(max(n, 16)), hip.DeviceBuffer(max(n, 16))
    for dec in (False, True):
        src, want = (want_ct, pt) if dec else (pt, want_ct)
        if n:
            d_in.upload(src)
        cuts = _cuts(rng, n, pieces)
        ctxs[0].stream_begin(iv, decrypt=dec)
        a_cut = aad_len // 2 // 16 * 16                          # the AAD in two beats as well, the second on the other context
        if aad_len:
            ctxs[0].stream_aad(aad[:a_cut])
            ctxs[1].stream_import(ctxs[0].stream_export())
            ctxs[1].stream_aad(aad[a_cut:])
            ctxs[0], ctxs[1] = ctxs[1], ctxs[0]
            with pytest.raises(hip.AesGcmError):                 # the context the state came from still has its session open: importing back into it is refused ...
                ctxs[1].stream_import(ctxs[0].stream_export())
            ctxs[1].stream_final()                               # ... until it is closed
        out = bytearray(n)
        for k in range(len(cuts) - 1):
            lo, hi = cuts[k], cuts[k + 1]
            c = ctxs[0]
            if hi - lo <= 4096 and rng.random() < 0.5:
                out[lo:hi] = c.stream_update(src[lo:hi])
            else:
                c.stream_update_dev(d_in.ptr + lo, hi - lo, d_out.ptr + lo)
                hip.dev_sync()
                out[lo:hi] = d_out.download(hi - lo, lo)
            if k + 2 < len(cuts):                                # hand over
                blob = c.stream_export()
                assert len(blob) == 64 and key not in blob and key[:8] not in blob
                ctxs[1].stream_import(blob)
                c.stream_final()                                 # the old context's session: closed with whatever tag it would have had
                ctxs[0], ctxs[1] = ctxs[1], ctxs[0]
        tag = ctxs[0].stream_final()
        assert bytes(out) == want and tag == want_tag, (dec, n, cuts)


def test_refusals(hip, orc):
    key, other_key, iv = splitmix_bytes(8100, 32), splitmix_bytes(8101, 32), splitmix_bytes(8102, 12)
    a, b, c = hip.Context(key), hip.Context(key), hip.Context(other_key)
    with pytest.raises(hip.AesGcmError) as ei:
        a.stream_export()                                        # no session
    assert ei.value.code == hip.ESTATE
    a.stream_begin(iv)
    a.stream_aad(b"header")
    a.stream_update(bytes(64))
    blob = a.stream_export()
    with pytest.raises(hip.AesGcmError) as ei:
        c.stream_import(blob)                                    # another key
    assert ei.value.code == hip.EARG
    for i in (0, 1, 20, 40, 63):
        bad = bytearray(blob); bad[i] ^= 0x10
        with pytest.raises(hip.AesGcmError) as ei:
            b.stream_import(bytes(bad))                          # damaged
        assert ei.value.code == hip.EARG
    b.stream_import(blob)
    with pytest.raises(hip.AesGcmError) as ei:
        b.stream_import(blob)                                    # a session is open
    assert ei.value.code == hip.ESTATE
    with pytest.raises(hip.AesGcmError) as ei:
        b.stream_aad(b"more")                                    # the imported state knows that data has begun
    assert ei.value.code == hip.ESTATE
    d = hip.DeviceBuffer(64)
    with pytest.raises(hip.AesGcmError) as ei:
        b.stream_update_dev(d.ptr + 4, 32, d.ptr)                # alignment
    assert ei.value.code == hip.EALIGN
    # both go on from the same point and agree with the oracle
    tail = splitmix_bytes(8103, 100)
    want_ct, want_tag = orc.Fast(key).encrypt(iv, b"header", bytes(64) + tail)
    for x in (a, b):
        assert x.stream_update(tail) == want_ct[64:] and x.stream_final() == want_tag
    a.stream_begin(iv)
    with pytest.raises(hip.AesGcmError) as ei:
        a.encrypt_pipelined(iv, b"", bytes(100))                 # the pipelined path keeps its state in the same slot: refused inside an open session
    assert ei.value.code == hip.ESTATE
    a.stream_final()
    assert a.encrypt_pipelined(iv, b"header", bytes(64) + tail) == (want_ct, want_tag)      # ... and is itself a session that closes behind it
    a.stream_begin(iv)
    a.stream_final()


CHILD = r"""
import json, sys
sys.path.insert(0, %(root)r)
import aesgcm_amd
from aesgcm_amd import lib
key, iv, aad = (bytes.fromhex(x) for x in sys.argv[1:4])
head = open(sys.argv[4], "rb").read()
c = lib.Context(key)
c.stream_begin(iv)
c.stream_aad(aad)
d = lib.DeviceBuffer(max(len(head), 16)); d.upload(head)
c.stream_update_dev(d.ptr, len(head), d.ptr)
blob = c.stream_export()
json.dump({"blob": blob.hex(), "ct": bytes(d.download(len(head))).hex()}, open(sys.argv[5], "w"))
"""


def test_across_two_processes_through_a_file(hip, orc, tmp_path):
    """process 1 absorbs the AAD and the first 1 MiB + 48 bytes on device pointers, writes the 64-byte state and its ciphertext to a file and exits; this process
    imports the state into a fresh context and finishes the message"""
    key, iv, aad = splitmix_bytes(8200, 32), splitmix_bytes(8201, 12), splitmix_bytes(8202, 28)
    pt = splitmix_bytes(8203, (1 << 20) + 48 + 70001)
    cut = (1 << 20) + 48
    path, head = str(tmp_path / "state.json"), str(tmp_path / "head.bin")
    with open(head, "wb") as fh:
        fh.write(pt[:cut])
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, key.hex(), iv.hex(), aad.hex(), head, path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    st = json.load(open(path))
    c = hip.Context(key)
    c.stream_import(bytes.fromhex(st["blob"]))
    rest = c.stream_update(pt[cut:])
    tag = c.stream_final()
    assert (bytes.fromhex(st["ct"]) + rest, tag) == orc.Fast(key).encrypt(iv, aad, pt)
