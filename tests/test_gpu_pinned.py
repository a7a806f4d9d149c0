"""GPU: the host-buffer entry points are synchronous even when the output buffer is page-locked.

aesgcm_encrypt / aesgcm_decrypt copy the result back with hipMemcpyAsync.  With pageable memory ROCm makes that copy
synchronous; with a buffer from the library's own aesgcm_host_alloc (what include/aesgcm.h recommends) it is truly
asynchronous, and the tag's generation number is published by k_combine BEFORE the copy starts -- so a call that only
polls for the tag can return while `ct` is still landing (round-2 verdict, Weak 2).  The reference model returns data
synchronously (tb/gcm_model.py:26).  Here: PinnedBuffer views as `out=`, 200 calls per size, the result compared with
the oracle IMMEDIATELY on return; consecutive calls alternate between two IVs so that a late copy shows up as the
previous call's bytes.
"""
import pytest

from util import splitmix_bytes

pytestmark = pytest.mark.gpu

SIZES = [64 << 10, 4 << 20, 64 << 20]


@pytest.mark.parametrize("n", SIZES)
def test_pinned_output_is_complete_on_return(hip, orc, n):
    import numpy as np
    key = splitmix_bytes(0x9101, 32)
    ivs = [splitmix_bytes(0x9102, 12), splitmix_bytes(0x9103, 12)]
    aad = splitmix_bytes(0x9104, 20)
    pt = np.frombuffer(orc.fill_splitmix64(n, 0x9105), dtype=np.uint8)
    f = orc.Fast(key)
    want = []
    for iv in ivs:
        ct = np.empty_like(pt)
        _, tag = f.crypt(False, iv, aad, pt, ct)
        want.append((ct, tag))
    ctx = hip.Context(key)
    out = hip.PinnedBuffer(n)
    back = hip.PinnedBuffer(n)
    view = np.frombuffer(out.view, dtype=np.uint8)
    bview = np.frombuffer(back.view, dtype=np.uint8)
    tail = min(n, 1 << 16)
    for it in range(200):
        k = it & 1
        _, tag = ctx.encrypt(ivs[k], aad, pt, out=out.view)
        # the end of the buffer lands last: look there first, then at everything
        assert np.array_equal(view[n - tail:], want[k][0][n - tail:]), "iteration %d: ciphertext tail not there on return" % it
        assert tag == want[k][1]
        if it % 20 == 0 or n <= (4 << 20):
            assert np.array_equal(view, want[k][0]), "iteration %d: ciphertext incomplete on return" % it
        # decrypt of the other IV's ciphertext into the second pinned buffer (alternating contents as well)
        _, t2 = ctx.decrypt(ivs[k], aad, want[k][0], tag=want[k][1], out=back.view)
        assert np.array_equal(bview[n - tail:], pt[n - tail:]), "iteration %d: plaintext tail not there on return" % it
        assert t2 == want[k][1]
        if it % 20 == 0 or n <= (4 << 20):
            assert np.array_equal(bview, pt)
        bview[n - tail:] = 0                      # the next decrypt must rewrite it
    ctx.close()
    view = bview = None
    out.free(); back.free()
