"""GPU: k_body as cyclic rows at its production sizes (4 .. 384 MiB): ONE launch takes the AAD (front rows of the row grid), data that starts at
any block (shards: head blocks in the front rows), the whole rows of the body and the partial last row -- against the oracle on the same inputs,
bit for bit, through the C ABI; encrypt, decrypt, in place, shards from odd first blocks, streaming chunks, and the ranges the layout refuses
(more AAD than one front row per strand) which must fall back to the other paths with the same result."""
import pytest

from util import splitmix_bytes

pytestmark = pytest.mark.gpu
MiB = 1 << 20


def _check(hip, f, ctx, iv, n, al, seed, aad_off=0):
    aad = splitmix_bytes(seed, al)
    d_in = hip.DeviceBuffer(n + 64); d_in.fill_splitmix64(seed + 1, 0, nbytes=n)
    d_out = hip.DeviceBuffer(n + 64)
    d_aad = hip.DeviceBuffer(al + 64)
    if al:
        d_aad.upload(bytes(aad_off) + aad)
    pt = bytes(d_in.download(n))
    want_ct, want_tag = f.encrypt(iv, aad, pt)
    tag = ctx.encrypt_dev(iv, d_in.ptr, n, d_out.ptr, d_aad=d_aad.ptr + aad_off if al else None, aad_len=al)
    assert tag == want_tag, ("enc tag", n, al)
    assert bytes(d_out.download(n)) == want_ct, ("enc ct", n, al)
    # decrypt in place, expected tag checked by the library
    tag2 = ctx.decrypt_dev(iv, d_out.ptr, n, d_out.ptr, d_aad=d_aad.ptr + aad_off if al else None, aad_len=al, tag=want_tag)
    assert tag2 == want_tag, ("dec tag", n, al)
    assert bytes(d_out.download(n)) == pt, ("dec pt", n, al)


@pytest.mark.parametrize("klen,close", [(16, 1), (24, 1), (32, 1), (16, 0), (24, 0), (32, 0)])
def test_whole_messages_of_any_shape_take_the_cyclic_launch(hip, orc, klen, close):
    """close = 1: the launch closes the tag itself (cyc_close: tree per workgroup, atomics across; what ships); 0: k_fold and k_combine behind it
    (the context option "cyc_close"; the form shards and streaming chunks always take)"""
    key, iv = splitmix_bytes(9100 + klen, klen), splitmix_bytes(9101, 12)
    ctx, f = hip.Context(key), orc.Fast(key)
    ctx.set_option("cyc_close", close)
    shapes = [(4 * MiB, 0), (4 * MiB + 5, 20), (5 * MiB - 3, 1000), (6 * MiB + 1023, 16 * 64), (7 * MiB + 16, 1), (9 * MiB + 1008, 16 * 64 * 3 + 7)]
    if klen == 32:
        shapes += [(16 * MiB + 1, 13), (33 * MiB - 17, 68), (64 * MiB + 4096 + 15, 4095)]
    for k, (n, al) in enumerate(shapes):
        head, body = ctx.split(n)
        assert head == 0 and body == (n // 1024) * 64, (n, head, body)           # every whole row is body: the cyclic launch took it
        _check(hip, f, ctx, iv, n, al, 9200 + 10 * k, aad_off=(k % 3) * 4)      # AAD pointer 16-, 4- and 8-byte aligned


@pytest.mark.parametrize("klen", [16, 24, 32])
def test_half_shape_of_the_cyclic_launch(hip, orc, klen):
    """context option "cyc_half": whole messages as k_bodyh -- 256 workgroups of 512 lanes, 2048 strands with the stride H^(2^17), the two-table round, eight
    items per workgroup in the closing (three tree levels, weights H^(512 (255 - g))) -- for callers with several messages in flight; every shape the full
    launch is tested on, AAD front rows (up to one per strand: 2 MiB), ragged ends, decrypt in place, and two contexts side by side on their own streams"""
    key, iv = splitmix_bytes(9150 + klen, klen), splitmix_bytes(9151, 12)
    ctx, f = hip.Context(key), orc.Fast(key)
    ctx.set_option("cyc_half", 1)
    shapes = [(64 * 1024, 0), (64 * 1024 + 1, 20), (96 * 1024 + 1013, 4095), (MiB + 5, 0), (2 * MiB + 1024 * 3 + 16, 16), (4 * MiB + 5, 20), (5 * MiB - 3, 1000),
              (6 * MiB + 1023, 16 * 64), (7 * MiB + 16, 1), (9 * MiB + 1008, 16 * 64 * 3 + 7), (3 * MiB, 2 * MiB - 64), (3 * MiB + 7, 2 * MiB + 4096)]
    if klen == 32:
        shapes += [(16 * MiB + 1, 13), (33 * MiB - 17, 68), (64 * MiB + 4096 + 15, 4095)]
    for k, (n, al) in enumerate(shapes):
        _check(hip, f, ctx, iv, n, al, 9250 + 10 * k, aad_off=(k % 3) * 4)
    # two contexts on their own streams, queued (tag = NULL), four launches each in flight
    c2 = hip.Context(key).set_option("cyc_half", 1)
    n = 3 * MiB + 4096 + 7
    d_in = hip.DeviceBuffer(n + 16); d_in.fill_splitmix64(9610 + klen, 0, nbytes=n)
    pt = bytes(d_in.download(n))
    o1, o2 = hip.DeviceBuffer(n + 16), hip.DeviceBuffer(n + 16)
    ivs = [splitmix_bytes(9630 + k, 12) for k in range(4)]
    for k, v in enumerate(ivs):
        ctx.encrypt_dev(v, d_in.ptr, n, o1.ptr, want_tag=False)
        c2.encrypt_dev(v, d_in.ptr, n - 4096 * (k & 1), o2.ptr, want_tag=False)
    t1, t2 = ctx.last_tag(), c2.last_tag()
    w1, w2 = f.encrypt(ivs[-1], b"", pt), f.encrypt(ivs[-1], b"", pt[:n - 4096])
    assert t1 == w1[1] and t2 == w2[1]
    assert bytes(o1.download(n)) == w1[0] and bytes(o2.download(n - 4096)) == w2[0]


def test_aad_longer_than_the_front_rows_falls_back(hip, orc):
    """more than 4096 front rows (4 MiB of AAD) do not fit one per strand: the range goes the other way, with the same tag"""
    key, iv = splitmix_bytes(9300, 32), splitmix_bytes(9301, 12)
    ctx, f = hip.Context(key), orc.Fast(key)
    _check(hip, f, ctx, iv, 4 * MiB + 100, 4 * MiB + 4096 + 9, 9310)
    _check(hip, f, ctx, iv, 4 * MiB + 100, 4 * MiB - 64, 9320)                # 4095.9 front rows: still the cyclic launch


def test_shards_and_streaming_chunks_from_odd_first_blocks(hip, orc):
    key, iv = splitmix_bytes(9400, 32), splitmix_bytes(9401, 12)
    ctx, f = hip.Context(key), orc.Fast(key)
    n, al, ranks = 23 * MiB + 11, 33, 3
    aad, pt = splitmix_bytes(9402, al), splitmix_bytes(9403, n)
    want_ct, want_tag = f.encrypt(iv, aad, pt)
    din, dout = hip.DeviceBuffer(n + 16), hip.DeviceBuffer(n + 16)
    din.upload(pt)
    d_aad = hip.DeviceBuffer(al); d_aad.upload(aad)
    parts = hip.DeviceBuffer(16 * ranks)
    total_blocks, first = (n + 15) // 16, 0
    for r in range(ranks):
        blocks = total_blocks // ranks + (1 if r < total_blocks % ranks else 0) + (7 if r == 0 else -7 if r == 1 else 0)     # odd cuts
        end = first + blocks
        ln = (n if end == total_blocks else 16 * end) - 16 * first
        sh, sb = ctx.split(ln, first)
        assert sb > 0 and sh == (-first) % 256, (first, sh, sb)
        ctx.shard_crypt_dev(False, iv, din.ptr + 16 * first, ln, dout.ptr + 16 * first, first, n, parts.ptr + 16 * r,
                            d_aad=d_aad.ptr if r == 0 else None, aad_len=al if r == 0 else 0)
        first = end
    assert ctx.shard_finalize_dev(iv, parts.ptr, ranks, al, n) == want_tag
    assert bytes(dout.download(n)) == want_ct
    # streaming: AAD, then data in chunks of 5 MiB + 48 (every chunk a cyclic launch with a carried state), decrypt direction too
    for dec in (False, True):
        src = want_ct if dec else pt
        ctx.stream_begin(iv, decrypt=dec)
        ctx.stream_aad(aad)
        out, step = [], 5 * MiB + 48
        for off in range(0, n, step):
            out.append(ctx.stream_update(src[off:off + step]))
        assert ctx.stream_final() == want_tag, dec
        assert b"".join(out) == (pt if dec else want_ct), dec


def test_back_to_back_messages_do_not_see_each_other(hip, orc):
    """the 4097 item slots are reused by every launch: alternate two messages of different shapes many times"""
    key = splitmix_bytes(9500, 16)
    ctx, f = hip.Context(key), orc.Fast(key)
    cases = []
    for k, (n, al) in enumerate(((4 * MiB + 77, 5), (6 * MiB, 0))):
        iv = splitmix_bytes(9501 + k, 12)
        aad = splitmix_bytes(9510 + k, al)
        d_in = hip.DeviceBuffer(n + 16); d_in.fill_splitmix64(9520 + k, 0, nbytes=n)
        d_aad = hip.DeviceBuffer(al + 16); d_aad.upload(aad)
        want = f.encrypt(iv, aad, bytes(d_in.download(n)))
        cases.append((iv, n, al, d_in, d_aad, want))
    d_out = hip.DeviceBuffer(6 * MiB + 64)
    for it in range(60):
        iv, n, al, d_in, d_aad, want = cases[it & 1]
        assert ctx.encrypt_dev(iv, d_in.ptr, n, d_out.ptr, d_aad=d_aad.ptr if al else None, aad_len=al) == want[1], it
    assert bytes(d_out.download(cases[1][1])) == cases[1][5][0]


def test_queued_launches_and_two_contexts(hip, orc):
    """tag = NULL: the calls only enqueue -- cyclic launches queue up behind each other on the stream (each leaves the accumulators zeroed for the
    next), two contexts on their own streams run side by side; the tags are read afterwards with aesgcm_last_tag"""
    k1, k2 = splitmix_bytes(9600, 32), splitmix_bytes(9601, 16)
    c1, c2 = hip.Context(k1), hip.Context(k2)
    f1, f2 = orc.Fast(k1), orc.Fast(k2)
    n = 3 * MiB + 4096 + 7
    d_in = hip.DeviceBuffer(n + 16); d_in.fill_splitmix64(9610, 0, nbytes=n)
    pt = bytes(d_in.download(n))
    o1, o2 = hip.DeviceBuffer(n + 16), hip.DeviceBuffer(n + 16)
    ivs = [splitmix_bytes(9620 + k, 12) for k in range(12)]
    for k, iv in enumerate(ivs):                                    # 12 launches per context in flight, nothing waited for
        c1.encrypt_dev(iv, d_in.ptr, n, o1.ptr, want_tag=False)
        c2.encrypt_dev(iv, d_in.ptr, n - 4096 * (k & 1), o2.ptr, want_tag=False)
    t1, t2 = c1.last_tag(), c2.last_tag()
    w1, w2 = f1.encrypt(ivs[-1], b"", pt), f2.encrypt(ivs[-1], b"", pt[:n - 4096])
    assert t1 == w1[1] and t2 == w2[1]
    assert bytes(o1.download(n)) == w1[0] and bytes(o2.download(n - 4096)) == w2[0]
    # and a waited call right behind them sees clean accumulators
    assert c1.encrypt_dev(ivs[0], d_in.ptr, n, o1.ptr) == f1.encrypt(ivs[0], b"", pt)[1]


def test_ciphertext_is_in_memory_when_the_tag_is(hip):
    """examples/early_read: a copy ordered behind nothing (its own non-blocking stream), issued the moment aesgcm_encrypt_dev / aesgcm_decrypt_dev return,
    reads the result of the call -- the in-launch tag of the cyclic rows appears only after every row has gone through the L2 to memory.  About 4200 calls:
    encrypt and decrypt, in place and not, AAD front rows and ragged byte stores, 64 KiB .. 1.17 GiB (the top of the cyclic range), and the half shape
    of the launch (k_bodyh) up to 79 MiB"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "examples"), "-s", "early_read"], check=True)
    r = subprocess.run([os.path.join(root, "examples", "early_read"), "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0 and "EARLY READ OK" in r.stdout, (r.stdout, r.stderr)
    assert r.stdout.count(": 0 of ") == 25, r.stdout


def test_many_random_shapes_back_to_back(hip, orc):
    """400 messages of random shape (64 KiB .. 6 MiB, any AAD up to 3 KiB, odd lengths) through two contexts in alternation: every tag against the oracle,
    every tenth ciphertext in full -- the accumulator slots, the arrival counter and the generation number of the host slot turn over 400 times"""
    import random
    rng = random.Random(31337)
    keys = [splitmix_bytes(9700, 16), splitmix_bytes(9701, 32)]
    ctxs = [hip.Context(k) for k in keys]
    fast = [orc.Fast(k) for k in keys]
    nmax = 6 * MiB
    d_in = hip.DeviceBuffer(nmax + 64); d_in.fill_splitmix64(9710, 0, nbytes=nmax)
    pt_all = bytes(d_in.download(nmax))
    d_out = hip.DeviceBuffer(nmax + 64)
    d_aad = hip.DeviceBuffer(4096)
    for it in range(400):
        k = it & 1
        n = rng.choice((rng.randint(64 << 10, nmax), rng.randint(64 << 10, 1 << 20), 1024 * rng.randint(64, 6144), 1024 * rng.randint(64, 6144) + rng.choice((1, 15, 16, 17, 1008, 1023))))
        n = min(n, nmax)
        al = rng.choice((0, 0, 1, 13, 16, 20, 64, 1000, 1024, rng.randint(1, 3072)))
        off = rng.randrange(0, nmax - n + 1, 16)
        aad = splitmix_bytes(9800 + it, al)
        if al:
            d_aad.upload(aad)
        iv = splitmix_bytes(10800 + it, 12)
        tag = ctxs[k].encrypt_dev(iv, d_in.ptr + off, n, d_out.ptr, d_aad=d_aad.ptr if al else None, aad_len=al)
        want_ct, want_tag = fast[k].encrypt(iv, aad, pt_all[off:off + n])
        assert tag == want_tag, (it, n, al, off)
        if it % 10 == 0:
            assert bytes(d_out.download(n)) == want_ct, (it, n, al, off)
