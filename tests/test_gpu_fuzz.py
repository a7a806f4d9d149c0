"""GPU: randomized differential test against the oracle -- key size, AAD/data lengths biased to block, row (64 blocks),
chunk and grid boundaries, chunk-size overrides (so both fold paths and ragged first chunks are hit), device
pointer paths with misaligned AAD, in-place, decrypt, shard splits at random block boundaries."""
import os
import random

import pytest

from util import splitmix_bytes

pytestmark = pytest.mark.gpu

# AESGCM_FUZZ_SEED=k re-draws every randomized test below (shapes, lengths AND the key / IV / data streams); AESGCM_FUZZ_SCALE multiplies the iteration
# counts.  The suite runs k = 0, scale 1; profiles/archive/runs/r04_run72.sh, 73, 78 went through k = 1 .. 32.
SEED = int(os.environ.get("AESGCM_FUZZ_SEED", "0"))
SCALE = float(os.environ.get("AESGCM_FUZZ_SCALE", "1"))
S0 = 1000003 * SEED                                    # offset of the SplitMix64 stream seeds


def _len(rng, cap):
    kind = rng.random()
    if kind < 0.15:
        return rng.choice((0, 1, 15, 16, 17, 31, 32, 33))
    if kind < 0.55:                                   # around row / chunk multiples
        base = rng.choice((64, 128, 1024, 2048, 16384, 65536)) * 16 * rng.randint(1, 6)
        return max(0, min(cap, base + rng.randint(-40, 40)))
    return int(rng.betavariate(.3, .3) * cap)


def test_fuzz_one_shot_and_device_paths(hip, orc):
    rng = random.Random(20260101 + SEED)
    if True:
        for it in range(int(220 * SCALE)):
            klen = rng.choice((16, 24, 32))
            tw = rng.choice((None, None, 1, 2, 3, 5, 16, 64))
            key, iv = splitmix_bytes(S0 + 9000 + it, klen), splitmix_bytes(S0 + 9500 + it, 12)
            al = _len(rng, 1 << 16) if rng.random() < 0.7 else 0
            n = _len(rng, 6 << 20)
            aad, pt = splitmix_bytes(S0 + 10000 + it, al), splitmix_bytes(S0 + 11000 + it, n)
            f = orc.Fast(key)
            want = f.encrypt(iv, aad, pt)
            ctx = hip.Context(key)
            if tw is not None:
                ctx.set_option("tw", tw)                              # rows per chunk of the dealt kernels
            # every other context keeps to k_main / k_fold / k_combine (the paths of messages below 64 KiB and of long AAD); the others take the
            # cyclic launch with its fused closing from 64 KiB
            if it & 1:
                ctx.set_option("cyc_min", 0).set_option("cyc_max", 0)
            mode = rng.choice(("host", "dev", "inplace", "pipe"))
            if mode == "host":
                got = ctx.encrypt(iv, aad, pt)
            elif mode == "pipe":
                got = ctx.encrypt_pipelined(iv, aad, pt, chunk_bytes=rng.choice((0, 1 << 16, 1 << 20)))
            else:
                d_in = hip.DeviceBuffer(n + 32); d_in.upload(pt)
                d_out = d_in if mode == "inplace" else hip.DeviceBuffer(n + 32)
                shift = rng.choice((0, 1, 7, 16))
                d_aad = hip.DeviceBuffer(al + 64); d_aad.upload(aad, offset=shift)
                tag = ctx.encrypt_dev(iv, d_in.ptr, n, d_out.ptr, d_aad=d_aad.ptr + shift if al else None, aad_len=al)
                got = (bytes(d_out.download(n)), tag)
            assert got == want, (it, klen, al, n, tw, mode)
            back, t2 = ctx.decrypt(iv, aad, want[0], tag=want[1])
            assert back == pt and t2 == want[1], (it, "dec")
            ctx.close()


def test_fuzz_random_shard_splits(hip, orc):
    rng = random.Random(77 + SEED)
    for it in range(int(40 * SCALE)):
        klen = rng.choice((16, 24, 32))
        key, iv = splitmix_bytes(S0 + 12000 + it, klen), splitmix_bytes(S0 + 12500 + it, 12)
        al, n = rng.choice((0, 13, 64, 1000)), _len(rng, 3 << 20)
        aad, pt = splitmix_bytes(S0 + 13000 + it, al), splitmix_bytes(S0 + 14000 + it, n)
        want = orc.Fast(key).encrypt(iv, aad, pt)
        ctx = hip.Context(key)
        nb = (n + 15) // 16
        ranks = rng.randint(1, 8)
        cuts = sorted(rng.randint(0, nb) for _ in range(ranks - 1))
        bounds = [0] + cuts + [nb]                                    # arbitrary (possibly empty) shards
        d_in, d_out = hip.DeviceBuffer(n + 32), hip.DeviceBuffer(n + 32)
        d_in.upload(pt)
        d_aad = hip.DeviceBuffer(al + 16); d_aad.upload(aad)
        parts = hip.DeviceBuffer(16 * ranks)
        for r in range(ranks):
            first, end = bounds[r], bounds[r + 1]
            ln = max(0, (n if end == nb else 16 * end) - 16 * first)
            ctx.shard_crypt_dev(False, iv, d_in.ptr + 16 * first, ln, d_out.ptr + 16 * first, first, n, parts.ptr + 16 * r,
                                d_aad=d_aad.ptr if (first == 0 and r == 0 and al) else None, aad_len=al if (first == 0 and r == 0) else 0)
        # the AAD belongs to the shard that starts at block 0; if rank 0 is empty the next rank starting at 0 must not re-add it
        tag = ctx.shard_finalize_dev(iv, parts.ptr, ranks, al, n)
        assert tag == want[1], (it, ranks, bounds, al, n)
        assert bytes(d_out.download(n)) == want[0]


@pytest.mark.parametrize("body", [False, True, "cyc"])
def test_fold_level_boundaries(hip, orc, body):
    """k_fold reduces up to 128 items per workgroup (8 waves x 1..16 items, fold_group) and k_combine folds the last 64:
    chunk counts on both sides of every boundary of that scheme (64, 128, 512 g for g = 1..16, 16 x 8192, 65536), with
    one-row chunks so that the count is the row count; once through k_main alone, once with the k_body cut forced
    (interleaved items, period-4 first level), once with the aligned middle as cyclic rows (k_body<.., true>: strands of 0 .. 17 rows,
    always 4096 items, rotated by the row count)."""
    key, iv = splitmix_bytes(4201, 32), splitmix_bytes(4202, 12)
    ctx, f = hip.Context(key), orc.Fast(key)
    ctx.set_option("tw", 1).set_option("body_min", 4096 if body is True else 1 << 59)
    ctx.set_option("cyc_min", 4096 if body == "cyc" else 0).set_option("cyc_max", 1 << 50 if body == "cyc" else 0)
    counts = (1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097, 8191, 8192, 8193,
              16383, 16384, 16385, 65535, 65536, 65537)
    for rows in counts:
        for extra, al in ((0, 0), (5, 20)):
            n = rows * 1024 + extra - (1024 if extra else 0) + (16 if extra else 0)      # ragged variant: one block + 5 bytes less/more
            n = max(n, 1)
            aad = splitmix_bytes(4300 + rows, al)
            d_in = hip.DeviceBuffer(n + 32); d_in.fill_splitmix64(4400 + rows, 0, nbytes=n)
            pt = bytes(d_in.download(n))
            want = f.encrypt(iv, aad, pt)
            d_aad = hip.DeviceBuffer(max(al, 16)); d_aad.upload(aad)
            tag = ctx.encrypt_dev(iv, d_in.ptr, n, d_in.ptr, d_aad=d_aad.ptr if al else None, aad_len=al)   # in place
            assert tag == want[1], (rows, n, al, body)
            assert bytes(d_in.download(n)) == want[0], (rows, n, al, body)
            if body and n >= 16 * 256 * 4:
                assert ctx.split(n)[1] > 0


def test_fuzz_packets_and_batches(hip, orc):
    """Randomized: many packets under one key (aesgcm_packets_crypt_dev) and with a key each (aesgcm_batch_crypt_var_dev) -- count, length mix (empty, ragged,
    MACsec-sized, a few long ones), AAD, fixed records or offset arrays with aligned or arbitrary starts, every kernel shape (forced through the debug build) or
    the library's own, taken in array order or by length class; encrypt, then decrypt in place with a few forged tags.  A sample of packets against the oracle."""
    import struct
    rng = random.Random(20260102 + SEED)
    shapes = {"wave": 64, "group16": 16, "g8": 8, "g4": 4, "lane": 1}

    def up(b):
        d = hip.DeviceBuffer(max(len(b), 16)); d.upload(b); return d
    with hip.debug_library() as dbg:
        for it in range(int(150 * SCALE)):
            per_key = it % 3 == 2
            klen = rng.choice((16, 24, 32))
            m = rng.choice((1, 7, 63, 64, 65, 300, 1000, 4097, 9000) + ((20000, 120000, 140000) if SEED else ()))      # re-seeded runs also cross the counts where the
            #                                                                 library itself changes shape (one lane per packet, its ILP form, the ordering)
            mix = rng.choice(("macsec", "tiny", "ragged", "long"))
            def one():
                if mix == "macsec":
                    return rng.choice((0, 46, 64, 128, 500, 1000, 1500, 1514, rng.randrange(0, 1515)))
                if mix == "tiny":
                    return rng.choice((0, 1, 15, 16, 17, 31, 32, 48))
                if mix == "ragged":
                    return rng.randrange(0, 700)
                return rng.choice((64, 1000, 4096, 9000, 70000 if m <= 300 else 5000)) if m <= 9000 else rng.choice((64, 1000, 2048))
            align = rng.choice((True, False))
            lens = [one() for _ in range(m)]
            if align:
                lens = [x // 16 * 16 for x in lens]
            aads = [rng.choice((0, 0, 8, 20, 28, 41)) for _ in range(m)]
            doff, aoff = [0], [0]
            for a, b in zip(lens, aads):
                doff.append(doff[-1] + a); aoff.append(aoff[-1] + b)
            key = splitmix_bytes(S0 + 30000 + it, klen)
            keys = splitmix_bytes(S0 + 31000 + it, klen * m) if per_key else key * m
            ivs, aad, pt = splitmix_bytes(S0 + 32000 + it, 12 * m), splitmix_bytes(S0 + 33000 + it, max(aoff[-1], 16)), splitmix_bytes(S0 + 34000 + it, max(doff[-1], 16))
            order = rng.choice((0, 1))
            shape = rng.choice((None,) + tuple(shapes))
            d_ivs, d_aad, d_buf = up(ivs), up(aad), up(pt)
            d_doff, d_aoff = up(struct.pack("<%dQ" % (m + 1), *doff)), up(struct.pack("<%dQ" % (m + 1), *aoff))
            d_tags, d_auth = hip.DeviceBuffer(16 * m), hip.DeviceBuffer(4 * m)
            if per_key:
                d_keys = up(keys)
                dbg.force(batch_lanes={None: 0, "wave": 64, "group16": 16, "g8": 8, "g4": 8, "lane": 16}[shape], batch_order=2 - order)

                def crypt(dec, d_exp=None):
                    hip.batch_crypt_var_dev(dec, m, klen, d_keys.ptr, d_ivs.ptr, d_buf.ptr, d_doff.ptr, d_buf.ptr, d_tags.ptr, d_aad=d_aad.ptr, d_aad_off=d_aoff.ptr,
                                            d_expect_tags=d_exp.ptr if d_exp else None, d_auth=d_auth.ptr if d_exp else None)
            else:
                ctx = hip.Context(key).set_option("pkt_order", order)
                dbg.force(pkt_lanes=shapes.get(shape, 0), batch_lanes=0, batch_order=0)

                def crypt(dec, d_exp=None):
                    ctx.packets_crypt_dev(dec, m, d_ivs.ptr, d_buf.ptr, d_buf.ptr, d_tags.ptr, d_data_off=d_doff.ptr, d_aad=d_aad.ptr, d_aad_off=d_aoff.ptr,
                                          d_expect_tags=d_exp.ptr if d_exp else None, d_auth=d_auth.ptr if d_exp else None)
            crypt(False)
            hip.dev_sync()
            ct, tags = bytes(d_buf.download(max(doff[-1], 16)))[:doff[-1]], bytes(d_tags.download())
            what = (it, "key each" if per_key else "one key", klen, m, mix, align, order, shape)
            for p in sorted(set(rng.sample(range(m), min(m, 120))) | {0, m - 1}):
                f = orc.Fast(keys[klen * p:klen * (p + 1)])
                want = f.encrypt(ivs[12 * p:12 * p + 12], aad[aoff[p]:aoff[p + 1]], pt[doff[p]:doff[p + 1]])
                assert (ct[doff[p]:doff[p + 1]], tags[16 * p:16 * p + 16]) == want, what + (p, lens[p], aads[p])
            forged = sorted(set(rng.sample(range(m), min(m, 3))))
            bad = bytearray(tags)
            for p in forged:
                bad[16 * p + rng.randrange(16)] ^= 1 << rng.randrange(8)
            d_exp = up(bytes(bad))
            crypt(True, d_exp)
            hip.dev_sync()
            assert bytes(d_buf.download(max(doff[-1], 16)))[:doff[-1]] == pt[:doff[-1]], what
            auth = struct.unpack("<%di" % m, bytes(d_auth.download()))
            assert [i for i, a in enumerate(auth) if not a] == forged, what
