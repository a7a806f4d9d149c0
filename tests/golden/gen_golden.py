#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run in the BUILD container, not on the GPU box).

    PYTHONDONTWRITEBYTECODE=1 python3 -B tests/golden/gen_golden.py [--large] [--huge]

Sources of truth, in the order the fixtures use them:
  * the reference's own pure-Python key schedule, imported from /root/reference/tb/key_exp.py
    (tb/key_exp.py:79-121) -> key_schedule.json, sbox.json;
  * the reference's two directed vectors (README.md:251, README.md:257: INPUTS only) with the
    outputs the linked IEEE 802.1AE document publishes, re-derived here by two independent builds
    of the published algorithm: system libcrypto (OpenSSL 3.0.x) through ctypes and node's bundled
    OpenSSL 1.1.1 -> kat.json;
  * the SP 800-38D / McGrew-Viega GCM specification test cases (published expected values typed in
    below and asserted against both libraries at generation time) -> kat.json;
  * seeded synthetic inputs (SplitMix64) through libcrypto, cross-checked with node
    for the small ones -> length_matrix.json, streams.json, shards.json, batch.json.
Nothing from the reference is copied: fixtures hold inputs/seeds and expected outputs only.
"""
import argparse
import ctypes
import hashlib
import json
import os
import subprocess
import sys
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import libcrypto_ref as R  # noqa: E402
from oracle import oracle as O         # noqa: E402  (only its SplitMix64 filler is used here)

REF_TB = "/root/reference/tb"

MASK64 = (1 << 64) - 1


def splitmix_bytes(seed, n_bytes, first_word=0):
    return bytes(O.fill_splitmix64(n_bytes, seed, first_word))


def dump(name, obj):
    path = os.path.join(HERE, name)
    with open(path, "w") as f:
        json.dump(obj, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", name, os.path.getsize(path), "bytes")


# ------------------------------------------------------------------ node cross-check (O2)
NODE_SRC = r"""
const crypto = require('crypto');
let raw = ''; process.stdin.on('data', d => raw += d); process.stdin.on('end', () => {
  const cases = JSON.parse(raw); const out = [];
  for (const c of cases) {
    const key = Buffer.from(c.key, 'hex'), iv = Buffer.from(c.iv, 'hex');
    const ci = crypto.createCipheriv('aes-' + (key.length * 8) + '-gcm', key, iv);
    ci.setAAD(Buffer.from(c.aad, 'hex'));
    const ct = Buffer.concat([ci.update(Buffer.from(c.pt, 'hex')), ci.final()]);
    out.push({ct: ct.toString('hex'), tag: ci.getAuthTag().toString('hex')});
  }
  process.stdout.write(JSON.stringify(out));
});
"""


def node_encrypt_many(cases):
    """cases: list of dict(key, iv, aad, pt) hex -> list of dict(ct, tag) hex, via node's OpenSSL."""
    p = subprocess.run(["node", "-e", NODE_SRC], input=json.dumps(cases).encode(), stdout=subprocess.PIPE, check=True)
    return json.loads(p.stdout.decode())


def both(key, iv, aad, pt):
    """libcrypto result, asserted equal to node's."""
    ct, tag = R.encrypt(key, iv, aad, pt)
    n = node_encrypt_many([dict(key=key.hex(), iv=iv.hex(), aad=aad.hex(), pt=pt.hex())])[0]
    assert n["ct"] == ct.hex() and n["tag"] == tag.hex(), "libcrypto and node disagree"
    return ct, tag


# ------------------------------------------------------------------ pure-python GF(2^128) (independent of oracle/)
def gf_mul_int(x, y):
    """SP 800-38D Algorithm 1 on 128-bit integers (bit 0 of the spec = MSB of the integer)."""
    R_ = 0xE1 << 120
    z, v = 0, y
    for i in range(127, -1, -1):
        if (x >> i) & 1:
            z ^= v
        v = (v >> 1) ^ (R_ if v & 1 else 0)
    return z


def gf_pow_int(h, e):
    r, b = 1 << 127, h
    while e:
        if e & 1:
            r = gf_mul_int(r, b)
        b = gf_mul_int(b, b)
        e >>= 1
    return r


def ghash_int(h, aad, ct):
    y = 0

    def eat(data):
        nonlocal y
        for off in range(0, len(data), 16):
            blk = data[off:off + 16].ljust(16, b"\0")
            y = gf_mul_int(y ^ int.from_bytes(blk, "big"), h)

    eat(aad)
    eat(ct)
    y = gf_mul_int(y ^ ((len(aad) * 8) << 64 | (len(ct) * 8)), h)
    return y


def ecb_block(key, block):
    """One AES-ECB block through libcrypto (to obtain H and E(J0) independently of oracle/)."""
    L = R._load()
    ciph = {16: "EVP_aes_128_ecb", 24: "EVP_aes_192_ecb", 32: "EVP_aes_256_ecb"}[len(key)]
    getattr(L, ciph).restype = ctypes.c_void_p
    ctx = L.EVP_CIPHER_CTX_new()
    assert L.EVP_CipherInit_ex(ctx, getattr(L, ciph)(), None, key, None, 1) == 1
    out = ctypes.create_string_buffer(32)
    n = ctypes.c_int(0)
    assert L.EVP_CipherUpdate(ctx, ctypes.addressof(out), ctypes.byref(n), block, 16) == 1
    L.EVP_CIPHER_CTX_free(ctx)
    return out.raw[:16]


# ------------------------------------------------------------------ fixtures
def gen_key_schedule_and_sbox():
    sys.path.insert(0, REF_TB)
    import key_exp  # the reference's own helper (pure Python)
    keys = {
        "fips197_a1_128": "2b7e151628aed2a6abf7158809cf4f3c",
        "fips197_a2_192": "8e73b0f7da0e6452c810f32b809079e562f8ead2522c6b7b",
        "fips197_a3_256": "603deb1015ca71be2b73aef0857d77811f352c073b6108d72d9810a30914dff4",
        "readme_251_128": "AD7A2BD03EAC835A6F620FDCB506B345",
        "readme_257_256": "691D3EE909D7F54167FD1CA0B5D769081F2BDE1AEE655FDBAB80BD5295AE6BE7",
        "readme_246_256": "92E11DCDAA866F5CE790FD24501F92509AACF4CB8B1339D50C9C1240935DD08B",
        "zero_128": "00" * 16, "zero_192": "00" * 24, "zero_256": "00" * 32,
        "ones_256": "ff" * 32,
    }
    for i in range(6):
        for nb in (16, 24, 32):
            keys["splitmix_%d_%d" % (nb * 8, i)] = splitmix_bytes(0x4B4559 + 101 * i + nb, nb).hex()
    out = {}
    for name, khex in keys.items():
        size = {32: "128", 48: "192", 64: "256"}[len(khex)]
        exp = key_exp.aes_expand_key(khex, size)
        out[name] = {"key": khex.lower(), "size": size, "expanded": bytes(exp).hex()}
    # FIPS-197 appendix A last round keys as a sanity anchor for the reference helper itself
    assert out["fips197_a1_128"]["expanded"][-32:] == "d014f9a8c9ee2589e13f0cc8b6630ca6"
    assert out["fips197_a3_256"]["expanded"][-32:] == "fe4890d1e6188d0b046df344706c631e"
    dump("key_schedule.json", {"source": "tb/key_exp.py:79-121 aes_expand_key", "vectors": out})
    dump("sbox.json", {"source": "tb/key_exp.py:23-54 exp_key.sbox", "sbox": bytes(key_exp.exp_key.sbox).hex()})
    sys.path.remove(REF_TB)


def gen_kat():
    kats = []

    def add(name, source, key, iv, aad, pt, ct=None, tag=None):
        key, iv, aad, pt = (bytes.fromhex(x) for x in (key, iv, aad, pt))
        c, t = both(key, iv, aad, pt)
        if ct is not None:
            assert c.hex() == ct.lower(), name
        if tag is not None:
            assert t.hex() == tag.lower(), name
        kats.append(dict(name=name, source=source, key=key.hex(), iv=iv.hex(), aad=aad.hex(), pt=pt.hex(),
                         ct=c.hex(), tag=t.hex(), published=ct is not None or tag is not None))

    # the reference's two directed vectors: inputs from README.md, outputs = IEEE 802.1AE published values
    add("readme_251_aes128", "README.md:251 (IEEE 802.1AE 2.2.1, 60-byte packet encryption GCM-AES-128)",
        "AD7A2BD03EAC835A6F620FDCB506B345", "12153524C0895E81B2C28465",
        "D609B1F056637A0D46DF998D88E52E00B2C2846512153524C0895E81",
        "08000F101112131415161718191A1B1C1D1E1F202122232425262728292A2B2C2D2E2F303132333435363738393A0002",
        "701AFA1CC039C0D765128A665DAB69243899BF7318CCDC81C9931DA17FBE8EDD7D17CB8B4C26FC81E3284F2B7FBA713D",
        "4F8D55E7D3F06FD5A13C0C29B9D5B880")
    add("readme_257_aes256", "README.md:257 (IEEE 802.1AE 2.3.2, 65-byte packet authentication GCM-AES-256)",
        "691D3EE909D7F54167FD1CA0B5D769081F2BDE1AEE655FDBAB80BD5295AE6BE7", "F0761E8DCD3D000176D457ED",
        "E20106D7CD0DF0761E8DCD3D88E5400076D457ED08000F101112131415161718191A1B1C1D1E1F202122232425262728292A2B2C2D2E2F303132333435363738393A0003",
        "", "", "35217C774BBC31B63166BCF9D4ABED07")
    # GCM specification test cases (McGrew & Viega; 96-bit-IV ones only: the RTL is 96-bit only, gcm_pkg.vhd:15-17)
    K3 = "feffe9928665731c6d6a8f9467308308"
    IV3 = "cafebabefacedbaddecaf888"
    P3 = ("d9313225f88406e5a55909c5aff5269a86a7a9531534f7da2e4c303d8a318a72"
          "1c3c0c95956809532fcf0e2449a6b525b16aedf5aa0de657ba637b391aafd255")
    A4 = "feedfacedeadbeeffeedfacedeadbeefabaddad2"
    spec = "GCM spec (McGrew-Viega) test case %d"
    add("gcmspec_tc1", spec % 1, "00" * 16, "00" * 12, "", "", "", "58e2fccefa7e3061367f1d57a4e7455a")
    add("gcmspec_tc2", spec % 2, "00" * 16, "00" * 12, "", "00" * 16, "0388dace60b6a392f328c2b971b2fe78",
        "ab6e47d42cec13bdf53a67b21257bddf")
    add("gcmspec_tc3", spec % 3, K3, IV3, "", P3,
        "42831ec2217774244b7221b784d0d49ce3aa212f2c02a4e035c17e2329aca12e"
        "21d514b25466931c7d8f6a5aac84aa051ba30b396a0aac973d58e091473f5985", "4d5c2af327cd64a62cf35abd2ba6fab4")
    add("gcmspec_tc4", spec % 4, K3, IV3, A4, P3[:120], None, "5bc94fbc3221a5db94fae95ae7121a47")
    add("gcmspec_tc7", spec % 7, "00" * 24, "00" * 12, "", "", "", "cd33b28ac773f74ba00ed1f312572435")
    add("gcmspec_tc8", spec % 8, "00" * 24, "00" * 12, "", "00" * 16, "98e7247c07f0fe411c267e4384b0f600",
        "2ff58d80033927ab8ef4d4587514f0fb")
    add("gcmspec_tc10", spec % 10, K3 + K3[:16], IV3, A4, P3[:120], None, None)
    add("gcmspec_tc13", spec % 13, "00" * 32, "00" * 12, "", "", "", "530f8afbc74536b9a963b4f1c4cb738b")
    add("gcmspec_tc14", spec % 14, "00" * 32, "00" * 12, "", "00" * 16, "cea7403d4d606b6e074ec5d3baf39d18",
        "d0d1c8a799996bf0265b98b5d48ab919")
    add("gcmspec_tc16", spec % 16, K3 + K3, IV3, A4, P3[:120], None, "76fc6ece0f4e1768cddf8853bb2d551b")
    dump("kat.json", {"oracles": [R.version(), "node " + subprocess.check_output(["node", "-p", "process.versions.openssl"]).decode().strip()],
                      "vectors": kats})


def gen_gfmul():
    """(H, X, Z) triples: Z from the independent big-int routine above, itself validated against
    libcrypto by recomputing whole GCM tags from H = E_K(0), E_K(J0) obtained through libcrypto ECB."""
    for seed, kl, al, pl in ((1, 16, 20, 100), (2, 24, 0, 33), (3, 32, 68, 0), (4, 32, 17, 4096)):
        key = splitmix_bytes(0x1000 + seed, kl)
        iv = splitmix_bytes(0x2000 + seed, 12)
        aad = splitmix_bytes(0x3000 + seed, al)
        pt = splitmix_bytes(0x4000 + seed, pl)
        ct, tag = R.encrypt(key, iv, aad, pt)
        h = int.from_bytes(ecb_block(key, b"\0" * 16), "big")
        ej0 = int.from_bytes(ecb_block(key, iv + b"\0\0\0\1"), "big")
        assert (ghash_int(h, aad, ct) ^ ej0).to_bytes(16, "big") == tag, "python GHASH disagrees with libcrypto"
    vec = []
    hs = [splitmix_bytes(0x5000 + i, 16) for i in range(4)] + [b"\x80" + b"\0" * 15, b"\0" * 15 + b"\x01", b"\xff" * 16]
    xs = [splitmix_bytes(0x6000 + i, 16) for i in range(6)]
    for bit in (0, 7, 8, 120, 127):   # GCM bit index (0 = MSB of byte 0)
        xs.append((1 << (127 - bit)).to_bytes(16, "big"))
    xs += [b"\0" * 16, b"\xff" * 16]
    for h in hs:
        for x in xs:
            z = gf_mul_int(int.from_bytes(x, "big"), int.from_bytes(h, "big"))
            vec.append(dict(h=h.hex(), x=x.hex(), z=z.to_bytes(16, "big").hex()))
    pows = []
    for h in hs[:3]:
        for e in (0, 1, 2, 3, 64, 65, 511, 512, 1 << 20, (1 << 32) - 3):
            pows.append(dict(h=h.hex(), e=e, z=gf_pow_int(int.from_bytes(h, "big"), e).to_bytes(16, "big").hex()))
    dump("gfmul.json", {"source": "SP 800-38D Algorithm 1 (= src/ghash_gfmul.vhd:37-64), python big-int, tag-validated against libcrypto",
                        "mul": vec, "pow": pows})


AAD_LENS = [0, 1, 15, 16, 17, 20, 28, 68, 4095]
PT_LENS = [0, 1, 15, 16, 17, 48, 255, 4096, 65535]


def matrix_inputs(kbits, al, pl):
    """Deterministic inputs of one length-matrix cell (shared with tests/util.py)."""
    tagv = (kbits << 40) | (al << 20) | pl
    key = splitmix_bytes(0xA0000000 + tagv, kbits // 8)
    iv = splitmix_bytes(0xB0000000 + tagv, 12)
    aad = splitmix_bytes(0xC0000000 + tagv, al)
    pt = splitmix_bytes(0xD0000000 + tagv, pl)
    return key, iv, aad, pt


def gen_length_matrix():
    cells, node_cases = [], []
    for kbits in (128, 192, 256):
        for al in AAD_LENS:
            for pl in PT_LENS:
                key, iv, aad, pt = matrix_inputs(kbits, al, pl)
                ct, tag = R.encrypt(key, iv, aad, pt)
                cells.append(dict(kbits=kbits, aad_len=al, pt_len=pl, tag=tag.hex(), ct_sha256=hashlib.sha256(ct).hexdigest(),
                                  ct_head=ct[:32].hex(), ct_tail=ct[-32:].hex()))
                node_cases.append(dict(key=key.hex(), iv=iv.hex(), aad=aad.hex(), pt=pt.hex()))
    for c, n in zip(cells, node_encrypt_many(node_cases)):
        assert n["tag"] == c["tag"] and hashlib.sha256(bytes.fromhex(n["ct"])).hexdigest() == c["ct_sha256"]
    dump("length_matrix.json", {"inputs": "SplitMix64 streams, seeds = 0x{A,B,C,D}0000000 + (kbits<<40 | aad_len<<20 | pt_len) for key/iv/aad/pt",
                                "cells": cells})


KEY_SEED, IV_SEED = 0x4B4559, 0x4956


def stream_case(name, kbytes, pt_seed, n_bytes, iv_tweak=0, first_word=0, aad=b"", chunk=64 << 20, threads=8):
    """libcrypto over a SplitMix64 plaintext stream generated chunk-wise; -> fixture dict."""
    key = splitmix_bytes(KEY_SEED, kbytes)
    iv = bytearray(splitmix_bytes(IV_SEED, 12))
    iv[11] = (iv[11] + iv_tweak) & 0xFF
    iv = bytes(iv)
    s = R.Stream(key, iv)
    s.aad(aad)
    sha = hashlib.sha256()
    head, tail = b"", b""
    done = 0
    import numpy as np
    ptbuf = np.empty(chunk, dtype=np.uint8)
    ctbuf = np.empty(chunk, dtype=np.uint8)
    while done < n_bytes:
        m = min(chunk, n_bytes - done)
        # parallel fill
        per = ((m + threads - 1) // threads + 7) // 8 * 8
        ths = []
        for t in range(threads):
            lo = t * per
            hi = min(m, lo + per)
            if lo >= hi:
                break
            th = threading.Thread(target=O.lib().orc_fill_splitmix64,
                                  args=(ptbuf.ctypes.data + lo, hi - lo, pt_seed, first_word + (done + lo) // 8))
            th.start()
            ths.append(th)
        for th in ths:
            th.join()
        s.update(ptbuf[:m], ctbuf[:m])
        sha.update(ctbuf[:m].data)
        if done == 0:
            head = bytes(ctbuf[:64])
        if done + m == n_bytes:
            tail = bytes(ctbuf[max(0, m - 64):m])
        done += m
    tag = s.final()
    print("  ", name, n_bytes, tag.hex())
    return dict(name=name, key_bytes=kbytes, key_seed=KEY_SEED, iv_seed=IV_SEED, iv_tweak=iv_tweak, pt_seed=pt_seed,
                first_word=first_word, n_bytes=n_bytes, aad=aad.hex(), tag=tag.hex(), ct_sha256=sha.hexdigest(),
                ct_head=head.hex(), ct_tail=tail.hex())


def gen_streams(large, huge):
    path = os.path.join(HERE, "streams.json")
    prev = {}
    if os.path.exists(path):
        prev = {c["name"]: c for c in json.load(open(path))["cases"]}
    cases = []

    def want(name, *a, **kw):
        if name in prev and not kw.pop("force", False):
            cases.append(prev[name])
        else:
            cases.append(stream_case(name, *a, **kw))

    MiB, GiB = 1 << 20, 1 << 30
    want("aes128_1MiB", 16, 0xAE5C0002, MiB)
    want("aes256_1MiB", 32, 0xAE5C0003, MiB)
    want("aes256_1MiB_aad20", 32, 0xAE5C0003, MiB, aad=splitmix_bytes(0x414144, 20))
    want("aes192_3MiB_plus5", 24, 0xAE5C0001, 3 * MiB + 5, aad=splitmix_bytes(0x414144, 37))
    want("aes256_64MiB", 32, 0xAE5C0003, 64 * MiB)
    want("aes256_64MiB_minus3", 32, 0xAE5C0003, 64 * MiB - 3)
    if large:
        want("cfg2_aes128_1GiB", 16, 0xAE5C0002, GiB)
        want("aes256_1GiB", 32, 0xAE5C0003, GiB)
        want("cfg3_aes256_16GiB", 32, 0xAE5C0003, 16 * GiB)
        # round 4: AES-192 at a size where the production split engages by itself (the dealt k_body<12, *> and FoldClose from 1 GiB), and the general
        # head / k_body / tail path with a carried state -- AAD in front, a ragged end -- at a size nobody forces (src/aes_pkg.vhd:31-33: three modes alike)
        want("aes192_1GiB", 24, 0xAE5C0007, GiB)
        want("aes256_4GiB_aad20_minus5", 32, 0xAE5C0008, 4 * GiB - 5, aad=splitmix_bytes(0x414144, 20))
    if huge:
        # cfg4: 128 GiB aggregate = 4 messages x 32 GiB of ONE SplitMix64 stream (seed 0xAE5C0004), IV last byte + m.
        for m in range(4):
            want("cfg4_aes256_msg%d_32GiB" % m, 32, 0xAE5C0004, 32 * GiB, iv_tweak=m, first_word=m * (32 * GiB // 8))
    if huge:
        # the largest message GCM (and the RTL's 32-bit block counter, src/aes_icb.vhd:114) allows: 2^36 - 32 bytes
        want("aes256_max_message", 32, 0xAE5C0006, (1 << 36) - 32, aad=splitmix_bytes(0x414144, 20))
    # keep previously generated big cases even when run without --large/--huge
    names = {c["name"] for c in cases}
    for n, c in prev.items():
        if n not in names:
            cases.append(c)
    dump("streams.json", {"inputs": "key = first key_bytes of SplitMix64(seed 0x4B4559); iv = first 12 bytes of SplitMix64(seed 0x4956) with "
                                    "last byte += iv_tweak; pt = SplitMix64(pt_seed) words [first_word, ...), little-endian",
                          "oracle": R.version(), "cases": cases})


def gen_shards():
    """Shard algebra fixture: 203 blocks + 5 bytes, 37-byte AAD, 8 shards."""
    key = splitmix_bytes(0x7001, 32)
    iv = splitmix_bytes(0x7002, 12)
    aad = splitmix_bytes(0x7003, 37)
    pt = splitmix_bytes(0x7004, 203 * 16 + 5)
    ct, tag = both(key, iv, aad, pt)
    h = int.from_bytes(ecb_block(key, b"\0" * 16), "big")
    ej0 = int.from_bytes(ecb_block(key, iv + b"\0\0\0\1"), "big")
    nblk = (len(ct) + 15) // 16
    bounds = [0, 25, 51, 76, 102, 127, 153, 178, nblk]      # block boundaries of the 8 shards
    shards = []
    fold = 0
    for g in range(8):
        s, e = bounds[g], bounds[g + 1]
        p = 0
        for i in range(s, e):
            blk = ct[16 * i:16 * i + 16].ljust(16, b"\0")
            p = gf_mul_int(p, h) ^ int.from_bytes(blk, "big")          # P_g = sum C_i H^(e-1-i)
        w = gf_mul_int(p, gf_pow_int(h, nblk - e))                     # weighted to the message end
        fold ^= w
        shards.append(dict(first_block=s, end_block=e, poly=p.to_bytes(16, "big").hex(), weight_exp=nblk - e,
                           weighted=w.to_bytes(16, "big").hex()))
    # PA = poly over AAD blocks
    pa = 0
    for off in range(0, len(aad), 16):
        pa = gf_mul_int(pa, h) ^ int.from_bytes(aad[off:off + 16].ljust(16, b"\0"), "big")
    y = gf_mul_int(pa, gf_pow_int(h, nblk)) ^ fold
    lb = (len(aad) * 8) << 64 | (len(ct) * 8)
    y = gf_mul_int(gf_mul_int(y, h) ^ lb, h)
    assert (y ^ ej0).to_bytes(16, "big") == tag
    dump("shards.json", dict(key=key.hex(), iv=iv.hex(), aad=aad.hex(), pt=pt.hex(), ct=ct.hex(), tag=tag.hex(),
                             h=h.to_bytes(16, "big").hex(), ej0=ej0.to_bytes(16, "big").hex(), n_blocks=nblk, shards=shards,
                             aad_poly=pa.to_bytes(16, "big").hex(), folded=fold.to_bytes(16, "big").hex()))


def batch_inputs(first_pkt, n_pkts, pkt_len, pt_seed=0xAE5C0005):
    """cfg5 definition: packet p has key = bytes 16p..16p+15 of stream 0x4B4559,
    iv = bytes 16p..16p+11 of stream 0x4956, pt = bytes p*pkt_len.. of stream pt_seed."""
    keys = splitmix_bytes(KEY_SEED, 16 * n_pkts, first_word=2 * first_pkt)
    ivw = splitmix_bytes(IV_SEED, 16 * n_pkts, first_word=2 * first_pkt)
    ivs = b"".join(ivw[16 * p:16 * p + 12] for p in range(n_pkts))
    pt = splitmix_bytes(pt_seed, pkt_len * n_pkts, first_word=first_pkt * pkt_len // 8)
    return keys, ivs, pt


def gen_batch(large):
    path = os.path.join(HERE, "batch.json")
    keys, ivs, pt = batch_inputs(0, 64, 4096)
    tags, ct_sha = [], hashlib.sha256()
    for p in range(64):
        ct, tag = R.encrypt(keys[16 * p:16 * p + 16], ivs[12 * p:12 * p + 12], b"", pt[4096 * p:4096 * (p + 1)])
        tags.append(tag.hex())
        ct_sha.update(ct)
    out = dict(definition="cfg5: key_p = stream(0x4B4559)[16p:16p+16], iv_p = stream(0x4956)[16p:16p+12], "
                          "pt_p = stream(0xAE5C0005)[4096p:4096p+4096], AES-128, empty AAD",
               first64_tags=tags, first64_ct_sha256=ct_sha.hexdigest())
    if os.path.exists(path):
        old = json.load(open(path))
        for k in ("full_n_pkts", "full_tags_sha256", "full_ct_sha256"):
            if k in old:
                out[k] = old[k]
    if large:
        n = 1 << 20
        tsha, csha = hashlib.sha256(), hashlib.sha256()
        step = 1 << 14
        for first in range(0, n, step):
            keys, ivs, pt = batch_inputs(first, step, 4096)
            for p in range(step):
                ct, tag = R.encrypt(keys[16 * p:16 * p + 16], ivs[12 * p:12 * p + 12], b"", pt[4096 * p:4096 * (p + 1)])
                tsha.update(tag)
                csha.update(ct)
        out.update(full_n_pkts=n, full_tags_sha256=tsha.hexdigest(), full_ct_sha256=csha.hexdigest())
    dump("batch.json", out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--large", action="store_true", help="also 1 GiB / 16 GiB streams and the full 2^20-packet batch")
    ap.add_argument("--huge", action="store_true", help="also the 4 x 32 GiB cfg4 messages")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    todo = a.only.split(",") if a.only else ["keys", "kat", "gfmul", "matrix", "streams", "shards", "batch"]
    if "keys" in todo:
        gen_key_schedule_and_sbox()
    if "kat" in todo:
        gen_kat()
    if "gfmul" in todo:
        gen_gfmul()
    if "matrix" in todo:
        gen_length_matrix()
    if "streams" in todo:
        gen_streams(a.large, a.huge)
    if "shards" in todo:
        gen_shards()
    if "batch" in todo:
        gen_batch(a.large)
