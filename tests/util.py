"""Shared helpers for the test-suite: golden fixture loading and the seeded input definitions the
fixtures were generated from (tests/golden/gen_golden.py)."""
import json
import os

from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KEY_SEED, IV_SEED = 0x4B4559, 0x4956


def golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def splitmix_bytes(seed, n, first_word=0):
    return bytes(O.fill_splitmix64(n, seed, first_word))


def matrix_inputs(kbits, al, pl):
    tagv = (kbits << 40) | (al << 20) | pl
    return (splitmix_bytes(0xA0000000 + tagv, kbits // 8), splitmix_bytes(0xB0000000 + tagv, 12),
            splitmix_bytes(0xC0000000 + tagv, al), splitmix_bytes(0xD0000000 + tagv, pl))


def stream_key_iv(case):
    key = splitmix_bytes(case["key_seed"], case["key_bytes"])
    iv = bytearray(splitmix_bytes(case["iv_seed"], 12))
    iv[11] = (iv[11] + case["iv_tweak"]) & 0xFF
    return key, bytes(iv)


def batch_inputs(first_pkt, n_pkts, pkt_len, pt_seed=0xAE5C0005):
    keys = splitmix_bytes(KEY_SEED, 16 * n_pkts, first_word=2 * first_pkt)
    ivw = splitmix_bytes(IV_SEED, 16 * n_pkts, first_word=2 * first_pkt)
    ivs = b"".join(ivw[16 * p:16 * p + 12] for p in range(n_pkts))
    pt = splitmix_bytes(pt_seed, pkt_len * n_pkts, first_word=first_pkt * pkt_len // 8)
    return keys, ivs, pt
