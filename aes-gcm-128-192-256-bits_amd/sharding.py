"""Host-side planning for one GCM message sharded over ranks (no arithmetic here).

A message of `total_len` bytes is cut at 16-byte block boundaries into `n_ranks` contiguous shards;
rank r owns blocks [first_block, end_block).  Every rank en/decrypts its shard with counters
2 + first_block + i (src/aes_icb.vhd:97-100: only the low 32 bits count) and produces the weighted GHASH
partial W_r = P_r * H^(n_blocks - end_block); the tag needs xor_r W_r, obtained with ONE 16-byte-per-rank
all-gather (RCCL has no XOR reduction).  Rank 0 also absorbs the AAD.

`plan_job` describes bench.py's workloads: N x bytes_per_gpu of plaintext as messages of at most 32 GiB
(one GCM message cannot exceed 2^36 - 32 bytes, src/aes_icb.vhd:114), every message sharded over all ranks.
"""

MAX_MESSAGE = (1 << 36) - 32


def shard_bounds(total_len, n_ranks, rank):
    """-> (first_block, end_block, byte_len) of `rank`'s shard; only the last shard may be ragged."""
    n_blocks = (total_len + 15) // 16
    base, extra = divmod(n_blocks, n_ranks)
    first = rank * base + min(rank, extra)
    end = first + base + (1 if rank < extra else 0)
    byte_end = total_len if end == n_blocks else 16 * end
    return first, end, max(0, byte_end - 16 * first)


def plan_job(n_ranks, bytes_per_gpu, rank, msg_bytes=None):
    """Messages of one bench step for `rank`.  Each entry: iv_tweak, total (bytes of the whole message),
    first_block / length of this rank's shard, offset of the shard in the rank's resident buffer, and
    stream_word = first 64-bit word of the shard in the job-wide SplitMix64 plaintext stream."""
    total = n_ranks * bytes_per_gpu
    if msg_bytes is None:
        msg_bytes = total if n_ranks == 1 else 2 * bytes_per_gpu
    msg_bytes = min(msg_bytes, total)
    if msg_bytes > MAX_MESSAGE:
        raise ValueError("a GCM message cannot exceed 2^36 - 32 bytes")
    n_msgs = total // msg_bytes
    if n_msgs > 256:
        raise ValueError("plan_job: more than 256 messages per job (tweak_iv advances one IV byte)")
    if n_msgs * msg_bytes != total or msg_bytes % (16 * n_ranks):
        raise ValueError("job does not split evenly")
    out = []
    off = 0
    for m in range(n_msgs):
        first, end, ln = shard_bounds(msg_bytes, n_ranks, rank)
        out.append(dict(msg=m, iv_tweak=m, total=msg_bytes, first_block=first, len=ln, off=off,
                        stream_word=(m * msg_bytes + 16 * first) // 8))
        off += ln
    return out


def tweak_iv(iv, tweak):
    """IV of message `tweak` of a bench job: the base IV with its last byte advanced by `tweak` (the rule the cfg4
    fixtures were generated with, tests/golden/gen_golden.py).  Bench/test helper only: it refuses to wrap, because a
    wrapped byte would silently reuse a (key, nonce) pair, which GCM does not survive."""
    b = bytearray(iv)
    if len(b) != 12 or tweak < 0 or b[11] + tweak > 0xFF:
        raise ValueError("tweak_iv: base IV byte 11 (0x%02x) + %d leaves the byte; pick another base IV or fewer messages" % (b[11] if len(b) == 12 else 0, tweak))
    b[11] += tweak
    return bytes(b)


def splitmix64_bytes(seed, nbytes, first_word=0):
    """The synthetic stream of SURVEY.md 8(d) on the host (keys, IVs): 64-bit little-endian word w of stream `seed`
    is SplitMix64 evaluated at position w.  Same definition as aesgcm_fill_splitmix64_dev."""
    M = (1 << 64) - 1
    out = bytearray()
    w = first_word
    while len(out) < nbytes:
        z = (seed + (w + 1) * 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        z ^= z >> 31
        out += z.to_bytes(8, "little")
        w += 1
    return bytes(out[:nbytes])
