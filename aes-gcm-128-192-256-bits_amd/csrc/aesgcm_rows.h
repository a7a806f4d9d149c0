// aesgcm_rows.h -- MANY messages under one key through k_body's row code (round 5): the lane pieces and the plan.
//
// The reference's deployment is frame after frame under one key (tb/gcm_test.py:76-85; H is kept while no new key is loaded,
// src/gcm_gctr.vhd:142-144).  Until round 4 a message-sized packet of aesgcm_packets_crypt_dev went through the wave-per-packet group
// kernel (two-table round, LDS full of tree tables: 4096 x 1 MiB 731 GiB/s against 980 for one message of the same bytes).  Here the rows
// of ALL messages of a call are one pool of work for k_body's row loop (four T-tables, rounds 1 - 2 from per-chunk lane constants and the
// scalar cache, the five-bit table of H^256):
//
//   * a message of `len` bytes is R = len / 1024 whole rows of 64 blocks, aligned to its first block (every packet starts at counter 2,
//     src/aes_icb.vhd:97-118, so every message IS an aligned body), plus a TAIL of tb <= 64 blocks (the last one ragged), plus its AAD;
//   * Q = R / 4 super-rows are cut into SUPER-CHUNKS of T super-rows; a CHUNK is one row phase v (0 .. 3) of a super-chunk: the rows
//     4 (q0 + i) + v, i < nrows -- exactly a strand of body_strand_lane, Horner stride H^256.  The R mod 4 rows behind the last whole
//     super-row are one more super-chunk of ONE super-row in which only the phases v < R mod 4 exist;
//   * waves pull chunks from dispensers (k_rows).  At the end of a chunk the wave does NOT leave its 64 lane accumulators: lane L's value
//     times H^(63 - L) through the key's per-lane Shoup tables (KeyMaterial::ltab, one multiply deep), XORed over the wave, is the
//     polynomial of the whole strand up to its last block -- 16 bytes per chunk (W_c) instead of 1 KiB, so the chunk size is free to
//     follow the load balance alone (HBM traffic 1.00 x algorithmic at any T);
//   * what is not a whole row is two more kinds of chunk in the same pool: a message's TAIL -- its tail blocks and the length block as one right-aligned row
//     (CTR from counter 2 + 64 R; the lane that holds the length block encrypts counter 1 instead, which is E_K(J0)), lane terms H^(64 - L), 16 bytes -- and
//     its AAD (rows of its own, lane terms H^(63 - L), 16 bytes).  The two-table round they use reads the same T0 | T2 image the row loop does;
//   * ONE small launch behind the rows closes every tag (k_rows_close): a LANE per chunk slot weights its 16 bytes with H^(blocks behind it + 2) bit-serially --
//     e = 64 (R - 1 - last row) + tb + 2 for a strand, 64 R + tb + 2 for the AAD, nothing for the tail: any exponent, all lanes in parallel -- and XORs the
//     product into the message's accumulator with memory-side atomics; every slot then counts itself arrived, and the one that counts a message's last
//     arrival holds its tag  P H^2 ^ L H ^ E_K(J0)  (gcm_ghash.vhd:257,293 re-associated), stores it and, for decrypt, compares.
//
// Fixed-size records need no plan: super-chunk sc belongs to message sc / S.  With offset arrays the lengths are on the device, so two small
// launches in front make the plan there: k_rows_plan (totals, the chunk size, the scan of super-chunks per message) and k_rows_expand (a
// 32-byte descriptor per super-chunk: where, which IV, which rows).
#pragma once
#include "aesgcm_dev.h"

#define ROWS_NQ AESGCM_NQ                     /* dispenser queues of k_rows (one cache line each) */
#define ROWS_T_MIN 4u                         /* fewest super-rows per super-chunk (a chunk then is 4 rows: small calls want parallelism) */
#define ROWS_T_MAX 64u                        /* most, unless the table of super-chunks would not hold the call (rows_pick_T) */
#define ROWS_SC_TARGET 8192u                  /* super-chunks a call is cut into when it is large enough: 32768 chunks = 8 per resident wave of k_rows */
#define ROWS_CAP_BASE 65536u                  /* offset-array form: the table holds ROWS_CAP_BASE + 2 n_pkts super-chunks (rows_pick_T doubles T until the call fits) */
#define ROWS_CLOSE_WG 256u                    /* lanes per k_rows_close workgroup */

struct RowsSc {                               // one super-chunk, 32 bytes: all a wave of k_rows needs for its chunk (the phase is chunk & 3)
    u64 off;                                  // byte offset of the message's data in `in` / `out`
    u32 iv0, iv1, iv2;                        // the message's IV, memory-order words
    u32 q0;                                   // first super-row
    u32 shape;                                // super-rows | phases << 28  (4; or R mod 4 for the single super-row behind the last whole one)
    u32 msg;
};
struct RowsHdr { u32 n_sc, T; u32 pad[14]; }; // made by k_rows_plan (offset-array form)

struct RowsParams {
    const unsigned char *ivs;                 // n_pkts * 12 bytes
    const unsigned char *aad;
    const unsigned char *in;
    unsigned char *out;
    unsigned char *tags;                      // n_pkts * 16
    const unsigned char *expect;              // dec: expected tags or NULL
    int *auth;                                // dec: per message 1 / 0, or NULL
    const u64 *data_off, *aad_off;            // n_pkts + 1 byte offsets, or NULL = fixed pkt_len / aad_len records
    u32 n_pkts, pkt_len, aad_len;
    u32 T, S, n_sc;                           // fixed-size form: super-rows per super-chunk, super-chunks per message, super-chunks of the call
    const RowsHdr *hdr;                       // offset-array form: the plan on the device ...
    const RowsSc *sc;                         // ... the super-chunk table (NULL in the fixed-size form) ...
    const u32 *msg_sc;                        // ... and the first super-chunk of every message (n_pkts + 1 entries)
    u32 cap_sc;                               // super-chunks the table (and wsum / 4) can hold
    G128 *wsum;                               // one strand polynomial per chunk (4 per super-chunk)
    G128 *wtail;                              // per message: (tail polynomial) H^2 ^ (length block) H ^ E_K(J0)
    G128 *waad;                               // per message: the AAD's polynomial (written only when the message has AAD)
    u32 has_aad;                              // the call has AAD (fixed aad_len > 0, or an offset array): k_rows deals an AAD chunk per message
    unsigned long long *acc;                  // per message {hi, lo}: the XOR of everything that makes its tag
    u32 *cnt;                                 // per message: contributors arrived
    u32 *queues;                              // ROWS_NQ dispensers, 16 u32 apart, zero when k_rows starts (k_rows_close leaves them so)
};

// ---- geometry (host, planner and kernels agree through these) -------------------------------------
struct RowsGeom { u32 R, Q, rho, tb, tail_bytes; };
HD RowsGeom rows_geom(u64 len) {
    RowsGeom g;
    g.R = (u32)(len >> 10); g.Q = g.R >> 2; g.rho = g.R & 3u;
    g.tail_bytes = (u32)(len & 1023u); g.tb = (g.tail_bytes + 15u) >> 4;
    return g;
}
HD u32 rows_nsc(u32 Q, u32 rho, u32 T) { return (Q + T - 1u) / T + (rho ? 1u : 0u); }
// super-chunk s (of rows_nsc) of a message: first super-row, super-rows, phases
HD void rows_sc_shape(u32 Q, u32 rho, u32 T, u32 s, u32 &q0, u32 &nrows, u32 &nphase) {
    const u32 S = (Q + T - 1u) / T;
    if (s < S) { q0 = s * T; nrows = Q - q0 < T ? Q - q0 : T; nphase = 4u; }
    else { q0 = Q; nrows = 1u; nphase = rho; }
}
// Super-rows per super-chunk of a call with total_q super-rows in n messages: about ROWS_SC_TARGET super-chunks when the call is large, never fewer than tmin
// super-rows (per-chunk work: a dispenser fetch, the descriptor, the round-1 constants, one Shoup multiply and a wave fold -- about one row's worth) nor more than
// tmax (the granularity of the dealing), unless the table would overflow.
HD u32 rows_pick_T(u64 total_q, u64 n, u64 cap_sc, u32 tmin, u32 tmax) {
    u64 T = (total_q + ROWS_SC_TARGET - 1u) / ROWS_SC_TARGET;
    if (T < tmin) T = tmin;
    if (T > tmax) T = tmax;
    if (T < 1) T = 1;
    while (cap_sc && total_q / T + 2 * n > cap_sc && T < (1u << 20)) T *= 2;
    return (u32)T;
}
// the message's lengths and offsets
struct RowsMsg { u64 doff, aoff; u32 len, alen; };
HD RowsMsg rows_msg(const RowsParams &p, u32 m) {
    RowsMsg q;
    q.len = p.pkt_len; q.alen = p.aad_len;
    q.doff = (u64)m * p.pkt_len; q.aoff = (u64)m * p.aad_len;
    if (p.data_off) { q.doff = p.data_off[m]; q.len = (u32)(p.data_off[m + 1] - q.doff); }
    if (p.aad_off) { q.aoff = p.aad_off[m]; q.alen = (u32)(p.aad_off[m + 1] - q.aoff); }
    return q;
}
// the descriptor of super-chunk sc (offset-array form: from the table; fixed-size form: arithmetic)
HD RowsSc rows_desc(const RowsParams &p, u32 sc) {
    if (p.sc) return p.sc[sc];
    RowsSc e;
    const u32 m = sc / p.S, s = sc - m * p.S;
    const RowsGeom g = rows_geom(p.pkt_len);
    u32 q0, nrows, nphase;
    rows_sc_shape(g.Q, g.rho, p.T, s, q0, nrows, nphase);
    const unsigned char *ivp = p.ivs + (size_t)m * 12;
    e.off = (u64)m * p.pkt_len;
    e.iv0 = load_le32(ivp); e.iv1 = load_le32(ivp + 4); e.iv2 = load_le32(ivp + 8);
    e.q0 = q0; e.shape = nrows | (nphase << 28); e.msg = m;
    return e;
}
// chunks of k_rows: [0, 4 n_sc) strands, then a tail chunk per message, then (has_aad) an AAD chunk per message
HD u32 rows_chunks(const RowsParams &p, u32 n_sc) { return 4u * n_sc + p.n_pkts * (p.has_aad ? 2u : 1u); }
// what arrives at a message's accumulator: every chunk slot of its super-chunks (whether the phase exists or not), its tail slot, its AAD slot
HD u32 rows_expected(u32 nsc) { return 4u * nsc + 2u; }

// ---- k_rows: one chunk ---------------------------------------------------------------------------
// lane `lane` of the wave that owns phase v of the super-chunk e: CTR over the rows 4 (q0 + i) + v and the lane's Horner accumulator (stride H^256)
template <int NR, int MODE>
HD uint4 rows_chunk_lane(const KeyMaterial *__restrict__ km, const DevTables *__restrict__ tb, const RowsParams &p, const RowsSc &e,
                         const unsigned char *smem, const CtrConsts &cc, u32 v, u32 lane) {
    return body_strand_rows<NR, MODE>(km, tb, p.in + e.off, p.out + e.off, 0u, smem, cc, e.q0, 1u, e.shape & 0x0FFFFFFFu, v, lane);
}
// the lane's term of the strand polynomial: B_L H^(63 - L) (XOR over the wave = the polynomial of the strand up to its last block)
HD G128 rows_chunk_term(const KeyMaterial *__restrict__ km, uint4 acc, u32 lane) { return shoup2_gmul_lds(mo_to_be(acc), km->ltab[63u - lane]); }

// ---- k_rows: the tail chunk and the AAD chunk of message m -------------------------------------------
// a further row of such a chunk (a tail of a full 64 blocks; AAD beyond 1 KiB): the accumulator times H^64, bit-serially -- k_rows has the table of H^256 in LDS, not of H^64,
// and these rows are rare
HD uint4 rows_mul_h64(const KeyMaterial *__restrict__ km, uint4 acc) { return gf_mul_mo(acc, km->pw[0][64]); }
// The tail blocks (data blocks 64 R ..., the last one ragged) and the length block [8 len(A)]_64 || [8 len(C)]_64 (gcm_ghash.vhd:257) as ONE right-aligned
// sequence of tb + 1 <= 65 slots: one row, or two when the tail is a full 64 blocks.  Every lane runs the cipher once per row: data lanes on counter
// 2 + block index (aes_icb.vhd:97-118), the lane of the length block on counter 1 -- E_K(IV || 0^31 1), the J0 block the RTL latches first
// (gcm_ghash.vhd:158-169) -- returned in *ej0 by that lane (lane 63).  Returns the lane's term  X_L H^(64 - L)  of  (tail polynomial) H^2 ^ (length block) H.
template <int NR, int DEC>
HD G128 rows_tail_lane(const KeyMaterial *__restrict__ km, const RowsParams &p, const RowsMsg &q, const unsigned char *smem, const CtrConsts &cc, u32 lane, uint4 *ej0) {
    const u32 *__restrict__ rk = km->rk;
    const u32 lb = (lane & 31u) << 2;
    const RowsGeom g = rows_geom(q.len);
    const u32 n_slots = g.tb + 1u, rows = (n_slots + 63u) >> 6, pad = 64u * rows - n_slots;
    const unsigned char *src = p.in + q.doff;
    unsigned char *dst = p.out + q.doff;
    uint4 acc = make_uint4(0, 0, 0, 0);
    *ej0 = make_uint4(0, 0, 0, 0);
    for (u32 k = 0; k < rows; k++) {
        if (k) acc = rows_mul_h64(km, acc);
        const u32 slot = k * 64u + lane;
        if (slot < pad) continue;
        const u32 j = slot - pad;
        const bool is_len = j == g.tb;
        const u32 i = 64u * g.R + j;                                                    // block index in the message
        u32 s0, s1, s2, s3;
        ctr_rounds_lds<NR>(bswap32(is_len ? 1u : 2u + i), cc, s0, s1, s2, s3, rk, smem, lb);
        uint4 gin;
        if (is_len) {
            *ej0 = make_uint4(s0, s1, s2, s3);
            const u64 a = (u64)q.alen * 8u, c = (u64)q.len * 8u;
            gin = make_uint4(bswap32((u32)(a >> 32)), bswap32((u32)a), bswap32((u32)(c >> 32)), bswap32((u32)c));
        } else {
            const u32 off = 16u * i, rem = q.len - off;
            const bool full = rem >= 16u;
            const uint4 x = full ? gload16_any(src + off) : load_block_bytes(src + off, rem);
            uint4 y = make_uint4(x.x ^ s0, x.y ^ s1, x.z ^ s2, x.w ^ s3);                // gcm_gctr.vhd:150
            if (!full) y = mask_block(y, rem);
            if (full) gstore16_any(dst + off, y); else store_block_bytes(dst + off, y, rem);
            gin = DEC ? x : y;                                                          // aes_gcm.vhd:207-211
        }
        acc = xor4(acc, gin);
    }
    return shoup2_gmul_lds(mo_to_be(acc), km->ltab[64u - lane]);
}
// The AAD of message m as rows of its own (right-aligned, Horner with H^64: one row up to 1 KiB of AAD): the lane's term  A_L H^(63 - L)  of the AAD's polynomial.  alen > 0.
HD G128 rows_aad_lane(const KeyMaterial *__restrict__ km, const RowsParams &p, const RowsMsg &q, u32 lane) {
    const u32 n_aad = (q.alen + 15u) >> 4, rows = (n_aad + 63u) >> 6, pad = 64u * rows - n_aad;
    const unsigned char *a = p.aad + q.aoff;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (u32 k = 0; k < rows; k++) {
        if (k) acc = rows_mul_h64(km, acc);
        const u32 slot = k * 64u + lane;
        if (slot < pad) continue;
        const u32 off = 16u * (slot - pad), rem = q.alen - off;
        acc = xor4(acc, rem >= 16u ? gload16_any(a + off) : load_block_bytes(a + off, rem));
    }
    return shoup2_gmul_lds(mo_to_be(acc), km->ltab[63u - lane]);
}

// ---- k_rows_close: the lane of slot `slot` -- [0, 4 cap_sc) chunk slots, then a tail slot and an AAD slot per message ----------------------------------
// *msg = the message it contributes to (or 0xFFFFFFFF: the slot is beyond the call), *z = its contribution (zero when the phase does not exist / no AAD)
HD void rows_weight_lane(const KeyMaterial *__restrict__ km, const RowsParams &p, u32 n_sc, u32 slot, u32 *msg, G128 *z) {
    z->w[0] = z->w[1] = z->w[2] = z->w[3] = 0;
    *msg = 0xFFFFFFFFu;
    G128 w;
    u64 e;
    if (slot < 4u * p.cap_sc) {
        if (slot >= 4u * n_sc) return;
        const u32 sc = slot >> 2, v = slot & 3u;
        u32 m, q0, shape;
        if (p.sc) { m = p.sc[sc].msg; q0 = p.sc[sc].q0; shape = p.sc[sc].shape; }
        else {
            m = sc / p.S;
            const RowsGeom g = rows_geom(p.pkt_len);
            u32 nrows, nphase;
            rows_sc_shape(g.Q, g.rho, p.T, sc - m * p.S, q0, nrows, nphase);
            shape = nrows | (nphase << 28);
        }
        *msg = m;
        if (v >= (shape >> 28)) return;
        const RowsGeom g = rows_geom(rows_msg(p, m).len);
        const u32 r_last = 4u * (q0 + (shape & 0x0FFFFFFFu) - 1u) + v;
        e = 64ull * (g.R - 1u - r_last) + g.tb + 2u;
        w = p.wsum[slot];
    } else {
        u32 m = slot - 4u * p.cap_sc;
        if (m < p.n_pkts) { *msg = m; *z = p.wtail[m]; return; }              // the tail: already weighted
        m -= p.n_pkts;
        if (m >= p.n_pkts) return;
        *msg = m;
        const RowsMsg q = rows_msg(p, m);
        if (!p.has_aad || !q.alen) return;
        const RowsGeom g = rows_geom(q.len);
        e = 64ull * g.R + g.tb + 2u;
        w = p.waad[m];
    }
    *z = gf_mul(w, gf_pow_h_serial(km, e));
}
