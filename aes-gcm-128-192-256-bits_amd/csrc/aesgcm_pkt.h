// aesgcm_pkt.h -- packets under ONE key: the lane code of k_pktg (lane groups) and k_pktl (a lane per packet) (part of aesgcm_dev.h).
#pragma once
#include "aesgcm_base.h"
#include "aesgcm_aes.h"
#include "aesgcm_ghash.h"
#include "aesgcm_stream.h"

// ================================================================================================
// Packets under ONE key (the RTL keeps H while no new key is loaded, src/gcm_gctr.vhd:142-144): per-packet IV,
// AAD and length; key schedule, H and every GHASH table come from the context.
//
// k_pktg<NR, DEC, LG>: G = 2^LG lanes per packet, 64 / G packets per wave (LG = 4: four packets of 16 lanes; LG = 6: one
// packet per wave).  Round 3 rework of k_pkt (one wave per packet), whose counters said (profiles/archive/r02g/pktw_1k: frac 0.05,
// 2.5 x its algorithmic traffic, LDS busy 0.37): the length block took a slot of the row structure (a 1 KiB packet was two
// rows, one of them 63/64 empty), the closing was a 128-step bit-serial multiply per lane per packet (~1300 VALU, three AES
// rows' worth), and the byte loops spilled to scratch inside the row loop.  Now:
//   * the packet's GHASH sequence (AAD blocks, then data blocks -- NOT the length block) is right-aligned into iterations of
//     G slots; lane l of the group takes slots l, l + G, ...: Horner with the key's five-bit LDS table of H^G (acc = acc *
//     H^G ^ X, ghash_mul_const_lds), the same row loop as k_main.  A 1 KiB packet is four iterations of its 16 lanes.
//   * closing without a slot and without a bit-serial step: P = sum_l B_l H^(G-1-l) and tag = P H^2 ^ L H ^ E_K(J0)
//     (gcm_ghash.vhd:257,293).  Every lane multiplies by H^2 (table); the length block L is XORed into lane G-2; then a
//     cross-lane tree of LG levels with WAVE-UNIFORM constants H, H^2, H^4, ... (level j: the lane with bit j set takes
//     partner * H^(2^j) ^ own).  Lane G-2's value meets exactly one more H on its way (level 0), every other path to lane G-1
//     carries the weights of P: lane G-1 ends with P H^2 ^ L H.  LG + 1 table multiplies per wave-iteration, shared by the
//     64 / G packets of the wave; the tables (H^(2^j), j < LG, five-bit form, 13.25 KiB each) sit behind the AES tables in LDS.
//   * E_K(IV || 1) of up to 64 packets comes from ONE extra AES pass per dispenser fetch: lane j encrypts packet j's J0 block
//     (gcm_ghash.vhd:158-169); the group's last lane picks its packet's value up with a lane shuffle.
// LDS: [0, 13568) five-bit table of H^G | [13568, +64 KiB) T0 | T2 | [79104 + j * 13568) tree table j | (lane groups) 1 KiB per wave for
// the E_K(J0) values of a dispenser block.  One workgroup per CU (LG = 4: 130.25 + 16 KiB at 1024 lanes, LG = 6: 156.75 KiB at 768).
// ================================================================================================
// What a lane needs to know of its packet, 48 bytes, written IN THE LAUNCH'S ORDER by the sort of a routed call (k_len_scatter, k_len_sort1; round 6) when the shape is
// a lane per packet: a wave reads 64 consecutive records -- three coalesced 16-byte loads per lane -- where it used to read its packet number, then four offsets and
// an IV at that number's places (messages wherever they live: five arrays): six cache lines per packet that nobody else touches, two dependent trips before the first
// byte of data is requested.  a / b / c: data offset, AAD offset, -- (offset arrays; the bases are the launch's) or the addresses of input, AAD and output.
struct PktDesc { u64 a, b, c; u32 len, alen, pkt, iv0, iv1, iv2; };
struct PktParams {
    const unsigned char *ivs;    // n_pkts * 12 bytes
    const unsigned char *aad;    // AAD bytes or NULL
    const unsigned char *in;
    unsigned char *out;
    unsigned char *tags;         // n_pkts * 16 (computed tags)
    const unsigned char *expect; // dec: expected tags or NULL
    int *auth;                   // dec: per-packet 1/0 or NULL
    const u64 *data_off;         // n_pkts + 1 offsets, or NULL = fixed pkt_len records
    const u64 *aad_off;          // n_pkts + 1 offsets, or NULL = fixed aad_len records
    u32 *counter; u32 counter_base;
    u32 deal;                    // packets per dispenser fetch (k_pktg: a multiple of the packets per wave, at most 64)
    u32 n_pkts, pkt_len, aad_len;
    u32 aligned;                 // in/out base pointers 16-byte aligned
    const u32 *perm;             // the order in which the launch takes the packets (k_len_*: by falling length), or NULL = as they come
    u32 plain;                   // k_pktg<.., 6 | 2>: fixed-size aligned records of whole group-iterations, no AAD
    // a ROUTED call (lengths on the device): how many packets the launch has (the first n_small entries of perm), whether this instance is the shape chosen for
    // that count, and its deal, are in the header k_len_scan left (aesgcm_rows.h RowsHdr); NULL = the host's numbers above.  For messages WHEREVER THEY LIVE
    // (aesgcm_messages_crypt_dev: the short ones of such a call are the packet kernels') the header also holds the arrays of addresses and lengths (sc_*); in / out /
    // aad are NULL then
    const struct RowsHdr *route;
    u32 scattered;               // 1: the packets' places come from route->sc_* (the host launches k_pktgs / k_pktls then)
    const PktDesc *desc;         // a routed call whose shape is a lane per packet: record i = the i-th packet of the launch (NULL: perm and the arrays)
};
HD u32 pkt_map(const PktParams &p, u32 i) { return p.perm ? p.perm[i] : i; }
HD const unsigned char *pkt_at(const unsigned char *base, u64 off) { return reinterpret_cast<const unsigned char *>((uintptr_t)base + off); }
HD unsigned char *pkt_at(unsigned char *base, u64 off) { return reinterpret_cast<unsigned char *>((uintptr_t)base + off); }
// Packets of mixed length (offset arrays).  The lanes (k_pktl) or lane groups (k_pktg, k_batch3) of a wave run to the longest packet among them: with frames
// of 64 .. 1514 bytes in arrival order a wave's 64 packets average 700 bytes and the wave takes as long as 1514 -- less than half the lanes work
// (profiles/r04/packets_sweep_mixed_*.txt: 2^20 frames 426 GiB/s against 854 for 2^20 x 1 KiB).  The launch therefore takes the packets in the order of a
// counting sort by length class (64 bytes per class, 256 classes, longest first so that the tail of the launch is short work): three small launches on the
// same stream in front of it -- histogram, scan, scatter; the order inside a class is whatever the atomics make it, results do not depend on it.
#define PKT_LEN_CLASSES 256u
HD u32 pkt_len_class(u64 len) { const u64 c = len >> 6; return c < PKT_LEN_CLASSES ? (u32)c : PKT_LEN_CLASSES - 1u; }
#ifndef AESGCM_PKTL_WG
#define AESGCM_PKTL_WG 768            // lanes per k_pktl workgroup: 3 waves per SIMD = 168 registers, what eight held blocks beside the table multiply need
#endif
// Packets of mixed length taken by length class: lanes per packet as log2 for n of them (0 = a lane per packet, k_pktl; 2 / 3 / 4 = lane groups, k_pktg) -- the
// host's rule for offset arrays (packets_pick_lg: measured in profiles/r04/packets_sweep_mixed_*.txt: by class the groups hold on until the packets fill k_pktl's
// resident lanes 4/3 times over) as a function the DEVICE can evaluate too: a routed call (round 6) learns the count only there (k_len_scan).
HD u32 route_pick_lg(u32 n_cu, u64 n) {
    const u64 lanes_total = (u64)n_cu * (AESGCM_PKT_WG / 64) * 64, lanes_l = (u64)n_cu * AESGCM_PKTL_WG;
    if (3 * n >= 4 * lanes_l) return 0u;
    const u64 fill = n ? lanes_total / n : lanes_total, cap = n >= 16384 ? 8 : 16;          // never more lanes than an eighth (a quarter, while the chip is not full) of a 1 KiB frame's blocks
    const u64 g = fill < cap ? fill : cap;
    return g >= 16 ? 4u : g >= 8 ? 3u : 2u;
}
// k_pktg: packets per dispenser fetch -- about 4 fetches per resident wave, a multiple of the packets per wave-iteration, at most 64 (one E_K(J0) pass per fetch)
HD u32 pktg_deal(u32 n_cu, u64 n, u32 lg) {
    const u32 P = 64u >> lg, waves_per_wg = (lg == 6u ? 768u : (u32)AESGCM_PKT_WG) / 64u;
    u32 deal = (u32)(n / ((u64)n_cu * waves_per_wg * 4));
    deal = deal / P * P;
    return deal < P ? P : deal > 64u ? 64u : deal;
}
#define PKTG_LDS_TREE_OFF (AESGCM_LDS_AES_OFF + AESGCM_LDS_AES)                       /* 79104 */
#define PKTG_LDS_BYTES(LG) (PKTG_LDS_TREE_OFF + (u32)(LG) * (u32)AESGCM_LDS_GH)
#define PKTG_MAX_DEAL 64u

// what thread `tid` of `nthreads` writes of k_pktg's LDS image
HD void pktg_fill_lds(unsigned char *smem, const KeyMaterial *km, const DevTables *tb, u32 tid, u32 nthreads, int lg) {
    main_fill_lds(smem, nullptr, tb, tid, false, nthreads);                                    // T0 | T2
    fill_lds_q5(smem, km->q5pow[lg], tid, nthreads, AESGCM_LDS_GH_OFF);                        // Horner stride H^G
    for (int j = 0; j < lg; j++) fill_lds_q5(smem, km->q5pow[j], tid, nthreads, PKTG_LDS_TREE_OFF + (u32)j * (u32)AESGCM_LDS_GH);
}

// the packets' data accesses.  DEC == 2 is the PROBE of a packet kernel (aesgcm_frames_ceiling_probe_dev): the same instruction stream WITHOUT the data's loads and
// stores -- IVs, AAD and tags still move --, what the formulation costs by itself on this chip at this moment's clocks; the "data" is then a value made of the block's
// number, so that nothing downstream folds away
// (-DAESGCM_PKT_NO_LOADS / -DAESGCM_PKT_NO_STORES: experiment builds in which the REAL instances lose one side of their data traffic -- which side the cycles between
// the probe and the real kernel belong to, profiles/r06/frames/loads_stores_ab.txt; no shipped library defines them)
template <int DEC>
HD uint4 pkt_ld(const unsigned char *p, u32 salt, bool aligned) {
#ifdef AESGCM_PKT_NO_LOADS
    return make_uint4(salt, salt * 3u, ~salt, 0x9E3779B9u ^ salt);
#endif
    if (DEC == 2) return make_uint4(salt, salt * 3u, ~salt, 0x9E3779B9u ^ salt);
    return aligned ? gload16(p) : gload16_any(p);
}
template <int DEC>
HD void pkt_st(unsigned char *p, uint4 v, bool aligned) {
#ifdef AESGCM_PKT_NO_STORES
    return;
#endif
    if (DEC == 2) return;
    if (aligned) gstore16(p, v); else gstore16_any(p, v);
}
// per-packet geometry and constants: uniform over the packet's lane group
struct PktInfo { u64 doff, ooff, aoff; u32 pkt_len, aad_len, iv0, iv1, iv2, aligned; };      // doff / ooff: where the packet's input / output lies (the same offset, except for messages in buffers of their own)
HD void pkt_place_scattered(const struct RowsHdr *h, u32 pkt, u64 *doff, u64 *ooff, u64 *aoff, u32 *pkt_len, u32 *aad_len);      // aesgcm_rows.h (the header's layout)
// SC (compile time): the packets' places come from the arrays of addresses behind p.route (messages wherever they live) -- kernels of their own (k_pktgs, k_pktls):
// as a run-time branch beside the offset form the two places of a packet stopped being one offset from two scalar bases, and k_pktg's 4-lane shape spilled (121 -> 128
// registers and 20 - 28 bytes of scratch)
template <bool SC>
HD void pkt_place(const PktParams &p, u32 pkt, u64 *doff, u64 *ooff, u64 *aoff, u32 *pkt_len, u32 *aad_len) {
    if (SC) { pkt_place_scattered(p.route, pkt, doff, ooff, aoff, pkt_len, aad_len); return; }     // addresses and lengths per message (in / out / aad are NULL: the "offsets" are addresses)
    *pkt_len = p.pkt_len; *aad_len = p.aad_len;
    *doff = (u64)pkt * p.pkt_len; *aoff = (u64)pkt * p.aad_len;
    if (p.data_off) { *doff = p.data_off[pkt]; *pkt_len = (u32)(p.data_off[pkt + 1] - *doff); }
    if (p.aad_off) { *aoff = p.aad_off[pkt]; *aad_len = (u32)(p.aad_off[pkt + 1] - *aoff); }
    *ooff = *doff;
}
template <bool SC = false>
HD PktInfo pkt_info(const PktParams &p, u32 pkt) {
    PktInfo q;
    pkt_place<SC>(p, pkt, &q.doff, &q.ooff, &q.aoff, &q.pkt_len, &q.aad_len);
    q.aligned = (p.aligned && (((q.doff | q.ooff) & 15) == 0)) ? 1u : 0u;
    const unsigned char *ivp = p.ivs + (size_t)pkt * 12;
    q.iv0 = load_le32(ivp); q.iv1 = load_le32(ivp + 4); q.iv2 = load_le32(ivp + 8);
    return q;
}
// iterations of a packet's lane group: its GHASH sequence (AAD blocks + data blocks) in slots of G
HD u32 pktg_iters(const PktInfo &q, u32 G) { return ((q.aad_len + 15) / 16 + (q.pkt_len + 15) / 16 + G - 1) / G; }

// E_K(IV || 0^31 1) of packet `pkt` (gcm_ghash.vhd:158-169), whole cipher from the IV: one lane per packet of a dispenser block
template <int NR>
HD uint4 pktg_ej0_lane(const KeyMaterial *__restrict__ km, const PktParams &p, const unsigned char *smem, u32 pkt, u32 lane) {
    const u32 *__restrict__ rk = km->rk;
    const unsigned char *ivp = p.ivs + (size_t)pkt * 12;
    u32 s0 = load_le32(ivp) ^ rk[0], s1 = load_le32(ivp + 4) ^ rk[1], s2 = load_le32(ivp + 8) ^ rk[2], s3 = 0x01000000u ^ rk[3];
    aes_rounds_lds<NR>(s0, s1, s2, s3, rk, smem, (lane & 31u) << 2);
    return make_uint4(s0, s1, s2, s3);
}

// lane l (0 .. G-1) of the group that owns packet `pkt`: CTR over its data blocks and the lane's Horner accumulator
// B_l = sum_k X[slot G k + l] (H^G)^(q-1-k) over the right-aligned sequence.  `iters` >= the packet's own q: the wave runs to
// the largest q of its groups under per-lane predicates (iterations beyond a packet's own come FIRST, as front padding).
template <int NR, int DEC, int LG>
HD uint4 pktg_lane(const KeyMaterial *__restrict__ km, const PktParams &p, const PktInfo &q, const unsigned char *smem, u32 l, u32 lane, u32 iters, bool act) {
    constexpr u32 G = 1u << LG;
    const u32 *__restrict__ rk = km->rk;
    const u32 lb = (lane & 31u) << 2;
    const CtrConsts cc = ctr_round1_consts(q.iv0, q.iv1, q.iv2, rk, smem, lb);      // key and IV only: uniform over the group
    const u32 n_aad = (q.aad_len + 15) / 16, n_ct = (q.pkt_len + 15) / 16, n_seq = n_aad + n_ct;
    const u32 pad = G * iters - n_seq;                                              // front padding slots (whole idle iterations included)
    const unsigned char *src = pkt_at(p.in, q.doff);
    unsigned char *dst = pkt_at(p.out, q.ooff);
    const bool aligned = q.aligned != 0;
    uint4 acc = make_uint4(0, 0, 0, 0);
    if ((LG == 6 || LG == 2) && p.plain) {   // a wave or four lanes per packet (the instances with registers to spare), records of one size, whole group-iterations, no AAD, aligned: no per-iteration tests
        const unsigned char *s = src + 16u * l;
        unsigned char *d = dst + 16u * l;
        for (u32 k = 0; k < iters; k++) {
            if (k) acc = ghash_mul_const_lds(acc, smem);
            const uint4 x = pkt_ld<DEC>(s, k, true);
            u32 s0, s1, s2, s3;
            ctr_rounds_lds<NR>(bswap32(2u + k * G + l), cc, s0, s1, s2, s3, rk, smem, lb);
            const uint4 y = make_uint4(x.x ^ s0, x.y ^ s1, x.z ^ s2, x.w ^ s3);
            if (act) pkt_st<DEC>(d, y, true);
            acc = xor4(acc, DEC == 1 ? x : y);
            s += 16u * G; d += 16u * G;
        }
        return acc;
    }
    for (u32 k = 0; k < iters; k++) {
        if (k) acc = ghash_mul_const_lds(acc, smem);
        const u32 v = k * G + l;
        if (v < pad) continue;
        const u32 j = v - pad;
        uint4 gin;
        if (j < n_aad) {
            const u32 off = 16 * j, rem = q.aad_len - off;
            const unsigned char *a = pkt_at(p.aad, q.aoff + off);
            gin = rem >= 16 ? gload16_any(a) : load_block_bytes(a, rem);
        } else {
            const u32 i = j - n_aad, off = 16 * i, rem = q.pkt_len - off;
            const bool full = rem >= 16;                                                // a whole block is one access at any address (gload16_any)
            uint4 x;
            if (full) x = pkt_ld<DEC>(src + off, i, aligned);
            else x = DEC == 2 ? mask_block(pkt_ld<DEC>(src + off, i, false), rem) : load_block_bytes(src + off, rem < 16 ? rem : 16);
            u32 s0, s1, s2, s3;
            ctr_rounds_lds<NR>(bswap32(2u + i), cc, s0, s1, s2, s3, rk, smem, lb);     // aes_icb.vhd:97-118: counter 2 + i
            uint4 y = make_uint4(x.x ^ s0, x.y ^ s1, x.z ^ s2, x.w ^ s3);                // gcm_gctr.vhd:150
            if (rem < 16) y = mask_block(y, rem);
            if (act && DEC != 2) {
                if (full) { if (aligned) gstore16(dst + off, y); else gstore16_any(dst + off, y); }
                else store_block_bytes(dst + off, y, rem < 16 ? rem : 16);
            }
            gin = DEC == 1 ? x : y;                                                          // aes_gcm.vhd:207-211
        }
        acc = xor4(acc, gin);
    }
    return acc;
}
// closing, step 1 (every lane): B_l * H^2, and the length block [8 len(A)]_64 || [8 len(C)]_64 (gcm_ghash.vhd:257) into lane G-2
template <int LG>
HD uint4 pktg_close_lane(uint4 acc, const PktInfo &q, const unsigned char *smem, u32 l) {
    constexpr u32 G = 1u << LG;
    acc = ghash_mul_q5_lds(acc, smem, PKTG_LDS_TREE_OFF + 1u * (u32)AESGCM_LDS_GH);
    if (l == G - 2u) acc = xor4(acc, make_uint4(0u, bswap32(q.aad_len * 8u), 0u, bswap32(q.pkt_len * 8u)));       // both < 2^32 bits by the ABI's limits
    return acc;
}
// closing, tree level j (every lane): the value offered to the partner lane l ^ 2^j, which takes it if its bit j is set
HD uint4 pktg_tree_offer(uint4 acc, const unsigned char *smem, int j) { return ghash_mul_q5_lds(acc, smem, PKTG_LDS_TREE_OFF + (u32)j * (u32)AESGCM_LDS_GH); }

#ifndef AESGCM_PKTL_GROUP
#define AESGCM_PKTL_GROUP 4
#endif
#ifndef AESGCM_PKTL_T4
#define AESGCM_PKTL_T4 1                 /* k_pktl: four T-tables in LDS (141 KiB; it is one workgroup per CU by its registers anyway), no rotates in rounds 2 .. NR-1 (round 4) */
#endif
#ifndef AESGCM_PKTL_CHAINS
#define AESGCM_PKTL_CHAINS 4              /* k_pktl's ILP form: keystream blocks computed side by side (two passes of four per 128-byte line) */
#endif
#ifndef AESGCM_PKTL_LINE
#define AESGCM_PKTL_LINE 1               /* k_pktl: a lane fetches its packet's whole 128-byte line at once (round 4) */
#endif
// One LANE per packet (k_pktl): the shape for MACsec-sized frames, where a 64-block row per packet would leave
// most lanes idle.  The lane runs the whole frame serially, as the reference core does (tb/gcm_test.py:76-85):
// AAD blocks, data blocks (CTR from 2, aes_icb.vhd:97-118; whole blocks as one access at whatever byte address the packet starts: gload16_any), the length block, Y = (Y ^ X) * H with the LDS
// nibble tables of H (main_fill_lds(GH_TAB_H)), tag = Y ^ E_K(IV || 1).  Nothing here is wave-uniform except the key.
template <int NR, int DEC, bool T4 = false, bool ILP = false, bool SC = false>
HD void pktl_lane(const KeyMaterial *__restrict__ km, const PktParams &p, const unsigned char *smem, u32 pkt, u32 lane, const PktDesc *d = nullptr) {
    const u32 *__restrict__ rk = km->rk;
    const u32 lb = (lane & 31u) << 2;
    u32 pkt_len, aad_len, iv0, iv1, iv2;
    u64 doff, ooff, aoff;
    if (d) {                                                                               // (uniform: the launch has records or it has not)
        const uint4 q0 = gload16(d), q1 = gload16(reinterpret_cast<const unsigned char *>(d) + 16), q2 = gload16(reinterpret_cast<const unsigned char *>(d) + 32);
        doff = ((u64)q0.y << 32) | q0.x; aoff = ((u64)q0.w << 32) | q0.z; ooff = SC ? (((u64)q1.y << 32) | q1.x) : doff;
        pkt_len = q1.z; aad_len = q1.w; pkt = q2.x; iv0 = q2.y; iv1 = q2.z; iv2 = q2.w;
    } else {
        pkt_place<SC>(p, pkt, &doff, &ooff, &aoff, &pkt_len, &aad_len);
        const unsigned char *ivp = p.ivs + (size_t)pkt * 12;
        iv0 = load_le32(ivp); iv1 = load_le32(ivp + 4); iv2 = load_le32(ivp + 8);
    }
    const CtrConsts cc = ctr_round1_consts(iv0, iv1, iv2, rk, smem, lb);
    uint4 acc = make_uint4(0, 0, 0, 0);
    const unsigned char *a = pkt_at(p.aad, aoff);
    for (u32 left = aad_len; left; ) {
        const u32 nb = left < 16 ? left : 16;
        acc = ghash_mul_const_lds(xor4(acc, nb == 16 ? gload16_any(a) : load_block_bytes(a, nb)), smem);
        a += nb; left -= nb;
    }
    const unsigned char *src = pkt_at(p.in, doff);
    unsigned char *dst = pkt_at(p.out, ooff);
    u32 ctr = 2, left = pkt_len;
    // whole groups of AESGCM_PKTL_GROUP blocks: the lane reads and writes 64 contiguous bytes at a time, so a cache
    // line is touched twice and not eight times (lanes of a wave are a packet apart: nothing coalesces across lanes).
    // Measured, 2^20 x 1 KiB, AES-256: 436 GiB/s block by block, 537 GiB/s in groups of 4 (8: the same).
#if AESGCM_PKTL_LINE
    // Round 4: the lane's whole 128-byte line at once, loads and stores.  With 64 bytes per step a line was touched twice, a tenth of a millisecond apart (a
    // lane needs ~0.2 ms for 128 bytes: 768 lanes share the CU's LDS), and the lines of all lanes in flight -- 32 CUs x 768 x (128 in + 128 out) = 6 MiB per
    // XCD -- turn the 4 MiB L2 over many times in between: the input was fetched 1.8 x (profiles/archive/r03e/pktl_1k), and output stored in two 64-byte groups left
    // the L2 as 1.22 x the ciphertext (block by block: 3.2 x; profiles/r04/pktl_store_ab.txt).  All eight loads are issued back to back, all eight stores
    // too; the blocks wait in 32 registers in between, which is why the workgroup is 768 lanes (3 waves per SIMD, 168 registers: at 1024 lanes AES-256
    // spilled, and decrypt -- whose GHASH runs on the loaded block while the plaintext waits -- did not fit at all).  The scheduling barrier keeps the
    // compiler from interleaving all eight blocks.  Measured, 2^20 packets under one key, same box (profiles/r04/pktl_768_ab.txt): HBM bytes = 1.000 - 1.005 x
    // algorithmic, encrypt and decrypt, 256 B ... 4 KiB (round 3: 1.41 x; decrypt until this change: 1.57 x); AES-256 encrypt 1 KiB 665 -> 679 GiB/s, 4 KiB 796
    // -> 823; decrypt 1 KiB 645 -> 662, 4 KiB 781 -> 750 (its register budget is full: 168).
    // ILP (k_pktl<.., 1>: 512-lane workgroups, 256 registers; what the host takes while the packets do not fill the chip): the eight keystream blocks of a line
    // as eight independent chains the compiler is free to interleave, then the eight multiplies.  With few waves per SIMD the wave itself must cover its LDS
    // latency -- one chain at a time it runs at the same 4 us per block whether 12 waves share the CU or 4 (65536 x 1 KiB: 256 GiB/s by lanes, 390 by groups of 4).
    while (ILP && left >= 128) {
        uint4 x[8], ks[8];
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = pkt_ld<DEC>(src + 16 * k, ctr + k, false);
        ctr_rounds_lds_n<NR, T4, AESGCM_PKTL_CHAINS>(ctr, cc, ks, rk, smem, lb);
        ctr_rounds_lds_n<NR, T4, AESGCM_PKTL_CHAINS>(ctr + AESGCM_PKTL_CHAINS, cc, ks + AESGCM_PKTL_CHAINS, rk, smem, lb);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint4 y = xor4(x[k], ks[k]);
            pkt_st<DEC>(dst + 16 * k, y, false);
            acc = ghash_mul_const_lds(xor4(acc, DEC == 1 ? x[k] : y), smem);
        }
        src += 128; dst += 128; left -= 128; ctr += 8;
    }
    // (Round 6 tried the lane's loads half a line AHEAD of its arithmetic -- the next 64 bytes requested before the current 64 are worked on, same 32 registers, stores in
    // groups of 64 bytes -- because the real kernel spends 16.5 % more cycles than its no-data twin, all of them waiting (SQ_WAIT_ANY + 256 M wave cycles for 2^20 frames,
    // instruction counts equal: profiles/r06/frames/probe_vs_real.txt).  It was SLOWER on the same box, 2^20 frames 1.58 ms against 1.49, AES-128 1.52 against 1.36
    // (profiles/r06/frames/pipe_ab.txt): the waiting is not a line's eight loads in front of its arithmetic.  Nor is it the packet's end: the loops below as ONE more
    // line under per-block predicates -- whole blocks and the ragged one requested together, dwords instead of byte loops, the IV as three dwords -- moved the call by
    // - 2 % .. + 3 % (tail_ab.txt, the patch beside it).  It is where the packets lie: 0.16 ms behind the twin on 64-byte boundaries, 0.33 ms byte-packed, loads and
    // stores in equal parts, and every other cache policy slower (align_probe.txt, loads_stores_ab.txt).)
    while (left >= 128) {
        uint4 xa[4], xb[4];
#pragma unroll
        for (int k = 0; k < 4; k++) xa[k] = pkt_ld<DEC>(src + 16 * k, ctr + k, false);
#pragma unroll
        for (int k = 0; k < 4; k++) xb[k] = pkt_ld<DEC>(src + 64 + 16 * k, ctr + 4 + k, false);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            u32 s0, s1, s2, s3;
            ctr_rounds_lds<NR, T4>(bswap32(ctr + k), cc, s0, s1, s2, s3, rk, smem, lb);
            if (DEC == 1) acc = ghash_mul_const_lds(xor4(acc, xa[k]), smem);
            xa[k] = make_uint4(xa[k].x ^ s0, xa[k].y ^ s1, xa[k].z ^ s2, xa[k].w ^ s3);
            if (DEC != 1) acc = ghash_mul_const_lds(xor4(acc, xa[k]), smem);
        }
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int k = 0; k < 4; k++) {
            u32 s0, s1, s2, s3;
            ctr_rounds_lds<NR, T4>(bswap32(ctr + 4 + k), cc, s0, s1, s2, s3, rk, smem, lb);
            if (DEC == 1) acc = ghash_mul_const_lds(xor4(acc, xb[k]), smem);
            xb[k] = make_uint4(xb[k].x ^ s0, xb[k].y ^ s1, xb[k].z ^ s2, xb[k].w ^ s3);
            if (DEC != 1) acc = ghash_mul_const_lds(xor4(acc, xb[k]), smem);
        }
        // all eight stores back to back: the 128 bytes meet in the L2 and leave it as one full line (WRITE_SIZE = the ciphertext, 1.00 x).  In two groups of
        // four, a tenth of a millisecond apart, 1.22 x; block by block 3.2 x -- the L2 turns over many times while a lane works through its line
        // (profiles/r04/pktl_store_ab.txt).
#pragma unroll
        for (int k = 0; k < 4; k++) pkt_st<DEC>(dst + 16 * k, xa[k], false);
#pragma unroll
        for (int k = 0; k < 4; k++) pkt_st<DEC>(dst + 64 + 16 * k, xb[k], false);
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_sched_barrier(0);
#endif
        src += 128; dst += 128; left -= 128; ctr += 8;
    }
#endif
    while (left >= 16 * AESGCM_PKTL_GROUP) {
        uint4 x[AESGCM_PKTL_GROUP];
#pragma unroll
        for (int k = 0; k < AESGCM_PKTL_GROUP; k++) x[k] = pkt_ld<DEC>(src + 16 * k, ctr + k, false);
#pragma unroll
        for (int k = 0; k < AESGCM_PKTL_GROUP; k++) {
            u32 s0, s1, s2, s3;
            ctr_rounds_lds<NR, T4>(bswap32(ctr + k), cc, s0, s1, s2, s3, rk, smem, lb);
            const uint4 y = make_uint4(x[k].x ^ s0, x[k].y ^ s1, x[k].z ^ s2, x[k].w ^ s3);
            acc = ghash_mul_const_lds(xor4(acc, DEC == 1 ? x[k] : y), smem);
            x[k] = y;
        }
#pragma unroll
        for (int k = 0; k < AESGCM_PKTL_GROUP; k++) pkt_st<DEC>(dst + 16 * k, x[k], false);
        src += 16 * AESGCM_PKTL_GROUP; dst += 16 * AESGCM_PKTL_GROUP; left -= 16 * AESGCM_PKTL_GROUP; ctr += AESGCM_PKTL_GROUP;
    }
    for (; left; ctr++) {
        const u32 nb = left < 16 ? left : 16;
        const bool full = nb == 16;
        const uint4 x = full ? pkt_ld<DEC>(src, ctr, false) : DEC == 2 ? mask_block(pkt_ld<DEC>(src, ctr, false), nb) : load_block_bytes(src, nb);
        u32 s0, s1, s2, s3;
        ctr_rounds_lds<NR, T4>(bswap32(ctr), cc, s0, s1, s2, s3, rk, smem, lb);
        uint4 y = make_uint4(x.x ^ s0, x.y ^ s1, x.z ^ s2, x.w ^ s3);
        if (full) pkt_st<DEC>(dst, y, false);
        else { y = mask_block(y, nb); if (DEC != 2) store_block_bytes(dst, y, nb); }
        acc = ghash_mul_const_lds(xor4(acc, DEC == 1 ? x : y), smem);
        src += nb; dst += nb; left -= nb;
    }
    // [8*len(A)]_64 || [8*len(C)]_64 (gcm_ghash.vhd:257) in memory order
    acc = ghash_mul_const_lds(xor4(acc, make_uint4(0u, bswap32(aad_len * 8u), 0u, bswap32(pkt_len * 8u))), smem);
    u32 s0, s1, s2, s3;
    ctr_rounds_lds<NR, T4>(bswap32(1u), cc, s0, s1, s2, s3, rk, smem, lb);
    const uint4 tag = make_uint4(acc.x ^ s0, acc.y ^ s1, acc.z ^ s2, acc.w ^ s3);       // gcm_ghash.vhd:293
    if ((((uintptr_t)p.tags) & 15) == 0) *reinterpret_cast<uint4 *>(p.tags + (size_t)pkt * 16) = tag;
    else store_block_bytes(p.tags + (size_t)pkt * 16, tag, 16);
    if (DEC == 1 && p.auth) {
        int ok = 1;
        if (p.expect) {
            const uint4 e = load_block_bytes(p.expect + (size_t)pkt * 16, 16);
            ok = ((e.x ^ tag.x) | (e.y ^ tag.y) | (e.z ^ tag.z) | (e.w ^ tag.w)) == 0;
        }
        p.auth[pkt] = ok;
    }
}

