// aesgcm_aes.h -- the AES rounds of the hot loops: T-table lookups from LDS (part of aesgcm_dev.h).
#pragma once
#include "aesgcm_base.h"

// ------------------------------------------------------------------------------------------------
// Hot loop piece 1: Nr AES rounds on one counter/ECB block per lane, T-table lookups from LDS.
//   lb   : (lane&31)*4.  One v_perm_b32 builds each address: (byte_k(s) << 8) | lb; the table base
//          and the T0/T2 select ride in the ds_read offset field.  A wave64 ds_read_b32 is served in
//          two 32-lane groups and lane l always hits bank l&31: conflict-free (MI355X_MICROARCH LDS).
//   rk   : round keys as memory-order words (wave-uniform -> scalar operands).
// Per column: out = T0[r0] ^ T2[r2] ^ rotl8(T0[r1] ^ T2[r3]) ^ rk   (T1 = rotl8 T0, T3 = rotl8 T2).
// This is aes_round's SB->SR->MC (config/config_aes_round.py:121-124) folded into the table, with
// the ARK of the NEXT round (:120) applied at the end, i.e. the standard FIPS-197 bracketing of the
// same cipher; the final round drops MC (:124 cnt = thr) and ends with aes_last_round.vhd:76.
// ------------------------------------------------------------------------------------------------
#define SEL_B(k) (0x0c0c0000u | ((4u + (k)) << 8))
#define T0_AT(lds, s, k, lb) LDS_LD32(lds, perm_b32(s, lb, SEL_B(k)) + AESGCM_LDS_AES_OFF)
#define T2_AT(lds, s, k, lb) LDS_LD32(lds, perm_b32(s, lb, SEL_B(k)) + (AESGCM_LDS_AES_OFF + 128u))

// one full round (SubBytes, ShiftRows, MixColumns, AddRoundKey(rkr)) on the whole state
HD void aes_round_lds(u32 &s0, u32 &s1, u32 &s2, u32 &s3, const u32 *__restrict__ rkr, const unsigned char *lds, u32 lb) {
    const u32 a0 = T0_AT(lds, s0, 0, lb), a1 = T0_AT(lds, s1, 1, lb), a2 = T2_AT(lds, s2, 2, lb), a3 = T2_AT(lds, s3, 3, lb);
    const u32 b0 = T0_AT(lds, s1, 0, lb), b1 = T0_AT(lds, s2, 1, lb), b2 = T2_AT(lds, s3, 2, lb), b3 = T2_AT(lds, s0, 3, lb);
    const u32 c0 = T0_AT(lds, s2, 0, lb), c1 = T0_AT(lds, s3, 1, lb), c2 = T2_AT(lds, s0, 2, lb), c3 = T2_AT(lds, s1, 3, lb);
    const u32 d0 = T0_AT(lds, s3, 0, lb), d1 = T0_AT(lds, s0, 1, lb), d2 = T2_AT(lds, s1, 2, lb), d3 = T2_AT(lds, s2, 3, lb);
    s0 = xor3(a0, a2, rkr[0]) ^ rotl32(a1 ^ a3, 8);
    s1 = xor3(b0, b2, rkr[1]) ^ rotl32(b1 ^ b3, 8);
    s2 = xor3(c0, c2, rkr[2]) ^ rotl32(c1 ^ c3, 8);
    s3 = xor3(d0, d2, rkr[3]) ^ rotl32(d1 ^ d3, 8);
}
// final round: SubBytes + ShiftRows + AddRoundKey.  S[x] sits in bytes 1,2 of T0[x] and bytes 0,3 of
// T2[x], so every output byte is already in place: row0 <- T2 byte0, row1 <- T0 byte1, row2 <- T0 byte2,
// row3 <- T2 byte3.
HD u32 merge_rows(u32 r0, u32 r1, u32 r2, u32 r3) {
    return (r0 & 0x000000ffu) | (r1 & 0x0000ff00u) | (r2 & 0x00ff0000u) | (r3 & 0xff000000u);
}
HD void aes_final_lds(u32 &s0, u32 &s1, u32 &s2, u32 &s3, const u32 *__restrict__ rkr, const unsigned char *lds, u32 lb) {
    const u32 a0 = T2_AT(lds, s0, 0, lb), a1 = T0_AT(lds, s1, 1, lb), a2 = T0_AT(lds, s2, 2, lb), a3 = T2_AT(lds, s3, 3, lb);
    const u32 b0 = T2_AT(lds, s1, 0, lb), b1 = T0_AT(lds, s2, 1, lb), b2 = T0_AT(lds, s3, 2, lb), b3 = T2_AT(lds, s0, 3, lb);
    const u32 c0 = T2_AT(lds, s2, 0, lb), c1 = T0_AT(lds, s3, 1, lb), c2 = T0_AT(lds, s0, 2, lb), c3 = T2_AT(lds, s1, 3, lb);
    const u32 d0 = T2_AT(lds, s3, 0, lb), d1 = T0_AT(lds, s0, 1, lb), d2 = T0_AT(lds, s1, 2, lb), d3 = T2_AT(lds, s2, 3, lb);
    s0 = merge_rows(a0, a1, a2, a3) ^ rkr[0];
    s1 = merge_rows(b0, b1, b2, b3) ^ rkr[1];
    s2 = merge_rows(c0, c1, c2, c3) ^ rkr[2];
    s3 = merge_rows(d0, d1, d2, d3) ^ rkr[3];
}
#ifndef AESGCM_T4
#define AESGCM_T4 1                      /* k_body uses four T-tables (136 KiB of LDS, one 1024-lane workgroup per CU); 0 = the two-table round */
#endif
// Four-table form of the same round (k_body with AESGCM_T4): T1 = rotl8(T0) and T3 = rotl8(T2) sit in a second 64 KiB
// LDS region exactly 65536 bytes above the first, reached by the SAME single v_perm per address: `lb2` = lb | 0x10000 and
// the selector also copies its byte 2.  A column is then two XOR3 -- no rotate (v_alignbit issues at about 0.6 of the plain
// VALU rate on this part, profiles/microbench) and no extra XOR: 8 instructions per round less, 12 % of the row's cycles.
#define SEL_B2(k) (0x0c020000u | ((4u + (k)) << 8))
#define T1_AT(lds, s, k, lb2) LDS_LD32(lds, perm_b32(s, lb2, SEL_B2(k)) + AESGCM_LDS_AES_OFF)
#define T3_AT(lds, s, k, lb2) LDS_LD32(lds, perm_b32(s, lb2, SEL_B2(k)) + (AESGCM_LDS_AES_OFF + 128u))
#define AESGCM_LDS_BYTES_T4 (AESGCM_LDS_BYTES + AESGCM_LDS_AES)
HD void aes_round_lds4(u32 &s0, u32 &s1, u32 &s2, u32 &s3, const u32 *__restrict__ rkr, const unsigned char *lds, u32 lb, u32 lb2) {
    const u32 a0 = T0_AT(lds, s0, 0, lb), a1 = T1_AT(lds, s1, 1, lb2), a2 = T2_AT(lds, s2, 2, lb), a3 = T3_AT(lds, s3, 3, lb2);
    const u32 b0 = T0_AT(lds, s1, 0, lb), b1 = T1_AT(lds, s2, 1, lb2), b2 = T2_AT(lds, s3, 2, lb), b3 = T3_AT(lds, s0, 3, lb2);
    const u32 c0 = T0_AT(lds, s2, 0, lb), c1 = T1_AT(lds, s3, 1, lb2), c2 = T2_AT(lds, s0, 2, lb), c3 = T3_AT(lds, s1, 3, lb2);
    const u32 d0 = T0_AT(lds, s3, 0, lb), d1 = T1_AT(lds, s0, 1, lb2), d2 = T2_AT(lds, s1, 2, lb), d3 = T3_AT(lds, s2, 3, lb2);
    s0 = xor3(xor3(a0, a1, a2), a3, rkr[0]);
    s1 = xor3(xor3(b0, b1, b2), b3, rkr[1]);
    s2 = xor3(xor3(c0, c1, c2), c3, rkr[2]);
    s3 = xor3(xor3(d0, d1, d2), d3, rkr[3]);
}
// generic: state already has rk[0..3] applied
template <int NR>
HD void aes_rounds_lds(u32 &s0, u32 &s1, u32 &s2, u32 &s3, const u32 *__restrict__ rk, const unsigned char *lds, u32 lb) {
#pragma unroll
    for (int r = 1; r < NR; r++) aes_round_lds(s0, s1, s2, s3, rk + 4 * r, lds, lb);
    aes_final_lds(s0, s1, s2, s3, rk + 4 * NR, lds, lb);
}

// CTR specialisation: in a counter block only the last word varies (IV || cnt, aes_icb.vhd:118), so 12 of
// round 1's 16 lookups are the same for every block of the message.  ctr_round1_consts() folds them (and
// round key 1) into four per-message constants once; ctr_rounds_lds() then does 4 lookups in round 1.
struct CtrConsts { u32 c0, c1, c2, c3; };
HD CtrConsts ctr_round1_consts(u32 iv0, u32 iv1, u32 iv2, const u32 *__restrict__ rk, const unsigned char *lds, u32 lb) {
    const u32 s0 = iv0 ^ rk[0], s1 = iv1 ^ rk[1], s2 = iv2 ^ rk[2];
    CtrConsts k;
    k.c0 = xor3(T0_AT(lds, s0, 0, lb), T2_AT(lds, s2, 2, lb), rk[4]) ^ rotl32(T0_AT(lds, s1, 1, lb), 8);   // + rotl8(T2[s3.b3])
    k.c1 = xor3(T0_AT(lds, s1, 0, lb), rotl32(T0_AT(lds, s2, 1, lb) ^ T2_AT(lds, s0, 3, lb), 8), rk[5]);  // + T2[s3.b2]
    k.c2 = xor3(T0_AT(lds, s2, 0, lb), T2_AT(lds, s0, 2, lb), rk[6]) ^ rotl32(T2_AT(lds, s1, 3, lb), 8);   // + rotl8(T0[s3.b1])
    k.c3 = xor3(T2_AT(lds, s1, 2, lb), rotl32(T0_AT(lds, s0, 1, lb) ^ T2_AT(lds, s2, 3, lb), 8), rk[7]);  // + T0[s3.b0]
    return k;
}
template <int NR, bool T4 = false>                                  // T4: rounds 2 .. NR-1 through four T-tables (aes_round_lds4: the kernel staged T1 | T3 as well)
HD void ctr_rounds_lds(u32 ctr_be_word, const CtrConsts &k, u32 &s0, u32 &s1, u32 &s2, u32 &s3,
                       const u32 *__restrict__ rk, const unsigned char *lds, u32 lb) {
    const u32 w3 = ctr_be_word ^ rk[3];
    s0 = k.c0 ^ rotl32(T2_AT(lds, w3, 3, lb), 8);
    s1 = k.c1 ^ T2_AT(lds, w3, 2, lb);
    s2 = k.c2 ^ rotl32(T0_AT(lds, w3, 1, lb), 8);
    s3 = k.c3 ^ T0_AT(lds, w3, 0, lb);
#pragma unroll
    for (int r = 2; r < NR; r++) {
        if (T4) aes_round_lds4(s0, s1, s2, s3, rk + 4 * r, lds, lb, lb | 0x10000u);
        else aes_round_lds(s0, s1, s2, s3, rk + 4 * r, lds, lb);
    }
    aes_final_lds(s0, s1, s2, s3, rk + 4 * NR, lds, lb);
}

// NB counter blocks of one key and IV at once, round by round: first the 16 lookups of every block, then the folds -- NB independent chains whose LDS latencies
// overlap inside ONE wave.  For kernels that run with few waves per SIMD (k_pktl's ILP form): there a wave has to cover the latency itself.
template <int NR, bool T4, int NB>
HD void ctr_rounds_lds_n(u32 ctr0, const CtrConsts &k, uint4 *out, const u32 *__restrict__ rk, const unsigned char *lds, u32 lb) {
    u32 s[NB][4];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const u32 w3 = bswap32(ctr0 + (u32)b) ^ rk[3];
        s[b][0] = k.c0 ^ rotl32(T2_AT(lds, w3, 3, lb), 8);
        s[b][1] = k.c1 ^ T2_AT(lds, w3, 2, lb);
        s[b][2] = k.c2 ^ rotl32(T0_AT(lds, w3, 1, lb), 8);
        s[b][3] = k.c3 ^ T0_AT(lds, w3, 0, lb);
    }
    const u32 lb2 = lb | 0x10000u;
#pragma unroll
    for (int r = 2; r < NR; r++) {
        u32 t[NB][16];
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const u32 s0 = s[b][0], s1 = s[b][1], s2 = s[b][2], s3 = s[b][3];
            if (T4) {
                t[b][0] = T0_AT(lds, s0, 0, lb); t[b][1] = T1_AT(lds, s1, 1, lb2); t[b][2] = T2_AT(lds, s2, 2, lb); t[b][3] = T3_AT(lds, s3, 3, lb2);
                t[b][4] = T0_AT(lds, s1, 0, lb); t[b][5] = T1_AT(lds, s2, 1, lb2); t[b][6] = T2_AT(lds, s3, 2, lb); t[b][7] = T3_AT(lds, s0, 3, lb2);
                t[b][8] = T0_AT(lds, s2, 0, lb); t[b][9] = T1_AT(lds, s3, 1, lb2); t[b][10] = T2_AT(lds, s0, 2, lb); t[b][11] = T3_AT(lds, s1, 3, lb2);
                t[b][12] = T0_AT(lds, s3, 0, lb); t[b][13] = T1_AT(lds, s0, 1, lb2); t[b][14] = T2_AT(lds, s1, 2, lb); t[b][15] = T3_AT(lds, s2, 3, lb2);
            } else {
                t[b][0] = T0_AT(lds, s0, 0, lb); t[b][1] = T0_AT(lds, s1, 1, lb); t[b][2] = T2_AT(lds, s2, 2, lb); t[b][3] = T2_AT(lds, s3, 3, lb);
                t[b][4] = T0_AT(lds, s1, 0, lb); t[b][5] = T0_AT(lds, s2, 1, lb); t[b][6] = T2_AT(lds, s3, 2, lb); t[b][7] = T2_AT(lds, s0, 3, lb);
                t[b][8] = T0_AT(lds, s2, 0, lb); t[b][9] = T0_AT(lds, s3, 1, lb); t[b][10] = T2_AT(lds, s0, 2, lb); t[b][11] = T2_AT(lds, s1, 3, lb);
                t[b][12] = T0_AT(lds, s3, 0, lb); t[b][13] = T0_AT(lds, s0, 1, lb); t[b][14] = T2_AT(lds, s1, 2, lb); t[b][15] = T2_AT(lds, s2, 3, lb);
            }
        }
#pragma unroll
        for (int b = 0; b < NB; b++) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const u32 *q = &t[b][4 * c];
                s[b][c] = T4 ? xor3(xor3(q[0], q[1], q[2]), q[3], rk[4 * r + c]) : (xor3(q[0], q[2], rk[4 * r + c]) ^ rotl32(q[1] ^ q[3], 8));
            }
        }
    }
#pragma unroll
    for (int b = 0; b < NB; b++) {
        u32 s0 = s[b][0], s1 = s[b][1], s2 = s[b][2], s3 = s[b][3];
        aes_final_lds(s0, s1, s2, s3, rk + 4 * NR, lds, lb);
        out[b] = make_uint4(s0, s1, s2, s3);
    }
}

