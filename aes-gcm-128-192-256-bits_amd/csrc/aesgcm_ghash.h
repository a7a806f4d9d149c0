// aesgcm_ghash.h -- GHASH multiplies by a launch constant through tables in LDS (part of aesgcm_dev.h).
#pragma once
#include "aesgcm_base.h"

// ------------------------------------------------------------------------------------------------
// Hot loop piece 2: multiply the lane's GHASH accumulator by the launch constant K = H^(lane stride)
// through tables in LDS: Y*K = xor_p T_p[group_p(Y)] (multiplication by a constant is GF(2)-linear -- the
// generalisation of the RTL's 2-way split, src/gcm_ghash.vhd:317-333).
//
// ghash_mul_const_lds (the row loops of k_main, k_body, k_pkt, k_pktl): 26 FIVE-bit tables read with ds_read_b64.
// The LDS array serves a wave64 ds_read_b64 in 2 cycles (32 lanes x 8 B = all 64 banks) and a ds_read_b128 in 4
// (MI355X_MICROARCH LDS table), so a table position costs 2 x 2 cycles for 5 bits against 4 cycles for 4 bits with
// 16-byte nibble-table entries: 104 array cycles per multiply instead of 128, in kernels whose binding unit is that
// array (round 2, profiles/archive/r02f/gh5_ab.txt: 886 -> 919 GiB/s on one box).  A five-bit table of 8-byte half entries is
// 32 x 8 B = one 256-byte bank row: two lanes of a 32-lane group read either the same address (broadcast) or different
// banks -- conflict-free by construction.  Groups are cut from the four memory-order dwords taken as one 128-bit integer
// (quint_elem_mo; any partition of the coordinates serves a linear map); three groups straddle a dword boundary (one
// v_alignbit each).  Layout from AESGCM_LDS_GH_OFF: row p = low halves (.x .y) of table p, row 27 + p = high halves
// (.z .w); row 26 stays empty so that the two halves are 6912 bytes apart, NOT a multiple of 512: otherwise the compiler
// fuses the pair into one ds_read2st64_b64, which the LDS serves as 2 x (4 x 16 lanes) = 8 cycles instead of 2 + 2.
// ------------------------------------------------------------------------------------------------
HD uint4 ghash_mul_q5_lds(uint4 y, const unsigned char *lds, const u32 base) {
    const u32 w[4] = {y.x, y.y, y.z, y.w};
    u32 r[4] = {0, 0, 0, 0}, t[4] = {0, 0, 0, 0};
#pragma unroll
    for (int p = 0; p < AESGCM_Q5_GROUPS; p++) {
        const int bit = 5 * p, wi = bit >> 5, sh = bit & 31;
        u32 x;                                                            // the group's value at bits 3..7
        if (sh > 27 && wi < 3) x = (u32)((((u64)w[wi + 1] << 32) | w[wi]) >> (sh - 3));
        else x = sh >= 3 ? w[wi] >> (sh - 3) : w[wi] << (3 - sh);
        const u32 a = x & 0xF8u;
        const u32x2_t l = LDS_LD64(lds, a + (base + (u32)p * 256u));
        const u32x2_t h = LDS_LD64(lds, a + (base + (u32)(AESGCM_Q5_HI_ROW + p) * 256u));
        if (p & 1) { r[0] = xor3(r[0], t[0], l.x); r[1] = xor3(r[1], t[1], l.y); r[2] = xor3(r[2], t[2], h.x); r[3] = xor3(r[3], t[3], h.y); }
        else { t[0] = l.x; t[1] = l.y; t[2] = h.x; t[3] = h.y; }
    }
    return make_uint4(r[0], r[1], r[2], r[3]);
}
// the launch constant's tables at LDS offset 0 (the row loops): the table base rides in the ds_read offset field
HD uint4 ghash_mul_const_lds(uint4 y, const unsigned char *lds) { return ghash_mul_q5_lds(y, lds, AESGCM_LDS_GH_OFF); }
// what thread `tid` of `nthreads` writes of the LDS image of a five-bit table set `src` (AESGCM_Q5_ENTRIES entries, p*32 + v)
HD void fill_lds_q5(unsigned char *smem, const uint4 *src, u32 tid, u32 nthreads, u32 base = AESGCM_LDS_GH_OFF) {
    for (u32 q = tid; q < AESGCM_Q5_ENTRIES; q += nthreads) {
        const uint4 e = src[q];
        const u32 p = q >> 5, v = q & 31u;
        u32 *lo = reinterpret_cast<u32 *>(smem + base + p * 256u + v * 8u);
        u32 *hi = reinterpret_cast<u32 *>(smem + base + (AESGCM_Q5_HI_ROW + p) * 256u + v * 8u);
        lo[0] = e.x; lo[1] = e.y; hi[0] = e.z; hi[1] = e.w;
    }
}

// The nibble-table form (k_fold, k_combine: constants that change per launch, tables of 512 x 16 B at any LDS offset):
// table p is one 256-byte LDS bank row (16 entries x 16 B), so within a ds_read_b128 lane group two lanes either read the
// same address (broadcast) or different 16-byte slots: conflict-free by construction.
HD uint4 ghash_mul_const_lds_at(uint4 y, const unsigned char *lds, u32 base) {
    u32x4_t r = {0, 0, 0, 0};
    const u32 w[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
    for (int wi = 0; wi < 4; wi++) {
#pragma unroll
        for (int bb = 0; bb < 4; bb++) {
            const int b = 4 * wi + bb;
            const u32 hi = (bb == 0) ? (w[wi] & 0xF0u) : ((w[wi] >> (8 * bb)) & 0xF0u);
            const u32 lo = (bb == 0) ? ((w[wi] << 4) & 0xF0u) : ((w[wi] >> (8 * bb - 4)) & 0xF0u);
            const u32x4_t a = LDS_LD128(lds, hi + (base + (2 * b) * 256));
            const u32x4_t c = LDS_LD128(lds, lo + (base + (2 * b + 1) * 256));
            r.x = xor3(r.x, a.x, c.x); r.y = xor3(r.y, a.y, c.y); r.z = xor3(r.z, a.z, c.z); r.w = xor3(r.w, a.w, c.w);
        }
    }
    return make_uint4(r.x, r.y, r.z, r.w);
}

// the same multiply as a LOOP over the four words of y (not unrolled: eight table loads, 32 registers of entries, in flight instead of up to 128) --
// for k_body's fused closing, which has 128 registers in all and must not spill: every wave of the launch runs it.  The words rotate through w0 so
// that nothing is indexed by the loop counter.
HD uint4 ghash_mul_const_lds_at_lean(uint4 y, const unsigned char *lds, u32 base) {
    u32x4_t r = {0, 0, 0, 0};
    u32 w0 = y.x, w1 = y.y, w2 = y.z, w3 = y.w;
#pragma unroll 1
    for (u32 wi = 0; wi < 4; wi++) {
#pragma unroll
        for (int bb = 0; bb < 4; bb++) {
            const u32 hi = (bb == 0) ? (w0 & 0xF0u) : ((w0 >> (8 * bb)) & 0xF0u);
            const u32 lo = (bb == 0) ? ((w0 << 4) & 0xF0u) : ((w0 >> (8 * bb - 4)) & 0xF0u);
            const u32x4_t a = LDS_LD128(lds, hi + (base + (2 * bb) * 256));
            const u32x4_t c = LDS_LD128(lds, lo + (base + (2 * bb + 1) * 256));
            r.x = xor3(r.x, a.x, c.x); r.y = xor3(r.y, a.y, c.y); r.z = xor3(r.z, a.z, c.z); r.w = xor3(r.w, a.w, c.w);
        }
        base += 2048u;
        w0 = w1; w1 = w2; w2 = w3;
    }
    return make_uint4(r.x, r.y, r.z, r.w);
}

