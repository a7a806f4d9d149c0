// aesgcm_batch.h -- packets with a key each (BASELINE config 5): the per-packet Shoup multiplies and key schedule of k_batch3 (part of aesgcm_dev.h).
#pragma once
#include "aesgcm_base.h"
#include "aesgcm_aes.h"
#include "aesgcm_ghash.h"
#include "aesgcm_stream.h"

// ================================================================================================
// Batch path (BASELINE config 5): many independent packets, each with its OWN key and IV (k_batch3: 8, 16 or 64 lanes per packet).  No per-key context exists,
// so everything key-dependent is rebuilt per packet inside the kernel: aes_kexp, H and E_K(J0), and -- because H-power tables cannot be amortised -- GHASH
// multiplies through Shoup tables of the packet's own constants built in LDS (the two-table form with the reduction delayed: shoup2_mul_dr, "k_batch3
// pieces" below).  Lane l of a packet's group runs Horner with H^(lanes per packet) over its slots; an LG-level cross-lane tree closes the packet.
// shoup_mul (one 16-entry table and a reduction-table read per nibble: the round-2 form) stays as the unit-tested reference of the later multiplies
// (tests/host_emul); BATCH_LDS_* is the LDS image that test gives it.
// ================================================================================================
#define BATCH_LDS_RTAB_OFF (AESGCM_LDS_AES_OFF + AESGCM_LDS_AES)            /* 16 x u32 reduction table of shoup_mul */
#define BATCH_LDS_BYTES (BATCH_LDS_RTAB_OFF + 64u)

struct BatchParams {
    const unsigned char *keys;   // n_pkts * key_len bytes
    const unsigned char *ivs;    // n_pkts * 12 bytes
    const unsigned char *aad;    // n_pkts * aad_len bytes or NULL
    const unsigned char *in;     // n_pkts * pkt_len bytes (pkt_len multiple of 16 => 16-byte aligned blocks)
    unsigned char *out;
    unsigned char *tags;         // enc: n_pkts * 16 written.  dec: computed tags written here too
    const unsigned char *expect; // dec: expected tags or NULL
    int *auth;                   // dec: per-packet 1 = tag ok, 0 = mismatch (NULL = skip)
    u32 *counter; u32 counter_base;
    u32 deal;                    // packets per dispenser fetch
    u32 n_pkts, pkt_len, aad_len;
    u32 aligned;                 // in/out rows are 16-byte aligned for every packet
    // variable-length form (all NULL = fixed pkt_len / aad_len, packets back to back):
    const u64 *data_off;         // n_pkts + 1 byte offsets into in/out: packet p = [data_off[p], data_off[p+1])
    const u64 *aad_off;          // n_pkts + 1 byte offsets into aad (or NULL = no AAD)
    const u32 *perm;             // variable-length form: the order in which the launch takes the packets (by falling length class, k_len_*), or NULL = as they come
    u32 plain;                   // k_batch3: fixed-size aligned records of a whole number of wave-iterations, no AAD: the loop without padding / AAD / ragged-block tests
};
HD u32 batch_map(const BatchParams &p, u32 i) { return p.perm ? p.perm[i] : i; }

// reduction of the 4 bits shifted out by Z*x^4: r(v) for v = Z's last nibble, as the top 16 bits of word 0.
// (bit k of v is GCM bit 124+k; after the shift it is x^(128+k'), reduced with R = 0xE1 || 0^120.)
HD u32 shoup_rem_calc(u32 v) {
    // multiply the nibble (as a 4-bit polynomial sitting at x^124..x^127) by x^4 and reduce: do it literally
    G128 z; z.w[0] = z.w[1] = z.w[2] = 0; z.w[3] = v;          // BE words: low nibble of w3 = GCM bits 124..127
    for (int k = 0; k < 4; k++) {                              // four multiplications by x
        const u32 lsb = 0u - (z.w[3] & 1u);
        z.w[3] = (z.w[3] >> 1) | (z.w[2] << 31); z.w[2] = (z.w[2] >> 1) | (z.w[1] << 31); z.w[1] = (z.w[1] >> 1) | (z.w[0] << 31);
        z.w[0] = (z.w[0] >> 1) ^ (lsb & 0xE1000000u);
    }
    return z.w[0];                                             // only the top 16 bits can be set
}
// squaring is linear over GF(2): spread the coefficients (x^i -> x^2i), then fold the upper 128 coefficients
// back with x^128 = 1 + x + x^2 + x^7 (R = 0xE1 || 0^120, src/ghash_gfmul.vhd:37-64).  About a hundred VALU
// operations and no table, against a full table multiply: used for the c_j = c_(j-1)^2 chain of k_batch.
HD u32 gf_spread16(u32 x) {                                  // bit b -> bit 2b
    x = (x | (x << 8)) & 0x00FF00FFu; x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u; x = (x | (x << 1)) & 0x55555555u;
    return x;
}
// a 256-coefficient polynomial (coefficient i in word i/32 at bit 31 - i%32) folded to 128: the upper half times x^128 = 1 + x + x^2 + x^7
HD G128 gf_reduce256(const u32 *W) {
    const u32 h0 = W[4], h1 = W[5], h2 = W[6], h3 = W[7];    // coefficients 128..255
    // Hh * (1 + x + x^2 + x^7): plain right shifts, the bits that fall off the end are folded once more
    u32 t0 = h0 ^ (h0 >> 1) ^ (h0 >> 2) ^ (h0 >> 7);
    u32 t1 = h1 ^ ((h1 >> 1) | (h0 << 31)) ^ ((h1 >> 2) | (h0 << 30)) ^ ((h1 >> 7) | (h0 << 25));
    u32 t2 = h2 ^ ((h2 >> 1) | (h1 << 31)) ^ ((h2 >> 2) | (h1 << 30)) ^ ((h2 >> 7) | (h1 << 25));
    u32 t3 = h3 ^ ((h3 >> 1) | (h2 << 31)) ^ ((h3 >> 2) | (h2 << 30)) ^ ((h3 >> 7) | (h2 << 25));
    const u32 v = (h3 << 31) ^ (h3 << 30) ^ (h3 << 25);       // overflow polynomial, degree <= 6
    t0 ^= v ^ (v >> 1) ^ (v >> 2) ^ (v >> 7);
    G128 r; r.w[0] = W[0] ^ t0; r.w[1] = W[1] ^ t1; r.w[2] = W[2] ^ t2; r.w[3] = W[3] ^ t3;
    return r;
}
HD G128 gf_sqr(G128 a) {
    // the upper half of word k spreads into product word 2k
    u32 W[8];
#pragma unroll
    for (int k = 0; k < 4; k++) { W[2 * k] = gf_spread16(a.w[k] >> 16) << 1; W[2 * k + 1] = gf_spread16(a.w[k] & 0xFFFFu) << 1; }
    return gf_reduce256(W);
}
// Y * c through the table at LDS byte offset `tab` (16 entries x 4 BE words) and the reduction table
HD G128 shoup_mul(G128 y, const unsigned char *lds, u32 tab) {
    u32 z0 = 0, z1 = 0, z2 = 0, z3 = 0;
#pragma unroll
    for (int wi = 3; wi >= 0; wi--) {
        const u32 yw = y.w[wi];
#pragma unroll
        for (int k = 0; k < 8; k++) {                          // nibbles of the word, last (lowest) first
            const u32 nib16 = (k == 0) ? ((yw << 4) & 0xF0u) : ((yw >> (4 * k - 4)) & 0xF0u);
            if (!(wi == 3 && k == 0)) {                        // Z = Z * x^4 (skipped while Z is still zero)
                const u32 rem = LDS_LD32(lds, ((z3 & 0xFu) << 2) + BATCH_LDS_RTAB_OFF);
                z3 = (z3 >> 4) | (z2 << 28); z2 = (z2 >> 4) | (z1 << 28); z1 = (z1 >> 4) | (z0 << 28);
                z0 = (z0 >> 4) ^ rem;
            }
            const u32x4_t t = LDS_LD128(lds, nib16 + tab);
            z0 ^= t.x; z1 ^= t.y; z2 ^= t.z; z3 ^= t.w;
        }
    }
    G128 z; z.w[0] = z0; z.w[1] = z1; z.w[2] = z2; z.w[3] = z3;
    return z;
}

// ---- k_batch3 pieces -----------------------------------------------------------------------------
// Byte-wise variant of the same method: TWO 16-entry tables per constant c, Th[v] = v*c and Tl[v] = v*c*x^4, so that
//     Y*c = Horner over Y's 16 bytes:  Z = Z*x^8 xor Th[high nibble] xor Tl[low nibble]
// halves the shift-and-reduce steps, and the 8 bits shifted out are reduced arithmetically (x^128 = 1 + x + x^2 + x^7,
// R = 0xE1 || 0^120, src/ghash_gfmul.vhd:37-64) instead of through a table: 16 steps of ~20 VALU + 2 ds_read_b128
// against 31 steps of ~14 VALU + ds_read_b32 + ds_read_b128 (round 2).  Round 4 delays the reduction (shoup2_mul_dr below).
// LDS of k_batch3 behind the T-tables: per packet one (8 lanes per packet) or two 512-byte table slots Th | Tl, then 32 bytes per packet for H and E_K(J0)
#define BATCH3_LDS_TAB_OFF (AESGCM_LDS_AES_OFF + AESGCM_LDS_AES)
#define BATCH3_GROUP_LDS_LG(LG) ((LG) >= 4 ? 1056u : 544u)
#define BATCH3_LDS_BYTES_LG(LG) (BATCH3_LDS_TAB_OFF + (AESGCM_WG / 64u) * (64u >> (LG)) * BATCH3_GROUP_LDS_LG(LG))
// Y * c through the two tables at LDS byte offsets tab (Th) and tab + 256 (Tl), entries = 4 BE words (the form with a reduction per byte: kept as the
// unit-tested reference of shoup2_mul_dr and for -DBATCH3_DR=0 builds)
HD G128 shoup2_mul(G128 y, const unsigned char *lds, u32 tab) {
    u32 z0 = 0, z1 = 0, z2 = 0, z3 = 0;
#pragma unroll
    for (int bi = 15; bi >= 0; bi--) {
        const u32 w = y.w[bi >> 2];
        const int sh = 8 * (3 - (bi & 3));
        const u32 hi = (sh ? (w >> sh) : w) & 0xF0u;
        const u32 lo = (sh ? (w >> (sh - 4)) : (w << 4)) & 0xF0u;
        if (bi != 15) gf_shift8(z0, z1, z2, z3);
        const u32x4_t a = LDS_LD128(lds, hi + tab);
        const u32x4_t c = LDS_LD128(lds, lo + (tab + 256u));
        z0 = xor3(z0, a.x, c.x); z1 = xor3(z1, a.y, c.y); z2 = xor3(z2, a.z, c.z); z3 = xor3(z3, a.w, c.w);
    }
    G128 z; z.w[0] = z0; z.w[1] = z1; z.w[2] = z2; z.w[3] = z3;
    return z;
}

// The same product with the reduction DELAYED (round 4).  shoup2_mul shifts its 128-bit accumulator by a byte -- and reduces the
// byte that falls out -- in front of every byte of y: 16 x 12 instructions that have nothing to do with the table.  Here byte
// 4w + k of y contributes E = Th[hi] ^ Tl[lo] shifted by w WORDS (a register choice, no instruction) and 8k bits, so the bytes
// are taken in the order k = 3..0, w = 0..3 into a 256-coefficient accumulator V that is shifted by a byte only between the four
// k groups (3 x 8 v_alignbit) and folded to 128 coefficients once (gf_reduce256, the tail of gf_sqr).  Same 32 ds_read_b128,
// ~180 VALU instead of ~320.  Degrees: E < 128, the largest shift is 120 -> V < 248 coefficients.
// Addresses: the entry offsets of all four bytes of a word at once -- hn = w & 0xF0F0F0F0 (high nibbles x 16), ln = (w << 4) & 0xF0F0F0F0 (low nibbles x 16) --
// and then ONE v_perm_b32 per table read: byte k of hn / ln under bytes 1, 2 of the slot address (tab is a multiple of 256 below 2^24), the selector
// SHOUP2_SEL(k) in a scalar register of the rolled loop; Tl rides in the ds_read offset field.  11 instructions per word instead of the 20 of shift / mask /
// or per byte; and every accumulator word takes its (up to four) entries in whole xor3s.  Round 4, second pass over k_batch3's issue count: 240 -> 180 VALU
// per multiply.
#define SHOUP2_SEL(k) (0x0c020104u + (u32)(k))
HD G128 shoup2_mul_dr(G128 y, const unsigned char *lds, u32 tab) {      // tab: a multiple of 256
    u32 V[8];
    const u32 h0 = y.w[0] & 0xF0F0F0F0u, h1 = y.w[1] & 0xF0F0F0F0u, h2 = y.w[2] & 0xF0F0F0F0u, h3 = y.w[3] & 0xF0F0F0F0u;
    const u32 l0 = (y.w[0] << 4) & 0xF0F0F0F0u, l1 = (y.w[1] << 4) & 0xF0F0F0F0u, l2 = (y.w[2] << 4) & 0xF0F0F0F0u, l3 = (y.w[3] << 4) & 0xF0F0F0F0u;
#define SHOUP2_DR_LOADS(sel) \
        const u32x4_t a0 = LDS_LD128(lds, perm_b32(h0, tab, sel)), c0 = LDS_LD128(lds, perm_b32(l0, tab, sel) + 256u); \
        const u32x4_t a1 = LDS_LD128(lds, perm_b32(h1, tab, sel)), c1 = LDS_LD128(lds, perm_b32(l1, tab, sel) + 256u); \
        const u32x4_t a2 = LDS_LD128(lds, perm_b32(h2, tab, sel)), c2 = LDS_LD128(lds, perm_b32(l2, tab, sel) + 256u); \
        const u32x4_t a3 = LDS_LD128(lds, perm_b32(h3, tab, sel)), c3 = LDS_LD128(lds, perm_b32(l3, tab, sel) + 256u);
    {   // k = 3, the lowest byte of every word: nothing to shift yet
        SHOUP2_DR_LOADS(SHOUP2_SEL(0))
        V[7] = 0;
        V[6] = a3.w ^ c3.w;
        V[5] = xor3(a3.z, c3.z, a2.w) ^ c2.w;
        V[4] = xor3(xor3(a3.y, c3.y, a2.z), c2.z, a1.w) ^ c1.w;
        V[3] = xor3(xor3(xor3(a3.x, c3.x, a2.y), c2.y, a1.z), c1.z, a0.w) ^ c0.w;
        V[2] = xor3(xor3(a2.x, c2.x, a1.y), c1.y, a0.z) ^ c0.z;
        V[1] = xor3(a1.x, c1.x, a0.y) ^ c0.y;
        V[0] = a0.x ^ c0.x;
    }
    // a real loop over the other three k groups (eight table loads in flight, not 32: fully unrolled, the kernels that use it spilled)
#pragma unroll 1
    for (int t = 1; t < 4; t++) {
        SHOUP2_DR_LOADS(SHOUP2_SEL(t))
        // V >>= 8, then the group's eight entries at word offsets 0 .. 3
        V[7] = (V[7] >> 8) | (V[6] << 24);
        V[6] = xor3((V[6] >> 8) | (V[5] << 24), a3.w, c3.w);
        V[5] = xor3(xor3((V[5] >> 8) | (V[4] << 24), a3.z, c3.z), a2.w, c2.w);
        V[4] = xor3(xor3(xor3((V[4] >> 8) | (V[3] << 24), a3.y, c3.y), a2.z, c2.z), a1.w, c1.w);
        V[3] = xor3(xor3(xor3(xor3((V[3] >> 8) | (V[2] << 24), a3.x, c3.x), a2.y, c2.y), a1.z, c1.z), a0.w, c0.w);
        V[2] = xor3(xor3(xor3((V[2] >> 8) | (V[1] << 24), a2.x, c2.x), a1.y, c1.y), a0.z, c0.z);
        V[1] = xor3(xor3((V[1] >> 8) | (V[0] << 24), a1.x, c1.x), a0.y, c0.y);
        V[0] = xor3(V[0] >> 8, a0.x, c0.x);
    }
#undef SHOUP2_DR_LOADS
    return gf_reduce256(V);
}

// Half of that product, for the split of ONE multiply over TWO lanes (the reference's own trick, src/gcm_ghash.vhd:317-333:
// X*H = (Xhi || 0)*H ^ (0 || Xlo)*H).  (s0, s1) are taken as words 0, 1 of the multiplicand: the owner of a value passes its words 0, 1,
// the helper lane passes words 2, 3 and its partial then stands two WORDS further down (shoup2_pair_join).  16 table reads, unreduced
// 6-word partial.  tab: a multiple of 256.
HD void shoup2_half_dr(u32 s0, u32 s1, const unsigned char *lds, u32 tab, u32 *V) {
    const u32 h0 = s0 & 0xF0F0F0F0u, h1 = s1 & 0xF0F0F0F0u, l0 = (s0 << 4) & 0xF0F0F0F0u, l1 = (s1 << 4) & 0xF0F0F0F0u;
#define SHOUP2_HALF_LOADS(sel) \
        const u32x4_t a0 = LDS_LD128(lds, perm_b32(h0, tab, sel)), c0 = LDS_LD128(lds, perm_b32(l0, tab, sel) + 256u); \
        const u32x4_t a1 = LDS_LD128(lds, perm_b32(h1, tab, sel)), c1 = LDS_LD128(lds, perm_b32(l1, tab, sel) + 256u);
    {
        SHOUP2_HALF_LOADS(SHOUP2_SEL(0))
        V[5] = 0;
        V[4] = a1.w ^ c1.w;
        V[3] = xor3(a1.z, c1.z, a0.w) ^ c0.w;
        V[2] = xor3(a1.y, c1.y, a0.z) ^ c0.z;
        V[1] = xor3(a1.x, c1.x, a0.y) ^ c0.y;
        V[0] = a0.x ^ c0.x;
    }
#pragma unroll 1
    for (int t = 1; t < 4; t++) {
        SHOUP2_HALF_LOADS(SHOUP2_SEL(t))
        V[5] = (V[5] >> 8) | (V[4] << 24);
        V[4] = xor3((V[4] >> 8) | (V[3] << 24), a1.w, c1.w);
        V[3] = xor3(xor3((V[3] >> 8) | (V[2] << 24), a1.z, c1.z), a0.w, c0.w);
        V[2] = xor3(xor3((V[2] >> 8) | (V[1] << 24), a1.y, c1.y), a0.z, c0.z);
        V[1] = xor3(xor3((V[1] >> 8) | (V[0] << 24), a1.x, c1.x), a0.y, c0.y);
        V[0] = xor3(V[0] >> 8, a0.x, c0.x);
    }
#undef SHOUP2_HALF_LOADS
}
// owner's partial (words 0, 1 of the value) and helper's partial (words 2, 3): the product
HD G128 shoup2_pair_join(const u32 *Vo, const u32 *Vh) {
    u32 R[8];
    R[0] = Vo[0]; R[1] = Vo[1]; R[2] = Vo[2] ^ Vh[0]; R[3] = Vo[3] ^ Vh[1]; R[4] = Vo[4] ^ Vh[2]; R[5] = Vo[5] ^ Vh[3]; R[6] = Vh[4]; R[7] = Vh[5];
    return gf_reduce256(R);
}

// aes_kexp per packet (config/config_aes_kexp.py:128-159 / tb/key_exp.py:79-114) on memory-order words, S-box
// taken from byte 1 of the LDS T0 entry.  Every lane computes the same words (uniform addresses broadcast).
template <int NR>
HD void batch_key_expand(const unsigned char *key, u32 *rk, const unsigned char *lds, u32 lb) {
    constexpr int NK = NR - 6;
#pragma unroll
    for (int w = 0; w < NK; w++) rk[w] = load_le32(key + 4 * w);
    u32 rcon = 1;
#pragma unroll
    for (int w = NK; w < 4 * (NR + 1); w++) {
        u32 t = rk[w - 1];
        if (w % NK == 0) {
            t = (t >> 8) | (t << 24);                                            // RotWord on a little-endian word
            t = ((T0_AT(lds, t, 0, lb) >> 8) & 0xFFu) | (T0_AT(lds, t, 1, lb) & 0xFF00u) |
                (T0_AT(lds, t, 2, lb) & 0xFF0000u) | ((T0_AT(lds, t, 3, lb) << 8) & 0xFF000000u);   // SubWord
            t ^= rcon;
            rcon = xtime2(rcon);
        } else if (NK == 8 && (w % NK) == 4) {
            t = ((T0_AT(lds, t, 0, lb) >> 8) & 0xFFu) | (T0_AT(lds, t, 1, lb) & 0xFF00u) |
                (T0_AT(lds, t, 2, lb) & 0xFF0000u) | ((T0_AT(lds, t, 3, lb) << 8) & 0xFF000000u);
        }
        rk[w] = rk[w - NK] ^ t;
    }
}

