// aesgcm_base.h -- types, global-memory accessors, the literal (byte / bit) AES and GF(2^128) arithmetic, LDS access (part of aesgcm_dev.h: see there).
//
// Everything here is __host__ __device__ so that the exact lane code the kernels run can also be
// driven by a CPU harness (tests/host_emul) in the GPU-less build container.  The kernels that
// compose these pieces are in aesgcm_kernels.hip.
//
// Data conventions (DESIGN.md "Layout"):
//   * a 16-byte block lives in registers as 4 dwords in MEMORY order ("mo"): d0 = bytes 0..3 loaded
//     little-endian, exactly what global_load_dwordx4 returns.  AES state columns are therefore
//     little-endian words (row 0 in the low byte); the T-table is built for that convention, so no
//     byte swap is ever needed on the data path.
//   * GF(2^128) arithmetic that needs shifts (the bit-serial multiply) works on big-endian words
//     ("be"): w[0] holds GCM bits 0..31 with bit 0 in the MSB (src/ghash_gfmul.vhd:44-57: VHDL bit
//     127 = leftmost).  mo <-> be is one byte swap per word.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32;
typedef uint64_t u64;

#ifndef AESGCM_LOG_WG
#define AESGCM_LOG_WG 10       /* log2(lanes per workgroup); 9 and 10 are supported */
#endif
#define AESGCM_WG (1 << AESGCM_LOG_WG)   /* lanes per workgroup = GHASH lane stride S = radix of the H-power tables */
#define AESGCM_GMAX 512        /* max workgroups per launch (2 per CU on 256 CUs) */
#ifndef AESGCM_MAIN_WG
/* lanes per k_main / k_body workgroup (waves are autonomous: any multiple of 64), two workgroups per CU.  768 = 6 waves
   per SIMD = an 80-register budget: at 1024 (8 per SIMD, 64 registers) both kernels spilled lane constants to scratch
   and reloaded them inside the row loop (round-1 ISA: ScratchSize 36/32, three scratch_load per row); at 768 and 896
   ScratchSize is 0.  Measured on one box, 16 GiB AES-256: 1024 -> 18.22/18.29 ms, 896 -> 18.50/18.59, 768 -> 17.72/17.77. */
#define AESGCM_MAIN_WG 768
#endif
#ifndef AESGCM_PKT_WG
#define AESGCM_PKT_WG AESGCM_WG           /* lanes per k_pkt / k_pktl workgroup */
#endif
#define AESGCM_NPW (AESGCM_WG + 1)       /* entries per power table: exponent digits 0..WG */
#define AESGCM_Q5_GROUPS 26     /* five-bit groups of a 128-bit value (the last has three bits) */
#define AESGCM_Q5_HI_ROW (AESGCM_Q5_GROUPS + 1)
#define AESGCM_Q5_ENTRIES (AESGCM_Q5_GROUPS * 32)
#define AESGCM_LDS_GH ((AESGCM_Q5_HI_ROW + AESGCM_Q5_GROUPS) * 256)   /* bytes: 13568 = 53 LDS rows: the five-bit GHASH tables of the launch constant (ghash_mul_const_lds) */
#define AESGCM_LDS_DRY_OFF (AESGCM_Q5_GROUPS * 256)  /* the spare row between the table halves: one u32 there is the workgroup's dry-queue mask (k_main / k_body dispensers) */
#define AESGCM_LDS_AES 65536   /* bytes: 256 entries x (32 replicas of T0 | 32 replicas of T2) */
#define AESGCM_LDS_BYTES (AESGCM_LDS_AES + AESGCM_LDS_GH)

#define HD __host__ __device__ __forceinline__

struct G128 { u32 w[4]; };     // big-endian words (math form)

HD u32 bswap32(u32 x) { return __builtin_bswap32(x); }
HD u32 rotl32(u32 x, int r) { return (x << r) | (x >> (32 - r)); }

HD G128 mo_to_be(uint4 m) { G128 g; g.w[0] = bswap32(m.x); g.w[1] = bswap32(m.y); g.w[2] = bswap32(m.z); g.w[3] = bswap32(m.w); return g; }
HD uint4 be_to_mo(G128 g) { return make_uint4(bswap32(g.w[0]), bswap32(g.w[1]), bswap32(g.w[2]), bswap32(g.w[3])); }
// 16-byte accesses to device memory that is known to be global: the pointers reach the kernels inside parameter
// structs as generic pointers, and a flat_load is served in 64-byte L2 requests where a global_load gets 128-byte
// ones (TCC_READ per byte: 1/62 vs 1/91, profiles/pmc_tcc.sh)
#if defined(__HIP_DEVICE_COMPILE__)
typedef u32 gvec4_t __attribute__((ext_vector_type(4)));
HD uint4 gload16(const void *p) {
    const gvec4_t v = *(const __attribute__((address_space(1))) gvec4_t *)(uintptr_t)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}
HD void gstore16(void *p, uint4 v) {
    gvec4_t w = {v.x, v.y, v.z, v.w};
    *(__attribute__((address_space(1))) gvec4_t *)(uintptr_t)p = w;
}
// ... at ANY byte address (packets packed back to back start wherever the previous one ended).  The target runs with unaligned access mode on (the compiler
// itself emits global_load_dwordx4 for an align-1 vector), so a whole block is one access whatever its address; only what is shorter than a block goes
// byte by byte.  Round 4: 2^20 packed frames under one key 133 -> 673 GiB/s (profiles/r04/packets_sweep_packed_*.txt) -- sixteen byte loads and sixteen byte
// stores per block before.
typedef u32 gvec4u_t __attribute__((ext_vector_type(4), aligned(1)));
HD uint4 gload16_any(const void *p) {
    const gvec4u_t v = *(const __attribute__((address_space(1))) gvec4u_t *)(uintptr_t)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}
HD void gstore16_any(void *p, uint4 v) {
    gvec4u_t w = {v.x, v.y, v.z, v.w};
    *(__attribute__((address_space(1))) gvec4u_t *)(uintptr_t)p = w;
}
// ... a dword at any byte address (the IVs of a call: 12 bytes apart from a base of any alignment)
typedef u32 gu32u_t __attribute__((aligned(1)));
HD u32 gload4_any(const void *p) { return *(const __attribute__((address_space(1))) gu32u_t *)(uintptr_t)p; }
// ... written THROUGH the XCD's L2 to memory (sc0 sc1): the line does not stay dirty in the L2, so nothing of it is left for a write-back at the end
// of the launch -- or, in a launch that publishes its result from inside (k_body's fused closing), before the result may be shown.  `base` is
// wave-uniform, `off` the lane's byte offset.  The s_nop covers the store-data hazard the compiler cannot see inside
// the asm (a VALU write of the data registers within two wait states of a store wider than 64 bits, gfx940 and later).
HD void gstore16_wt(unsigned char *base, u32 off, uint4 v) {
    gvec4_t w = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1\n\ts_nop 1" :: "v"(off), "v"(w), "s"(base) : "memory");
}
// ... at a per-lane address (the general rows of k_main's lane code), whole blocks, dwords and single bytes
HD void gstore16_wt_at(void *p, uint4 v) {
    gvec4_t w = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(w) : "memory");
}
HD void gstore4_wt_at(void *p, u32 v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory"); }
HD void gstore1_wt_at(void *p, u32 v) { asm volatile("global_store_byte %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory"); }
#else
HD uint4 gload16(const void *p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }           // (host harness: the row code of k_rows runs on packets packed from any byte address)
HD void gstore16(void *p, uint4 v) { __builtin_memcpy(p, &v, 16); }
HD uint4 gload16_any(const void *p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }
HD void gstore16_any(void *p, uint4 v) { __builtin_memcpy(p, &v, 16); }
HD u32 gload4_any(const void *p) { u32 v; __builtin_memcpy(&v, p, 4); return v; }
HD void gstore16_wt(unsigned char *base, u32 off, uint4 v) { __builtin_memcpy(base + off, &v, 16); }
HD void gstore16_wt_at(void *p, uint4 v) { __builtin_memcpy(p, &v, 16); }
HD void gstore4_wt_at(void *p, u32 v) { *reinterpret_cast<u32 *>(p) = v; }
HD void gstore1_wt_at(void *p, u32 v) { *reinterpret_cast<unsigned char *>(p) = (unsigned char)v; }
#endif
HD uint4 xor4(uint4 a, uint4 b) { return make_uint4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w); }

// v_perm_b32: result byte i = pool[sel.byte[i]] with pool = {src1 bytes 0..3, src0 bytes 4..7},
// selector 0x0c = constant 0x00.
HD u32 perm_b32(u32 src0, u32 src1, u32 sel) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(src0, src1, sel);
#else
    u64 pool = ((u64)src0 << 32) | src1;
    u32 r = 0;
    for (int i = 0; i < 4; i++) {
        u32 s = (sel >> (8 * i)) & 0xff;
        u32 b = (s <= 7) ? (u32)((pool >> (8 * s)) & 0xff) : 0u;   // only 0..7 and 0x0c are used here
        r |= b << (8 * i);
    }
    return r;
#endif
}

// ------------------------------------------------------------------------------------------------
// GF(2^8) helpers and the S-box (src/aes_func.vhd:187-210 xtime2/xtime3, :228-301 sbox).  The S-box
// is computed from its FIPS-197 definition by a 256-thread init kernel, never typed in.
// ------------------------------------------------------------------------------------------------
HD u32 xtime2(u32 d) { return ((d << 1) ^ ((d & 0x80) ? 0x1Bu : 0u)) & 0xff; }
HD u32 gf8_mul(u32 a, u32 b) { u32 r = 0; for (int i = 0; i < 8; i++) { if (b & 1) r ^= a; a = xtime2(a); b >>= 1; } return r; }
HD u32 sbox_calc(u32 x) {
    u32 inv = 0;
    if (x) { u32 p = 1, b = x; for (int e = 254; e; e >>= 1) { if (e & 1) p = gf8_mul(p, b); b = gf8_mul(b, b); } inv = p; }
    u32 s = inv, r = inv;
    for (int k = 0; k < 4; k++) { r = ((r << 1) | (r >> 7)) & 0xff; s ^= r; }
    return s ^ 0x63;
}
// T0 in the memory-order convention: column bytes (row0..row3) = (2s, s, s, 3s) -> LE word.
// = mix_columns(aes_func.vhd:159-169) applied to a column whose row 0 holds sbox(x).
HD u32 te0_calc(u32 s) { u32 s2 = xtime2(s); return s2 | (s << 8) | (s << 16) | ((s2 ^ s) << 24); }

// ------------------------------------------------------------------------------------------------
// Literal single-block AES and key schedule, byte oriented, in the reference's own bracketing.
// Used only in one-off setup lanes (key expansion, H, E_K(J0)); the bulk path is aes_rounds_lds().
// ------------------------------------------------------------------------------------------------
// FIPS-197 KeyExpansion (tb/key_exp.py:79-114; config/config_aes_kexp.py:128-159: RotWord/SubWord,
// rcon doubled by xtime2 :150, 256-bit "skip" step = SubWord only :147-152).  rk = 16*(nr+1) bytes.
HD int key_expand_bytes(const uint8_t *key, int key_len, const uint8_t *sbox, uint8_t *rk) {
    int nk = key_len / 4, nr = nk + 6, total = 4 * (nr + 1);
    for (int i = 0; i < key_len; i++) rk[i] = key[i];
    u32 rcon = 1;
    for (int w = nk; w < total; w++) {
        uint8_t t0 = rk[4 * w - 4], t1 = rk[4 * w - 3], t2 = rk[4 * w - 2], t3 = rk[4 * w - 1];
        if (w % nk == 0) {
            uint8_t r0 = sbox[t1], r1 = sbox[t2], r2 = sbox[t3], r3 = sbox[t0];      // rot_word then sub_word
            t0 = (uint8_t)(r0 ^ rcon); t1 = r1; t2 = r2; t3 = r3;
            rcon = xtime2(rcon);
        } else if (nk == 8 && (w % nk) == 4) {
            t0 = sbox[t0]; t1 = sbox[t1]; t2 = sbox[t2]; t3 = sbox[t3];
        }
        rk[4 * w + 0] = rk[4 * (w - nk) + 0] ^ t0; rk[4 * w + 1] = rk[4 * (w - nk) + 1] ^ t1;
        rk[4 * w + 2] = rk[4 * (w - nk) + 2] ^ t2; rk[4 * w + 3] = rk[4 * (w - nk) + 3] ^ t3;
    }
    return nr;
}
// round r = 1..Nr: s = MC?(SR(SB(s ^ k[r-1]))), MC skipped at r = Nr (config/config_aes_round.py:120-126);
// then out = s ^ k[Nr] (src/aes_last_round.vhd:76).  State byte 4*c + r = column c, row r.
HD void aes_block_bytes(const uint8_t *rk, int nr, const uint8_t *sbox, const uint8_t in[16], uint8_t out[16]) {
    uint8_t s[16], t[16];
    for (int i = 0; i < 16; i++) s[i] = in[i];
    for (int r = 1; r <= nr; r++) {
        for (int i = 0; i < 16; i++) s[i] = sbox[s[i] ^ rk[16 * (r - 1) + i]];                 // ARK, SubBytes
        for (int c = 0; c < 4; c++) for (int q = 0; q < 4; q++) t[4 * c + q] = s[4 * ((c + q) & 3) + q];   // ShiftRows
        if (r != nr) {
            for (int c = 0; c < 4; c++) {                                                      // MixColumns
                u32 a0 = t[4 * c], a1 = t[4 * c + 1], a2 = t[4 * c + 2], a3 = t[4 * c + 3];
                s[4 * c + 0] = (uint8_t)(xtime2(a0) ^ xtime2(a1) ^ a1 ^ a2 ^ a3);
                s[4 * c + 1] = (uint8_t)(a0 ^ xtime2(a1) ^ xtime2(a2) ^ a2 ^ a3);
                s[4 * c + 2] = (uint8_t)(a0 ^ a1 ^ xtime2(a2) ^ xtime2(a3) ^ a3);
                s[4 * c + 3] = (uint8_t)(xtime2(a0) ^ a0 ^ a1 ^ a2 ^ xtime2(a3));
            }
        } else {
            for (int i = 0; i < 16; i++) s[i] = t[i];
        }
    }
    for (int i = 0; i < 16; i++) out[i] = s[i] ^ rk[16 * nr + i];
}

// ------------------------------------------------------------------------------------------------
// GF(2^128): bit-serial multiply, SP 800-38D Algorithm 1 as src/ghash_gfmul.vhd:37-64 states it
// (V starts as the second operand, is shifted right once per bit of the first, R = 0xE1 || 0^120).
// Variable x variable; used off the hot loop only (setup tables, per-lane tail power, combine).
// ------------------------------------------------------------------------------------------------
HD G128 gf_mul(G128 x, G128 v) {
    u32 z0 = 0, z1 = 0, z2 = 0, z3 = 0;
    u32 v0 = v.w[0], v1 = v.w[1], v2 = v.w[2], v3 = v.w[3];
    u32 x0 = x.w[0], x1 = x.w[1], x2 = x.w[2], x3 = x.w[3];
#pragma unroll 1
    for (int wi = 0; wi < 4; wi++) {
        u32 xw = x0; x0 = x1; x1 = x2; x2 = x3;
#pragma unroll 8
        for (int b = 0; b < 32; b++) {
            u32 m = (u32)((int32_t)xw >> 31);      // GCM bit order: MSB first
            xw <<= 1;
            z0 ^= v0 & m; z1 ^= v1 & m; z2 ^= v2 & m; z3 ^= v3 & m;
            u32 lsb = 0u - (v3 & 1u);
            v3 = (v3 >> 1) | (v2 << 31); v2 = (v2 >> 1) | (v1 << 31); v1 = (v1 >> 1) | (v0 << 31);
            v0 = (v0 >> 1) ^ (lsb & 0xE1000000u);
        }
    }
    G128 z; z.w[0] = z0; z.w[1] = z1; z.w[2] = z2; z.w[3] = z3;
    return z;
}
// multiply a field element by x (one right shift with reduction)
HD G128 gf_mulx(G128 v) {
    const u32 lsb = 0u - (v.w[3] & 1u);
    G128 r;
    r.w[3] = (v.w[3] >> 1) | (v.w[2] << 31); r.w[2] = (v.w[2] >> 1) | (v.w[1] << 31); r.w[1] = (v.w[1] >> 1) | (v.w[0] << 31);
    r.w[0] = (v.w[0] >> 1) ^ (lsb & 0xE1000000u);
    return r;
}
HD G128 gf_mulx4(G128 v) { return gf_mulx(gf_mulx(gf_mulx(gf_mulx(v)))); }
// entry v (0..15) of the Shoup table of constant c: (v as polynomial v3 + v2 x + v1 x^2 + v0 x^3, GCM bit order:
// the nibble's MSB is x^0) times c
HD G128 shoup_entry(G128 c, u32 v) {
    G128 r; r.w[0] = r.w[1] = r.w[2] = r.w[3] = 0;
    G128 t = c;
    for (int k = 3; k >= 0; k--) {                             // bit 3 of v <-> x^0, bit 0 <-> x^3
        const u32 m = 0u - ((v >> k) & 1u);
        r.w[0] ^= t.w[0] & m; r.w[1] ^= t.w[1] & m; r.w[2] ^= t.w[2] & m; r.w[3] ^= t.w[3] & m;
        t = gf_mulx(t);
    }
    return r;
}
HD uint4 gf_mul_mo(uint4 a, uint4 b) { return be_to_mo(gf_mul(mo_to_be(a), mo_to_be(b))); }
HD uint4 gf_one_mo() { return make_uint4(0x80u, 0u, 0u, 0u); }   // the field's 1: byte 0 = 0x80

// element whose nibble position p (0 = high nibble of byte 0 ... 31 = low nibble of byte 15) holds v
HD uint4 nibble_elem_mo(int p, u32 v) {
    u32 w[4] = {0, 0, 0, 0};
    int b = p >> 1;
    u32 byte = (p & 1) ? v : (v << 4);
    w[b >> 2] = byte << (8 * (b & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// element whose REGISTER bits [5p, 5p+5) hold v, reading the four memory-order dwords as one 128-bit little-endian
// integer (p = 0..25; group 25 has three bits).  Any partition of the 128 coordinates serves a GF(2)-linear map.
HD uint4 quint_elem_mo(int p, u32 v) {
    u32 w[4] = {0, 0, 0, 0};
    const int bit = 5 * p, wi = bit >> 5, sh = bit & 31;
    w[wi] = v << sh;
    if (sh > 27 && wi < 3) w[wi + 1] = v >> (32 - sh);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// ------------------------------------------------------------------------------------------------
// LDS access.  The kernels' dynamic LDS segment starts at LDS address 0 (k_main has no static LDS), so
// table addresses are plain integers: this lets the compiler put the table base into the 16-bit
// `offset:` field of ds_read_* instead of spending a v_add per lookup.  Layout of the 77.25 KiB segment:
//   [0, 13568)        the 26 five-bit GHASH tables of the launch constant K in 8-byte halves, one 256 B LDS bank row
//                     per table half (ghash_mul_const_lds)
//   [13568, +64 KiB)  AES: entry for byte value x at 13568 + x*256 + sel*128 + (lane&31)*4
//                     (sel 0 = T0, sel 1 = T2 = rotl16(T0)), 32 replicas so lane l always reads bank l&31
// On the host (tests/host_emul) `lds` is an ordinary array with the same layout.
// ------------------------------------------------------------------------------------------------
#define AESGCM_LDS_GH_OFF 0u
#define AESGCM_LDS_AES_OFF ((u32)AESGCM_LDS_GH)       /* 13568: a multiple of 128, so lane l still reads bank l&31 */
typedef u32 u32x4_t __attribute__((ext_vector_type(4)));
typedef u32 u32x2_t __attribute__((ext_vector_type(2)));
#if defined(__HIP_DEVICE_COMPILE__)
#define LDS_LD32(lds, off) (*(const __attribute__((address_space(3))) u32 *)(uintptr_t)(off))
#define LDS_LD64(lds, off) (*(const __attribute__((address_space(3))) u32x2_t *)(uintptr_t)(off))
#define LDS_LD128(lds, off) (*(const __attribute__((address_space(3))) u32x4_t *)(uintptr_t)(off))
HD u32 xor3(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
#else
#define LDS_LD32(lds, off) (*(const u32 *)((lds) + (off)))
#define LDS_LD64(lds, off) (*(const u32x2_t *)((lds) + (off)))
#define LDS_LD128(lds, off) (*(const u32x4_t *)((lds) + (off)))
HD u32 xor3(u32 a, u32 b, u32 c) { return a ^ b ^ c; }
#endif

// Z * x^8: shift right by one byte; the byte b that falls out (bit k of b = GCM bit 127 - k) comes back as
// b * (1 + x + x^2 + x^7) at the top of word 0
HD void gf_shift8(u32 &z0, u32 &z1, u32 &z2, u32 &z3) {
    const u32 b = z3 & 0xFFu;
    z3 = (z3 >> 8) | (z2 << 24); z2 = (z2 >> 8) | (z1 << 24); z1 = (z1 >> 8) | (z0 << 24);
    z0 = xor3(z0 >> 8, b << 24, b << 23) ^ (b << 22) ^ (b << 17);
}

// SplitMix64 at word position w: a definition, restated independently by the CPU checker.
HD u64 splitmix64_at(u64 seed, u64 w) {
    u64 z = seed + (w + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

