// aesgcm_stream.h -- device-resident structures, the lane code and the host-side plans of the stream kernels: k_setup, k_main, k_fold, k_body (dealt chunks, cyclic rows and their closings), k_combine (part of aesgcm_dev.h).
#pragma once
#include "aesgcm_base.h"
#include "aesgcm_aes.h"
#include "aesgcm_ghash.h"

// ================================================================================================
// Device-resident structures and the per-lane bodies of the kernels.  The __global__ wrappers in
// aesgcm_kernels.hip only add LDS staging, barriers and cross-lane reductions around these, so the
// CPU harness (tests/host_emul) executes the same lane code against the same structures.
// ================================================================================================
struct DevTables {           // per device
    uint8_t sbox[256];
    u32 te0[256];
    u32 te1[256], te2[256], te3[256];   // te0 rotated left by 8, 16, 24: k_body's scalar-cache reads need no rotate
};

#define AESGCM_NPTAB 26
#define AESGCM_NLTAB 130
#define AESGCM_NQ5POW 7
struct KeyMaterial {         // per context (device memory)
    u32 rk[60];              // expanded key, memory-order words
    u32 nr;
    u32 _pad[3];
    uint4 h;                 // H = E_K(0^128)
    uint4 pw[4][AESGCM_NPW]; // pw[d][k] = H^(k * WG^d)
    uint4 q5pow[AESGCM_NQ5POW][AESGCM_Q5_ENTRIES];   // five-bit tables of H^(2^j), j = 0 .. 6: entry p*32+v = quint_elem_mo(p, v) * H^(2^j).  [6] = K = H^64, the lane
                                     // stride of a wave (k_main); [0] = H itself (k_pktl: one lane per packet, serial Horner); k_pktg: Horner stride H^(lanes per packet)
                                     // and the constants of the cross-lane tree H, H^2, H^4, ...
    uint4 k4tab[AESGCM_Q5_ENTRIES];  // ... of H^256 (k_body: a wave takes every fourth row)
    uint4 k18tab[AESGCM_Q5_ENTRIES]; // ... of H^(2^18) (k_body, cyclic rows: a wave takes every 4096th row)
    uint4 k17tab[AESGCM_Q5_ENTRIES]; // ... of H^(2^17) (k_body, cyclic rows in the half shape: 2048 waves, round 4)
    uint4 pwh[256];                  // H^(512 j), j = 0 .. 255: the weight of workgroup 255 - j's item in the closing of the half shape (cyc_close_half)
    uint4 ptab[AESGCM_NPTAB][512]; // nibble tables of H^(2^k), k = 6 .. 31: the Horner constants of k_fold when chunk sizes are powers of two
    uint4 ltab[AESGCM_NLTAB][32];  // two-table Shoup form of H^e, e = 0 .. 129 (65 - lane, plus up to 64 blocks of a separate last row behind the items: CombineParams::tail_blocks): [e][v] = v*H^e, [e][16 + v] = v*H^e*x^4 (per-lane constant multiplies of the closing steps)
    uint8_t rk_bytes[240];   // expanded key as the byte string tb/key_exp.py produces
};

enum { MODE_ENC = 0, MODE_DEC = 1, MODE_KS = 2, MODE_ECB = 3, MODE_PROBE = 4 };   // PROBE (k_body only): ENC without the global load and store

// per key, once (k_setup_ptab): the two Shoup tables of H^e for e = 0 .. 65 in device memory.  The closing steps of a tag need
// lane L's value times H^(65-L) (or H^(63-L) for a shard partial) -- 64 different constants at once; with these tables a
// lane's multiply is 32 independent 16-byte loads (indices = the nibbles of ITS value, all known up front) and a 16-step
// shift-and-xor chain, instead of building a table per launch or running 128 bit-serial steps.
// entry tid (0 .. 31) of the two-table Shoup form of the constant c: [v] = v*c, [16 + v] = v*c*x^4
HD uint4 shoup2_entry(G128 c, u32 tid) {
    G128 t = shoup_entry(c, tid & 15u);
    if (tid >= 16) t = gf_mulx4(t);
    return make_uint4(t.w[0], t.w[1], t.w[2], t.w[3]);
}
HD void setup_ltab_lane(KeyMaterial *km, u32 e, u32 tid) {
    if (tid >= 32 || e >= AESGCM_NLTAB) return;
    km->ltab[e][tid] = shoup2_entry(mo_to_be(km->pw[0][e]), tid);
}
// y * H^e through km->ltab[e] (device / host memory, not LDS)
HD G128 shoup2_gmul(G128 y, const uint4 *__restrict__ tab) {
    uint4 a[16], c[16];
#pragma unroll
    for (int bi = 0; bi < 16; bi++) {
        const u32 w = y.w[bi >> 2];
        const int sh = 8 * (3 - (bi & 3));
        const u32 byte = (w >> sh) & 0xFFu;
        a[bi] = tab[byte >> 4];
        c[bi] = tab[16u + (byte & 15u)];
    }
    u32 z0 = 0, z1 = 0, z2 = 0, z3 = 0;
#pragma unroll
    for (int bi = 15; bi >= 0; bi--) {
        if (bi != 15) gf_shift8(z0, z1, z2, z3);
        z0 = xor3(z0, a[bi].x, c[bi].x); z1 = xor3(z1, a[bi].y, c[bi].y); z2 = xor3(z2, a[bi].z, c[bi].z); z3 = xor3(z3, a[bi].w, c[bi].w);
    }
    G128 z; z.w[0] = z0; z.w[1] = z1; z.w[2] = z2; z.w[3] = z3;
    return z;
}

// the same multiply as a LOOP over the four words of y, last word first (not unrolled: eight table loads in flight instead of 32 -- for tables in LDS,
// where a load costs little to wait for, inside kernels that have no registers to spare: k_body's fused closing)
HD G128 shoup2_gmul_lds(G128 y, const uint4 *__restrict__ tab) {
    u32 z0 = 0, z1 = 0, z2 = 0, z3 = 0;
    u32 w0 = y.w[3], w1 = y.w[2], w2 = y.w[1], w3 = y.w[0];
#pragma unroll 1
    for (u32 q = 0; q < 4; q++) {
        uint4 a[4], c[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 byte = (w0 >> (8 * (3 - k))) & 0xFFu;
            a[k] = tab[byte >> 4];
            c[k] = tab[16u + (byte & 15u)];
        }
#pragma unroll
        for (int k = 3; k >= 0; k--) {
            if (k != 3) gf_shift8(z0, z1, z2, z3);
            else if (q) gf_shift8(z0, z1, z2, z3);
            z0 = xor3(z0, a[k].x, c[k].x); z1 = xor3(z1, a[k].y, c[k].y); z2 = xor3(z2, a[k].z, c[k].z); z3 = xor3(z3, a[k].w, c[k].w);
        }
        w0 = w1; w1 = w2; w2 = w3;
    }
    G128 z; z.w[0] = z0; z.w[1] = z1; z.w[2] = z2; z.w[3] = z3;
    return z;
}

// One atomic address serves ~87 M fetches/s on MI355X (measured): a single dispenser caps a launch at one chunk per
// 11.5 ns, i.e. chunks shorter than ~10 rows run at the dispenser's speed, not the kernel's.  Chunks are therefore
// dealt from AESGCM_NQ queues on separate cache lines; a wave starts at its home queue and walks on when one runs dry;
// dry queues are remembered per workgroup in LDS, so only the first wave of a workgroup to find one pays a failing fetch.
// Round 1 never reset the queues: every wave made one FAILING fetch on every queue so that the next launch knew the base
// values -- waves x queues serialized atomics (6144 x 16 at 11.5 ns per address = 70 - 100 us) at the end of every dynamic
// launch, which is why chunk counts above the wave count cost mid-size messages +100 us (profiles/archive/r02f/tw_sweep_before.txt).
// Now there are two sets of queues: a launch uses one and zeroes the other for the next launch on the stream.
#define AESGCM_NQ 16
struct MainParams {
    const unsigned char *in;     // data in (16-byte aligned) or NULL (MODE_KS)
    unsigned char *out;
    const unsigned char *aad;    // AAD bytes or NULL
    uint4 *parts;                // one GHASH partial per chunk (GHASH modes)
    u32 *counter;                // chunk dispensers: queue q is the u32 at counter[16 q] (one cache line each)
    u32 nq, seg;                 // queue q hands out chunks [q seg, (q+1) seg); the value fetched is the index in the queue (queues start at 0)
    u32 *counter_zero;           // the other set of queues: zeroed by this launch for the next one
    u64 aad_len;                 // bytes
    u64 n_aad;                   // AAD blocks
    u64 len;                     // data bytes
    u64 n_seq;                   // n_aad + data blocks
    u64 rows;                    // R = ceil(n_seq / 64): one row = one 64-lane wave iteration
    u32 pad;                     // 64*R - n_seq front-padding slots (< 64, all in row 0)
    u32 Tw;                      // rows per chunk
    u32 C;                       // chunks = ceil(R / Tw); chunk 0 is the short one (R0 rows)
    u32 R0;                      // rows in chunk 0 = R - (C-1)*Tw
    u32 row_lo, row_hi;          // rows [row_lo, row_hi) hold 64 full data blocks each (fast path)
    u32 ctr0;                    // counter of data block 0 (2 + first_block)
    u32 iv0, iv1, iv2;           // IV as memory-order words
    u32 aad_aligned;             // AAD pointer 16-byte aligned
    u64 *trace;                  // optional per-workgroup {start, end, HW_ID | XCC_ID << 32, chunks done} (measurement support)
    uint4 *ej0;                  // GHASH modes: where the wave that owns chunk 0 leaves E_K(IV || 1) for k_combine (or NULL)
    u32 tail;                    // 1: single-chunk whole message -- the wave that owns chunk 0 also finishes the tag (no k_combine launch)
    uint4 *tag_out, *tag_host;   // tail: where the tag goes (device slot, and the pinned host slot or NULL)
    u64 gen;                     // tail: generation number published behind the host copy (see CombineParams::gen)
};

#define AESGCM_MAX_CHUNKS (1u << 18)   /* 64 lane accumulators (1 KiB) per chunk: at most 256 MiB of them; k_fold's first level measured ~0.5 ns per chunk (135 us for 2^18; about 2x its LDS-array floor of 1.25 table multiplies per item) */

// Chunking of a GHASH sequence of n_seq blocks: rows of 64 blocks, Tw rows per chunk.  A chunk costs its rows, a
// dispenser fetch and a 1 KiB item store; ~8k waves are resident and a lone wave needs ~10 us per row when the CU is
// full.  Small inputs want MANY short chunks (parallelism); large ones enough chunks per resident wave for the
// dynamic dealing to level the age-ordered issue arbitration, but not so many that the dispensers (16 queues x
// ~87 M fetches/s) or k_fold show up.  Measured (profiles/tw_sweep.py; round 2 after the dispenser and k_fold changes:
// profiles/archive/r02f/tw_sweep_after.txt): 16 .. 64 MiB best at 8 rows, 100 .. 256 MiB at 16, beyond at 32; never more than
// AESGCM_MAX_CHUNKS chunks.
HD void main_geometry(u64 n_seq, u32 tw_override, u64 *rows, u32 *Tw, u32 *C) {
    const u64 R = (n_seq + 63) / 64;
    u64 t;
    if (tw_override) t = tw_override;
    else if (R <= 2) t = R;                                        // <= 2 KiB: ONE chunk; its wave finishes the tag itself (k_main's tail: a single launch).  A lone wave needs ~2.2 us per row, so longer messages are faster as one row per wave + k_combine
    else if (R <= 256) t = (R + 63) / 64;                          // <= 256 KiB: at most 64 chunks, which k_combine folds itself (no k_fold launch)
    else if (R <= 16384) { t = R / 2048; if (t < 1) t = 1; }      // <= 16 MiB: ~2k chunks, a wave each (static assignment)
    else if (R <= 65536) t = 8;                                    // <= 64 MiB: measured best (profiles/archive/r02f/tw_sweep_after.txt): 4096 static chunks at 32 MiB, 8192 dealt ones at 64 MiB
    else t = R < (1u << 18) ? 16 : 32;
    const u64 tmin = (R + AESGCM_MAX_CHUNKS - 1) / AESGCM_MAX_CHUNKS;
    if (t < tmin) t = tmin;
    if (t < 1) t = 1;
    if (!tw_override) { u64 p2 = 1; while (p2 < t) p2 <<= 1; t = p2; }      // powers of two: k_fold's constants are then precomputed tables
    *rows = R; *Tw = (u32)t; *C = (u32)((R + t - 1) / t);
}

HD u32 load_le32(const uint8_t *b) { return (u32)b[0] | ((u32)b[1] << 8) | ((u32)b[2] << 16) | ((u32)b[3] << 24); }

// ---- k_setup pieces ----------------------------------------------------------------------------
// lane 0: key schedule (aes_kexp) or pre-expanded copy, H = E_K(0) (gcm_gctr.vhd:141-144); seeds tab[0..1]
HD void setup_lane0(KeyMaterial *km, const uint8_t *sbox, const uint8_t *key, int key_len, int preexpanded_nr, u32 G, uint4 *tab) {
    int nr;
    if (preexpanded_nr) {
        nr = preexpanded_nr;
        for (int i = 0; i < 16 * (nr + 1); i++) km->rk_bytes[i] = key[i];
    } else {
        uint8_t k[32];
        for (int i = 0; i < key_len; i++) k[i] = key[i];
        nr = key_expand_bytes(k, key_len, sbox, km->rk_bytes);
    }
    for (int w = 0; w < 4 * (nr + 1); w++) km->rk[w] = load_le32(km->rk_bytes + 4 * w);
    for (int w = 4 * (nr + 1); w < 60; w++) km->rk[w] = 0;
    km->nr = (u32)nr;
    (void)G;
    uint8_t zero[16] = {0}, hb[16];
    aes_block_bytes(km->rk_bytes, nr, sbox, zero, hb);
    km->h = make_uint4(load_le32(hb), load_le32(hb + 4), load_le32(hb + 8), load_le32(hb + 12));
    tab[0] = gf_one_mo();
    tab[1] = km->h;
}
// doubling level j of a power table: tab[2^j + k] = tab[k] * tab[2^j] for k = 1..2^j
HD bool setup_level(const uint4 *tab, int j, int tid, uint4 *prod) {
    const int base = 1 << j;
    if (tid < 1 || tid > base) return false;
    *prod = gf_mul_mo(tab[tid], tab[base]);
    return true;
}
// after the beta table (d == 1) is complete: the five-bit tables of the fixed Horner / tree constants H^(2^j) (j = 0 .. 6), H^256 and H^(2^18)
HD void setup_beta_lane(KeyMaterial *km, const uint4 *tab, int tid) {
    (void)tab;
    for (int q = tid; q < (AESGCM_NQ5POW + 3) * AESGCM_Q5_ENTRIES; q += AESGCM_WG) {     // 8320 entries over the workgroup: at most nine each
        const int which = q / AESGCM_Q5_ENTRIES, e = q % AESGCM_Q5_ENTRIES;
        const uint4 c = which < AESGCM_NQ5POW ? km->pw[0][1u << which] : which == AESGCM_NQ5POW ? km->pw[0][256] :
                        which == AESGCM_NQ5POW + 1 ? km->pw[1][256] : km->pw[1][128];                            // H^(2^18) = H^(256 * 1024), H^(2^17) = H^(128 * 1024)
        (which < AESGCM_NQ5POW ? km->q5pow[which] : which == AESGCM_NQ5POW ? km->k4tab : which == AESGCM_NQ5POW + 1 ? km->k18tab : km->k17tab)[e] =
            gf_mul_mo(quint_elem_mo(e >> 5, (u32)(e & 31)), c);
    }
    for (int j = tid; j < 256; j += AESGCM_WG)                        // H^(512 j) = H^(1024 (j >> 1)) * H^(512 (j & 1)); pw[0] and pw[1] are complete here (d == 1)
        km->pwh[j] = (j & 1) ? gf_mul_mo(km->pw[1][j >> 1], km->pw[0][512]) : km->pw[1][j >> 1];
}

// after all four power tables exist: ptab[k] = nibble tables of H^(2^(k+6)); H^(2^j) = pw[j / LOG_WG][2^(j % LOG_WG)]
HD void setup_ptab_lane(KeyMaterial *km, u32 k, u32 tid) {
    const u32 j = k + 6;
    if (tid < 512) km->ptab[k][tid] = gf_mul_mo(nibble_elem_mo((int)(tid >> 4), tid & 15u), km->pw[j / AESGCM_LOG_WG][1u << (j % AESGCM_LOG_WG)]);
}

// ---- k_main pieces -----------------------------------------------------------------------------
// LDS image of one workgroup: what thread `tid` of AESGCM_MAIN_WG writes
enum { GH_TAB_K64 = 0, GH_TAB_H = 1, GH_TAB_K256 = 2, GH_TAB_K2P18 = 3, GH_TAB_K2P17 = 4 };       // which constant's five-bit tables go to LDS
HD void main_fill_lds(unsigned char *smem, const KeyMaterial *km, const DevTables *tb, u32 tid, bool gh, u32 nthreads = AESGCM_MAIN_WG, int which = GH_TAB_K64) {
    if (gh) {
        fill_lds_q5(smem, which == GH_TAB_H ? km->q5pow[0] : which == GH_TAB_K256 ? km->k4tab : which == GH_TAB_K2P18 ? km->k18tab : which == GH_TAB_K2P17 ? km->k17tab : km->q5pow[6], tid, nthreads);
    }
    uint4 *dst = reinterpret_cast<uint4 *>(smem + AESGCM_LDS_AES_OFF);
    for (u32 q = tid; q < AESGCM_LDS_AES / 16; q += nthreads) {
        const u32 t0 = tb->te0[q >> 4];
        const u32 v = ((q >> 3) & 1) ? rotl32(t0, 16) : t0;
        dst[q] = make_uint4(v, v, v, v);
    }
}

// region B of the LDS image: what thread `tid` writes (row x: 32 replicas of rotl8(T0[x]) | 32 replicas of rotl8(T2[x]))
HD void fill_lds_t4(unsigned char *smem, const DevTables *tb, u32 tid, u32 nthreads) {
    uint4 *dst = reinterpret_cast<uint4 *>(smem + AESGCM_LDS_AES_OFF + AESGCM_LDS_AES);
    for (u32 q = tid; q < AESGCM_LDS_AES / 16; q += nthreads) {
        const u32 t0 = tb->te0[q >> 4];
        const u32 v = ((q >> 3) & 1) ? rotl32(t0, 24) : rotl32(t0, 8);
        dst[q] = make_uint4(v, v, v, v);
    }
}
// block loads/stores with the ragged last block handled bytewise (gcm_ghash.vhd:225-246 byte-valid
// mask = zero padding on the right; gcm_gctr.vhd:184 byte-valid passthrough on the data output)
HD uint4 load_block_bytes(const unsigned char *p, u32 nbytes) {
    if (nbytes == 16) return gload16_any(p);                       // a whole block: one access at any byte address (unaligned access mode, gload16_any)
    // fully unrolled with constant word indices: a loop over w[k >> 2] with a run-time trip count made the compiler keep the
    // four words in scratch memory (round-2 ISA: scratch_* inside the packet kernels' row loops)
    u32 w0 = 0, w1 = 0, w2 = 0, w3 = 0;
#pragma unroll
    for (u32 k = 0; k < 16; k++) {
        if (k < nbytes) {
            const u32 b = (u32)p[k] << (8 * (k & 3));
            if (k < 4) w0 |= b; else if (k < 8) w1 |= b; else if (k < 12) w2 |= b; else w3 |= b;
        }
    }
    return make_uint4(w0, w1, w2, w3);
}
HD void store_block_bytes(unsigned char *p, uint4 v, u32 nbytes, bool wt = false) {      // wt: through the L2 (gstore16_wt)
    if (nbytes == 16) {                                            // a whole block: one access at any byte address
        if (wt) gstore16_wt_at(p, v); else gstore16_any(p, v);
        return;
    }
#pragma unroll
    for (u32 k = 0; k < 16; k++) {
        if (k < nbytes) {
            const u32 w = k < 4 ? v.x : k < 8 ? v.y : k < 12 ? v.z : v.w;
            if (wt) gstore1_wt_at(p + k, (w >> (8 * (k & 3))) & 0xFFu);
            else p[k] = (unsigned char)(w >> (8 * (k & 3)));
        }
    }
}
HD u32 mask_word(u32 w, int keep) { return keep <= 0 ? 0u : keep < 4 ? (w & ((1u << (8 * keep)) - 1u)) : w; }   // keep = valid bytes of the word
HD uint4 mask_block(uint4 v, u32 nbytes) {
    return make_uint4(mask_word(v.x, (int)nbytes), mask_word(v.y, (int)nbytes - 4), mask_word(v.z, (int)nbytes - 8), mask_word(v.w, (int)nbytes - 12));
}

// The hot loop: lane `lane` (0..63) of the wave that owns chunk `c`.  Rows of the chunk are consecutive
// 64-block groups; the lane runs Horner with K = H^64 (its own blocks are 64 apart).  Returns the lane's
// GHASH accumulator for the chunk: sum_r X[row_r, lane] * K^(rows-1-r).
// Almost every row is a "pure data row" (64 full data blocks): those take the fast path, whose addresses are a
// wave-uniform 64-bit base plus a 32-bit lane offset (scalar base + vector offset addressing, no 64-bit vector
// arithmetic, no per-lane branches).  Rows that contain front padding, AAD blocks or the ragged last block take
// the general path.
template <int NR, int MODE>
HD uint4 main_block(const u32 *__restrict__ rk, const unsigned char *smem, const CtrConsts &cc, u32 lb, uint4 x, u32 ctr) {
    u32 s0, s1, s2, s3;
    if (MODE == MODE_ECB) {
        s0 = x.x ^ rk[0]; s1 = x.y ^ rk[1]; s2 = x.z ^ rk[2]; s3 = x.w ^ rk[3];
        aes_rounds_lds<NR>(s0, s1, s2, s3, rk, smem, lb);
        return make_uint4(s0, s1, s2, s3);
    }
    // counter block IV || cnt, cnt big-endian, low 32 bits only (aes_icb.vhd:97-100,118)
    ctr_rounds_lds<NR>(bswap32(ctr), cc, s0, s1, s2, s3, rk, smem, lb);
    return make_uint4(x.x ^ s0, x.y ^ s1, x.z ^ s2, x.w ^ s3);                  // gcm_gctr.vhd:150
}

// make wave-uniform values visibly scalar to the compiler (host: identity)
HD u32 uniform32(u32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (u32)__builtin_amdgcn_readfirstlane(x);
#else
    return x;
#endif
}
HD u64 uniform64(u64 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return ((u64)(u32)__builtin_amdgcn_readfirstlane((u32)(x >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((u32)x);
#else
    return x;
#endif
}

template <int NR, int MODE, bool WT = false>      // WT: stores through the L2 (the generic rows of a cyclic k_body launch)
HD uint4 main_chunk_lane(const KeyMaterial *__restrict__ km, const MainParams &p, const unsigned char *smem, const CtrConsts &cc, u32 c, u32 lane) {
    constexpr bool GH = (MODE == MODE_ENC || MODE == MODE_DEC);
    const u32 *__restrict__ rk = km->rk;
    const u32 lb = (lane & 31u) << 2;
    const u64 n_data_blocks = p.n_seq - p.n_aad;
    const u32 tail_bytes = (u32)(p.len & 15);          // 0 = last data block is full
    const u32 aad_tail = (u32)(p.aad_len & 15);
    const u32 row0 = c ? p.R0 + (c - 1) * p.Tw : 0;                       // rows fit 32 bits (R <= 2^30)
    const u32 nrows = c ? p.Tw : p.R0;
    const u64 first_data_slot = (u64)p.pad + p.n_aad;                     // virtual slot of data block 0
    const u32 lane16 = lane * 16u;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (u32 r = 0; r < nrows; ++r) {
        const u32 row = row0 + r;                                          // wave-uniform
        if (GH && r > 0) acc = ghash_mul_const_lds(acc, smem);
        if (row >= p.row_lo && row < p.row_hi) {
            // ---- fast path: 64 full data blocks; uniform 64-bit base + 32-bit lane offset
            const u64 i0 = (u64)row * 64 - first_data_slot;                // data block index of lane 0
            const unsigned char *src = reinterpret_cast<const unsigned char *>(uniform64((u64)(uintptr_t)p.in + 16 * i0));
            unsigned char *dst = reinterpret_cast<unsigned char *>(uniform64((u64)(uintptr_t)p.out + 16 * i0));
            uint4 x = make_uint4(0, 0, 0, 0);
            if (MODE != MODE_KS) x = gload16(src + lane16);
            const uint4 y = main_block<NR, MODE>(rk, smem, cc, lb, x, p.ctr0 + (u32)i0 + lane);
            if (WT) gstore16_wt(dst, lane16, y); else gstore16(dst + lane16, y);
            if (GH) acc = xor4(acc, (MODE == MODE_DEC) ? x : y);          // aes_gcm.vhd:207-211
            continue;
        }
        // ---- general path
        const u64 v = (u64)row * 64 + lane;
        if (v < p.pad) continue;                       // front padding: contributes zero
        const u64 j = v - p.pad;                       // index in the GHASH sequence (AAD blocks then data blocks)
        uint4 gin;
        if (GH && j < p.n_aad) {
            const unsigned char *ap = p.aad + 16 * j;
            if (j == p.n_aad - 1 && aad_tail) gin = load_block_bytes(ap, aad_tail);
            else if (p.aad_aligned) gin = *reinterpret_cast<const uint4 *>(ap);
            else gin = load_block_bytes(ap, 16);
        } else {
            const u64 i = j - p.n_aad;                 // data block index within this launch
            const bool ragged = tail_bytes && (i == n_data_blocks - 1);
            uint4 x = make_uint4(0, 0, 0, 0);
            if (MODE != MODE_KS) {
                if (ragged) x = load_block_bytes(p.in + 16 * i, tail_bytes);
                else x = *reinterpret_cast<const uint4 *>(p.in + 16 * i);
            }
            uint4 y = main_block<NR, MODE>(rk, smem, cc, lb, x, p.ctr0 + (u32)i);
            if (ragged) { y = mask_block(y, tail_bytes); store_block_bytes(p.out + 16 * i, y, tail_bytes, WT); }
            else if (WT) gstore16_wt_at(p.out + 16 * i, y);
            else *reinterpret_cast<uint4 *>(p.out + 16 * i) = y;
            gin = (MODE == MODE_DEC) ? x : y;           // aes_gcm.vhd:207-211
        }
        if (GH) acc = xor4(acc, gin);
    }
    return acc;
}
// per-message constants a lane computes once (round-1 hoisting)
template <int MODE>
HD CtrConsts main_lane_consts(const KeyMaterial *__restrict__ km, const MainParams &p, const unsigned char *smem, u32 lane) {
    CtrConsts cc = {0, 0, 0, 0};
    if (MODE != MODE_ECB) cc = ctr_round1_consts(p.iv0, p.iv1, p.iv2, km->rk, smem, (lane & 31u) << 2);
    return cc;
}
// ---- k_fold pieces -----------------------------------------------------------------------------
// A chunk leaves its 64 raw lane accumulators behind (one ITEM = 64 x 16 B, lane L = sum_r X[r, L] * K^(rows-1-r));
// nothing is multiplied per chunk any more (a per-lane bit-serial multiply per chunk used to cost ~2.4 rows).
// The polynomial of the range is  sum_L H^(63-L) * B_L,  B_L = sum_i item_i[L] * H^(blocks between item i and the end),
// and B_L is a Horner recurrence per lane with WAVE-UNIFORM constants, i.e. the cheap LDS-table multiply.
// One k_fold launch reduces n items to ceil(n / (8 g)): stage a, a wave folds g consecutive items (constant A);
// stage b, wave 0 folds the workgroup's <= 8 results (constant C = A^g).  Groups are cut from the END, so the
// first group / first wave is the short one and group ends stay equally spaced.  g = 16 for long inputs; for up
// to 16384 items g is the smallest power of two that leaves k_combine at most 64 items (fold_group): a workgroup's
// multiplies all go through one CU's LDS array (128 array cycles each), so a full workgroup costs
// ~10-16 us however few workgroups there are -- with g = 1 .. 8 a mid-size message spreads over up to 64 CUs instead of 4
// (round 2: k_fold 32 us -> see profiles/README.md).  k_combine applies H^(63-L) to the last item and XOR-folds the lanes.
// k_body's chunks are interleaved (item 4s+v, v = row phase, 64 blocks apart; super-chunks 256 T apart):
// period = 4 folds the four phases with A = H^64 and the super-chunks with B = H^(256 T).
// Constants that are H^(2^k) (chunk sizes are powers of two unless the context option "tw" says otherwise) come from the
// key's precomputed tables (tab* = device pointer); others are built in the kernel from the exponent.
// k_fold can close the message itself (whole messages whose dealt k_body launch is the whole range: BASELINE configs 2 and 3): the workgroups of the
// FIRST level then do with their output items what cyc_close does with a workgroup's item -- lane terms H^(65 - L), the weight H^(step (G - 1 - g)) as the
// product of its radix-1024 digit powers (pw[d][digit], one two-table Shoup multiply per non-zero digit), 16 bytes into the accumulator slots, the
// last arrival publishes -- and the second level, k_combine and their two launch gaps disappear (1 GiB: 47 -> 28 us behind k_body).
struct FoldClose {
    u32 on;
    u64 step;                    // blocks between the ends of consecutive output items (fold_out_step)
    u64 aad_len, ct_len;         // bytes, for the length block
    const uint4 *ej0;            // E_K(IV || 1), left by the k_body launch in front
    unsigned long long *acc;     // accumulator slots + arrival counter (as BodyParams::acc)
    uint4 *tag_out, *tag_host; u64 gen;
};
#define FOLD_LDS_LTAB 32768u                 /* closing: the lanes' Shoup tables of H^(65 - L), 64 x 528 bytes */
#define FOLD_LDS_WTAB (FOLD_LDS_LTAB + 64u * 528u)   /* closing: the weight digit's two-table Shoup form */
#define FOLD_LDS_CLOSE_BYTES (FOLD_LDS_WTAB + 512u)
struct FoldParams {
    FoldClose close;
    const uint4 *in; uint4 *out;
    u32 n;                       // items in
    u32 period;                  // 1: plain Horner with A.  4: inner Horner with A over each 4 items, outer with B
    u32 group;                   // items per wave (1, 2, 4, 8 or FOLD_GROUP; a multiple of the period)
    u64 eA, eB, eC;              // exponents (blocks) of the three constants; eC = blocks between stage-a outputs
    const uint4 *tabA, *tabB, *tabC;   // precomputed nibble tables or NULL
};
#define COMBINE_MAX_ITEMS 64u
#define FOLD_GROUP 16u           /* most items per wave */
#ifndef FOLD_WAVES
/* waves per workgroup: at most 128 items per workgroup.  With 512 lanes the compiler takes 176 VGPRs (batches of 8 items in flight):
   two waves per SIMD, i.e. ONE such workgroup per CU (profiles/archive/r03/isa_census.txt; forcing 128 registers, -DFOLD_WPS=4, spills 188 bytes into
   the item loop) -- round 2's "two workgroups share a CU" was wrong, its measurement stands: half as many items behind one CU's LDS array.
   Measured per k_fold launch over a 16 GiB message's 2^18 items (profiles/archive/r02f/fold_waves.txt): 16 waves 115 us, 8 waves 73 us,
   4 waves 66 us; a rolled loop at 64 VGPRs with two 16-wave workgroups per CU: 172 us. */
#define FOLD_WAVES 8u
#endif
#define FOLD_WG (64u * FOLD_WAVES)
#define FOLD_LDS_TAB 24576u      /* three 8 KiB tables */
#define FOLD_LDS_BYTES (FOLD_LDS_TAB + FOLD_WAVES * 1024u)
HD G128 gf_pow_h_serial(const KeyMaterial *km, u64 e);
// nibble tables of H^e at LDS byte offset `base` (what thread tid of nthreads writes): copied or built
HD void fold_fill_lds(unsigned char *smem, const KeyMaterial *km, const uint4 *tab, u64 e, u32 base, u32 tid, u32 nthreads) {
    if (tab) { for (u32 q = tid; q < 512; q += nthreads) reinterpret_cast<uint4 *>(smem + base)[q] = tab[q]; return; }
    const uint4 c = be_to_mo(gf_pow_h_serial(km, e));
    for (u32 q = tid; q < 512; q += nthreads) reinterpret_cast<uint4 *>(smem + base)[q] = gf_mul_mo(nibble_elem_mo((int)(q >> 4), q & 15u), c);
}
// items per wave of a level over n items
HD u32 fold_group(u32 n, u32 period) {
    if (period > 1) return FOLD_GROUP;
    u32 g = 1;
    while (g < FOLD_GROUP && n > COMBINE_MAX_ITEMS * FOLD_WAVES * g) g <<= 1;
    return g;
}
HD u32 fold_wgs(u32 n, u32 group) { return (n + group * FOLD_WAVES - 1) / (group * FOLD_WAVES); }
// capacity of the two ping-pong buffers in items (levels alternate between them, the first writes A): a first level leaves at
// most MAX_CHUNKS/(FOLD_GROUP FOLD_WAVES); from there on fold_group keeps a level's output at COMBINE_MAX_ITEMS or below, where
// k_combine takes over (tests/host_emul checks every n)
#define FOLD_A_ITEMS (AESGCM_MAX_CHUNKS / (FOLD_GROUP * FOLD_WAVES))
#define FOLD_B_ITEMS (AESGCM_MAX_CHUNKS / 65536u + COMBINE_MAX_ITEMS)
// the items [*start, *end) of workgroup g, and how many waves have work
HD u32 fold_wg_range(u32 n, u32 group, u32 g, u32 *start, u32 *end) {
    const u32 per = group * FOLD_WAVES;
    *end = n - per * (fold_wgs(n, group) - 1 - g);
    *start = *end > per ? *end - per : 0;
    return (*end - *start + group - 1) / group;
}
// stage a: lane `lane` of wave w (of J active waves) of the workgroup that owns items [start, end)
HD uint4 fold_wave_lane(const FoldParams &p, const unsigned char *smem, u32 start, u32 end, u32 J, u32 w, u32 lane) {
    const u32 e = end - p.group * (J - 1 - w);
    const u32 s = e > start + p.group ? e - p.group : start;
    // loads in batches of 8 ahead of their multiplies (they do not depend on the accumulator; the items come from
    // HBM, ~1-2 us away)
    const u32 cnt = e - s;
    uint4 outer = make_uint4(0, 0, 0, 0), inner = make_uint4(0, 0, 0, 0);
    for (u32 k0 = 0; k0 < cnt; k0 += 8) {
        uint4 it[8];
#pragma unroll
        for (u32 k = 0; k < 8; ++k) it[k] = (k0 + k < cnt) ? p.in[(size_t)(s + k0 + k) * 64 + lane] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (u32 k = 0; k < 8; ++k) {
            if (k0 + k < cnt) {
                if (p.period <= 1) {
                    if (k0 + k) outer = ghash_mul_const_lds_at(outer, smem, 0u);
                    outer = xor4(outer, it[k]);
                } else {
                    const u32 ph = (k0 + k) % p.period;                // s is a multiple of the period
                    if (ph) inner = ghash_mul_const_lds_at(inner, smem, 0u);
                    inner = xor4(inner, it[k]);
                    if (ph == p.period - 1) {
                        if (k0 + k + 1 > p.period) outer = ghash_mul_const_lds_at(outer, smem, 8192u);
                        outer = xor4(outer, inner);
                        inner = make_uint4(0, 0, 0, 0);
                    }
                }
            }
        }
    }
    return outer;
}
// stage b: lane `lane` of wave 0 folds the J wave results staged in LDS (item j at FOLD_LDS_TAB + 1024 j)
HD uint4 fold_wg_lane(const unsigned char *smem, u32 J, u32 lane) {
    uint4 acc = *reinterpret_cast<const uint4 *>(smem + FOLD_LDS_TAB + lane * 16u);
    for (u32 j = 1; j < J; ++j) acc = xor4(ghash_mul_const_lds_at(acc, smem, 16384u), *reinterpret_cast<const uint4 *>(smem + FOLD_LDS_TAB + j * 1024u + lane * 16u));
    return acc;
}
// blocks between the ends of consecutive output items of a launch
static inline u64 fold_out_step(const FoldParams &p) { return FOLD_WAVES * p.eC; }
// fill the constants of a level: items eA blocks apart (period 1), or phases eA apart and periods eB apart
static inline void plan_fold(FoldParams &p, const uint4 *in, uint4 *out, u32 n, u32 period, u64 eA, u64 eB) {
    p.close = FoldClose{};                                       // (all of it: a level that does not close carries no stale pointers -- the fake runtime checks every pointer a launch carries)
    p.in = in; p.out = out; p.n = n; p.period = period; p.eA = eA; p.eB = period > 1 ? eB : 0;
    p.group = fold_group(n, period);
    p.eC = period > 1 ? (p.group / period) * eB : p.group * eA;
    p.tabA = p.tabB = p.tabC = nullptr;
}
// index into KeyMaterial::ptab of the table of H^e, or -1
static inline int ptab_index(u64 e) {
    if (!e || (e & (e - 1))) return -1;
    int k = 0; while ((e >> k) != 1) k++;
    return (k >= 6 && k < 6 + AESGCM_NPTAB) ? k - 6 : -1;
}

// ---- k_body pieces -----------------------------------------------------------------------------
// The aligned middle of a large message: data blocks from a block index (within the message) that is a multiple
// of 256, in whole super-chunks of 4*T rows; row m holds blocks [64 m, 64 m + 64) -- aligned in memory as well
// as in the message.  A wave owns one PHASE v of a super-chunk: rows 4q + v, q = s*T .. s*T + T-1.
// The counter of block i is i + 2 (aes_icb.vhd:97-118: IV || cnt, cnt in bytes 12..15 big-endian), so in a row
//   lanes 0..61  have counter 256 q + 64 v + lane + 2:      byte 15 = 64 v + lane + 2,   bytes 12..14 = hi24(q)
//   lanes 62, 63 have counter 256 q + 64 (v+1) + lane - 62:  byte 15 = 64 (v+1) + lane - 62 (mod 256),
//                                                            bytes 12..14 = hi24(q), or hi24(q + 1) when v = 3.
// Byte 15 of a lane is a constant of the chunk, bytes 12..14 are wave-uniform per row (two values in phase 3).
// After round 1 only column 0 of the state depends on byte 15, and every round-2 output column is
// P_j(lane) ^ U_j(row):  P_j = the one table value that comes from column 0 (four VGPRs, computed once per chunk),
// U_j = the three row-uniform table values, the round key and the round-1 constants (scalar: table reads with
// wave-uniform indices go through the scalar cache, not LDS).  Rounds 1 and 2 therefore cost no LDS lookup in the
// row loop: 192 instead of 212 per AES-256 block.
// GHASH: the lane's blocks are 256 apart, Horner constant H^256 (main_fill_lds(GH_TAB_K256)).
#ifndef AESGCM_BODY_RKV_FROM_HALF
#define AESGCM_BODY_RKV_FROM_HALF(NR) (4 * ((NR) + 1))                    /* ... in the half shape (two-table round: more temporaries) */
#endif
#ifndef AESGCM_BODY_RKV_FROM
#define AESGCM_BODY_RKV_FROM(NR) ((NR) == 14 ? 28 : 4 * ((NR) + 1))     /* first round-key word k_body keeps in a vector register (none for AES-128 / 192) */
#endif
HD u32 pin_vgpr(u32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    u32 r;
    asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(x));
    return r;
#else
    return x;
#endif
}
struct BodyParams {
    const unsigned char *in;     // first body block (16-byte aligned)
    unsigned char *out;
    uint4 *parts;                // one GHASH partial per chunk, chunk c = 4*s + v
    u32 *counter; u32 nq, seg; u32 *counter_zero;         // as in MainParams
    u32 T;                       // rows per chunk (iterations of a wave), super-chunk = 4*T rows = 256*T blocks
    u32 C;                       // chunks = 4 * super-chunks
    u32 ctr_hi0;                 // (message block index of body block 0) >> 8; the index is a multiple of 256
    u32 iv0, iv1, iv2;
    u64 *trace;
    uint4 *ej0;                  // where the wave that owns chunk 0 leaves E_K(IV || 1) for k_combine (or NULL)
    // cyclic rows (k_body<.., true>, body_cyc_lane): T, C and the queues are unused
    u32 cyc;
    u32 cw;                      // waves of the cyclic launch = the stride of a strand in rows: BODY_CYC_WAVES (4096), or BODY_CYC_WAVES_HALF (2048) in the half shape
    u32 F, R;                    // rows in front of the body (AAD blocks and the data blocks up to the body, front-padded: `front`) and whole rows of the body
    u32 tb;                      // blocks of the last, partial row behind the body (`last`; 0 = none): its item goes to parts[BODY_CYC_WAVES]
    MainParams front, last;      // the two generic pieces as one-row chunks of main_chunk_lane
    // fused closing (whole messages, body_cyc_* below): the launch folds its own items and leaves the tag -- no k_fold, no k_combine
    u32 prio_rows;               // rotate the waves' issue priorities every so many rows (0 = leave them alone)
    u32 fuse;                    // 1: the launch closes the tag itself (cyc_close).  Its rows reach memory before the tag is shown: through-the-L2 stores (AESGCM_BODY_WT) or, without them, an L2 write-back
    u64 aad_len, ct_len;         // bytes, for the length block
    unsigned long long *acc;     // CYC_ACC_SLOTS x {hi, lo} XOR accumulators and the arrival counter behind them (device memory, zero between launches)
    uint4 *tag_out, *tag_host;   // where the tag goes (device slot, and the pinned host slot or NULL)
    u64 gen;                     // generation number published behind the host copy (see CombineParams::gen)
};
struct BodyLane { u32 p0, p1, p2, p3; };
// wave-uniform table values (host: plain loads; device: scalar loads from the global T0 table)
HD u32 tu0(const DevTables *__restrict__ tb, u32 x) { return tb->te0[x & 0xFFu]; }
HD u32 tu1(const DevTables *__restrict__ tb, u32 x) { return tb->te1[x & 0xFFu]; }
HD u32 tu2(const DevTables *__restrict__ tb, u32 x) { return tb->te2[x & 0xFFu]; }
HD u32 tu3(const DevTables *__restrict__ tb, u32 x) { return tb->te3[x & 0xFFu]; }
// per-chunk lane constants: column 0 after round 1, and its four round-2 table values
HD BodyLane body_lane_consts(const u32 *__restrict__ rk, const CtrConsts &cc, const unsigned char *lds, u32 v, u32 lane) {
    const u32 lb = (lane & 31u) << 2;
    const u32 b15 = lane < 62 ? 64u * v + lane + 2u : (64u * (v + 1u) + lane - 62u) & 0xFFu;
    const u32 w3 = (b15 << 24) ^ rk[3];                                // only byte 3 (= counter byte 15) is used
    const u32 s0 = cc.c0 ^ rotl32(T2_AT(lds, w3, 3, lb), 8);
    BodyLane b;
    b.p0 = T0_AT(lds, s0, 0, lb);
    b.p1 = rotl32(T2_AT(lds, s0, 3, lb), 8);
    b.p2 = T2_AT(lds, s0, 2, lb);
    b.p3 = rotl32(T0_AT(lds, s0, 1, lb), 8);
    return b;
}
// the row-uniform part of the state after round 2 for counter bytes 12..14 = hi24
struct BodyRow { u32 U0, U1, U2, U3; };
HD BodyRow body_uniform(u32 hi24, const CtrConsts &cc, const u32 *__restrict__ rk, const DevTables *__restrict__ tb) {
    // counter bytes 12, 13, 14 = hi24 big-endian; memory-order word 3 holds them in bytes 0, 1, 2 (byte 3 is the lane's)
    const u32 k3 = rk[3];
    const u32 u1 = cc.c1 ^ tu2(tb, hi24 ^ (k3 >> 16));                 // columns 1..3 after round 1
    const u32 u2 = cc.c2 ^ tu1(tb, (hi24 >> 8) ^ (k3 >> 8));
    const u32 u3 = cc.c3 ^ tu0(tb, (hi24 >> 16) ^ k3);
    const u32 *__restrict__ k2 = rk + 8;
    BodyRow r;
    r.U0 = tu2(tb, u2 >> 16) ^ k2[0] ^ tu1(tb, u1 >> 8) ^ tu3(tb, u3 >> 24);
    r.U1 = tu0(tb, u1) ^ tu2(tb, u3 >> 16) ^ k2[1] ^ tu1(tb, u2 >> 8);
    r.U2 = tu0(tb, u2) ^ k2[2] ^ tu1(tb, u3 >> 8) ^ tu3(tb, u1 >> 24);
    r.U3 = tu0(tb, u3) ^ tu2(tb, u1 >> 16) ^ k2[3] ^ tu3(tb, u2 >> 24);
    return r;
}
// rounds 3..NR from the state after round 2
template <int NR, bool T4 = (AESGCM_T4 != 0)>                   // T4: four T-tables in LDS (136 KiB); else the two-table round (the half shape of the cyclic rows, 77 KiB)
HD void body_rounds(u32 &s0, u32 &s1, u32 &s2, u32 &s3, const u32 *__restrict__ rk, const unsigned char *lds, u32 lb) {
#pragma unroll
    for (int r = 3; r < NR; r++) {
        if (T4) aes_round_lds4(s0, s1, s2, s3, rk + 4 * r, lds, lb, lb | 0x10000u);
        else aes_round_lds(s0, s1, s2, s3, rk + 4 * r, lds, lb);
    }
    aes_final_lds(s0, s1, s2, s3, rk + 4 * NR, lds, lb);
}
#ifndef AESGCM_BODY_WT
#define AESGCM_BODY_WT 1                 /* k_body's rows store their ciphertext through the L2 (gstore16_wt): nothing of it is left dirty for the end of the launch -- the cyclic launch shows
                                            its tag from inside (cyc_close), and a dealt 16 GiB launch ends 0.1 ms sooner (profiles/archive/r03c/body_wt_ab: step 16.90 -> 16.79 ms).  0: plain stores;
                                            cyc_close then writes the XCD's L2 back (an agent-scope release, 5 us) before the workgroup counts itself arrived */
#endif
// the state of super-row q after round 2: per-chunk lane constants xor the row-uniform part (two values in phase 3)
HD void body_state(u32 &s0, u32 &s1, u32 &s2, u32 &s3, const BodyLane &b, u32 hi24, u32 v, u32 lane, const CtrConsts &cc,
                   const u32 *__restrict__ rk, const DevTables *__restrict__ tb) {
    const BodyRow u = body_uniform(hi24, cc, rk, tb);
    s0 = b.p0 ^ u.U0; s1 = b.p1 ^ u.U1; s2 = b.p2 ^ u.U2; s3 = b.p3 ^ u.U3;
    if (v == 3) {                                                  // wave-uniform: lanes 62, 63 are already in the next 256-block
        const BodyRow n = body_uniform(hi24 + 1, cc, rk, tb);
        const u32 m = lane >= 62 ? 0xFFFFFFFFu : 0u;
        s0 ^= m & (u.U0 ^ n.U0); s1 ^= m & (u.U1 ^ n.U1); s2 ^= m & (u.U2 ^ n.U2); s3 ^= m & (u.U3 ^ n.U3);
    }
}
// The CU's issue arbitration is priority first, age second: with equal shares of rows the oldest wave of a SIMD runs ahead and the youngest is left to
// finish alone (a launch of cyclic rows loses 8 % at 512 MiB and 16 % at 4 GiB that way).  Rotating the priorities -- every `rows` rows wave `slot` of
// its SIMD (0 .. 3) takes priority (slot + i / rows) mod 4 -- gives every wave every rank for the same share of the time.
HD void body_prio(u32 i, u32 rows, u32 slot) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (i % rows == 0) {
        switch ((slot + i / rows) & 3u) {                              // s_setprio takes an immediate
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
        }
    }
#else
    (void)i; (void)rows; (void)slot;
#endif
}
// lane `lane` of a wave that takes the n super-rows q0, q0 + qstep, ... in row phase v: returns sum_i X[row 4(q0 + i qstep) + v, lane] * K^(n-1-i),
// K = H^(256 qstep) = the constant whose tables the launch staged in LDS
// (body_strand_rows: the same over an explicit range -- `in` / `out` = the first block of the aligned body, ctr_hi0 = that block's message index >> 8 -- for callers
// whose body is not described by a BodyParams: the chunks of k_rows, aesgcm_rows.h)
template <int NR, int MODE, bool T4 = (AESGCM_T4 != 0)>
HD uint4 body_strand_rows(const KeyMaterial *__restrict__ km, const DevTables *__restrict__ tb, const unsigned char *in, unsigned char *out, u32 ctr_hi0,
                          const unsigned char *smem, const CtrConsts &cc, u32 q0, u32 qstep, u32 n, u32 v, u32 lane,
                          uint4 acc_in = make_uint4(0, 0, 0, 0), bool continued = false,        // continued: acc_in is the strand so far (one more multiply in front of the first row)
                          u32 prio_rows = 0, u32 prio_slot = 0) {                               // prio_rows > 0: rotate the wave's issue priority every prio_rows rows (body_prio)
    const u32 *__restrict__ rk0 = km->rk;
    const u32 lb = (lane & 31u) << 2, lane16 = lane * 16u;
    const BodyLane b = body_lane_consts(rk0, cc, smem, v, lane);
    // AES-256 has 60 round-key words; all of them in scalar registers, with the row-uniform round-2 state, the pointers and the loop state, overflow the
    // 102 SGPRs: the compiler parked 31 scalars in the lanes of a VGPR and fetched ten of them back with v_readlane in EVERY row (round-3 ISA census) --
    // VALU issue slots in a loop that is bound by them.  The keys of the late rounds therefore live in VECTOR registers (a VALU operand either way; the
    // kernel uses 85 of its 128): pin_vgpr hides the copy from the compiler so that it stays one.
    u32 rk[4 * (NR + 1)];
#pragma unroll
    for (int w = 0; w < 4 * (NR + 1); w++) rk[w] = (w >= (T4 ? AESGCM_BODY_RKV_FROM(NR) : AESGCM_BODY_RKV_FROM_HALF(NR))) ? pin_vgpr(rk0[w]) : rk0[w];
    uint4 acc = acc_in;
    u32 i = 0;
    for (; i < n; ++i) {
        const u32 q = q0 + i * qstep;                                  // super-row: counters [256 q, 256 q + 255] of the body
        if (prio_rows) body_prio(i, prio_rows, prio_slot);
        if (i || continued) acc = ghash_mul_const_lds(acc, smem);
        const u64 off = ((u64)q * 4 + v) * 1024;                       // byte offset of the row in the body
        const unsigned char *src = reinterpret_cast<const unsigned char *>(uniform64((u64)(uintptr_t)in + off));
        unsigned char *dst = reinterpret_cast<unsigned char *>(uniform64((u64)(uintptr_t)out + off));
        // MODE_PROBE: the same instruction stream without HBM traffic -- the ceiling of the formulation itself
        // (aesgcm_ctx_ceiling_probe); the "plaintext" is a lane/row pattern and the ciphertext only feeds GHASH
        const uint4 x = (MODE == MODE_PROBE) ? make_uint4(lane, q, v, 0u) : gload16(src + lane16);
        u32 s0, s1, s2, s3;
        body_state(s0, s1, s2, s3, b, ctr_hi0 + q, v, lane, cc, rk, tb);
        body_rounds<NR, T4>(s0, s1, s2, s3, rk, smem, lb);
        const uint4 y = make_uint4(x.x ^ s0, x.y ^ s1, x.z ^ s2, x.w ^ s3);                  // gcm_gctr.vhd:150
        if (MODE != MODE_PROBE) {
#if AESGCM_BODY_WT
            gstore16_wt(dst, lane16, y);
#else
            gstore16(dst + lane16, y);
#endif
        }
        acc = xor4(acc, (MODE == MODE_DEC) ? x : y);                  // aes_gcm.vhd:207-211
    }
    return acc;
}
template <int NR, int MODE, bool T4 = (AESGCM_T4 != 0)>
HD uint4 body_strand_lane(const KeyMaterial *__restrict__ km, const DevTables *__restrict__ tb, const BodyParams &p,
                          const unsigned char *smem, const CtrConsts &cc, u32 q0, u32 qstep, u32 n, u32 v, u32 lane,
                          uint4 acc_in = make_uint4(0, 0, 0, 0), bool continued = false, u32 prio_rows = 0, u32 prio_slot = 0) {
    return body_strand_rows<NR, MODE, T4>(km, tb, p.in, p.out, p.ctr_hi0, smem, cc, q0, qstep, n, v, lane, acc_in, continued, prio_rows, prio_slot);
}
// The same row code over CONSECUTIVE rows r0, r0 + 1, ... of an aligned body (k_rows: a run of rows of one of many messages, aesgcm_rows.h): the row phase
// v = r & 3 changes with every row, so the wave holds the lane constants of all four phases (sixteen registers instead of four; one select per row and word)
// and ONE Horner accumulator with the stride H^64 of consecutive rows (main_fill_lds(GH_TAB_K64)) -- a run of any length is one piece of work with one value
// to leave behind, where four strands would be four.  Returns sum_i X[row r0 + i, lane] * (H^64)^(n-1-i).
template <int NR, int MODE, bool T4 = (AESGCM_T4 != 0)>
HD uint4 body_rows_lane(const KeyMaterial *__restrict__ km, const DevTables *__restrict__ tb, const unsigned char *in, unsigned char *out, u32 ctr_hi0,
                        const unsigned char *smem, const CtrConsts &cc, u32 r0, u32 n, u32 lane, u32 prio_rows = 0, u32 prio_slot = 0) {
    const u32 *__restrict__ rk0 = km->rk;
    const u32 lb = (lane & 31u) << 2, lane16 = lane * 16u;
#ifndef AESGCM_ROWS_HOLD_PHASES
#define AESGCM_ROWS_HOLD_PHASES(NR) ((NR) != 14)                        /* AES-256 keeps 32 round-key words in vector registers: with sixteen more for the phases its row loop spilled (12 scratch accesses per row); it computes the row's lane constants per row instead (five lookups) */
#endif
    constexpr bool HOLD = AESGCM_ROWS_HOLD_PHASES(NR);
    BodyLane b0 = {0, 0, 0, 0}, b1 = b0, b2 = b0, b3 = b0;
    if (HOLD) { b0 = body_lane_consts(rk0, cc, smem, 0u, lane); b1 = body_lane_consts(rk0, cc, smem, 1u, lane); b2 = body_lane_consts(rk0, cc, smem, 2u, lane); b3 = body_lane_consts(rk0, cc, smem, 3u, lane); }
#ifndef AESGCM_ROWS_RKV_FROM
#define AESGCM_ROWS_RKV_FROM(NR) ((NR) == 14 ? 36 : 4 * ((NR) + 1))
#endif
    u32 rk[4 * (NR + 1)];                                              // the keys of the late rounds in vector registers: see body_strand_rows
#pragma unroll
    for (int w = 0; w < 4 * (NR + 1); w++) rk[w] = (w >= AESGCM_ROWS_RKV_FROM(NR)) ? pin_vgpr(rk0[w]) : rk0[w];
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (u32 i = 0; i < n; ++i) {
        const u32 r = r0 + i, v = r & 3u, q = r >> 2;                  // wave-uniform
        if (prio_rows) body_prio(i, prio_rows, prio_slot);
        if (i) acc = ghash_mul_const_lds(acc, smem);
        const u64 off = (u64)r * 1024;
        const unsigned char *src = reinterpret_cast<const unsigned char *>(uniform64((u64)(uintptr_t)in + off));
        unsigned char *dst = reinterpret_cast<unsigned char *>(uniform64((u64)(uintptr_t)out + off));
        const uint4 x = (MODE == MODE_PROBE) ? make_uint4(lane, q, v, 0u) : gload16_any(src + lane16);      // (k_rows feeds this loop rows of packets packed from ANY byte address: the accessor whose vector type promises no alignment -- the same global_load_dwordx4; the store below is inline asm and promises nothing)
        const bool hi = (v & 2u) != 0, odd = (v & 1u) != 0;
        BodyLane b;
        if (HOLD) {
            b.p0 = hi ? (odd ? b3.p0 : b2.p0) : (odd ? b1.p0 : b0.p0); b.p1 = hi ? (odd ? b3.p1 : b2.p1) : (odd ? b1.p1 : b0.p1);
            b.p2 = hi ? (odd ? b3.p2 : b2.p2) : (odd ? b1.p2 : b0.p2); b.p3 = hi ? (odd ? b3.p3 : b2.p3) : (odd ? b1.p3 : b0.p3);
        } else b = body_lane_consts(rk0, cc, smem, v, lane);
        u32 s0, s1, s2, s3;
        body_state(s0, s1, s2, s3, b, ctr_hi0 + q, v, lane, cc, rk, tb);
        body_rounds<NR, T4>(s0, s1, s2, s3, rk, smem, lb);
        const uint4 y = make_uint4(x.x ^ s0, x.y ^ s1, x.z ^ s2, x.w ^ s3);                  // gcm_gctr.vhd:150
        if (MODE != MODE_PROBE) {
#if AESGCM_BODY_WT
            gstore16_wt(dst, lane16, y);
#else
            gstore16(dst + lane16, y);
#endif
        }
        acc = xor4(acc, (MODE == MODE_DEC) ? x : y);                  // aes_gcm.vhd:207-211
    }
    return acc;
}
// dealt chunks: lane `lane` of the wave that owns chunk c = 4*s + v: returns sum_i X[row 4(sT+i)+v, lane] * (H^256)^(T-1-i)
template <int NR, int MODE>
HD uint4 body_chunk_lane(const KeyMaterial *__restrict__ km, const DevTables *__restrict__ tb, const BodyParams &p,
                         const unsigned char *smem, const CtrConsts &cc, u32 c, u32 lane) {
    return body_strand_lane<NR, MODE>(km, tb, p, smem, cc, (c >> 2) * p.T, 1u, p.T, c & 3u, lane);      // (rotating priorities here was tried: nothing, profiles/archive/r03c/body_prio_dealt.txt)
}
// Cyclic rows (mid-size ranges, BodyParams::cyc): no dispenser, no item per chunk, and the whole range -- AAD, odd first block, ragged end -- in ONE launch.
// The GHASH sequence of the range is laid on a grid of 64-block rows that is aligned to the BODY: F front rows (the AAD blocks and the data blocks up to the
// first block whose message index is a multiple of 256, padded with zero blocks in FRONT, which GHASH does not see), R whole rows of the body, and behind them at
// most one partial row of tb blocks.  Wave s of the launch's 4096 takes the rows  rho, rho + 4096, rho + 8192, ...  of the F + R grid rows
// (rho = (s + F + R) mod 4096) as ONE Horner with the constant H^(64 * 4096) = H^(2^18) (main_fill_lds(GH_TAB_K2P18)) and leaves item s: the last
// rows of the 4096 strands are the last 4096 rows of the grid in the order of s, so the items are 64 blocks apart like the chunks of a k_main
// launch with one row per chunk, whatever the length (strands without a row leave a zero item in front, which a Horner fold passes through) --
// always 4096 items, one k_fold level, k_combine.  A strand's front row, if it has one (F <= 4096: at most its first), goes through k_main's
// general row code (main_chunk_lane on a one-row chunk); every other row is a body row with the round-1/2 shortcuts, and all body rows of a wave
// have the same phase v = (rho - F) mod 4.  The partial row behind the body cannot sit on the grid (its end is the end of the sequence): wave 0
// (a strand of the shorter kind) takes it as a front-padded row of its own and leaves item 4096, which k_combine weights separately
// (CombineParams::tail_item).  At any moment the waves of the launch work on 4096 consecutive rows (a 4 MiB window), and every wave has the
// same number of rows to within one: nothing to balance as long as the launch is short against the drift of the issue arbitration (which is
// what the dealt chunks of a long launch are for).
#define BODY_CYC_WAVES 4096u             /* waves of the launch = 256 workgroups x 16: the stride of a strand in rows */
#define BODY_CYC_MAX_FRONT BODY_CYC_WAVES /* front rows the layout admits (4 MiB of AAD): one per strand */
// The HALF shape (round 4, for callers that keep several messages in flight): 256 workgroups of 512 lanes with the two-table round (77 KiB of LDS, 128
// registers), i.e. 2048 strands with the stride H^(64 * 2048) = H^(2^17) -- half a CU's wave slots, registers and LDS, so that the workgroup of ANOTHER message
// shares the CU: one's table staging and closing (10 us of a 32 us launch at 16 MiB, profiles/r04/cyc_timeline_*.txt) run beside the other's rows.  Alone on
// the chip such a launch has two waves per SIMD and is slow; the library uses it only on request (context option "cyc_half").
#define BODY_CYC_WAVES_HALF 2048u
template <u32 W = BODY_CYC_WAVES>
HD u32 body_cyc_residue(const BodyParams &p, u32 s) { return (s + p.F + p.R) & (W - 1u); }
template <int NR, int MODE, bool T4 = (AESGCM_T4 != 0), u32 W = BODY_CYC_WAVES>      // W = waves of the launch (BodyParams::cw says the same to the host side)
HD uint4 body_cyc_lane(const KeyMaterial *__restrict__ km, const DevTables *__restrict__ tb, const BodyParams &p,
                       const unsigned char *smem, const CtrConsts &cc, u32 s, u32 lane) {
    const u32 Rt = p.F + p.R;
    u32 u = body_cyc_residue<W>(p, s);
    uint4 acc = make_uint4(0, 0, 0, 0);
    bool started = false;
    if (u < p.F) {                                                     // wave-uniform
        acc = main_chunk_lane<NR, MODE, AESGCM_BODY_WT != 0>(km, p.front, smem, cc, u, lane);
        started = true;
        u += W;
    }
    if (u >= Rt) return acc;
    const u32 r0 = u - p.F, n = (Rt - u + W - 1u) / W;
    return body_strand_lane<NR, MODE, T4>(km, tb, p, smem, cc, r0 >> 2, W / 4u, n, r0 & 3u, lane, acc, started, p.prio_rows, (s >> 2) & 3u);   // wave s of a workgroup sits on SIMD s & 3
}
// the partial row behind the body (lane values = one-row item, right-aligned)
template <int NR, int MODE>
HD uint4 body_cyc_last_lane(const KeyMaterial *__restrict__ km, const BodyParams &p, const unsigned char *smem, const CtrConsts &cc, u32 lane) {
    return main_chunk_lane<NR, MODE, AESGCM_BODY_WT != 0>(km, p.last, smem, cc, 0u, lane);
}

// Fused closing of a cyclic launch (BodyParams::fuse: whole messages).  Every workgroup folds its own sixteen items -- they are consecutive,
// 64 blocks apart -- in a binary tree, four table multiplies deep, with the constants H^64, H^128, H^256, H^512 (the key's nibble tables ptab[0 .. 3],
// staged where the T-tables were): level k leaves  y = x_even * H^(64 * 2^k) ^ x_odd.  Wave 0 then closes the workgroup's item G the way k_combine
// closes a message -- lane L contributes G_L * H^(65 - L + tb) through the per-lane Shoup tables, tb = blocks of the partial last row -- and weights
// the sum with H^(1024 (255 - g)), the blocks between the end of its items and the end of the grid (pw[1][255 - g], as a two-table Shoup form built
// in LDS).  Workgroup 0 adds the terms that occur once: the partial last row (lane L: T_L * H^(65 - L)), the length block times H and E_K(J0).  The
// workgroup's 16 bytes are XORed into one of CYC_ACC_SLOTS accumulators with memory-side atomics; the workgroup that finds all others arrived XORs
// the slots together, zeroes them for the next launch and publishes the tag.  Nothing but atomics crosses workgroups, which is what the dispensers
// already rely on (the XCDs' L2s are not coherent with each other for plain loads).
#define CYC_ACC_SLOTS 16u
#define CYC_LDS_TREE_TAB 0u                  /* four nibble tables of 8 KiB: H^64, H^128, H^256, H^512 */
#define CYC_LDS_STAGE 32768u                 /* 16 + 8 + 4 + 2 items of 1 KiB: the levels of the tree */
#define CYC_LDS_WTAB (CYC_LDS_STAGE + 30u * 1024u)   /* two-table Shoup form of the workgroup's weight (512 B) */
#define CYC_LDS_LTAB (CYC_LDS_WTAB + 512u)       /* lane L's Shoup tables of H^(65 - L + tb): 64 x 32 entries, 528 bytes apart (the 16 spare bytes spread the lanes over the banks) */
#define CYC_LDS_LTAB_STRIDE 528u
#define CYC_LDS_LTAB0 (CYC_LDS_LTAB + 64u * CYC_LDS_LTAB_STRIDE)   /* workgroup 0, when there is a partial last row: the same for H^(65 - L) */
#define CYC_LDS_END (CYC_LDS_LTAB0 + 64u * CYC_LDS_LTAB_STRIDE)
#define CYC_LDS_PARK AESGCM_LDS_BYTES_T4      /* behind the row loop's tables: wave 0 of workgroup 0 parks its last-row item (64 x 16 B) and E_K(J0) here until the closing */
#define CYC_LDS_PARK_BYTES 1040u
// ... and of the half shape (cyc_close_half: eight items per workgroup, 77 KiB of LDS in all): three tree tables, 8 + 4 + 2 staged items, the weight, the lanes' tables
#define CYCH_LDS_TREE_TAB 0u                 /* three nibble tables of 8 KiB: H^64, H^128, H^256 */
#define CYCH_LDS_STAGE 24576u
#define CYCH_LDS_WTAB (CYCH_LDS_STAGE + 14u * 1024u)
#define CYCH_LDS_LTAB (CYCH_LDS_WTAB + 512u)
#define CYCH_LDS_END (CYCH_LDS_LTAB + 64u * CYC_LDS_LTAB_STRIDE)
#define CYCH_LDS_PARK AESGCM_LDS_BYTES       /* behind the row loop's tables (T0 | T2 and the five-bit GHASH tables) */
HD u32 cych_stage_off(u32 level) { return CYCH_LDS_STAGE + (level == 0 ? 0u : level == 1 ? 8192u : 12288u); }
HD uint4 cych_tree_lane(const unsigned char *smem, u32 level, u32 k, u32 lane);
HD u32 cyc_stage_off(u32 level) { return CYC_LDS_STAGE + (level == 0 ? 0u : level == 1 ? 16384u : level == 2 ? 24576u : 28672u); }   // where the inputs of tree level `level` sit
// tree level `level` (0 .. 3), pair k: lane `lane`
HD uint4 cyc_tree_lane(const unsigned char *smem, u32 level, u32 k, u32 lane) {
    const u32 in = cyc_stage_off(level);
    const uint4 xe = *reinterpret_cast<const uint4 *>(smem + in + (2u * k) * 1024u + lane * 16u);
    const uint4 xo = *reinterpret_cast<const uint4 *>(smem + in + (2u * k + 1u) * 1024u + lane * 16u);
    return xor4(ghash_mul_const_lds_at_lean(xe, smem, CYC_LDS_TREE_TAB + level * 8192u), xo);
}
HD uint4 cych_tree_lane(const unsigned char *smem, u32 level, u32 k, u32 lane) {
    const u32 in = cych_stage_off(level);
    const uint4 xe = *reinterpret_cast<const uint4 *>(smem + in + (2u * k) * 1024u + lane * 16u);
    const uint4 xo = *reinterpret_cast<const uint4 *>(smem + in + (2u * k + 1u) * 1024u + lane * 16u);
    return xor4(ghash_mul_const_lds_at_lean(xe, smem, CYCH_LDS_TREE_TAB + level * 8192u), xo);
}
// lane L's term of a workgroup item (tb = blocks behind the grid) or of the partial last row (tb = 0)
HD G128 cyc_lane_term(const KeyMaterial *__restrict__ km, uint4 item, u32 lane, u32 tb) { return shoup2_gmul(mo_to_be(item), km->ltab[65u - lane + tb]); }
// ... of a workgroup item, from the copy of the lanes' tables that the workgroup staged in LDS (entry k of 2048: lane k >> 5, entry k & 31)
HD uint4 cyc_ltab_entry(const KeyMaterial *__restrict__ km, u32 k, u32 tb) { return km->ltab[65u - (k >> 5) + tb][k & 31u]; }
HD u32 cyc_ltab_off(u32 k, u32 base = CYC_LDS_LTAB) { return base + (k >> 5) * CYC_LDS_LTAB_STRIDE + (k & 31u) * 16u; }
HD G128 cyc_lane_term_lds(const unsigned char *smem, uint4 item, u32 lane, u32 base = CYC_LDS_LTAB) { return shoup2_gmul_lds(mo_to_be(item), reinterpret_cast<const uint4 *>(smem + base + lane * CYC_LDS_LTAB_STRIDE)); }

// ---- k_combine pieces --------------------------------------------------------------------------
// One workgroup per message (or per shard / streaming step).  Round 2: the launch also folds up to 64 chunk items itself
// (two Horner stages with the key's precomputed H^(2^k) tables, as k_fold does for more) and then needs ONE level of
// per-lane constant multiplies through the key's precomputed Shoup tables (KeyMaterial::ltab) instead of three 128-step
// bit-serial multiplies in sequence:  tag = sum_L B_L*H^(65-L) ^ L*H ^ E_K(J0)  (gcm_ghash.vhd:257,293 re-associated).
// A 64 KiB message is now k_main + k_combine (it was k_main + k_fold + k_combine + a 16-byte copy kernel).
#define COMBINE_THREADS 1024u                /* 16 waves: GMAX gathered-partial lanes, 16 level-1 fold groups, the single-term wave */
enum { PARTS_NONE = 0, PARTS_GATHERED = 1, PARTS_ITEM = 2 };
#define COMBINE_FOLD_GROUP 4u                /* fan-in of every level of the in-launch fold: 64 -> 16 -> 4 -> 1, three multiplies deep each */
#define CMB_LDS_TABA 0u                      /* nibble tables of H^(eA): items are eA blocks apart */
#define CMB_LDS_TABB 8192u                   /* ... of H^(4 eA): level-1 results */
#define CMB_LDS_TABC 16384u                  /* ... of H^(16 eA): level-2 results */
#define CMB_LDS_STAGE1 24576u                /* 16 level-1 results x 1 KiB */
#define CMB_LDS_STAGE2 40960u                /* 4 level-2 results x 1 KiB */
#define CMB_LDS_SBOX 45056u
#define CMB_LDS_RED 45312u                   /* one 16-byte slot per wave */
#define CMB_LDS_BYTES (CMB_LDS_RED + 16u * (COMBINE_THREADS / 64u))
struct CombineParams {
    const uint4 *parts; u32 np; u32 kind;   // GATHERED: np weighted 16-byte partials (shards); ITEM: np <= 64 chunk items of 64 lanes, eA blocks apart
    u32 stride;                  // GATHERED: distance between consecutive partials in 16-byte units (0 = 1): an all-gather of M messages' partials leaves [rank][message]
    u64 eA;                      // ITEM, np > 1: blocks between the ends of consecutive items (a power of two: the tables come from KeyMaterial::ptab)
    const uint4 *tail_item; u32 tail_blocks;   // ITEM: one more item of 64 lanes whose end is the end of the sequence, tail_blocks (<= 64) blocks behind the end of the others (k_body's cyclic rows: the partial last row)
    const uint4 *tabA, *tabB, *tabC;   // device pointers to the nibble tables of H^eA, H^(4 eA), H^(16 eA) (filled in by the host side)
    u32 want_tag;                // 1 = TAG, 0 = POLY
    u64 e;                       // POLY: exponent applied to the folded partials
    const uint4 *carry; u64 e_carry; u32 has_carry;
    u64 aad_len, ct_len;         // bytes, for the length block
    u32 iv0, iv1, iv2;
    const uint4 *ej0;            // E_K(IV || 1) left by k_main, or NULL: k_combine computes it (one lane, bytewise: ~15 us)
    uint4 *out;
    uint4 *out_host;             // optional second copy of the result in host-visible (pinned, mapped) memory: no copy kernel for the tag
    u64 gen;                     // written to out_host[1] AFTER the result (system-scope fence between): the host polls it
};
// In-launch fold of n <= 64 chunk items: three levels of fan-in 4 (64 -> 16 -> 4 -> 1), each level a Horner over at most four
// values with a wave-uniform constant (three dependent table multiplies of ~0.6 us for a lone wave, where two levels of
// fan-in 8 were fourteen).  At every level the groups are cut from the END, so only the first group is short and group ends
// stay equally spaced: 4 eA after level 1, 16 eA after level 2.
HD u32 fold4_groups(u32 n) { return (n + COMBINE_FOLD_GROUP - 1) / COMBINE_FOLD_GROUP; }
HD void fold4_range(u32 n, u32 g, u32 *s, u32 *e) {        // the values [*s, *e) of group g of fold4_groups(n)
    const u32 J = fold4_groups(n);
    *e = n - COMBINE_FOLD_GROUP * (J - 1 - g);
    *s = g == 0 ? 0 : *e - COMBINE_FOLD_GROUP;
}
// level 1 loads: they depend on nothing and each is an L2 round trip, so k_combine issues them before it stages its tables
struct CombineItems { uint4 it[COMBINE_FOLD_GROUP]; u32 n; };
HD CombineItems combine_fold_load(const CombineParams &p, u32 g, u32 lane) {
    u32 s, e;
    fold4_range(p.np, g, &s, &e);
    CombineItems c;
    c.n = e - s;
#pragma unroll
    for (u32 k = 0; k < COMBINE_FOLD_GROUP; ++k) c.it[k] = (s + k < e) ? p.parts[(size_t)(s + k) * 64 + lane] : make_uint4(0, 0, 0, 0);
    return c;
}
HD uint4 combine_fold_items(const CombineItems &c, const unsigned char *smem, u32 tab) {
    uint4 acc = c.it[0];
#pragma unroll
    for (u32 k = 1; k < COMBINE_FOLD_GROUP; ++k)
        if (k < c.n) acc = xor4(ghash_mul_const_lds_at(acc, smem, tab), c.it[k]);
    return acc;
}
// levels 2 and 3: group g of the n values staged at `stage` (1 KiB each), constant table at `tab`
HD uint4 combine_fold_staged(const unsigned char *smem, u32 stage, u32 n, u32 g, u32 tab, u32 lane) {
    u32 s, e;
    fold4_range(n, g, &s, &e);
    CombineItems c;
    c.n = e - s;
#pragma unroll
    for (u32 k = 0; k < COMBINE_FOLD_GROUP; ++k) c.it[k] = (s + k < e) ? *reinterpret_cast<const uint4 *>(smem + stage + (s + k) * 1024u + lane * 16u) : make_uint4(0, 0, 0, 0);
    return combine_fold_items(c, smem, tab);
}
// The closing of a tag without closing multiplies: tag = P*H^2 ^ L*H ^ E_K(J0) with P = sum_L B_L * H^(63-L), so lane L
// contributes B_L * H^(65-L) and the length block L contributes L * H^1 -- 65 per-lane constant multiplies in parallel through
// the key's precomputed tables (km->ltab[e], shoup2_gmul), one multiply deep.  Shared by k_combine and k_main's tail.
HD G128 tag_lane_term(const KeyMaterial *__restrict__ km, uint4 b, u32 lane) { return shoup2_gmul(mo_to_be(b), km->ltab[65 - lane]); }
HD G128 tag_len_term(const KeyMaterial *__restrict__ km, u64 aad_len, u64 ct_len) {
    G128 L; const u64 a = aad_len * 8, c = ct_len * 8;
    L.w[0] = (u32)(a >> 32); L.w[1] = (u32)a; L.w[2] = (u32)(c >> 32); L.w[3] = (u32)c;
    return shoup2_gmul(L, km->ltab[1]);
}
// E_K(IV || 0^31 1): the J0 block the RTL latches first (gcm_ghash.vhd:158-169), bytewise (only when k_main left none behind)
HD G128 combine_ej0_bytes(const KeyMaterial *__restrict__ km, const uint8_t *sbox, const CombineParams &p) {
    uint8_t j0[16], o[16];
    const u32 ivw[3] = {p.iv0, p.iv1, p.iv2};
    for (int k = 0; k < 12; k++) j0[k] = (uint8_t)(ivw[k >> 2] >> (8 * (k & 3)));
    j0[12] = 0; j0[13] = 0; j0[14] = 0; j0[15] = 1;
    aes_block_bytes(km->rk_bytes, (int)km->nr, sbox, j0, o);
    return mo_to_be(make_uint4(load_le32(o), load_le32(o + 4), load_le32(o + 8), load_le32(o + 12)));
}
// length block [8*len(A)]_64 || [8*len(C)]_64 (gcm_ghash.vhd:257)
HD G128 combine_len_block(const CombineParams &p) {
    G128 L; const u64 a = p.aad_len * 8, c = p.ct_len * 8;
    L.w[0] = (u32)(a >> 32); L.w[1] = (u32)a; L.w[2] = (u32)(c >> 32); L.w[3] = (u32)c;
    return L;
}
// H^e as the product of its four radix-WG digit entries (e < WG^4 >= 2^36)
HD G128 gf_pow_h_digit(const KeyMaterial *km, u64 e, u32 d) { return mo_to_be(km->pw[d][(e >> (AESGCM_LOG_WG * d)) & (u64)(AESGCM_WG - 1)]); }

HD G128 gf_pow_h_serial(const KeyMaterial *km, u64 e) {
    G128 v = gf_pow_h_digit(km, e, 0);
    for (u32 d = 1; d < 4; d++) { const u64 dig = (e >> (AESGCM_LOG_WG * d)) & (u64)(AESGCM_WG - 1); if (dig) v = gf_mul(v, gf_pow_h_digit(km, e, d)); }
    return v;
}

// ---- host-side planning (shared by the C ABI and the CPU harness) -------------------------------
static inline void iv_to_words(const uint8_t iv[12], u32 w[3]) { for (int q = 0; q < 3; q++) w[q] = load_le32(iv + 4 * q); }

// how many dispenser queues a launch of C chunks uses, and the chunks per queue
HD void plan_queues(u32 C, u32 *nq, u32 *seg) {
    u32 n = C >= 4096 ? AESGCM_NQ : C >= 512 ? 4 : 1;
    *nq = n; *seg = (C + n - 1) / n;
}
// Fill MainParams for one launch; returns the number of chunks (0 = nothing to launch).
static inline u32 plan_main(MainParams &p, int mode, u32 tw_override, const uint8_t *iv, const void *aad, u64 aad_len,
                            const void *in, u64 len, void *out, u64 first_block, uint4 *parts) {
    const bool gh = (mode == MODE_ENC || mode == MODE_DEC);
    const u64 n_aad = gh ? (aad_len + 15) / 16 : 0;
    const u64 n_seq = n_aad + (len + 15) / 16;
    if (n_seq == 0) return 0;
    u64 R; u32 Tw, C;
    main_geometry(n_seq, tw_override, &R, &Tw, &C);
    p.in = (const unsigned char *)in; p.out = (unsigned char *)out; p.aad = (const unsigned char *)aad;
    p.parts = parts;
    p.aad_len = gh ? aad_len : 0; p.n_aad = n_aad; p.len = len; p.n_seq = n_seq;
    p.rows = R; p.pad = (u32)(64 * R - n_seq); p.Tw = Tw; p.C = C; p.R0 = (u32)(R - (u64)(C - 1) * Tw);
    {
        const u64 fds = (u64)p.pad + n_aad;                            // virtual slot of data block 0
        const u64 full = (len / 16);                                     // data blocks that are 16 bytes long
        const u64 lo = (fds + 63) / 64, hi = (fds + full) / 64;
        p.row_lo = (u32)lo; p.row_hi = (u32)(hi > lo ? hi : lo);
    }
    p.ctr0 = (u32)(2 + first_block);
    u32 w[3] = {0, 0, 0};
    if (iv) iv_to_words(iv, w);
    p.iv0 = w[0]; p.iv1 = w[1]; p.iv2 = w[2];
    p.aad_aligned = (((uintptr_t)aad) & 15) == 0;
    return C;
}
// Split of a data range for k_body: [head blocks][body = S super-chunks of 256*T blocks][tail].  The body starts at
// the first block whose index in the message (first_block + i) is a multiple of 256 -- no head at all for a whole
// message or a shard cut at such an index -- and holds only whole 16-byte blocks.  Returns false when the range is too small to be worth three launches (min_bytes).
struct BodySplit { u64 head_blocks, body_blocks; u32 T, S; };
static inline bool plan_body_split(u64 len, u64 first_block, u32 tw_override, u64 min_bytes, BodySplit *b) {
    const u64 nfull = len / 16;
    const u64 head = (256 - (first_block & 255)) & 255;                // to the next multiple of 256 of the message block index
    if (nfull <= head) return false;
    const u64 rows = (nfull - head) / 64;
    u64 R; u32 Tw, C;
    main_geometry(rows * 64, tw_override, &R, &Tw, &C);
    if (!Tw) return false;
    const u64 S = rows / (4ull * Tw);
    if (!S || S * 4 > 0xFFFFFFFFull / 2) return false;
    const u64 body = S * 256ull * Tw;
    if (body * 16 < min_bytes) return false;
    b->head_blocks = head; b->body_blocks = body; b->T = Tw; b->S = (u32)S;
    return true;
}
static inline void plan_body(BodyParams &p, const BodySplit &b, const uint8_t *iv, const void *in, void *out, u64 first_block, uint4 *parts) {
    p.in = (const unsigned char *)in + 16 * b.head_blocks; p.out = (unsigned char *)out + 16 * b.head_blocks;
    p.parts = parts; p.T = b.T; p.C = 4 * b.S;
    p.ctr_hi0 = (u32)((first_block + b.head_blocks) >> 8);
    u32 w[3]; iv_to_words(iv, w); p.iv0 = w[0]; p.iv1 = w[1]; p.iv2 = w[2];
}
// A whole range (AAD, data from any first block, ragged end) as ONE k_body launch of cyclic rows (body_cyc_lane), when its body -- the whole 64-block
// rows from the first block whose message index is a multiple of 256 -- has [min_bytes, max_bytes) bytes and the blocks in front of it fit one front
// row per strand.  Fills p (parts: BODY_CYC_WAVES + 1 items) and returns true; `tail_blocks` = blocks of the partial row behind the body.
static inline bool plan_body_cyc(BodyParams &p, int mode, const uint8_t *iv, const void *aad, u64 aad_len, const void *in, u64 len, void *out,
                                 u64 first_block, uint4 *parts, u64 min_bytes, u64 max_bytes, u32 waves = BODY_CYC_WAVES) {
    const u64 nfull = len / 16;
    const u64 head = (256 - (first_block & 255)) & 255;
    if (nfull <= head) return false;
    const u64 R = (nfull - head) / 64;
    if (!R || R * 1024 < min_bytes || R * 1024 >= max_bytes) return false;
    const u64 n_aad = (aad_len + 15) / 16, F = (n_aad + head + 63) / 64;
    if (F > waves || F + R > 0x7FFFFFFFull) return false;             // one front row per strand at most
    { const BodyParams zero = {}; p = zero; }
    const u64 done = head + 64 * R;                                    // data blocks in front of the partial row
    p.cyc = 1; p.cw = waves; p.F = (u32)F; p.R = (u32)R;
    p.in = (const unsigned char *)in + 16 * head; p.out = (unsigned char *)out + 16 * head;
    p.parts = parts;
    p.ctr_hi0 = (u32)((first_block + head) >> 8);
    u32 w[3]; iv_to_words(iv, w); p.iv0 = w[0]; p.iv1 = w[1]; p.iv2 = w[2];
    // the generic pieces as chunks of one row (tw_override = 1): chunk c of `front` is grid row c
    const u32 Cf = plan_main(p.front, mode, 1, iv, aad, aad_len, in, 16 * head, out, first_block, nullptr);
    const u32 Cl = plan_main(p.last, mode, 1, iv, nullptr, 0, (const unsigned char *)in + 16 * done, len - 16 * done,
                             (unsigned char *)out + 16 * done, first_block + done, nullptr);
    if (Cf != (u32)F || Cl > 1) return false;                          // cannot happen: the geometry above is plan_main's
    p.tb = Cl ? (u32)p.last.n_seq : 0;
    return true;
}
// whole-message tag from the folded item:  P*H^2 ^ L*H ^ E_K(J0)
static inline CombineParams plan_combine_tag(const uint4 *parts, u32 np, u32 kind, const uint8_t iv[12],
                                             u64 aad_len, u64 ct_len, uint4 *out) {
    CombineParams q = {};
    q.parts = parts; q.np = np; q.kind = np ? kind : (u32)PARTS_NONE; q.want_tag = 1;
    q.aad_len = aad_len; q.ct_len = ct_len;
    u32 w[3]; iv_to_words(iv, w); q.iv0 = w[0]; q.iv1 = w[1]; q.iv2 = w[2];
    q.out = out;
    return q;
}
// chunk items handed to k_combine unfolded: their spacing (the host side adds the table pointers)
static inline CombineParams combine_with_items(CombineParams q, u64 eA, const uint4 *tail_item = nullptr, u32 tail_blocks = 0) { q.eA = eA; q.tail_item = tail_item; q.tail_blocks = tail_blocks; return q; }
// polynomial value of local partials times H^e (shard partial, aesgcm_ghash)
static inline CombineParams plan_combine_poly(const uint4 *parts, u32 np, u32 kind, u64 e, uint4 *out) {
    CombineParams q = {};
    q.parts = parts; q.np = np; q.kind = np ? kind : (u32)PARTS_NONE; q.e = e; q.out = out;
    return q;
}
// streaming: Y' = Y * H^nb ^ P(new blocks)
static inline CombineParams plan_combine_carry(const uint4 *parts, u32 np, u32 kind, uint4 *state, u64 nb) {
    CombineParams q = {};
    q.parts = parts; q.np = np; q.kind = np ? kind : (u32)PARTS_NONE; q.carry = state; q.has_carry = 1; q.e_carry = nb; q.out = state;
    return q;
}
// streaming final: tag = Y*H^2 ^ L*H ^ E_K(J0)
static inline CombineParams plan_combine_final(uint4 *state, const uint8_t iv[12], u64 aad_len, u64 ct_len, uint4 *out) {
    CombineParams q = plan_combine_tag(nullptr, 0, PARTS_NONE, iv, aad_len, ct_len, out);
    q.carry = state; q.has_carry = 1; q.e_carry = 0;
    return q;
}

