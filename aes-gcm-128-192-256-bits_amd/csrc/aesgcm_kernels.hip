// aesgcm_kernels.hip -- HIP kernels (gfx950) + the C ABI of include/aesgcm.h.
//
// Kernels (DESIGN.md section 5 says what binds each)
//   k_init_tables   per device, once: S-box (256 B) and T0 (1 KiB) computed from their definitions.
//   k_setup, k_setup_ptab   per key: aes_kexp (or pre-expanded load), H = E_K(0), H-power tables, the five-bit / nibble / Shoup tables of the launch constants.
//   k_main<NR,MODE> fused AES-CTR + GHASH over dealt or owned chunks: messages below 64 KiB, heads and tails of larger ones; ECB / keystream modes.
//   k_body<NR,MODE,CYC>   rounds 1-2 without LDS lookups, four T-tables.  CYC = false: dealt chunks (messages and shards from 1 GiB); CYC = true: cyclic rows,
//                   one launch per message of 64 KiB .. 1 GiB that closes the tag itself (cyc_close).
//   k_bodyh<NR,MODE>      the cyclic rows in a half shape (512 lanes, two-table round, two workgroups per CU) for messages in flight beside each other.
//   k_fold          reduces the chunks' items (64 lane accumulators each) by Horner with wave-uniform constants; its first level may close the tag (FoldClose).
//   k_combine       per message: H^(65-L) on the last item, lane fold, optional H^e weighting / chaining value
//                   (shards, streaming), length block, E_K(J0) -> tag.  k_combine_batch: up to 8 messages, one workgroup each.
//   k_batch3<NR,DEC,LG>   packets with their OWN key: 8, 16 or 64 lanes per packet, one pass (per-packet aes_kexp, CTR, GHASH with the packet's own tables).
//   k_pktg<NR,DEC,LG>, k_pktl      packets under the context's key: 2^LG lanes per packet (4, 8, 16, 64) / one lane per packet.
//   k_len_hist, k_len_scan, k_len_scatter   the order in which a launch takes packets of mixed length: a counting sort by falling length class.
//   k_gfmul, k_fill_splitmix64, k_copy16   small utility kernels.
//
// GHASH re-association (DESIGN.md "GHASH as a polynomial"): the GHASH input sequence
// A_0..A_{u-1}, C_0..C_{c-1} (n = u + c blocks) is right-aligned into rows of 64 slots (front padding =
// zeros, which do not change a polynomial) and cut into chunks of Tw rows.  Waves pull chunks from atomic
// dispensers; inside a chunk lane L runs Horner over its column with the per-key constant K = H^64
// (acc = acc*K ^ X) and the wave stores the 64 accumulators as the chunk's item.  k_fold folds the items
// lane-wise (B_L = sum_i item_i[L] * H^(blocks to the end)), k_combine forms P = sum_L B_L * H^(63-L).
// The tag is (P*H ^ L)*H ^ E_K(J0) = P*H^2 ^ L*H ^ E_K(J0).
#include "aesgcm_dev.h"
#include "aesgcm_rows.h"
#include "../../include/aesgcm.h"

#include <algorithm>
#include <mutex>
#include <new>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <vector>

// ------------------------------------------------------------------------------------------------
__global__ void k_init_tables(DevTables *t) {
    u32 x = threadIdx.x;
    u32 s = sbox_calc(x);
    t->sbox[x] = (uint8_t)s;
    t->te0[x] = te0_calc(s);
    t->te1[x] = rotl32(te0_calc(s), 8); t->te2[x] = rotl32(te0_calc(s), 16); t->te3[x] = rotl32(te0_calc(s), 24);
}

// one GF multiply per thread: z[i] = x[i] * h[i]   (aesgcm_gfmul; replaces src/ghash_gfmul.vhd:37-64)
__global__ void k_gfmul(const uint4 *h, const uint4 *x, uint4 *z, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) z[i] = gf_mul_mo(x[i], h[i]);
}

// plain copy, 16 bytes per lane, ONE element per thread and no loop: the measured HBM read+write rate bench.py prints beside
// the peak.  Round 2's grid-stride form (8192 x 256 threads looping) reached 4.9 TB/s; this form 6.16 TB/s over the same two
// 16 GiB buffers on the same box (profiles/r03/copy_variants.txt: tiles of 4 .. 16 loads in flight per lane, nontemporal
// accesses and hipMemcpyAsync all sit between 4.6 and 5.6) -- the guide's float4-copy figure is 6.29.
__global__ __launch_bounds__(256) void k_copy16(uint4 *__restrict__ dst, const uint4 *__restrict__ src, u64 n) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < n) gstore16(dst + i, gload16(src + i));
}

__global__ void k_fill_splitmix64(u64 *buf, size_t n_words, size_t tail_bytes, u64 seed, u64 first_word) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride)
        buf[i] = splitmix64_at(seed, first_word + i);
    if (tail_bytes && blockIdx.x == 0 && threadIdx.x == 0) {
        u64 z = splitmix64_at(seed, first_word + n_words);
        unsigned char *p = reinterpret_cast<unsigned char *>(buf + n_words);
        for (size_t k = 0; k < tail_bytes; k++) p[k] = (unsigned char)(z >> (8 * k));
    }
}

// ------------------------------------------------------------------------------------------------
// k_setup: one workgroup of 512 lanes, once per key.
//   lane 0      : key schedule (aes_kexp) or pre-expanded copy, H = E_K(0)  (gcm_gctr.vhd:141-144)
//   all lanes   : four 1025-entry power tables by doubling, nibble tables of H^64, H, H^256 (k_setup_ptab adds H^(2^k)).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AESGCM_WG) void k_setup(KeyMaterial *km, const DevTables *tb, const uint8_t *key, int key_len,
                                                     int preexpanded_nr, u32 G) {
    __shared__ uint4 tab[AESGCM_NPW];
    __shared__ uint8_t s_sbox[256];
    const int tid = threadIdx.x;
    if (tid < 256) s_sbox[tid] = tb->sbox[tid];
    __syncthreads();
    if (tid == 0) setup_lane0(km, s_sbox, key, key_len, preexpanded_nr, G, tab);
    __syncthreads();
    for (int d = 0; d < 4; d++) {
        for (int j = 0; j < AESGCM_LOG_WG; j++) {
            uint4 prod;
            const bool act = setup_level(tab, j, tid, &prod);
            __syncthreads();
            if (act) tab[(1 << j) + tid] = prod;
            __syncthreads();
        }
        for (int k = tid; k < AESGCM_NPW; k += AESGCM_WG) km->pw[d][k] = tab[k];
        __syncthreads();
        if (d == 1) setup_beta_lane(km, tab, tid);
        if (d < 3) {
            uint4 next = tab[AESGCM_WG];
            __syncthreads();
            if (tid == 0) { tab[0] = gf_one_mo(); tab[1] = next; }
            __syncthreads();
        }
    }
}

// Next chunk of a dynamic launch for the calling wave, or DISPENSER_DONE.  `q` is the wave's current queue (wave-uniform).
// The common case is one atomicAdd on the wave's own queue.  A queue found dry is recorded in a per-WORKGROUP bit mask in
// LDS (the spare row of the GHASH table region), so the workgroup's other waves skip it without touching memory: a launch
// makes at most (workgroups x queues) failing fetches instead of (waves x queues).  (Reading the counters with plain
// agent-scope loads instead was tried and is wrong for this part: such loads are served by the XCD's own L2, which is not
// coherent with the memory-side atomics of other XCDs -- waves saw stale "work left" values and spun on dry queues; k_body
// got 25 % slower.)  Every queue keeps its home waves until it is dry, so every chunk is handed out whatever the others do.
#define DISPENSER_DONE 0xFFFFFFFFu
__device__ __forceinline__ u32 next_chunk(u32 *counter, unsigned char *smem, u32 nq, u32 seg, u32 C, u32 &q, u32 lane) {
    u32 *dry = reinterpret_cast<u32 *>(smem + AESGCM_LDS_DRY_OFF);
    const u32 all = nq >= 32 ? 0xFFFFFFFFu : (1u << nq) - 1u;
    for (u32 tries = 0; tries < 4 * AESGCM_NQ; ++tries) {
        const u32 mask = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile u32 *>(dry));
        if ((mask & all) == all) return DISPENSER_DONE;
        if ((mask >> q) & 1u) {                                     // known dry: the next queue (cyclically) that is not
            const u32 live = ~mask & all, above = live & ~((2u << q) - 1u);
            q = (u32)__builtin_ctz(above ? above : live);
        }
        u32 v = 0;
        if (lane == 0) v = atomicAdd(counter + 16 * q, 1u);
        v = __builtin_amdgcn_readfirstlane(v);
        if (v < seg) {
            const u32 c = q * seg + v;
            if (c < C) return c;
            continue;                                               // the last queue is padded to seg (fewer than nq entries)
        }
        if (lane == 0) atomicOr(dry, 1u << q);
    }
    return DISPENSER_DONE;
}

// ------------------------------------------------------------------------------------------------
// k_main: the fused hot path.  Persistent workgroups; after the LDS tables are staged every WAVE is
// autonomous: it pulls chunk indices from the dispenser and processes one 16-byte block per lane per row
// (lane body: main_chunk_lane()).  No barrier after the staging one, so the age-ordered issue arbitration
// of the CU (older waves first) only changes WHO does the work, never how long the kernel's tail is.
// ------------------------------------------------------------------------------------------------
#ifndef AESGCM_WAVES_PER_SIMD
#define AESGCM_WAVES_PER_SIMD (2 * AESGCM_MAIN_WG / 256)   /* two workgroups per CU */
#endif
__device__ __forceinline__ G128 wave_xor_fold(G128 z) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        z.w[0] ^= __shfl_xor(z.w[0], off); z.w[1] ^= __shfl_xor(z.w[1], off);
        z.w[2] ^= __shfl_xor(z.w[2], off); z.w[3] ^= __shfl_xor(z.w[3], off);
    }
    return z;
}
// value of lane (lane ^ MASK).  For MASK < 32 this is ds_swizzle in bit mode (and 0x1F, or 0, xor MASK: no address register);
// __shfl_xor lowers to ds_bpermute with a per-lane index, and the compiler hoists those index registers out of the packet
// loops -- in k_pktg at 128 registers they were 7 of the ~20 dwords it then spilled to scratch (round-3 ISA).
template <int MASK>
__device__ __forceinline__ u32 lane_xor(u32 x) {
    if constexpr (MASK < 32) return (u32)__builtin_amdgcn_ds_swizzle((int)x, (MASK << 10) | 0x1F);
    else return (u32)__shfl_xor((int)x, MASK);
}
// the lane's index in its wave from nothing but the execution mask (no input register, opaque to common-subexpression elimination)
__device__ __forceinline__ u32 lane_id_fresh() {
    u32 x;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x));
    return x;
}
__device__ __forceinline__ u32 lane_xor_pow2(u32 x, int j) {           // lane ^ (1 << j); j is a constant after unrolling
    switch (j) {
    case 0: return lane_xor<1>(x);
    case 1: return lane_xor<2>(x);
    case 2: return lane_xor<4>(x);
    case 3: return lane_xor<8>(x);
    case 4: return lane_xor<16>(x);
    default: return lane_xor<32>(x);
    }
}
// result -> pinned host slot, then (behind a system-scope fence) the generation number the host is polling for
__device__ __forceinline__ void publish_host(uint4 *slot, uint4 v, u64 gen) {
    *slot = v;
    __threadfence_system();
    __hip_atomic_store(reinterpret_cast<u64 *>(slot + 1), gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <int NR, int MODE>
__global__ __launch_bounds__(AESGCM_MAIN_WG, AESGCM_WAVES_PER_SIMD) void k_main(const KeyMaterial *__restrict__ km, const DevTables *__restrict__ tb, const MainParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool GH = (MODE == MODE_ENC || MODE == MODE_DEC);
    const u32 tid = threadIdx.x, lane = tid & 63u;
    if (p.trace && tid == 0) {
        u64 *tr = p.trace + 4 * (u64)blockIdx.x;
        tr[0] = wall_clock64();
        tr[2] = (u64)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((u64)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF) << 32);
    }
    const u64 cyc0 = p.trace ? clock64() : 0;
    main_fill_lds(smem, km, tb, tid, GH);
    if (tid == 0) *reinterpret_cast<u32 *>(smem + AESGCM_LDS_DRY_OFF) = 0;                      // dry-queue mask of the workgroup (next_chunk)
    __syncthreads();
    // round-1 constants depend on key and IV only (the lane merely picks which table replica it reads), so they
    // are wave-uniform: keep them in scalar registers, the vector file is full at 8 waves per SIMD
    CtrConsts cc = main_lane_consts<MODE>(km, p, smem, lane);
    cc.c0 = __builtin_amdgcn_readfirstlane(cc.c0); cc.c1 = __builtin_amdgcn_readfirstlane(cc.c1);
    cc.c2 = __builtin_amdgcn_readfirstlane(cc.c2); cc.c3 = __builtin_amdgcn_readfirstlane(cc.c3);
    u32 done = 0;
    // bounded on purpose: no wave can own more than C chunks (plus one dry fetch per queue), so a dispenser problem can
    // never turn into a hang
    if (p.nq && blockIdx.x == 0 && tid < AESGCM_NQ) p.counter_zero[16 * tid] = 0;             // the next dynamic launch's queues
    u32 q = p.nq ? (blockIdx.x * (AESGCM_MAIN_WG / 64) + (tid >> 6)) % p.nq : 0;                // home queue
    q = __builtin_amdgcn_readfirstlane(q);
    const u32 wave_id = __builtin_amdgcn_readfirstlane(blockIdx.x * (AESGCM_MAIN_WG / 64) + (tid >> 6));
    for (u32 guard = 0; guard <= p.C; ++guard) {
        u32 c;
        if (p.nq == 0) {
            // small launch: at least as many waves as chunks, wave i owns chunk i -- no dispenser round trips on the
            // latency path of a short message
            if (guard) break;
            if (wave_id >= p.C) {
                // a spare wave (the launch has at least C + 1 of them) computes E_K(IV || 1) off the critical path of chunk 0
                if (GH && wave_id == p.C && p.ej0 && !p.tail) {
                    u32 s0, s1, s2, s3;
                    ctr_rounds_lds<NR>(bswap32(1u), cc, s0, s1, s2, s3, km->rk, smem, (lane & 31u) << 2);
                    if (lane == 0) *p.ej0 = make_uint4(s0, s1, s2, s3);
                }
                break;
            }
            c = wave_id;
        } else {
            c = next_chunk(p.counter, smem, p.nq, p.seg, p.C, q, lane);
            if (c == DISPENSER_DONE) break;
        }
        const uint4 acc = main_chunk_lane<NR, MODE>(km, p, smem, cc, c, lane);
        if (GH) p.parts[(size_t)c * 64 + lane] = acc;          // the chunk's item: 64 raw lane accumulators (k_fold / k_combine take over)
        if (GH && c == 0 && (p.ej0 || p.tail) && (p.nq != 0 || p.tail)) {   // E_K(IV || 1) for the tag (gcm_ghash.vhd:158-169), once per launch (static launches: a spare wave does it)
            u32 s0, s1, s2, s3;
            ctr_rounds_lds<NR>(bswap32(1u), cc, s0, s1, s2, s3, km->rk, smem, (lane & 31u) << 2);
            if (lane == 0 && p.ej0) *p.ej0 = make_uint4(s0, s1, s2, s3);
            if (p.tail) {
                // single-chunk message: this wave holds the whole polynomial (lane L: B_L); finish the tag here instead of
                // launching k_combine: tag = sum_L B_L*H^(65-L) ^ L*H ^ E_K(IV || 1), every term one table multiply deep
                G128 term = tag_lane_term(km, acc, lane);
                if (lane == 0) {
                    const G128 lt = tag_len_term(km, p.aad_len, p.len);
                    term.w[0] ^= lt.w[0] ^ bswap32(s0); term.w[1] ^= lt.w[1] ^ bswap32(s1); term.w[2] ^= lt.w[2] ^ bswap32(s2); term.w[3] ^= lt.w[3] ^ bswap32(s3);
                }
                const G128 t = wave_xor_fold(term);
                if (lane == 0) { *p.tag_out = be_to_mo(t); if (p.tag_host) publish_host(p.tag_host, be_to_mo(t), p.gen); }
            }
        }
        ++done;
    }
    if (p.trace && lane == 0) {
        u64 *tr = p.trace + 4 * (u64)blockIdx.x;
        atomicMax((unsigned long long *)&tr[1], (unsigned long long)wall_clock64());
        atomicAdd((unsigned long long *)&tr[3], (unsigned long long)done | ((unsigned long long)((clock64() - cyc0) >> 10) << 32));
    }
}

// ------------------------------------------------------------------------------------------------
// k_body: the aligned middle of a large message (lane body: body_chunk_lane()); rounds 1-2 of every counter
// block come from per-lane chunk constants and scalar-cache table reads, not from LDS.
// ------------------------------------------------------------------------------------------------
#if AESGCM_T4
#ifndef AESGCM_BODY_WG_T4
#define AESGCM_BODY_WG_T4 1024               /* lanes of k_body's workgroup (the cyclic rows and their closing need 1024; 768 = 3 waves per SIMD was the round-4 energy A/B, profiles/r04/energy_ab.txt) */
#endif
#define AESGCM_BODY_WG AESGCM_BODY_WG_T4     /* one workgroup per CU (136 KiB of LDS), 4 waves per SIMD, 128 registers */
#define AESGCM_BODY_WPS ((AESGCM_BODY_WG + 255) / 256)
#define AESGCM_BODY_LDS AESGCM_LDS_BYTES_T4
#else
#define AESGCM_BODY_WG AESGCM_MAIN_WG
#define AESGCM_BODY_WPS AESGCM_WAVES_PER_SIMD
#define AESGCM_BODY_LDS AESGCM_LDS_BYTES
#endif
// the layout assumptions the kernels rely on, checked where they are used
static_assert(2u * AESGCM_LDS_BYTES <= 160u * 1024u, "k_main / k_pktl: two workgroups must share a CU's 160 KiB of LDS");
static_assert(PKTG_LDS_BYTES(6) <= 160u * 1024u && PKTG_LDS_BYTES(4) + 16u * 1024u <= 160u * 1024u && AESGCM_NQ5POW >= 7, "k_pktg: Horner table, T0 | T2, the tree tables (and the E_K(J0) slots of up to 16 waves at 16 lanes per packet) in one CU's LDS");
static_assert(AESGCM_BODY_LDS + CYC_LDS_PARK_BYTES <= 160u * 1024u && CYC_LDS_END <= CYC_LDS_PARK, "k_body: one workgroup per CU; the fused closing's tables end in front of the parked items");
static_assert(AESGCM_LDS_AES_OFF % 128u == 0, "T-table replicas: lane l must read bank l & 31");
static_assert(AESGCM_LDS_DRY_OFF >= AESGCM_Q5_GROUPS * 256u && AESGCM_LDS_DRY_OFF + 4u <= AESGCM_Q5_HI_ROW * 256u, "the dry-queue mask sits in the spare row between the table halves");
static_assert((AESGCM_Q5_HI_ROW * 256u) % 512u != 0 && AESGCM_Q5_HI_ROW * 256u > 2040u, "the two halves of a five-bit table entry must not be fusable into one ds_read2[st64]_b64");
static_assert(FOLD_B_ITEMS >= COMBINE_MAX_ITEMS && FOLD_A_ITEMS >= COMBINE_MAX_ITEMS, "k_fold ping-pong buffers");
// host-visible tag without a full fence: the slot is pinned host memory (stores go out over the fabric, not into L2), so ordering the generation
// number behind the tag takes a wait for the tag's stores, not a write-back of the XCD's L2 -- which in a launch that has just streamed the message
// through that L2 would be megabytes on the critical path
__device__ __forceinline__ void publish_host_lean(uint4 *slot, uint4 v, u64 gen) {
    unsigned long long *q = reinterpret_cast<unsigned long long *>(slot);
    __hip_atomic_store(q, (unsigned long long)v.x | ((unsigned long long)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(q + 1, (unsigned long long)v.z | ((unsigned long long)v.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(q + 2, (unsigned long long)gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// XOR of a value over the lanes of the wave (all lanes get the sum)
__device__ __forceinline__ G128 wave_xor(G128 z) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        z.w[0] ^= __shfl_xor(z.w[0], off); z.w[1] ^= __shfl_xor(z.w[1], off);
        z.w[2] ^= __shfl_xor(z.w[2], off); z.w[3] ^= __shfl_xor(z.w[3], off);
    }
    return z;
}
// One workgroup's 16 bytes of a tag into the accumulator slots, and the tag out of them when this was the launch's last arrival (one lane calls this).
// Memory-side atomics only: the XORs return before the arrival is counted (the increment depends on their results), so the workgroup that counts the
// last arrival finds every contribution in the slots; it zeroes slots and counter for the next launch.
__device__ __forceinline__ void acc_arrive(unsigned long long *acc, u32 g, G128 z, uint4 *tag_out, uint4 *tag_host, u64 gen) {
    const u32 slot = g & (CYC_ACC_SLOTS - 1u);
    const unsigned long long ohi = atomicXor(acc + 2u * slot, ((unsigned long long)z.w[0] << 32) | z.w[1]);
    const unsigned long long olo = atomicXor(acc + 2u * slot + 1u, ((unsigned long long)z.w[2] << 32) | z.w[3]);
    u32 dep;
    asm volatile("v_and_b32 %0, 0, %1" : "=v"(dep) : "v"((u32)(ohi ^ olo) | (u32)((ohi ^ olo) >> 32)));
    const unsigned long long arrived = atomicAdd(acc + 2u * CYC_ACC_SLOTS, 1ull + dep);
    if (arrived + 1ull != gridDim.x) return;
    unsigned long long hi = 0, lo = 0;
#pragma unroll
    for (u32 k = 0; k < CYC_ACC_SLOTS; ++k) { hi ^= atomicExch(acc + 2u * k, 0ull); lo ^= atomicExch(acc + 2u * k + 1u, 0ull); }
    atomicExch(acc + 2u * CYC_ACC_SLOTS, 0ull);
    G128 t; t.w[0] = (u32)(hi >> 32); t.w[1] = (u32)hi; t.w[2] = (u32)(lo >> 32); t.w[3] = (u32)lo;
    *tag_out = be_to_mo(t);
    if (tag_host) publish_host_lean(tag_host, be_to_mo(t), gen);
}
// the fused closing of a cyclic launch (lane pieces and the algebra: aesgcm_dev.h, "Fused closing"); acc = the wave's item; wave 0 of workgroup 0 has left its
// partial last row and E_K(IV || 1) at CYC_LDS_PARK
__device__ __forceinline__ void cyc_close(const KeyMaterial *__restrict__ km, const BodyParams &p, unsigned char *smem, uint4 acc) {
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6, g = blockIdx.x;
    // the tree's tables and the weight, requested before the barrier so that they travel while the workgroup's last waves finish their rows
    const uint4 *pt = &km->ptab[0][0];
    const uint4 t0 = pt[tid], t1 = pt[1024u + tid];
    const uint4 l0 = cyc_ltab_entry(km, tid, p.tb), l1 = cyc_ltab_entry(km, 1024u + tid, p.tb);
    const uint4 wc = km->pw[1][gridDim.x - 1u - g];                            // H^(1024 (255 - g)); the launch has 256 workgroups (enqueue_cyc)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // the rows' stores (the compiler does not count those issued from asm) have been acknowledged
    const bool once = g == 0 && p.tb;                                         // workgroup 0 closes the partial last row as well: H^(65 - L)
    uint4 m0 = make_uint4(0, 0, 0, 0), m1 = m0;
    if (once) { m0 = cyc_ltab_entry(km, tid, 0u); m1 = cyc_ltab_entry(km, 1024u + tid, 0u); }
    __syncthreads();                                                          // every wave is done with the T-tables
    if (once) { *reinterpret_cast<uint4 *>(smem + cyc_ltab_off(tid, CYC_LDS_LTAB0)) = m0; *reinterpret_cast<uint4 *>(smem + cyc_ltab_off(1024u + tid, CYC_LDS_LTAB0)) = m1; }
    reinterpret_cast<uint4 *>(smem + CYC_LDS_TREE_TAB)[tid] = t0;
    reinterpret_cast<uint4 *>(smem + CYC_LDS_TREE_TAB)[1024u + tid] = t1;
    *reinterpret_cast<uint4 *>(smem + cyc_ltab_off(tid)) = l0;
    *reinterpret_cast<uint4 *>(smem + cyc_ltab_off(1024u + tid)) = l1;
    *reinterpret_cast<uint4 *>(smem + cyc_stage_off(0) + wv * 1024u + lane * 16u) = acc;
    __syncthreads();
    uint4 y = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (u32 level = 0; level < 4; ++level) {
        if (wv < (8u >> level)) {
            y = cyc_tree_lane(smem, level, wv, lane);
            if (level < 3) *reinterpret_cast<uint4 *>(smem + cyc_stage_off(level + 1) + wv * 1024u + lane * 16u) = y;
        }
        if (level < 3) __syncthreads();
    }
    if (wv != 0) return;
    // The workgroup's ciphertext has reached its XCD's L2 (the barriers above waited for the stores); an agent-scope release writes that L2 back, so that
    // when the last arrival publishes the tag every byte of the message is in memory -- for the copy engines, the other XCDs and the host -- although the
    // launch itself retires a few microseconds later.  With through-the-L2 row stores (AESGCM_BODY_WT, the default build) nothing is dirty and there is nothing to write back.
#if !AESGCM_BODY_WT
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
    G128 z = wave_xor(cyc_lane_term_lds(smem, y, lane));
    if (g + 1u != gridDim.x) {                                                // weight H^(1024 (255 - g)) through a two-table Shoup form in LDS
        if (lane < 32) *reinterpret_cast<uint4 *>(smem + CYC_LDS_WTAB + 16u * lane) = shoup2_entry(mo_to_be(wc), lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        z = shoup2_gmul_lds(z, reinterpret_cast<const uint4 *>(smem + CYC_LDS_WTAB));
    }
    if (g == 0) {                                                             // the terms that occur once
        G128 x; x.w[0] = x.w[1] = x.w[2] = x.w[3] = 0;
        if (p.tb) x = cyc_lane_term_lds(smem, *reinterpret_cast<const uint4 *>(smem + CYC_LDS_PARK + lane * 16u), lane, CYC_LDS_LTAB0);
        if (lane == 0) {
            const uint4 ej0 = *reinterpret_cast<const uint4 *>(smem + CYC_LDS_PARK + 1024u);
            G128 L; const u64 la = p.aad_len * 8, lc = p.ct_len * 8;       // the length block times H (tag_len_term with the batched multiply: no registers to spare here)
            L.w[0] = (u32)(la >> 32); L.w[1] = (u32)la; L.w[2] = (u32)(lc >> 32); L.w[3] = (u32)lc;
            L = shoup2_gmul_lds(L, km->ltab[1]);
            const G128 e = mo_to_be(ej0);
            x.w[0] ^= L.w[0] ^ e.w[0]; x.w[1] ^= L.w[1] ^ e.w[1]; x.w[2] ^= L.w[2] ^ e.w[2]; x.w[3] ^= L.w[3] ^ e.w[3];
        }
        x = wave_xor(x);
        z.w[0] ^= x.w[0]; z.w[1] ^= x.w[1]; z.w[2] ^= x.w[2]; z.w[3] ^= x.w[3];
    }
    if (lane != 0) return;
    acc_arrive(p.acc, g, z, p.tag_out, p.tag_host, p.gen);
    if (p.trace) {                                                             // timing mode: how long the closing took behind the workgroup's last row, in 10 ns units, bits 52 .. 63 of word 2
        unsigned long long *tr = reinterpret_cast<unsigned long long *>(p.trace + 4 * (u64)g);
        const u64 rows_end = __hip_atomic_load(tr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u64 dt = wall_clock64() - rows_end;
        atomicOr(tr + 2, (unsigned long long)(dt > 0xFFFu ? 0xFFFu : dt) << 52);
    }
}

template <int NR, int MODE, bool CYC>
__global__ __launch_bounds__(AESGCM_BODY_WG, AESGCM_BODY_WPS) void k_body(const KeyMaterial *__restrict__ km, const DevTables *__restrict__ tb, const BodyParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 tid = threadIdx.x, lane = tid & 63u;
    if (p.trace && tid == 0) {
        u64 *tr = p.trace + 4 * (u64)blockIdx.x;
        tr[0] = wall_clock64();
        tr[2] = (u64)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((u64)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF) << 32);
    }
    const u64 cyc0 = p.trace ? clock64() : 0;
    main_fill_lds(smem, km, tb, tid, true, AESGCM_BODY_WG, CYC ? GH_TAB_K2P18 : GH_TAB_K256);
#if AESGCM_T4
    fill_lds_t4(smem, tb, tid, AESGCM_BODY_WG);
#endif
    if (tid == 0) *reinterpret_cast<u32 *>(smem + AESGCM_LDS_DRY_OFF) = 0;   // dry-queue mask of the workgroup (next_chunk)
    __syncthreads();
    if (p.trace && tid == 0) {                                                // timing mode: when the tables were staged, 10 ns units behind the workgroup's start, bits 40 .. 51 of word 2
        unsigned long long *tr = reinterpret_cast<unsigned long long *>(p.trace + 4 * (u64)blockIdx.x);
        const u64 dt = wall_clock64() - tr[0];
        atomicOr(tr + 2, (unsigned long long)(dt > 0xFFFu ? 0xFFFu : dt) << 40);
    }
    CtrConsts cc = ctr_round1_consts(p.iv0, p.iv1, p.iv2, km->rk, smem, (lane & 31u) << 2);   // key and IV only: wave-uniform
    cc.c0 = __builtin_amdgcn_readfirstlane(cc.c0); cc.c1 = __builtin_amdgcn_readfirstlane(cc.c1);
    cc.c2 = __builtin_amdgcn_readfirstlane(cc.c2); cc.c3 = __builtin_amdgcn_readfirstlane(cc.c3);
    u32 done = 0;
    if (CYC) {                                                                // cyclic rows: one strand and one item per wave, no dispenser
        const u32 w = __builtin_amdgcn_readfirstlane(blockIdx.x * (AESGCM_BODY_WG / 64) + (tid >> 6));
        const uint4 acc = body_cyc_lane<NR, MODE>(km, tb, p, smem, cc, w, lane);
        uint4 last = make_uint4(0, 0, 0, 0), ej0 = make_uint4(0, 0, 0, 0);
        if (w == 0) {                                                         // a strand of the shorter kind: the partial last row and E_K(IV || 1)
            if (p.tb) last = body_cyc_last_lane<NR, MODE>(km, p, smem, cc, lane);
            if (p.ej0 || p.fuse) {
                u32 s0, s1, s2, s3;
                ctr_rounds_lds<NR>(bswap32(1u), cc, s0, s1, s2, s3, km->rk, smem, (lane & 31u) << 2);
                ej0 = make_uint4(s0, s1, s2, s3);
            }
        }
        if (p.trace && lane == 0) {
            u64 *tr = p.trace + 4 * (u64)blockIdx.x;
            atomicMax((unsigned long long *)&tr[1], (unsigned long long)wall_clock64());
            atomicAdd((unsigned long long *)&tr[3], (unsigned long long)((p.F + p.R) / BODY_CYC_WAVES) | ((unsigned long long)((clock64() - cyc0) >> 10) << 32));
        }
        if (!p.fuse) {                                                        // items for k_fold / k_combine
            p.parts[(size_t)w * 64 + lane] = acc;
            if (w == 0) {
                if (p.tb) p.parts[(size_t)BODY_CYC_WAVES * 64 + lane] = last;
                if (p.ej0 && lane == 0) *p.ej0 = ej0;
            }
            return;
        }
        if (w == 0) {                                                         // parked in LDS (not in registers: the closing has none to spare) until workgroup 0's wave 0 closes
            *reinterpret_cast<uint4 *>(smem + CYC_LDS_PARK + lane * 16u) = last;
            if (lane == 0) *reinterpret_cast<uint4 *>(smem + CYC_LDS_PARK + 1024u) = ej0;
        }
        cyc_close(km, p, smem, acc);
        return;
    }
    if (blockIdx.x == 0 && tid < AESGCM_NQ) p.counter_zero[16 * tid] = 0;    // the next dynamic launch's queues
    u32 q = (blockIdx.x * (AESGCM_BODY_WG / 64) + (tid >> 6)) % p.nq;
    q = __builtin_amdgcn_readfirstlane(q);
    for (u32 guard = 0; guard <= p.C; ++guard) {                             // bounded, as every dispenser loop here
        const u32 c = next_chunk(p.counter, smem, p.nq, p.seg, p.C, q, lane);
        if (c == DISPENSER_DONE) break;
        const uint4 acc = body_chunk_lane<NR, MODE>(km, tb, p, smem, cc, c, lane);
        p.parts[(size_t)c * 64 + lane] = acc;
        if (c == 0 && p.ej0) {                                  // E_K(IV || 1) for the tag, once per launch (as in k_main)
            u32 s0, s1, s2, s3;
            ctr_rounds_lds<NR>(bswap32(1u), cc, s0, s1, s2, s3, km->rk, smem, (lane & 31u) << 2);
            if (lane == 0) *p.ej0 = make_uint4(s0, s1, s2, s3);
        }
        ++done;
    }
    if (p.trace && lane == 0) {
        u64 *tr = p.trace + 4 * (u64)blockIdx.x;
        atomicMax((unsigned long long *)&tr[1], (unsigned long long)wall_clock64());
        atomicAdd((unsigned long long *)&tr[3], (unsigned long long)done | ((unsigned long long)((clock64() - cyc0) >> 10) << 32));
    }
}

// ------------------------------------------------------------------------------------------------
// k_bodyh: the cyclic rows in the HALF shape (round 4; aesgcm_dev.h, BODY_CYC_WAVES_HALF) -- 256 workgroups of 512 lanes, the two-table round, 77 KiB of LDS and
// 128 registers, so that TWO workgroups share a CU: those of two messages in flight on two streams.  What one launch spends outside its row loop -- 1.8 us of
// table staging, 8 us of closing (tree, lane terms, weight, two atomic round trips; profiles/r04/cyc_timeline_aes256.txt) -- leaves the CU's issue slots and its
// LDS array to the other message's rows.  Whole messages with the tag closed in the launch only (cyc_close_half); same strands, same items, same algebra as
// k_body<.., true> with 2048 waves instead of 4096.
// ------------------------------------------------------------------------------------------------
#define AESGCM_BODYH_WG 512
static_assert(2u * (AESGCM_LDS_BYTES + CYC_LDS_PARK_BYTES) <= 160u * 1024u && CYCH_LDS_END <= CYCH_LDS_PARK, "k_bodyh: two workgroups per CU; the closing's tables end in front of the parked items");
__device__ __forceinline__ void cyc_close_half(const KeyMaterial *__restrict__ km, const BodyParams &p, unsigned char *smem, uint4 acc) {
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6, g = blockIdx.x;
    const uint4 *pt = &km->ptab[0][0];                                        // ptab[0 .. 2]: nibble tables of H^64, H^128, H^256 (512 entries each)
    const uint4 t0 = pt[tid], t1 = pt[512u + tid], t2 = pt[1024u + tid];
    uint4 l[4];
#pragma unroll
    for (u32 k = 0; k < 4; ++k) l[k] = cyc_ltab_entry(km, k * 512u + tid, p.tb);
    const uint4 wc = km->pwh[gridDim.x - 1u - g];                             // H^(512 (255 - g)): the blocks between the end of this workgroup's eight items and the end of the grid
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // the rows' stores (the compiler does not count those issued from asm) have been acknowledged
    __syncthreads();                                                          // every wave is done with the T-tables
    reinterpret_cast<uint4 *>(smem + CYCH_LDS_TREE_TAB)[tid] = t0;
    reinterpret_cast<uint4 *>(smem + CYCH_LDS_TREE_TAB)[512u + tid] = t1;
    reinterpret_cast<uint4 *>(smem + CYCH_LDS_TREE_TAB)[1024u + tid] = t2;
#pragma unroll
    for (u32 k = 0; k < 4; ++k) *reinterpret_cast<uint4 *>(smem + cyc_ltab_off(k * 512u + tid, CYCH_LDS_LTAB)) = l[k];
    *reinterpret_cast<uint4 *>(smem + cych_stage_off(0) + wv * 1024u + lane * 16u) = acc;
    __syncthreads();
    uint4 y = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (u32 level = 0; level < 3; ++level) {
        if (wv < (4u >> level)) {
            y = cych_tree_lane(smem, level, wv, lane);
            if (level < 2) *reinterpret_cast<uint4 *>(smem + cych_stage_off(level + 1) + wv * 1024u + lane * 16u) = y;
        }
        if (level < 2) __syncthreads();
    }
    if (wv != 0) return;
#if !AESGCM_BODY_WT
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
    G128 z = wave_xor(cyc_lane_term_lds(smem, y, lane, CYCH_LDS_LTAB));
    if (g + 1u != gridDim.x) {                                                // the weight through a two-table Shoup form in LDS
        if (lane < 32) *reinterpret_cast<uint4 *>(smem + CYCH_LDS_WTAB + 16u * lane) = shoup2_entry(mo_to_be(wc), lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        z = shoup2_gmul_lds(z, reinterpret_cast<const uint4 *>(smem + CYCH_LDS_WTAB));
    }
    if (g == 0) {                                                             // the terms that occur once
        G128 x; x.w[0] = x.w[1] = x.w[2] = x.w[3] = 0;
        if (p.tb) {
            // the partial last row wants the lanes' tables of H^(65 - L): they go where the workgroup's own were -- this wave is the only one left, and it is done with them
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (u32 k = 0; k < 32u; ++k) *reinterpret_cast<uint4 *>(smem + cyc_ltab_off(k * 64u + lane, CYCH_LDS_LTAB)) = cyc_ltab_entry(km, k * 64u + lane, 0u);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            x = cyc_lane_term_lds(smem, *reinterpret_cast<const uint4 *>(smem + CYCH_LDS_PARK + lane * 16u), lane, CYCH_LDS_LTAB);
        }
        if (lane == 0) {
            const uint4 ej0 = *reinterpret_cast<const uint4 *>(smem + CYCH_LDS_PARK + 1024u);
            G128 L; const u64 la = p.aad_len * 8, lc = p.ct_len * 8;
            L.w[0] = (u32)(la >> 32); L.w[1] = (u32)la; L.w[2] = (u32)(lc >> 32); L.w[3] = (u32)lc;
            L = shoup2_gmul_lds(L, km->ltab[1]);
            const G128 e = mo_to_be(ej0);
            x.w[0] ^= L.w[0] ^ e.w[0]; x.w[1] ^= L.w[1] ^ e.w[1]; x.w[2] ^= L.w[2] ^ e.w[2]; x.w[3] ^= L.w[3] ^ e.w[3];
        }
        x = wave_xor(x);
        z.w[0] ^= x.w[0]; z.w[1] ^= x.w[1]; z.w[2] ^= x.w[2]; z.w[3] ^= x.w[3];
    }
    if (lane != 0) return;
    acc_arrive(p.acc, g, z, p.tag_out, p.tag_host, p.gen);
    if (p.trace) {
        unsigned long long *tr = reinterpret_cast<unsigned long long *>(p.trace + 4 * (u64)g);
        const u64 rows_end = __hip_atomic_load(tr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u64 dt = wall_clock64() - rows_end;
        atomicOr(tr + 2, (unsigned long long)(dt > 0xFFFu ? 0xFFFu : dt) << 52);
    }
}

template <int NR, int MODE>
__global__ __launch_bounds__(AESGCM_BODYH_WG, 4) void k_bodyh(const KeyMaterial *__restrict__ km, const DevTables *__restrict__ tb, const BodyParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 tid = threadIdx.x, lane = tid & 63u;
    if (p.trace && tid == 0) {
        u64 *tr = p.trace + 4 * (u64)blockIdx.x;
        tr[0] = wall_clock64();
        tr[2] = (u64)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((u64)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF) << 32);
    }
    const u64 cyc0 = p.trace ? clock64() : 0;
    main_fill_lds(smem, km, tb, tid, true, AESGCM_BODYH_WG, GH_TAB_K2P17);    // T0 | T2 and the five-bit tables of the stride H^(2^17)
    __syncthreads();
    if (p.trace && tid == 0) {
        unsigned long long *tr = reinterpret_cast<unsigned long long *>(p.trace + 4 * (u64)blockIdx.x);
        const u64 dt = wall_clock64() - tr[0];
        atomicOr(tr + 2, (unsigned long long)(dt > 0xFFFu ? 0xFFFu : dt) << 40);
    }
    CtrConsts cc = ctr_round1_consts(p.iv0, p.iv1, p.iv2, km->rk, smem, (lane & 31u) << 2);
    cc.c0 = __builtin_amdgcn_readfirstlane(cc.c0); cc.c1 = __builtin_amdgcn_readfirstlane(cc.c1);
    cc.c2 = __builtin_amdgcn_readfirstlane(cc.c2); cc.c3 = __builtin_amdgcn_readfirstlane(cc.c3);
    const u32 w = __builtin_amdgcn_readfirstlane(blockIdx.x * (AESGCM_BODYH_WG / 64) + (tid >> 6));
    const uint4 acc = body_cyc_lane<NR, MODE, false, BODY_CYC_WAVES_HALF>(km, tb, p, smem, cc, w, lane);
    if (w == 0) {                                                             // a strand of the shorter kind: the partial last row and E_K(IV || 1), parked in LDS for the closing
        uint4 last = make_uint4(0, 0, 0, 0);
        if (p.tb) last = body_cyc_last_lane<NR, MODE>(km, p, smem, cc, lane);
        u32 s0, s1, s2, s3;
        ctr_rounds_lds<NR>(bswap32(1u), cc, s0, s1, s2, s3, km->rk, smem, (lane & 31u) << 2);
        *reinterpret_cast<uint4 *>(smem + CYCH_LDS_PARK + lane * 16u) = last;
        if (lane == 0) *reinterpret_cast<uint4 *>(smem + CYCH_LDS_PARK + 1024u) = make_uint4(s0, s1, s2, s3);
    }
    if (p.trace && lane == 0) {
        u64 *tr = p.trace + 4 * (u64)blockIdx.x;
        atomicMax((unsigned long long *)&tr[1], (unsigned long long)wall_clock64());
        atomicAdd((unsigned long long *)&tr[3], (unsigned long long)((p.F + p.R) / BODY_CYC_WAVES_HALF) | ((unsigned long long)((clock64() - cyc0) >> 10) << 32));
    }
    cyc_close_half(km, p, smem, acc);
}

// k_fold: up to FOLD_GROUP x FOLD_WAVES = 128 items per workgroup (lane bodies: fold_wave_lane(), fold_wg_lane()).  No static LDS: table offsets
// are absolute.
#ifndef FOLD_WPS
#define FOLD_WPS 2                       /* waves per SIMD the register budget is sized for (2 = one 8-wave workgroup per CU by registers) */
#endif
__global__ __launch_bounds__(FOLD_WG, FOLD_WPS) void k_fold(const KeyMaterial *__restrict__ km, const FoldParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    fold_fill_lds(smem, km, p.tabA, p.eA, 0u, tid, FOLD_WG);
    if (p.period > 1) fold_fill_lds(smem, km, p.tabB, p.eB, 8192u, tid, FOLD_WG);
    fold_fill_lds(smem, km, p.tabC, p.eC, 16384u, tid, FOLD_WG);
    if (p.close.on) for (u32 k = tid; k < 2048u; k += FOLD_WG) *reinterpret_cast<uint4 *>(smem + cyc_ltab_off(k, FOLD_LDS_LTAB)) = cyc_ltab_entry(km, k, 0u);   // the lanes' tables of H^(65 - L)
    __syncthreads();
    u32 start, end;
    const u32 J = fold_wg_range(p.n, p.group, blockIdx.x, &start, &end);
    if (w < J) *reinterpret_cast<uint4 *>(smem + FOLD_LDS_TAB + w * 1024u + lane * 16u) = fold_wave_lane(p, smem, start, end, J, w, lane);
    __syncthreads();
    if (w != 0) return;
    const uint4 item = fold_wg_lane(smem, J, lane);
    if (!p.close.on) { p.out[(size_t)blockIdx.x * 64 + lane] = item; return; }
    // closing (FoldClose): this workgroup's item ends step (G - 1 - g) blocks in front of the end of the message
    const u32 g = blockIdx.x;
    G128 z = wave_xor(cyc_lane_term_lds(smem, item, lane, FOLD_LDS_LTAB));
    const u64 e = p.close.step * (u64)(gridDim.x - 1u - g);
#pragma unroll 1
    for (u32 d = 0; d < 4; ++d) {
        const u32 dig = (u32)(e >> (AESGCM_LOG_WG * d)) & (u32)(AESGCM_WG - 1);
        if (!dig) continue;                                                   // wave-uniform
        if (lane < 32) *reinterpret_cast<uint4 *>(smem + FOLD_LDS_WTAB + 16u * lane) = shoup2_entry(mo_to_be(km->pw[d][dig]), lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        z = shoup2_gmul_lds(z, reinterpret_cast<const uint4 *>(smem + FOLD_LDS_WTAB));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                // the table is rewritten by the next digit
        __builtin_amdgcn_wave_barrier();
    }
    if (g + 1u == gridDim.x && lane == 0) {                                   // the terms that occur once: the length block times H, E_K(J0)
        G128 L; const u64 la = p.close.aad_len * 8, lc = p.close.ct_len * 8;
        L.w[0] = (u32)(la >> 32); L.w[1] = (u32)la; L.w[2] = (u32)(lc >> 32); L.w[3] = (u32)lc;
        L = shoup2_gmul_lds(L, km->ltab[1]);
        const G128 ej = mo_to_be(*p.close.ej0);
        z.w[0] ^= L.w[0] ^ ej.w[0]; z.w[1] ^= L.w[1] ^ ej.w[1]; z.w[2] ^= L.w[2] ^ ej.w[2]; z.w[3] ^= L.w[3] ^ ej.w[3];
    }
    if (lane == 0) acc_arrive(p.close.acc, g, z, p.close.tag_out, p.close.tag_host, p.close.gen);
}
// nibble tables of H^(2^k), k = 6..31, once per key (after k_setup)
__global__ __launch_bounds__(512) void k_setup_ptab(KeyMaterial *km) {
    if (blockIdx.x < AESGCM_NPTAB) setup_ptab_lane(km, blockIdx.x, threadIdx.x);
    else setup_ltab_lane(km, blockIdx.x - AESGCM_NPTAB, threadIdx.x);      // Shoup tables of H^e, e = 0 .. 65
}

// ------------------------------------------------------------------------------------------------
// k_combine: one workgroup, per message.
//   acc = sum_L item[L] * H^(63-L)  (the last k_fold item)  or  sum_g parts[g]  (gathered shard partials, already weighted)
//   POLY: out = acc * H^e                                (shard partial W_g, streaming state, aesgcm_ghash)
//   TAG : out = acc*H^2 ^ L*H ^ E_K(IV||1)               (gcm_ghash.vhd:257 length block, :293 tag)
//   a previous chaining value `carry` (streaming) enters as carry * H^(e_carry).
// ------------------------------------------------------------------------------------------------
// product of the four radix-512 digit entries of H^e, computed by lanes 0..3 of a wave + 2 tree levels
__device__ __forceinline__ G128 gf_pow_h(const KeyMaterial *km, u64 e, u32 lane) {
    G128 v = mo_to_be(gf_one_mo());
    if (lane < 4) v = gf_pow_h_digit(km, e, lane);
    for (int off = 1; off <= 2; off <<= 1) {
        G128 o;
        o.w[0] = __shfl_xor(v.w[0], off); o.w[1] = __shfl_xor(v.w[1], off);
        o.w[2] = __shfl_xor(v.w[2], off); o.w[3] = __shfl_xor(v.w[3], off);
        v = gf_mul(v, o);
    }
    return v;       // lanes 0..3 all hold the product
}

__device__ __forceinline__ void combine_body(const KeyMaterial *__restrict__ km, const DevTables *__restrict__ tb, const CombineParams &p, unsigned char *smem) {
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const bool tag = p.want_tag != 0, items = p.kind == PARTS_ITEM;
    const u32 J1 = items ? fold4_groups(p.np) : 0, J2 = items ? fold4_groups(J1) : 0;      // groups at level 1 (<= 16) and level 2 (<= 4)
    // ---- the item loads first (L2 round trips that depend on nothing), then the tables this launch needs
    CombineItems ci;
    ci.n = 0;
    if (w < J1) ci = combine_fold_load(p, w, lane);
    if (items && p.np > 1) for (u32 q = tid; q < 512; q += COMBINE_THREADS) reinterpret_cast<uint4 *>(smem + CMB_LDS_TABA)[q] = p.tabA[q];
    if (J1 > 1) for (u32 q = tid; q < 512; q += COMBINE_THREADS) reinterpret_cast<uint4 *>(smem + CMB_LDS_TABB)[q] = p.tabB[q];
    if (J2 > 1) for (u32 q = tid; q < 512; q += COMBINE_THREADS) reinterpret_cast<uint4 *>(smem + CMB_LDS_TABC)[q] = p.tabC[q];
    if (tid < 256) smem[CMB_LDS_SBOX + tid] = tb->sbox[tid];
    __syncthreads();
    // ---- chunk items: three Horner levels of fan-in 4
    if (w < J1) *reinterpret_cast<uint4 *>(smem + CMB_LDS_STAGE1 + w * 1024u + lane * 16u) = combine_fold_items(ci, smem, CMB_LDS_TABA);
    __syncthreads();
    if (w < J2) *reinterpret_cast<uint4 *>(smem + CMB_LDS_STAGE2 + w * 1024u + lane * 16u) = combine_fold_staged(smem, CMB_LDS_STAGE1, J1, w, CMB_LDS_TABB, lane);
    __syncthreads();
    // ---- every lane's term of the result.  TAG: B_L*H^(65-L), gathered W_g*H^2, L*H, E_K(J0), carry*H^2 -- all one
    // table multiply deep (km->ltab).  POLY: B_L*H^(63-L), W_g; the weighting by H^e and the carry follow below.
    G128 z; z.w[0] = z.w[1] = z.w[2] = z.w[3] = 0;
    if (items && w == 0) {
        const G128 b = mo_to_be(combine_fold_staged(smem, CMB_LDS_STAGE2, J2, 0, CMB_LDS_TABC, lane));
        z = shoup2_gmul(b, km->ltab[(tag ? 65u : 63u) - lane + (p.tail_item ? p.tail_blocks : 0u)]);
    } else if (items && w == 1 && p.tail_item) {                   // the partial last row of k_body's cyclic rows: it ends where the sequence ends
        z = shoup2_gmul(mo_to_be(p.tail_item[lane]), km->ltab[(tag ? 65u : 63u) - lane]);
    } else if (p.kind == PARTS_GATHERED && tid < p.np) {
        z = mo_to_be(p.parts[(size_t)tid * (p.stride ? p.stride : 1u)]);
        if (tag) z = shoup2_gmul(z, km->ltab[2]);
    } else if (tag && w == COMBINE_THREADS / 64 - 1) {             // the last wave carries the three single terms of a tag
        if (lane == 0) z = tag_len_term(km, p.aad_len, p.ct_len);
        else if (lane == 1) z = p.ej0 ? mo_to_be(*p.ej0) : combine_ej0_bytes(km, smem + CMB_LDS_SBOX, p);
        else if (lane == 2 && p.has_carry && !p.e_carry) z = shoup2_gmul(mo_to_be(*p.carry), km->ltab[2]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        z.w[0] ^= __shfl_xor(z.w[0], off); z.w[1] ^= __shfl_xor(z.w[1], off);
        z.w[2] ^= __shfl_xor(z.w[2], off); z.w[3] ^= __shfl_xor(z.w[3], off);
    }
    if (lane == 0) *reinterpret_cast<uint4 *>(smem + CMB_LDS_RED + 16u * w) = be_to_mo(z);
    __syncthreads();
    if (tid < 64) {
        uint4 r = *reinterpret_cast<const uint4 *>(smem + CMB_LDS_RED);
        for (u32 k = 1; k < COMBINE_THREADS / 64; k++) r = xor4(r, *reinterpret_cast<const uint4 *>(smem + CMB_LDS_RED + 16u * k));
        G128 acc = mo_to_be(r);                                   // TAG: the tag itself (unless a weighted carry is still due); POLY: the polynomial
        if (!tag && p.e) acc = gf_mul(acc, gf_pow_h(km, p.e, tid));
        if (p.has_carry && (!tag || p.e_carry)) {                 // bit-serial path: shard / streaming steps, off the one-shot latency path
            G128 c = mo_to_be(*p.carry);
            if (p.e_carry) c = gf_mul(c, gf_pow_h(km, p.e_carry, tid));
            if (tag) c = gf_mul(c, mo_to_be(km->pw[0][2]));
            acc.w[0] ^= c.w[0]; acc.w[1] ^= c.w[1]; acc.w[2] ^= c.w[2]; acc.w[3] ^= c.w[3];
        }
        if (tid == 0) {
            *p.out = be_to_mo(acc);
            if (p.out_host) publish_host(p.out_host, be_to_mo(acc), p.gen);
        }
    }
}

__global__ __launch_bounds__(COMBINE_THREADS) void k_combine(const KeyMaterial *__restrict__ km, const DevTables *__restrict__ tb, const CombineParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // no static LDS: table offsets are absolute (CMB_LDS_*)
    combine_body(km, tb, p, smem);
}
// several independent messages in ONE launch, one workgroup each (aesgcm_shard_finalize_batch_dev: the M tags of a multi-GPU step)
#define COMBINE_BATCH_MAX 8
struct CombineBatch { CombineParams p[COMBINE_BATCH_MAX]; };
__global__ __launch_bounds__(COMBINE_THREADS) void k_combine_batch(const KeyMaterial *__restrict__ km, const DevTables *__restrict__ tb, const CombineBatch b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    combine_body(km, tb, b.p[blockIdx.x], smem);
}

// the two-table Shoup form of a constant c at LDS offset `tab` (Th at tab, Tl = Th * x^4 at tab + 256), built by the 2^LG lanes that share it
template <int LG>
__device__ __forceinline__ void shoup2_build(unsigned char *smem, u32 tab, G128 c, u32 l) {
#pragma unroll
    for (u32 v = l; v < 16; v += (1u << LG)) {               // 16 entries per table, built by the group's own lanes
        const G128 e = shoup_entry(c, v), el = gf_mulx4(e);
        *reinterpret_cast<uint4 *>(smem + tab + 16 * v) = make_uint4(e.w[0], e.w[1], e.w[2], e.w[3]);
        *reinterpret_cast<uint4 *>(smem + tab + 256 + 16 * v) = make_uint4(el.w[0], el.w[1], el.w[2], el.w[3]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// ------------------------------------------------------------------------------------------------
// Lane-group helpers of the packet kernels (G = 2^LG lanes per packet, 64 / G packets per wave)
// ------------------------------------------------------------------------------------------------
// the largest value of a group-uniform quantity over the wave's 64 >> LG packet groups (wave-uniform result)
template <int LG>
__device__ __forceinline__ u32 groups_max(u32 v) {
    u32 m = 0;
#pragma unroll
    for (u32 g = 0; g < (64u >> LG); g++) { const u32 x = (u32)__builtin_amdgcn_readlane((int)v, (int)(g << LG)); m = x > m ? x : m; }
    return m;
}

// ------------------------------------------------------------------------------------------------
// k_batch3: BASELINE config 5 with G = 2^LG lanes per packet, i.e. 64 / G packets per wave (LG = 4: four packets of 16 lanes, LG = 3: eight of 8), ONE pass
// over the data.  What a packet pays besides its AES and one GHASH multiply per block -- key schedule, H and E_K(J0), the table of the Horner stride, the
// closing -- is paid per WAVE, so the more packets share a wave the cheaper it gets; each lane group has its own key, so round keys live in vector registers.
// Lane l of the packet's group owns slots l, l + G, ... of the right-aligned GHASH sequence in both roles: one loop does AES-CTR on the block and
// acc = acc * H^G ^ block (Shoup tables of the per-packet constant H^G, LG linear squarings of H).  The closing is k_pktg's: every lane times H^2, the
// length block into lane G-2, an LG-level tree with the group-uniform constants H, H^2, H^4, (H^8) -- q + LG + 1 table multiplies per wave-iteration, no
// ciphertext read-back, no fence.  Decrypt is the same pass (the lane reads its ciphertext block before it writes the plaintext: in place is safe).
// (Rounds 2 and 3 kept a two-phase predecessor, k_batch2 -- encrypt, fence, read the ciphertext back for GHASH: 1.57 x the algorithmic HBM traffic,
// profiles/r02g/cfg5_batch -- for A/B runs; round 4 deleted it.  HISTORY.md.)
// ------------------------------------------------------------------------------------------------
// Lanes per k_batch3 workgroup (one per CU).  The first round-3 build (1024 lanes, 128 registers, round keys in VGPRs) spilled 104 - 124 bytes around its
// packet loop and moved 11.1e9 bytes against 8.64e9 algorithmic; with 768-lane workgroups (160 registers, no scratch) 8.68e9 at the same speed -- the
// extra traffic was scratch (profiles/r03/batch3_wg768_ab.txt).  What was being spilled was bookkeeping, as in k_pktg: the packet's H and E_K(J0) held
// across the block loop (now in the group's LDS slot), the lane's position (lane_id_fresh behind the loop), ds_bpermute index registers (ds_swizzle).
// Without them every instance fits 98 - 115 registers at 1024 lanes with ScratchSize 0.  BATCH3_WG forces another geometry.
#ifdef BATCH3_WG
#define BATCH3_LANES(NR) BATCH3_WG
#else
#define BATCH3_LANES(NR) AESGCM_WG
#endif
// LG = 4: 16 lanes per packet, four packets per wave, two table slots per packet (the closing alternates between them).  LG = 3: 8 lanes per packet, eight
// packets per wave -- what a wave-iteration pays once (key schedule, H and E_K(J0), the H^8 table, the closing) now serves eight packets, and the tree is a
// level shorter; 128 packets per workgroup leave LDS for ONE table slot each, so the closing rebuilds that slot between its multiplies.
// Which lanes are a packet (round 4).  A ds_read_b128 is served in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the
// same + 32 (MI355X_MICROARCH.md, LDS) -- and a packet's Shoup table is a full 256-byte bank row, so lanes of DIFFERENT packets in one service group
// collide whenever they pick the same bank with different entries.  With packets on consecutive lanes a service group holds FOUR packets at 8 lanes per
// packet (lanes 0-3, 12-15, 20-23, 24-27) and two at 16: profiles/r03e/cfg5_n1 counted 9.1e8 conflict cycles in 2.83e9 LDS-array cycles, a table read
// at 10.3 cycles instead of 4.  BATCH3_PERM=1 makes the packets unions of service-group quads: at 16 lanes per packet a packet IS a service group (no
// collision possible), at 8 a service group holds two packets.  With b = the lane's bits: 16 lanes: grp = b5 | b2^b3^b4, l = b4 b3 b1 b0; 8 lanes:
// grp = b5 b4 | b2^b3, l = b3 b1 b0.  The tree partner l ^ 2^j is then lane ^ {1, 2, 12, 20}[j]: still a ds_swizzle, no index register.
#ifndef BATCH3_PERM
#define BATCH3_PERM 1
#endif
#ifndef BATCH3_DR
#define BATCH3_DR 1                        /* shoup2_mul_dr: the table multiply with its reduction delayed */
#endif
template <int LG>
__device__ __forceinline__ void batch3_pos(u32 lane, u32 &grp, u32 &l) {
    if (BATCH3_PERM && LG == 4) { grp = ((lane >> 4) & 2u) | (((lane >> 2) ^ (lane >> 3) ^ (lane >> 4)) & 1u); l = ((lane >> 1) & 12u) | (lane & 3u); }
    else if (BATCH3_PERM && LG == 3) { grp = ((lane >> 3) & 6u) | (((lane >> 2) ^ (lane >> 3)) & 1u); l = ((lane >> 1) & 4u) | (lane & 3u); }
    else { grp = lane >> LG; l = lane & ((1u << LG) - 1u); }
}
template <int LG>
__device__ __forceinline__ constexpr u32 batch3_first_lane(u32 g) {          // lane of position 0 of packet group g
    return (BATCH3_PERM && LG == 4) ? (g >> 1) * 32u + (g & 1u) * 4u : (BATCH3_PERM && LG == 3) ? (g >> 2) * 32u + ((g >> 1) & 1u) * 16u + (g & 1u) * 4u : g << LG;
}
template <int LG>
__device__ __forceinline__ u32 batch3_groups_max(u32 v) {                     // the largest value of a group-uniform quantity over the wave's packets
    u32 m = 0;
#pragma unroll
    for (u32 g = 0; g < (64u >> LG); g++) { const u32 x = (u32)__builtin_amdgcn_readlane((int)v, (int)batch3_first_lane<LG>(g)); m = x > m ? x : m; }
    return m;
}
template <int LG>
__device__ __forceinline__ u32 batch3_partner(u32 x, int j) {                 // the value of the lane whose position differs in bit j
    if (LG == 6) return lane_xor_pow2(x, j);                                   // a wave per packet: positions are the lanes
#if BATCH3_PERM
    switch (j) {
    case 0: return lane_xor<1>(x);
    case 1: return lane_xor<2>(x);
    case 2: return lane_xor<12>(x);
    default: return lane_xor<20>(x);
    }
#else
    return lane_xor_pow2(x, j);
#endif
}
#if BATCH3_DR
#define BATCH3_MUL shoup2_mul_dr
#else
#define BATCH3_MUL shoup2_mul
#endif
// 8 lanes per packet: a service group still holds TWO packets, and their table reads collide (23.8 % of the LDS-array cycles, profiles/r04/batch_ab.txt).
// BATCH3_PAIR=1 splits every multiply over the two lanes lane and lane ^ 20 of the two packets (the reference's split multiplier, src/gcm_ghash.vhd:317-333,
// over lanes instead of over two multiplier halves): in a first pass ALL sixteen lanes of the service group read the table of the packet with lane bit 4
// clear -- its own lanes for words 0, 1 of their accumulators, the partner lanes for words 2, 3 of the same accumulators -- in a second pass the other
// packet's.  Same 32 reads per lane, never two tables in one service group; the partials (6 words each way) cross by ds_swizzle.
#ifndef BATCH3_PAIR
#define BATCH3_PAIR 1
#endif
__device__ __forceinline__ G128 batch3_mul_pair(G128 y, const unsigned char *smem, u32 tab_mine, u32 tab_partner, bool first) {
    const u32 p2 = lane_xor<20>(y.w[2]), p3 = lane_xor<20>(y.w[3]);            // the partner's accumulator, words 2, 3
    u32 V1[6], V2[6];
    shoup2_half_dr(first ? y.w[0] : p2, first ? y.w[1] : p3, smem, first ? tab_mine : tab_partner, V1);     // pass 1: the table of the `first` packet
    shoup2_half_dr(first ? p2 : y.w[0], first ? p3 : y.w[1], smem, first ? tab_partner : tab_mine, V2);     // pass 2: the other packet's
    u32 Vo[6], Vh[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        Vo[j] = first ? V1[j] : V2[j];                                         // the pass in which this lane worked on its own accumulator
        Vh[j] = lane_xor<20>(first ? V2[j] : V1[j]);                           // what the partner computed for this lane's accumulator
    }
    return shoup2_pair_join(Vo, Vh);
}
template <int NR, int DEC, int LG>
__global__ __launch_bounds__(BATCH3_LANES(NR), (BATCH3_LANES(NR) + 255) / 256) void k_batch3(const DevTables *__restrict__ tb, const BatchParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr u32 G = 1u << LG, P = 64u >> LG;
    // LDS behind the T-tables: one (8 lanes per packet) or two 512-byte table slots per packet, 256-byte aligned (shoup2_mul_dr ORs the entry offset into
    // the slot address), then 32 bytes per packet for its H and E_K(J0)
    constexpr u32 GRP_TAB = BATCH3_GROUP_LDS_LG(LG) - 32u, WAVES = BATCH3_LANES(NR) / 64u, HSLOTS = BATCH3_LDS_TAB_OFF + WAVES * P * GRP_TAB;
    static_assert(BATCH3_LDS_TAB_OFF % 256u == 0 && GRP_TAB % 256u == 0, "k_batch3: table slots are 256-byte aligned");
    constexpr bool ONE_TAB = LG < 4;
    constexpr bool PAIR = BATCH3_PAIR && BATCH3_PERM && BATCH3_DR && LG == 3;
    const u32 tid = threadIdx.x, lane = tid & 63u;
    main_fill_lds(smem, nullptr, tb, tid, false, BATCH3_LANES(NR));
    __syncthreads();
    const u32 lb = (lane & 31u) << 2;
    const u32 wave_id = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));                                          // scalar
    const u32 wave_tab = BATCH3_LDS_TAB_OFF + wave_id * P * GRP_TAB, wave_hs = HSLOTS + wave_id * P * 32u;
    constexpr u32 KEYLEN = 4 * (NR - 6);
    const u32 K = p.deal, nb = (p.n_pkts + K - 1) / K;
    u32 pk0 = 0, pk_end = 0;
    for (u32 guard = 0; guard <= p.n_pkts; ++guard, pk0 += P) {      // bounded on purpose (as every dispenser loop)
        if (pk0 >= pk_end) {
            u32 b = 0;
            if (lane == 0) b = atomicAdd(p.counter, 1u) - p.counter_base;
            b = __builtin_amdgcn_readfirstlane(b);
            if (b >= nb) break;
            pk0 = b * K;
            pk_end = pk0 + K < p.n_pkts ? pk0 + K : p.n_pkts;
        }
        // the lane's position from a fresh lane id here and again behind the block loop (lane_id_fresh), so that none of it stays in a register across the loop
        u32 grp, l;
        batch3_pos<LG>(lane_id_fresh(), grp, l);
        const u32 tabA = wave_tab + grp * GRP_TAB, hsA = wave_hs + grp * 32u;
        const u32 tabAp = wave_tab + (grp ^ 3u) * GRP_TAB;             // PAIR: the table slot of the packet on lanes ^ 20
        const bool pair_first = (grp & 2u) == 0;                       // lane bit 4 clear
        const bool act = pk0 + grp < pk_end;                 // groups past the end shadow the first packet; their stores are masked
        const u32 pkt = batch_map(p, act ? pk0 + grp : pk0);
        const unsigned char *key = p.keys + (size_t)pkt * KEYLEN;
        const unsigned char *ivp = p.ivs + (size_t)pkt * 12;
        u32 pkt_len = p.pkt_len, aad_len = p.aad_len;
        u64 doff = (u64)pkt * p.pkt_len, aoff = (u64)pkt * p.aad_len;
        if (p.data_off) { doff = p.data_off[pkt]; pkt_len = (u32)(p.data_off[pkt + 1] - doff); }
        if (p.aad_off) { aoff = p.aad_off[pkt]; aad_len = (u32)(p.aad_off[pkt + 1] - aoff); }
        const bool aligned = p.aligned && ((doff & 15) == 0);
        const unsigned char *aad = p.aad ? p.aad + aoff : nullptr;
        const unsigned char *in = p.in + doff;
        unsigned char *out = p.out + doff;
        const u32 n_aad = (aad_len + 15) / 16, n_ct = (pkt_len + 15) / 16, n_seq = n_aad + n_ct;
        const u32 iters = batch3_groups_max<LG>((n_seq + G - 1) / G);          // the wave runs to its longest packet; shorter ones idle FIRST (front padding)
        const u32 pad = G * iters - n_seq;

        // ---- aes_kexp for this lane's packet (config/config_aes_kexp.py:128-159); every lane of a group computes the same words
        u32 rk[4 * (NR + 1)];
        batch_key_expand<NR>(key, rk, smem, lb);
        const u32 iv0 = load_le32(ivp), iv1 = load_le32(ivp + 4), iv2 = load_le32(ivp + 8);
        // ---- H = E_K(0^128) on lane 0 and E_K(IV || 1) on lane 1 of the group (gcm_gctr.vhd:141-145), one pass for both
        // Both go to the group's LDS slot (32 bytes behind its tables) and are read back where they are needed: H now and at the closing, E_K(J0) at the very
        // end -- held in registers across the block loop they were part of what the 128-register build spilled.
        {
            u32 s0 = (l == 0 ? 0u : iv0) ^ rk[0], s1 = (l == 0 ? 0u : iv1) ^ rk[1], s2 = (l == 0 ? 0u : iv2) ^ rk[2];
            u32 s3 = (l == 0 ? 0u : 0x01000000u) ^ rk[3];
            aes_rounds_lds<NR>(s0, s1, s2, s3, rk, smem, lb);
            const G128 e = mo_to_be(make_uint4(s0, s1, s2, s3));
            if (l < 2) *reinterpret_cast<uint4 *>(smem + hsA + 16u * l) = make_uint4(e.w[0], e.w[1], e.w[2], e.w[3]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        {
            const uint4 hv = *reinterpret_cast<const uint4 *>(smem + hsA);
            G128 h; h.w[0] = hv.x; h.w[1] = hv.y; h.w[2] = hv.z; h.w[3] = hv.w;
            G128 hs = gf_sqr(gf_sqr(gf_sqr(h)));                                       // Horner stride H^(lanes per packet): LG squarings (linear: gf_sqr, no table)
#pragma unroll
            for (int j = 3; j < LG; j++) hs = gf_sqr(hs);
            shoup2_build<LG>(smem, tabA, hs, l);
        }

        // ---- one pass: CTR on the lane's blocks and Horner over its slots
        G128 acc; acc.w[0] = acc.w[1] = acc.w[2] = acc.w[3] = 0;
        const CtrConsts cc = ctr_round1_consts(iv0, iv1, iv2, rk, smem, lb);
        // Records of one size that is a whole number of wave-iterations, no AAD, aligned (cfg5's shape): no padding slot, no AAD slot, no ragged block -- the
        // same work without the per-iteration tests and masks of the general loop below (launch-uniform: BatchParams::plain)
        if (p.plain) {
            const unsigned char *src = in + 16u * l;
            unsigned char *dst = out + 16u * l;
            for (u32 k = 0; k < iters; k++) {
                if (k) acc = PAIR ? batch3_mul_pair(acc, smem, tabA, tabAp, pair_first) : BATCH3_MUL(acc, smem, tabA);
                const uint4 x = gload16(src);
                u32 s0, s1, s2, s3;
                ctr_rounds_lds<NR>(bswap32(2u + k * G + l), cc, s0, s1, s2, s3, rk, smem, lb);
                const uint4 y = make_uint4(x.x ^ s0, x.y ^ s1, x.z ^ s2, x.w ^ s3);
                if (act) gstore16(dst, y);
                const G128 b = mo_to_be(DEC ? x : y);                // aes_gcm.vhd:207-211
                acc.w[0] ^= b.w[0]; acc.w[1] ^= b.w[1]; acc.w[2] ^= b.w[2]; acc.w[3] ^= b.w[3];
                src += 16u * G; dst += 16u * G;
            }
        } else
        for (u32 k = 0; k < iters; k++) {
            if (k) acc = PAIR ? batch3_mul_pair(acc, smem, tabA, tabAp, pair_first) : BATCH3_MUL(acc, smem, tabA);
            const u32 v = k * G + l;
            if (v < pad) continue;
            const u32 j = v - pad;
            uint4 gin;
            if (j < n_aad) {
                const u32 off = 16 * j, rem = aad_len - off;
                gin = rem >= 16 ? gload16_any(aad + off) : load_block_bytes(aad + off, rem);
            } else {
                const u32 i = j - n_aad, off = 16 * i, rem = pkt_len - off;
                const bool full = rem >= 16;                    // a whole block is one access at any address
                uint4 x;
                if (full) x = aligned ? gload16(in + off) : gload16_any(in + off);
                else x = load_block_bytes(in + off, rem < 16 ? rem : 16);
                u32 s0, s1, s2, s3;
                ctr_rounds_lds<NR>(bswap32(2u + i), cc, s0, s1, s2, s3, rk, smem, lb);
                uint4 y = make_uint4(x.x ^ s0, x.y ^ s1, x.z ^ s2, x.w ^ s3);
                if (rem < 16) y = mask_block(y, rem);
                if (act) {
                    if (full) { if (aligned) gstore16(out + off, y); else gstore16_any(out + off, y); }
                    else store_block_bytes(out + off, y, rem < 16 ? rem : 16);
                }
                gin = DEC ? x : y;                               // aes_gcm.vhd:207-211
            }
            const G128 b = mo_to_be(gin);
            acc.w[0] ^= b.w[0]; acc.w[1] ^= b.w[1]; acc.w[2] ^= b.w[2]; acc.w[3] ^= b.w[3];
        }

        // ---- closing: P = sum_l B_l H^(15-l);  tag = P H^2 ^ L H ^ E_K(J0)  (gcm_ghash.vhd:257,293), as in k_pktg
        u32 grp2, l2;
        batch3_pos<LG>(lane_id_fresh(), grp2, l2);
        const u32 tabA2 = wave_tab + grp2 * GRP_TAB, tabB2 = ONE_TAB ? tabA2 : tabA2 + 512u, hsA2 = wave_hs + grp2 * 32u;
        const bool act2 = pk0 + grp2 < pk_end;
        const u32 pkt2 = batch_map(p, act2 ? pk0 + grp2 : pk0);
        G128 h;
        { const uint4 hv = *reinterpret_cast<const uint4 *>(smem + hsA2); h.w[0] = hv.x; h.w[1] = hv.y; h.w[2] = hv.z; h.w[3] = hv.w; }
        G128 c = gf_sqr(h);                                     // H^2
        if (ONE_TAB) __builtin_amdgcn_wave_barrier();           // every lane is done with the Horner table
        shoup2_build<LG>(smem, tabB2, c, l2);
        const u32 tabAp2 = wave_tab + (grp2 ^ 3u) * GRP_TAB;
        const bool pair_first2 = (grp2 & 2u) == 0;
        acc = PAIR ? batch3_mul_pair(acc, smem, tabB2, tabAp2, pair_first2) : BATCH3_MUL(acc, smem, tabB2);
        if (l2 == G - 2u) { acc.w[1] ^= aad_len * 8u; acc.w[3] ^= pkt_len * 8u; }     // the length block: both < 2^32 bits by the ABI's limits
        if (ONE_TAB) __builtin_amdgcn_wave_barrier();
        shoup2_build<LG>(smem, tabA2, h, l2);                   // the Horner table is no longer needed
#pragma unroll
        for (int j = 0; j < LG; j++) {
            // level j: constant H^(2^j); two slots: H in tabA, H^2 in tabB, then H^4 -> tabA, H^8 -> tabB; one slot: each level rebuilds it (c = H^2 is still at hand for level 1)
            if (ONE_TAB) { if (j >= 1) { if (j >= 2) c = gf_sqr(c); __builtin_amdgcn_wave_barrier(); shoup2_build<LG>(smem, tabA2, c, l2); } }
            else if (j >= 2) { c = gf_sqr(c); shoup2_build<LG>(smem, (j & 1) ? tabB2 : tabA2, c, l2); }
            const G128 t = PAIR ? batch3_mul_pair(acc, smem, tabA2, tabAp2, pair_first2) : BATCH3_MUL(acc, smem, (j & 1) ? tabB2 : tabA2);
            G128 o;
            o.w[0] = batch3_partner<LG>(t.w[0], j); o.w[1] = batch3_partner<LG>(t.w[1], j);
            o.w[2] = batch3_partner<LG>(t.w[2], j); o.w[3] = batch3_partner<LG>(t.w[3], j);
            if (l2 & (1u << j)) { acc.w[0] ^= o.w[0]; acc.w[1] ^= o.w[1]; acc.w[2] ^= o.w[2]; acc.w[3] ^= o.w[3]; }
        }
        { const uint4 ev = *reinterpret_cast<const uint4 *>(smem + hsA2 + 16u); acc.w[0] ^= ev.x; acc.w[1] ^= ev.y; acc.w[2] ^= ev.z; acc.w[3] ^= ev.w; }
        if (l2 == G - 1u && act2) {
            const uint4 tag = be_to_mo(acc);
            store_block_bytes(p.tags + (size_t)pkt2 * 16, tag, 16);
            if (DEC && p.auth) {
                int ok = 1;
                if (p.expect) {
                    const uint4 e = load_block_bytes(p.expect + (size_t)pkt2 * 16, 16);
                    ok = ((e.x ^ tag.x) | (e.y ^ tag.y) | (e.z ^ tag.z) | (e.w ^ tag.w)) == 0;
                }
                p.auth[pkt2] = ok;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_len_hist / k_len_scan / k_len_scatter: the order in which a launch takes packets of mixed length (aesgcm_dev.h, pkt_len_class): a counting sort of the
// packet numbers by falling length class.  LEN_SORT_WGS workgroups own a slice of the packets each; counts[(255 - class) * LEN_SORT_WGS + workgroup] holds a
// workgroup's count of a class, then (after the scan over that array in its own order) the position of its first packet of that class.  Atomics only in LDS:
// a first form with one global cursor per class spent milliseconds on 2^20 atomics to two dozen addresses (call 25).
// ------------------------------------------------------------------------------------------------
#define LEN_SORT_WGS 256u
#define LEN_SORT_ENTRIES (PKT_LEN_CLASSES * LEN_SORT_WGS)
__device__ __forceinline__ void len_sort_slice(u32 n, u32 &lo, u32 &hi) {
    const u32 per = (n + LEN_SORT_WGS - 1u) / LEN_SORT_WGS;
    lo = blockIdx.x * per < n ? blockIdx.x * per : n;
    hi = lo + per < n ? lo + per : n;
}
__global__ __launch_bounds__(256) void k_len_hist(const u64 *__restrict__ off, u32 n, u32 *__restrict__ counts) {
    __shared__ u32 h[PKT_LEN_CLASSES];
    h[threadIdx.x] = 0;
    __syncthreads();
    u32 lo, hi;
    len_sort_slice(n, lo, hi);
    for (u32 i = lo + threadIdx.x; i < hi; i += 256u) atomicAdd(&h[pkt_len_class(off[i + 1] - off[i])], 1u);
    __syncthreads();
    counts[(PKT_LEN_CLASSES - 1u - threadIdx.x) * LEN_SORT_WGS + blockIdx.x] = h[threadIdx.x];
}
__global__ __launch_bounds__(1024) void k_len_scan(u32 *__restrict__ counts) {         // exclusive prefix sums over the 65536 entries, in place; one workgroup
    __shared__ u32 part[1024];
    constexpr u32 PER = LEN_SORT_ENTRIES / 1024u;
    u32 *mine = counts + threadIdx.x * PER;
    u32 s = 0;
    for (u32 k = 0; k < PER; ++k) s += mine[k];
    part[threadIdx.x] = s;
    __syncthreads();
    for (u32 d = 1; d < 1024u; d <<= 1) {                                              // Hillis-Steele over the 1024 partial sums
        const u32 v = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    u32 run = part[threadIdx.x] - s;
    for (u32 k = 0; k < PER; ++k) { const u32 c = mine[k]; mine[k] = run; run += c; }
}
__global__ __launch_bounds__(256) void k_len_scatter(const u64 *__restrict__ off, u32 n, const u32 *__restrict__ base, u32 *__restrict__ perm) {
    __shared__ u32 cur[PKT_LEN_CLASSES];
    cur[threadIdx.x] = base[(PKT_LEN_CLASSES - 1u - threadIdx.x) * LEN_SORT_WGS + blockIdx.x];
    __syncthreads();
    u32 lo, hi;
    len_sort_slice(n, lo, hi);
    for (u32 i = lo + threadIdx.x; i < hi; i += 256u) perm[atomicAdd(&cur[pkt_len_class(off[i + 1] - off[i])], 1u)] = i;
}

// ------------------------------------------------------------------------------------------------
// k_pktg: many packets under the context's key, 2^LG lanes per packet (lane bodies: pktg_lane(), pktg_close_lane(),
// pktg_tree_offer(); see "Packets under ONE key" in aesgcm_dev.h).  One 1024-lane workgroup per CU.
// ------------------------------------------------------------------------------------------------
#define PKTG_WAVE_SLOT 1280u                                                                                /* per wave: 64 E_K(J0) values and the 64 packet numbers of its dispenser block */
#define PKTG_LDS_TOTAL(LG) (PKTG_LDS_BYTES(LG) + ((LG) <= 4 ? (u32)(PKTG_WG(LG) / 64) * PKTG_WAVE_SLOT : 0u))
// the cross-lane tree of a packet's group, level J .. LG-1 (compile-time recursion: lane_xor needs its mask as a constant)
template <int LG, int J>
__device__ __forceinline__ void pktg_tree(uint4 &acc, const unsigned char *smem, u32 l) {
    if constexpr (J < LG) {
        const uint4 o = pktg_tree_offer(acc, smem, J);
        const u32 ox = lane_xor<(1 << J)>(o.x), oy = lane_xor<(1 << J)>(o.y), oz = lane_xor<(1 << J)>(o.z), ow = lane_xor<(1 << J)>(o.w);
        if (l & (1u << J)) { acc.x ^= ox; acc.y ^= oy; acc.z ^= oz; acc.w ^= ow; }
        pktg_tree<LG, J + 1>(acc, smem, l);
    }
}
// Lanes per k_pktg workgroup (one workgroup per CU).  The first round-3 build (1024 lanes, 128 registers) spilled 68 - 88 bytes around its packet
// loop, and that scratch is what its extra HBM traffic was: 2^20 x 1 KiB at 16 lanes per packet read 1.658e9 bytes against 1.086e9 algorithmic, all
// 128-byte requests; with 768-lane workgroups (148 - 165 registers, no scratch) 1.104e9 (profiles/r03/pktg_wg768_ab.txt).  What was being spilled was
// bookkeeping, and it is gone at 1024 lanes too: the index registers of ds_bpermute exchanges (now ds_swizzle, lane_xor), the 64 E_K(J0) values held
// across the packet loop (now in the wave's LDS slot) and the lane's position (recomputed from lane_id_fresh after the loop).  Lane groups therefore
// run 1024-lane workgroups again (4 waves per SIMD: 16 lanes per packet 517 -> 530, 681 -> 700 GiB/s at 1 / 4 KiB against 768 lanes); one packet per
// wave keeps its E_K(J0) values in registers and stays at 768 lanes, where it needs no scratch.  AESGCM_PKTG_WG forces one geometry for all.
#ifdef AESGCM_PKTG_WG
#define PKTG_WG(LG) AESGCM_PKTG_WG
#else
#define PKTG_WG(LG) ((LG) == 6 ? 768 : AESGCM_PKT_WG)
#endif
template <int NR, int DEC, int LG>
__global__ __launch_bounds__(PKTG_WG(LG), (PKTG_WG(LG) + 255) / 256) void k_pktg(const KeyMaterial *__restrict__ km, const DevTables *__restrict__ tb, const PktParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr u32 G = 1u << LG, P = 64u >> LG;
    const u32 tid = threadIdx.x, lane = tid & 63u;
    pktg_fill_lds(smem, km, tb, tid, PKTG_WG(LG), LG);
    __syncthreads();
    const u32 wave_slot = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6)) * PKTG_WAVE_SLOT;       // scalar
    // packets are dealt to the waves in blocks of p.deal (a multiple of P, at most 64) from a dispenser: one atomic per block
    // keeps the single dispenser address far below its ~87 M fetches/s ceiling (measured), and late waves still level the
    // tail.  The loop is bounded on purpose (a wave can never own more than nb blocks).
    const u32 K = p.deal, nb = (p.n_pkts + K - 1) / K;
    for (u32 guard = 0; guard <= nb; ++guard) {
        u32 b = 0;
        if (lane == 0) b = atomicAdd(p.counter, 1u) - p.counter_base;
        b = __builtin_amdgcn_readfirstlane(b);
        if (b >= nb) break;
        const u32 p0 = b * K, cnt = (p0 + K < p.n_pkts ? p0 + K : p.n_pkts) - p0;
        // E_K(IV || 1) of the block's packets, one lane each, in ONE AES pass.  Lane groups park the 64 values in the wave's own 1 KiB of LDS
        // (behind the tree tables): held in registers across the packet loop they were spilled at 128 registers; one packet per wave keeps them.
        constexpr bool EJ_LDS = LG <= 4;
        unsigned char *ej_slot = smem + PKTG_LDS_BYTES(LG) + wave_slot;
        // the block's packet numbers (pkt_map: the launch order of packets of mixed length) beside them, read back per group: the pointer chase stays out of the packet loop
        const u32 mine = pkt_map(p, p0 + (lane < cnt ? lane : 0u));
        uint4 ej = pktg_ej0_lane<NR>(km, p, smem, mine, lane);
        if (EJ_LDS) {                                           // addresses from a fresh lane id: hoisted out of the dispenser loop they were two more registers held across it (spilled at 128)
            const u32 lf = lane_id_fresh();
            *reinterpret_cast<uint4 *>(ej_slot + lf * 16u) = ej; ej = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<u32 *>(ej_slot + 1024u + lf * 4u) = mine;
        }
        for (u32 t = 0; t * P < cnt; ++t) {
            const u32 lane1 = lane_id_fresh(), l = lane1 & (G - 1u), idx = t * P + (lane1 >> LG);
            const bool act = idx < cnt;                          // groups past the end shadow the block's first packet; their stores are masked
            const u32 pkt = EJ_LDS ? *reinterpret_cast<const u32 *>(ej_slot + 1024u + (act ? idx : 0u) * 4u) : pkt_map(p, p0 + (act ? idx : 0u));
            const PktInfo q = pkt_info(p, pkt);
            // the wave runs to the longest packet of its groups
            u32 iters = pktg_iters(q, G);
            iters = groups_max<LG>(iters);                      // wave-uniform: the first lane of every group, through scalar registers
            uint4 acc = pktg_lane<NR, DEC, LG>(km, p, q, smem, l, lane1, iters, act);
            // what follows needs the lane's position again.  Taken from a FRESH lane id (lane_id_fresh: the compiler cannot tie it to the one above), so that
            // l / grp / idx / pkt need not stay in registers across the packet loop -- at 128 registers they were spilled there
            const u32 lane2 = lane_id_fresh(), l2 = lane2 & (G - 1u), idx2 = t * P + (lane2 >> LG);
            const bool act2 = idx2 < cnt;
            const u32 pkt2 = EJ_LDS ? *reinterpret_cast<const u32 *>(ej_slot + 1024u + (act2 ? idx2 : 0u) * 4u) : pkt_map(p, p0 + (act2 ? idx2 : 0u));
            acc = pktg_close_lane<LG>(acc, q, smem, l2);
            pktg_tree<LG, 0>(acc, smem, l2);
            // lane G-1 of the group holds P H^2 ^ L H; its packet's E_K(IV || 1) sits in lane idx of `ej`
            const int srcl = (int)(act2 ? idx2 : 0u);
            const uint4 e = EJ_LDS ? *reinterpret_cast<const uint4 *>(ej_slot + (u32)srcl * 16u)
                                   : make_uint4((u32)__shfl((int)ej.x, srcl), (u32)__shfl((int)ej.y, srcl), (u32)__shfl((int)ej.z, srcl), (u32)__shfl((int)ej.w, srcl));
            if (l2 == G - 1u && act2) {
                const uint4 tag = xor4(acc, e);                  // gcm_ghash.vhd:293
                store_block_bytes(p.tags + (size_t)pkt2 * 16, tag, 16);
                if (DEC && p.auth) {
                    int ok = 1;
                    if (p.expect) {
                        const uint4 x = load_block_bytes(p.expect + (size_t)pkt2 * 16, 16);
                        ok = ((x.x ^ tag.x) | (x.y ^ tag.y) | (x.z ^ tag.z) | (x.w ^ tag.w)) == 0;
                    }
                    p.auth[pkt2] = ok;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_pktl: many packets under the context's key, one LANE per packet (lane body: pktl_lane()); waves take
// blocks of 64 consecutive packets from the dispenser.
// ------------------------------------------------------------------------------------------------
#ifndef AESGCM_PKTL_WG
#define AESGCM_PKTL_WG 768            // lanes per k_pktl workgroup: 3 waves per SIMD = 168 registers, what eight held blocks beside the table multiply need
#endif
#define AESGCM_PKTL_LDS (AESGCM_PKTL_T4 ? AESGCM_LDS_BYTES_T4 : AESGCM_LDS_BYTES)
#ifndef AESGCM_PKTL_WAVES
#define AESGCM_PKTL_WAVES ((AESGCM_PKTL_WG + 255) / 256)          // waves per SIMD the register budget is sized for (one workgroup per CU)
#endif
// ILP = 1: the same lane code compiled for 512-lane workgroups (two waves per SIMD, 256 registers) with the eight keystream blocks of a line as independent
// chains: for batches that do not fill the chip, where a wave has to hide its own LDS latency (pktl_lane, AESGCM_PKTL_WG_ILP).
#define AESGCM_PKTL_WG_ILP 512
template <int NR, int DEC, int ILP>
__global__ __launch_bounds__(ILP ? AESGCM_PKTL_WG_ILP : AESGCM_PKTL_WG, ILP ? 2 : AESGCM_PKTL_WAVES) void k_pktl(const KeyMaterial *__restrict__ km, const DevTables *__restrict__ tb, const PktParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr u32 WG = ILP ? AESGCM_PKTL_WG_ILP : AESGCM_PKTL_WG;
    const u32 tid = threadIdx.x, lane = tid & 63u;
    main_fill_lds(smem, km, tb, tid, true, WG, GH_TAB_H);
#if AESGCM_PKTL_T4
    fill_lds_t4(smem, tb, tid, WG);
#endif
    __syncthreads();
    const u32 nb = (p.n_pkts + 63u) / 64u;
    for (u32 guard = 0; guard <= nb; ++guard) {                // bounded, as every dispenser loop here
        u32 b = 0;
        if (lane == 0) b = atomicAdd(p.counter, 1u) - p.counter_base;
        b = __builtin_amdgcn_readfirstlane(b);
        if (b >= nb) break;
        const u32 idx = b * 64u + lane;
        if (idx < p.n_pkts) pktl_lane<NR, DEC, AESGCM_PKTL_T4 != 0, ILP != 0>(km, p, smem, pkt_map(p, idx), lane);
    }
}

// ------------------------------------------------------------------------------------------------
// k_rows / k_rows_close (+ k_rows_plan with offset arrays): MANY messages under the context's key through k_body's row code; the algebra, the cut into blocks and
// pieces and the lane code are in aesgcm_rows.h.  k_rows has k_body's LDS image and row loop (body_strand_rows); a wave takes a block of the call's unit axis --
// its own (one block per wave) or the next from the dispensers -- and walks the pieces in it; what it leaves per piece is a 32-byte record.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 opaque_sgpr(u32 x) { u32 r; asm volatile("s_mov_b32 %0, %1" : "=s"(r) : "s"(x)); return r; }   // the same value, of unknown origin to the compiler
template <int NR, int MODE>
__global__ __launch_bounds__(AESGCM_BODY_WG, AESGCM_BODY_WPS) void k_rows(const KeyMaterial *__restrict__ km, const DevTables *__restrict__ tb, const RowsParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 tid = threadIdx.x, lane = tid & 63u;
    main_fill_lds(smem, km, tb, tid, true, AESGCM_BODY_WG, GH_TAB_K64);            // consecutive rows: Horner stride H^64
#if AESGCM_T4
    fill_lds_t4(smem, tb, tid, AESGCM_BODY_WG);
#endif
    if (tid == 0) *reinterpret_cast<u32 *>(smem + AESGCM_LDS_DRY_OFF) = 0;   // dry-queue mask of the workgroup (next_chunk)
    __syncthreads();
    const u64 G = uniform64(p.hdr ? p.hdr->G : p.G);
    const u32 D = uniform32(p.hdr ? p.hdr->D : p.D), NB = uniform32(p.hdr ? p.hdr->NB : p.NB), dyn = uniform32(p.hdr ? p.hdr->dyn : p.dyn);
    const u32 wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (AESGCM_BODY_WG / 64) + (tid >> 6));
    u32 nq, seg;
    plan_queues(NB, &nq, &seg);
    u32 q = __builtin_amdgcn_readfirstlane(wave % nq);
    for (u32 guard = 0; guard <= NB; ++guard) {                              // bounded, as every dispenser loop here
        u32 b;
        if (dyn) { b = next_chunk(p.queues, smem, nq, seg, NB, q, lane); if (b == DISPENSER_DONE) break; }
        else { if (guard || wave >= NB) break; b = wave; }
        u64 g = (u64)b * D;
        const u64 g_end = g + D < G ? g + D : G;
        u32 m = uniform32(rows_find_msg(p, g));
        // The pieces of the block.  Every iteration derives what it needs from (g, m) alone -- lengths, offsets, the IV's constants are loaded again per piece, and
        // the record's header is written BEFORE the rows -- so that almost nothing but g, m and the record's address is live across the row loop (an earlier form
        // that carried the message's geometry through it spilled 114 scalars and 252 bytes of scratch).
        for (u32 guard2 = 0; g < g_end && guard2 <= 2u * D + 4u; ++guard2) {
            m = opaque_sgpr(m);
            RowsMsg mq = rows_msg(p, m);
            mq.doff = uniform64(mq.doff); mq.aoff = uniform64(mq.aoff); mq.len = uniform32(mq.len); mq.alen = uniform32(mq.alen);
            const RowsGeom geo = rows_geom(mq.len);
            const u64 g0 = uniform64(rows_unit_base(p, m));
            if (g >= g0 + rows_units(geo, p.has_aad)) { ++m; continue; }     // the next message (every message has at least its tail unit)
            const RowsPiece pc = rows_piece(geo, uniform32(rows_slot_base(p, m)), g0, (u32)(g - g0), g_end - g, D);
            RowsRec *rr = p.rec + pc.slot;
            if (lane_id_fresh() == 0) {
                rr->e = pc.e; rr->msg = m; rr->flags = pc.kind == ROWS_TAIL ? ROWS_REC_VALID : (ROWS_REC_VALID | ROWS_REC_WEIGH);    // (the tail is weighted already: lane terms H^(64 - L))
                atomicAdd(p.npieces + m, 1u);                                 // what k_rows_close waits for
            }
            // every kind of piece takes the lane's index FRESH (lane_id_fresh: opaque to the compiler), so that nothing lane-dependent of the tail and AAD code -- table
            // addresses, byte masks -- is hoisted out of the piece loop and kept in registers across the row loop (first build: 26 scratch accesses per row)
            G128 z;
            if (pc.kind == ROWS_AAD) {
                z = wave_xor(rows_aad_lane(km, p, mq, smem, lane_id_fresh()));
            } else {
                const unsigned char *ivp = p.ivs + (size_t)m * 12;
                CtrConsts cc = ctr_round1_consts(uniform32(load_le32(ivp)), uniform32(load_le32(ivp + 4)), uniform32(load_le32(ivp + 8)), km->rk, smem, (lane_id_fresh() & 31u) << 2);   // key and IV only: wave-uniform
                cc.c0 = __builtin_amdgcn_readfirstlane(cc.c0); cc.c1 = __builtin_amdgcn_readfirstlane(cc.c1);
                cc.c2 = __builtin_amdgcn_readfirstlane(cc.c2); cc.c3 = __builtin_amdgcn_readfirstlane(cc.c3);
                if (pc.kind == ROWS_RUN) {
                    const uint4 acc = rows_run_lane<NR, MODE>(km, tb, p, mq, pc, smem, cc, lane_id_fresh(), dyn ? 0u : p.prio_rows, (tid >> 8) & 3u);
                    z = wave_xor(rows_run_term(km, acc, lane_id_fresh()));
                } else {
                    uint4 ej0;
                    z = wave_xor(rows_tail_lane<NR, MODE == MODE_DEC>(km, p, mq, smem, cc, lane_id_fresh(), &ej0));
                    const G128 e = mo_to_be(make_uint4((u32)__builtin_amdgcn_readlane((int)ej0.x, 63), (u32)__builtin_amdgcn_readlane((int)ej0.y, 63),
                                                       (u32)__builtin_amdgcn_readlane((int)ej0.z, 63), (u32)__builtin_amdgcn_readlane((int)ej0.w, 63)));
                    z.w[0] ^= e.w[0]; z.w[1] ^= e.w[1]; z.w[2] ^= e.w[2]; z.w[3] ^= e.w[3];
                }
            }
            if (lane_id_fresh() == 0) rr->w = z;
            g += pc.len;
        }
    }
}

// a lane per record slot: the record's contribution to its message's tag and its arrival; the lane that counts the message's last piece holds the tag: it stores
// it and (decrypt) compares.  Memory-side atomics only, as acc_arrive: the XORs have returned before the arrival is counted.  Zero at rest: the lane puts its
// record's flags back to zero, the closing lane the message's accumulator and counts, workgroup 0 the dispensers.
template <int DEC>
__global__ __launch_bounds__(ROWS_CLOSE_WG) void k_rows_close(const KeyMaterial *__restrict__ km, const RowsParams p) {
    if (blockIdx.x == 0 && threadIdx.x < ROWS_NQ) p.queues[16u * threadIdx.x] = 0;
    const u32 slot = blockIdx.x * ROWS_CLOSE_WG + threadIdx.x;
    if (slot >= p.slot_cap) return;
    const RowsRec r = p.rec[slot];
    if (!(r.flags & ROWS_REC_VALID)) return;
    p.rec[slot].flags = 0;
    const G128 z = rows_weigh(km, r);
    const u32 m = r.msg, expected = p.npieces[m];
    const unsigned long long ohi = atomicXor(p.acc + 2u * m, ((unsigned long long)z.w[0] << 32) | z.w[1]);
    const unsigned long long olo = atomicXor(p.acc + 2u * m + 1u, ((unsigned long long)z.w[2] << 32) | z.w[3]);
    u32 dep;
    asm volatile("v_and_b32 %0, 0, %1" : "=v"(dep) : "v"((u32)(ohi ^ olo) | (u32)((ohi ^ olo) >> 32)));
    const u32 arrived = atomicAdd(p.cnt + m, 1u + dep);
    if (arrived + 1u != expected) return;
    const unsigned long long hi = atomicExch(p.acc + 2u * m, 0ull), lo = atomicExch(p.acc + 2u * m + 1u, 0ull);
    p.cnt[m] = 0; p.npieces[m] = 0;
    G128 t; t.w[0] = (u32)(hi >> 32); t.w[1] = (u32)hi; t.w[2] = (u32)(lo >> 32); t.w[3] = (u32)lo;
    const uint4 tag = be_to_mo(t);
    store_block_bytes(p.tags + (size_t)m * 16, tag, 16);
    if (DEC && p.auth) {
        int ok = 1;
        if (p.expect) {
            const uint4 x = load_block_bytes(p.expect + (size_t)m * 16, 16);
            ok = ((x.x ^ tag.x) | (x.y ^ tag.y) | (x.z ^ tag.z) | (x.w ^ tag.w)) == 0;
        }
        p.auth[m] = ok;
    }
}

// The cut of a call with offset arrays, on the device (the host does not know the lengths): ONE workgroup.  Units per message -> prefix[0 .. n] and G; the cut
// (rows_cut); record slots per message -> slot_base[0 .. n]; the header.
__device__ __forceinline__ u64 block_scan_u64(unsigned long long *part, u64 mine, u32 tid) {      // exclusive prefix of `mine` over the 1024 threads; part[1023] = the total afterwards
    part[tid] = mine;
    __syncthreads();
    for (u32 d = 1; d < 1024u; d <<= 1) {                                    // Hillis-Steele
        const u64 v = tid >= d ? part[tid - d] : 0ull;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    return part[tid] - mine;
}
__global__ __launch_bounds__(1024) void k_rows_plan(const u64 *__restrict__ off, u32 n, u32 has_aad, u32 waves, u32 force_d, u32 nb_cap, u32 slot_cap,
                                                    RowsHdr *hdr, u64 *prefix, u32 *slot_base) {
    __shared__ unsigned long long part[1024];
    const u32 tid = threadIdx.x, per = (n + 1023u) / 1024u;
    const u32 lo = tid * per < n ? tid * per : n, hi = lo + per < n ? lo + per : n;
    u64 s = 0;
    for (u32 m = lo; m < hi; ++m) s += rows_units(rows_geom(off[m + 1] - off[m]), has_aad);
    u64 run = block_scan_u64(part, s, tid);
    const u64 G = part[1023];
    __syncthreads();
    u32 D, NB, dyn;
    rows_cut(G, waves, force_d, nb_cap, &D, &NB, &dyn);
    u64 t = 0;
    for (u32 m = lo; m < hi; ++m) {
        const RowsGeom g = rows_geom(off[m + 1] - off[m]);
        prefix[m] = run;
        t += rows_slots(g, has_aad, run, D);
        run += rows_units(g, has_aad);
    }
    u64 srun = block_scan_u64(part, t, tid);
    const u64 slots = part[1023];
    run = prefix[lo < n ? lo : 0];
    for (u32 m = lo; m < hi; ++m) {
        const RowsGeom g = rows_geom(off[m + 1] - off[m]);
        slot_base[m] = (u32)srun;
        srun += rows_slots(g, has_aad, run, D);
        run += rows_units(g, has_aad);
    }
    if (tid == 0) {
        prefix[n] = G; slot_base[n] = (u32)slots;
        hdr->G = slots <= slot_cap ? G : 0ull;                               // (the host sizes the scratch for the worst case; a cut that does not fit would be its bug: then nothing runs)
        hdr->D = D; hdr->NB = slots <= slot_cap ? NB : 0u; hdr->dyn = dyn;
    }
}

// k_wipe_failed: a wave per packet; packets whose auth[] says 0 get their output bytes zeroed (context option "wipe_on_auth_fail", aesgcm_wipe_failed_dev)
__global__ __launch_bounds__(256) void k_wipe_failed(unsigned char *out, const int *auth, const u64 *data_off, u32 n_pkts, u32 pkt_len) {
    const u32 pkt = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (pkt >= n_pkts || auth[pkt]) return;
    const u64 lo = data_off ? data_off[pkt] : (u64)pkt * pkt_len, hi = data_off ? data_off[pkt + 1] : lo + pkt_len;
    unsigned char *p = out + lo;
    const u64 len = hi - lo, head = len < 16 ? len : ((16u - ((uintptr_t)p & 15u)) & 15u);
    if (lane < head) p[lane] = 0;
    const u64 nvec = (len - head) / 16;
    for (u64 i = lane; i < nvec; i += 64) gstore16(p + head + 16 * i, make_uint4(0, 0, 0, 0));
    const u64 done = head + 16 * nvec;
    if (done + lane < len) p[done + lane] = 0;
}

// ================================================================================================
// host side
// ================================================================================================
static thread_local char g_err[256] = "";
static int hip_fail(hipError_t e, const char *what) {
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    return AESGCM_EHIP;
}
#define HIPCHK(call) do { hipError_t _e = (call); if (_e != hipSuccess) return hip_fail(_e, #call); } while (0)

#define BATCH_DISPENSERS 256
// The launch order of packets of mixed length (k_len_*): scratch of one launch.  `done` is recorded behind the packet kernel that reads the order, and the next
// user of the slot makes its stream wait for it: slots may be reused by launches on other streams at any rate.
struct OrderSlot { u32 *perm = nullptr; size_t cap = 0; u32 *bins = nullptr; hipEvent_t done = nullptr; };
// `streams`: the streams of destroyed contexts, for the next context of the device -- hipStreamCreate takes 2 ms and hipStreamDestroy half a millisecond on this
// runtime (profiles/microbench/runtime_costs.cpp), more than everything else a context costs together (k_setup: 0.4 ms).
struct DeviceState { DevTables *tables = nullptr; int n_cu = 0; bool attrs = false; u32 *batch_counter = nullptr; u32 batch_slot = 0; OrderSlot order[4]; unsigned order_next = 0;
                     std::vector<hipStream_t> streams; };   // ring of dispensers: concurrent batch launches never share one
static std::mutex &g_mu = *new std::mutex();
static std::vector<DeviceState> &g_dev = *new std::vector<DeviceState>();     // never destroyed (as g_ctxs): contexts may outlive this library's static destructors

static int device_state(int device, DeviceState **out) {
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) { snprintf(g_err, sizeof g_err, "device %d out of range (%d visible)", device, n); return AESGCM_EHIP; }
    if ((int)g_dev.size() < n) g_dev.resize(n);
    DeviceState &d = g_dev[device];
    if (!d.tables) {
        HIPCHK(hipSetDevice(device));
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, device));
        d.n_cu = prop.multiProcessorCount;
        DevTables *t = nullptr;
        HIPCHK(hipMalloc(&t, sizeof(DevTables)));
        hipLaunchKernelGGL(k_init_tables, dim3(1), dim3(256), 0, 0, t);
        HIPCHK(hipGetLastError());
        HIPCHK(hipDeviceSynchronize());
        d.tables = t;
        HIPCHK(hipMalloc(&d.batch_counter, 4 * BATCH_DISPENSERS));
        HIPCHK(hipMemset(d.batch_counter, 0, 4 * BATCH_DISPENSERS));
    }
    *out = &d;
    return AESGCM_OK;
}

struct aesgcm_ctx {
    int device = 0;
    int nr = 0;
    int G = 0;                         // workgroups per full launch
    DevTables *tables = nullptr;
    KeyMaterial *km = nullptr;
    uint4 *parts = nullptr;            // one item (64 lane accumulators) per chunk, grown on demand
    size_t parts_cap = 0;              // items
    uint4 *fold_a = nullptr, *fold_b = nullptr;   // k_fold ping-pong: MAX_CHUNKS/128 items; the second level leaves at most max(MAX_CHUNKS/65536, COMBINE_MAX_ITEMS) (fold_group)
    u32 *d_counter = nullptr;          // chunk dispenser
    u32 counter_base = 0;              // value the packet dispenser (d_counter[0]) holds before the next launch
    u32 qset = 0;                      // which of the two sets of chunk queues (d_counter[16 (1 + 16 set + q)]) the next dynamic launch of k_main / k_body uses; that launch zeroes the other set
    u32 tw_override = 0;               // option "tw": rows per chunk of the dealt kernels, 0 = the library's rule (main_geometry)
    u64 body_min = (u64)256 << 20;     // ranges with an aligned middle of at least this many bytes go through k_body (option "body_min").  Since k_main
                                       // got cheaper below 256 MiB (dispensers, k_fold: profiles/r02f/split_threshold.txt) the cut pays from 256 MiB:
                                       // 128 MiB -16 %, 256 MiB +0.8 %, 512 MiB +4 %, 1 GiB +11 %, 2 GiB +7 %; it was 128 MiB before, 0.7 GiB in round 1
    long poll_ns = 200000L;            // how long fetch_tag polls the host slot before it blocks in the runtime (option "poll_us")
    unsigned long long *d_cyc = nullptr;   // the accumulators and the arrival counter of the fused closing of a cyclic launch (zero between launches)
    // The tag of a fused cyclic launch appears while the launch is still running, and the call's contract is that the ciphertext is in memory by then.  Three ways were
    // built and measured in round 3 (profiles/r03c/cyc_end.txt, us per message at 64 KiB / 16 MiB): the rows store THROUGH the L2 (sc0 sc1), so no line is left dirty --
    // 24 / 40, what ships (AESGCM_BODY_WT); every workgroup writes its XCD's L2 back before it counts itself arrived -- 29 / 46 (what a -DAESGCM_BODY_WT=0 build does);
    // the host waits for the end of the launch behind the tag -- 38 / 54 (deleted in round 4 with the run-time switch between the three).
    bool fold_close = true;            // whole messages through the dealt k_body: k_fold's first level closes the tag (FoldClose; option "fold_close" 0: further levels and k_combine)
    u32 cyc_prio = 2;                  // rows between rotations of the waves' issue priorities in a cyclic launch (body_prio; option "cyc_prio", 0 = off).  Without it the oldest wave of
                                       // every SIMD runs ahead and the youngest finishes alone: 256 MiB 321 -> 291 us, 1 GiB 1238 -> 1105 (dealt chunks: 1090), profiles/r03c/cyc_prio_*.txt
    int cyc_half = 2;                  // option "cyc_half": whole messages below cyc_half_max bytes take the HALF shape of the cyclic rows (k_bodyh: 256 workgroups of 512 lanes, two per
                                       // CU) -- for callers that keep two or more messages in flight on contexts of their own, where one message's staging and closing then run
                                       // beside another's rows; alone on the chip the half shape is slower than the full one.  0 = never, 1 = always, 2 (default) = when another
                                       // context of the device has a message under way at the moment of the call (others_in_flight)
    u64 cyc_half_max = (u64)80 << 20;  // sustained GiB/s, AES-256, full shape with 2 in flight / half shape with 3 (profiles/r04/inflight_threshold.txt): 8 MiB 409 / 562, 24 MiB 677 / 797,
                                       // 32 MiB 730 / 816, 48 MiB 810 / 836, 64 MiB 828 / 846, 96 MiB 870 / 862, 128 MiB 877 / 867 -- the two-table round costs what the overlap buys from there
    bool cyc_fuse = true;              // whole messages: the cyclic launch closes the tag itself (option "cyc_close" 0: k_fold + k_combine behind it, as for shards and streaming chunks)
    // Which ranges go through k_body as cyclic rows (body_cyc_lane: one launch for AAD, data and ragged end, no dispenser, 4096 items whatever the size).  options "cyc_min" / "cyc_max"
    // (bytes; both 0 = never); needs one k_body workgroup per CU on 256 CUs.  Whole messages close their tag inside the launch (cyc_close): 24 us from 16 KiB to 2 MiB where
    // k_main + k_fold + k_combine take 27 (64 KiB) .. 39 (256 KiB) .. 34 (1 MiB), profiles/r03c/cyc_small.txt -- from 64 KiB.  Shards and streaming chunks keep k_fold + k_combine
    // behind the launch and start at 4 MiB (2 MiB: 35 -> 37 us, 4 MiB: 39 -> 38).  The upper end: with the waves' priorities rotating (cyc_prio) equal shares hold up to about
    // 1 GiB -- AES-256, us per message, dealt chunks / cyclic rows: 512 MiB 577 / 555, 768 MiB 824 / 818, 896 MiB 970 / 932, 1 GiB 1069 / 1099, 1.25 GiB 1370 / 1381
    // (profiles/r03c/cyc_prio_fine_*.txt); a range with pieces around its body costs the dealt form a launch pair per piece (+45 .. 80 us), so those stay cyclic a little longer.
    u64 cyc_min_fused = (u64)64 << 10, cyc_min = (u64)4 << 20;
    u64 cyc_max = (u64)1 << 30, cyc_max_fused = (u64)1 << 30, cyc_max_pieces = (u64)1280 << 20;
    uint4 *h_tag = nullptr;            // 64 bytes of pinned, device-mapped host memory: k_combine leaves the tag here too, so fetching it
    uint4 *h_tag_dev = nullptr;        //   is a host read -- no copy kernel, no interrupt-driven stream wait (its device address)
    u64 tag_gen = 0;                   // generation number of the last result sent to the host slot (the kernel publishes it behind the tag)
    uint4 *h_mtag = nullptr, *h_mtag_dev = nullptr;   // COMBINE_BATCH_MAX slots of {tag, generation} in pinned host memory (batched finalize); created on first use
    uint4 *d_mtag = nullptr;
    uint4 *d_tag = nullptr;            // [0] tag / poly result, [1] streaming state Y
    int last_shape = AESGCM_LAUNCH_NONE;   // which launch structure the context's last whole-message call took (aesgcm_ctx_last_launch)
    bool wipe_on_auth_fail = false;    // option "wipe_on_auth_fail": decrypt calls that verify a tag zero the output of what fails (the reference's model returns the plaintext and raises: default off)
    u64 *d_trace = nullptr;            // per-workgroup trace of the last k_main launch (timing mode only)
    u32 last_np = 0;
    uint8_t *d_keystage = nullptr;     // 256 bytes: where a key (or schedule) waits for k_setup; zeroed behind it
    hipStream_t stream = nullptr;
    hipEvent_t ev_sync = nullptr;      // aesgcm_ctx_wait: marks "everything enqueued so far on this context's stream"
    hipEvent_t ev_fused = nullptr;     // aesgcm_ctx_wait_fused: recorded behind every fused-kernel launch once somebody has asked for it
    // host-API staging
    unsigned char *st_in = nullptr, *st_out = nullptr, *st_aad = nullptr;
    size_t st_in_cap = 0, st_out_cap = 0, st_aad_cap = 0;
    // pipelined host path: two device chunk slots, copy streams and events
    unsigned char *pl_buf[2] = {nullptr, nullptr};
    size_t pl_cap = 0;
    hipStream_t pl_in = nullptr, pl_out = nullptr;
    hipEvent_t pl_ev_h2d[2] = {nullptr, nullptr}, pl_ev_k[2] = {nullptr, nullptr}, pl_ev_d2h[2] = {nullptr, nullptr};
    // packets of mixed length: the launch order by length class (k_len_*).  A ring of slots, so that calls on different streams do not share one.
    OrderSlot order[4];
    unsigned order_next = 0;
    size_t order_min = 98304;          // packets from which the order pays (context option "pkt_order"; 0 = never)
    // many messages through the row kernel (k_rows, aesgcm_rows.h): one block of device scratch, grown on demand
    unsigned char *rows_buf = nullptr;
    size_t rows_cap_slots = 0, rows_cap_n = 0;
    bool rows_dirty = true;            // the scratch is not known to be zero (fresh, or a launch failed between k_rows and k_rows_close)
    u64 rows_min = (u64)64 << 10;      // packets of at least this many bytes go by rows (option "rows_min"; 0 = never).  With offset arrays the caller's pkt_len is the hint that says so
    u32 rows_block = 0;                // option "rows_block": units per dealt block of k_rows (0 = the library's cut: one block per wave, or blocks of ROWS_DYN_BLOCK for large calls)
    // streaming state
    bool s_active = false, s_data = false, s_ragged = false;
    int s_dec = 0;
    uint8_t s_iv[12];
    u64 s_aad_len = 0, s_len = 0, s_blocks = 0;   // s_blocks = GHASH blocks absorbed so far
    // timing
    bool timing = false;
    bool timing_mute = false;          // head / tail launches beside k_body are not the measured kernel
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
};

static void pipeline_release(aesgcm_ctx *c);
// the generation number of the context's host slot: written by the thread that launches, read by other contexts' launches (others_in_flight) -- atomics both ways;
// a launch that fails after taking a number gives it back, so that no context is ever taken for "under way" on account of a launch that never ran
static inline u64 gen_take(aesgcm_ctx *c) { return __atomic_add_fetch(&c->tag_gen, 1, __ATOMIC_RELAXED); }
static inline void gen_give_back(aesgcm_ctx *c) { __atomic_sub_fetch(&c->tag_gen, 1, __ATOMIC_RELAXED); }
static inline u64 gen_now(const aesgcm_ctx *c) { return __atomic_load_n(&c->tag_gen, __ATOMIC_RELAXED); }
static inline hipStream_t pick_stream(aesgcm_ctx *c, void *s) { return s ? (hipStream_t)s : c->stream; }

static const u64 MAX_DATA = (((u64)1) << 36) - 32;      // aes_icb.vhd:114
static const u64 MAX_SEQ_BLOCKS = ((u64)1) << 36;

template <int MODE>
static hipError_t launch_main_nr(int nr, dim3 grid, hipStream_t st, const KeyMaterial *km, const DevTables *tb, const MainParams &p) {
    const unsigned lds = AESGCM_LDS_BYTES;
    switch (nr) {
    case 10: hipLaunchKernelGGL((k_main<10, MODE>), grid, dim3(AESGCM_MAIN_WG), lds, st, km, tb, p); break;
    case 12: hipLaunchKernelGGL((k_main<12, MODE>), grid, dim3(AESGCM_MAIN_WG), lds, st, km, tb, p); break;
    default: hipLaunchKernelGGL((k_main<14, MODE>), grid, dim3(AESGCM_MAIN_WG), lds, st, km, tb, p); break;
    }
    return hipGetLastError();
}
static hipError_t launch_main(int mode, int nr, dim3 grid, hipStream_t st, const KeyMaterial *km, const DevTables *tb, const MainParams &p) {
    switch (mode) {
    case MODE_ENC: return launch_main_nr<MODE_ENC>(nr, grid, st, km, tb, p);
    case MODE_DEC: return launch_main_nr<MODE_DEC>(nr, grid, st, km, tb, p);
    case MODE_KS:  return launch_main_nr<MODE_KS>(nr, grid, st, km, tb, p);
    default:       return launch_main_nr<MODE_ECB>(nr, grid, st, km, tb, p);
    }
}

static int set_lds_attrs(int device, DeviceState *ds) {
    // 72 KiB of dynamic LDS per workgroup exceeds the 64 KiB default cap: opt in once per kernel instance and device.
    std::lock_guard<std::mutex> lk(g_mu);
    if (ds->attrs) return AESGCM_OK;
    HIPCHK(hipSetDevice(device));
#define SETATTR(NR, MODE) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_main<NR, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, AESGCM_LDS_BYTES))
    SETATTR(10, MODE_ENC); SETATTR(12, MODE_ENC); SETATTR(14, MODE_ENC);
    SETATTR(10, MODE_DEC); SETATTR(12, MODE_DEC); SETATTR(14, MODE_DEC);
    SETATTR(10, MODE_KS);  SETATTR(12, MODE_KS);  SETATTR(14, MODE_KS);
    SETATTR(10, MODE_ECB); SETATTR(12, MODE_ECB); SETATTR(14, MODE_ECB);
#undef SETATTR
#define SETATTRY(NR, MODE, CYC) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_body<NR, MODE, CYC>), hipFuncAttributeMaxDynamicSharedMemorySize, AESGCM_BODY_LDS + (CYC ? CYC_LDS_PARK_BYTES : 0u)))
    SETATTRY(10, MODE_ENC, false); SETATTRY(12, MODE_ENC, false); SETATTRY(14, MODE_ENC, false); SETATTRY(10, MODE_DEC, false); SETATTRY(12, MODE_DEC, false); SETATTRY(14, MODE_DEC, false);
    SETATTRY(10, MODE_ENC, true); SETATTRY(12, MODE_ENC, true); SETATTRY(14, MODE_ENC, true); SETATTRY(10, MODE_DEC, true); SETATTRY(12, MODE_DEC, true); SETATTRY(14, MODE_DEC, true);
    SETATTRY(10, MODE_PROBE, false); SETATTRY(12, MODE_PROBE, false); SETATTRY(14, MODE_PROBE, false);
#undef SETATTRY
#define SETATTRH(NR, MODE) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bodyh<NR, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, AESGCM_LDS_BYTES + CYC_LDS_PARK_BYTES))
    SETATTRH(10, MODE_ENC); SETATTRH(12, MODE_ENC); SETATTRH(14, MODE_ENC); SETATTRH(10, MODE_DEC); SETATTRH(12, MODE_DEC); SETATTRH(14, MODE_DEC);
#undef SETATTRH
#define SETATTRR(NR, MODE) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rows<NR, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, AESGCM_BODY_LDS))
    SETATTRR(10, MODE_ENC); SETATTRR(12, MODE_ENC); SETATTRR(14, MODE_ENC); SETATTRR(10, MODE_DEC); SETATTRR(12, MODE_DEC); SETATTRR(14, MODE_DEC);
#undef SETATTRR
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fold), hipFuncAttributeMaxDynamicSharedMemorySize, FOLD_LDS_CLOSE_BYTES));
#define SETATTRB(NR, D) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pktg<NR, D, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, PKTG_LDS_TOTAL(2))); \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pktg<NR, D, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, PKTG_LDS_TOTAL(3))); \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pktg<NR, D, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, PKTG_LDS_TOTAL(4))); \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pktg<NR, D, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, PKTG_LDS_TOTAL(6))); \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pktl<NR, D, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, AESGCM_PKTL_LDS)); \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pktl<NR, D, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, AESGCM_PKTL_LDS))
    SETATTRB(10, 0); SETATTRB(12, 0); SETATTRB(14, 0); SETATTRB(10, 1); SETATTRB(12, 1); SETATTRB(14, 1);
#undef SETATTRB
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_combine), hipFuncAttributeMaxDynamicSharedMemorySize, CMB_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_combine_batch), hipFuncAttributeMaxDynamicSharedMemorySize, CMB_LDS_BYTES));
#define SETATTRB3(NR, D) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_batch3<NR, D, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, BATCH3_LDS_BYTES_LG(6))); \
                         HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_batch3<NR, D, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, BATCH3_LDS_BYTES_LG(4))); \
                         HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_batch3<NR, D, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, BATCH3_LDS_BYTES_LG(3)))
    SETATTRB3(10, 0); SETATTRB3(12, 0); SETATTRB3(14, 0); SETATTRB3(10, 1); SETATTRB3(12, 1); SETATTRB3(14, 1);
#undef SETATTRB3
    ds->attrs = true;
    return AESGCM_OK;
}

// What the fold stage needs to know about the partials a launch produced.
struct Partials { const uint4 *ptr = nullptr; u32 np = 0; u32 kind = PARTS_NONE; const uint4 *ej0 = nullptr; u64 eA = 0; bool done = false; const uint4 *tail_item = nullptr; u32 tail_blocks = 0; };   // eA: blocks between chunk items when k_combine folds them itself (np > 1)

static int grow_parts(aesgcm_ctx *c, size_t need) {
    if (need <= c->parts_cap) return AESGCM_OK;
    if (c->parts) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(c->parts)); c->parts = nullptr; c->parts_cap = 0; }   // rare: first big message
    size_t n = need < 4096 ? 4096 : need;
    hipError_t e = hipMalloc(&c->parts, n * 64 * sizeof(uint4));
    if (e == hipErrorOutOfMemory) return AESGCM_ENOMEM;
    if (e != hipSuccess) return hip_fail(e, "hipMalloc");
    c->parts_cap = n;
    return AESGCM_OK;
}

// device address of the key's precomputed table of H^e, or NULL
static const uint4 *ptab_ptr(const aesgcm_ctx *c, u64 e) {
    const int k = ptab_index(e);
    return k < 0 ? nullptr : reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(c->km) + offsetof(KeyMaterial, ptab)) + (size_t)k * 512;
}
// k_fold levels: n items (period, eA, eB as in FoldParams) -> one item (left in parts, fold_a or fold_b)
#define FOLD_CLOSE_MAX_WGS 512u
static int enqueue_fold(aesgcm_ctx *c, const uint4 *items, u32 n, u32 period, u64 eA, u64 eB, hipStream_t st, Partials *po, const FoldClose *close = nullptr) {
    const uint4 *cur = items;
    int which = 0;
    while (n > 1) {
        // the last level(s) can be k_combine's own: up to 64 items whose spacing has precomputed tables
        if (period <= 1 && n <= COMBINE_MAX_ITEMS && ptab_ptr(c, eA) && (n <= 4 || ptab_ptr(c, 4 * eA)) && (n <= 16 || ptab_ptr(c, 16 * eA))) {
            po->ptr = cur; po->np = n; po->kind = PARTS_ITEM; po->eA = eA;
            return AESGCM_OK;
        }
        FoldParams f;
        plan_fold(f, cur, which ? c->fold_b : c->fold_a, n, period, eA, eB);
        f.tabA = ptab_ptr(c, f.eA); f.tabB = ptab_ptr(c, f.eB); f.tabC = ptab_ptr(c, f.eC);
        const u32 G = fold_wgs(n, f.group);
        if (close && G <= FOLD_CLOSE_MAX_WGS) {
            // a whole message: this level closes the tag itself (FoldClose) -- no further level, no k_combine.  Every closing workgroup stages the lanes' tables (33 KB)
            // and spends ~4 us: the 256 workgroups of a 1 GiB message's first level are one round on the chip and the step gains 11 us (cfg2: 976 -> 963 us); the
            // 2048 of 16 GiB would be eight rounds and cost what they save (profiles/r03c/fold_close_ab.txt), so there the first level stays plain and the second
            // (64 workgroups) closes
            f.close = *close;
            f.close.on = 1; f.close.step = fold_out_step(f);
            hipLaunchKernelGGL(k_fold, dim3(G), dim3(FOLD_WG), FOLD_LDS_CLOSE_BYTES, st, c->km, f);
            HIPCHK(hipGetLastError());
            po->done = true;
            return AESGCM_OK;
        }
        if (G > (which ? FOLD_B_ITEMS : FOLD_A_ITEMS)) { snprintf(g_err, sizeof g_err, "k_fold: %u output items do not fit the level's buffer", G); return AESGCM_EHIP; }
        hipLaunchKernelGGL(k_fold, dim3(G), dim3(FOLD_WG), FOLD_LDS_BYTES, st, c->km, f);
        HIPCHK(hipGetLastError());
        eA = fold_out_step(f); eB = 0; period = 1;
        cur = f.out; n = G; which ^= 1;
    }
    po->ptr = cur; po->np = 1; po->kind = PARTS_ITEM;
    return AESGCM_OK;
}

// the context's cut of a range into head / k_body (dealt chunks) / tail
static bool ctx_body_split(const aesgcm_ctx *c, u64 len, u64 first_block, BodySplit *b) {
    return plan_body_split(len, first_block, c->tw_override, c->body_min, b);
}
// Enqueue the fused kernel over (aad, data) and the k_fold levels over its chunk items; describe the result for k_combine.  mode ENC/DEC: GHASH partials.  mode KS/ECB: no GHASH.
static int enqueue_main(aesgcm_ctx *c, int mode, const uint8_t iv[12], const void *d_aad, u64 aad_len,
                        const void *d_in, u64 len, void *d_out, u64 first_block, hipStream_t st, Partials *po, bool want_tail = false) {
    const bool gh = (mode == MODE_ENC || mode == MODE_DEC);
    if (po) *po = Partials();
    if (((uintptr_t)d_in & 15) || ((uintptr_t)d_out & 15)) return AESGCM_EALIGN;
    MainParams p;
    memset(&p, 0, sizeof p);
    const u32 C = plan_main(p, mode, c->tw_override, iv, d_aad, aad_len, d_in, len, d_out, first_block, nullptr);
    if (!C) return AESGCM_OK;
    int rc;
    if (gh && (rc = grow_parts(c, C))) return rc;
    p.parts = c->parts;
    u32 wgs = (C + 1 + AESGCM_MAIN_WG / 64 - 1) / (AESGCM_MAIN_WG / 64);      // one wave per chunk is enough for small inputs (+ one spare for E_K(J0))
    if (wgs > (u32)c->G) wgs = (u32)c->G;
    plan_queues(C, &p.nq, &p.seg);
    if ((u64)wgs * (AESGCM_MAIN_WG / 64) >= (u64)C + 1) p.nq = 0;   // a wave per chunk and a spare: static assignment, the dispensers are not touched
    p.counter = c->d_counter + 16 * (1 + AESGCM_NQ * c->qset);
    p.counter_zero = c->d_counter + 16 * (1 + AESGCM_NQ * (c->qset ^ 1u));
    if (gh && po) { p.ej0 = c->d_tag + 3; po->ej0 = p.ej0; }
    if (gh && po && want_tail && C == 1) { p.tail = 1; p.tag_out = c->d_tag; p.tag_host = c->h_tag_dev; p.gen = gen_take(c); po->done = true; }
    p.trace = nullptr;
    const bool timed = c->timing && !c->timing_mute;
    if (timed) {
        p.trace = c->d_trace;
        HIPCHK(hipMemsetAsync(c->d_trace, 0, sizeof(u64) * 4 * AESGCM_GMAX, st));
    }
    if (!c->timing_mute) c->last_np = wgs;
    std::pair<hipEvent_t, hipEvent_t> evp;
    if (timed) {
        if (!c->ev_pool.empty()) { evp = c->ev_pool.back(); c->ev_pool.pop_back(); }
        else { HIPCHK(hipEventCreate(&evp.first)); HIPCHK(hipEventCreate(&evp.second)); }
        HIPCHK(hipEventRecord(evp.first, st));
    }
    {
        const hipError_t le = launch_main(mode, c->nr, dim3(wgs), st, c->km, c->tables, p);
        if (le != hipSuccess) {                      // nothing ran: the queues were not touched on the device
            if (timed) c->ev_pool.push_back(evp);
            if (p.tail) gen_give_back(c);
            return hip_fail(le, "k_main launch");
        }
        if (p.nq) c->qset ^= 1u;                      // the launch leaves the other set zeroed for the next dynamic one
        if (c->ev_fused) HIPCHK(hipEventRecord(c->ev_fused, st));
    }
    if (timed) { HIPCHK(hipEventRecord(evp.second, st)); c->ev.push_back(evp); }
    if (gh && po && po->done) return AESGCM_OK;                   // the launch finished the tag itself
    if (gh && po) {
        const u64 eA = (u64)64 * p.Tw;
        if (C <= COMBINE_MAX_ITEMS && (C == 1 || (ptab_ptr(c, eA) && (C <= 4 || ptab_ptr(c, 4 * eA)) && (C <= 16 || ptab_ptr(c, 16 * eA))))) {
            po->ptr = c->parts; po->np = C; po->kind = PARTS_ITEM; po->eA = eA;   // few chunks: k_combine folds them, no k_fold launch
            return AESGCM_OK;
        }
        return enqueue_fold(c, c->parts, C, 1, eA, 0, st, po);
    }
    return AESGCM_OK;
}

static int enqueue_combine(aesgcm_ctx *c, const CombineParams &p, hipStream_t st);

// one k_body launch (dealt chunks or cyclic rows) with the context's timing and event bookkeeping
static int launch_body(aesgcm_ctx *c, int mode, BodyParams &p, u32 wgs, hipStream_t st) {
    const bool cyc = p.cyc != 0, half = cyc && p.cw == BODY_CYC_WAVES_HALF;
    if (cyc && mode == MODE_PROBE) return AESGCM_EARG;
    if (half && !p.fuse) return AESGCM_EARG;                    // the half shape exists with the in-launch closing only
    if (c->timing) { p.trace = c->d_trace; HIPCHK(hipMemsetAsync(c->d_trace, 0, sizeof(u64) * 4 * AESGCM_GMAX, st)); }
    c->last_np = wgs;
    std::pair<hipEvent_t, hipEvent_t> evp;
    if (c->timing) {
        if (!c->ev_pool.empty()) { evp = c->ev_pool.back(); c->ev_pool.pop_back(); }
        else { HIPCHK(hipEventCreate(&evp.first)); HIPCHK(hipEventCreate(&evp.second)); }
        HIPCHK(hipEventRecord(evp.first, st));
    }
#define LY(NR, M, CYC) hipLaunchKernelGGL((k_body<NR, M, CYC>), dim3(wgs), dim3(AESGCM_BODY_WG), AESGCM_BODY_LDS + (CYC ? CYC_LDS_PARK_BYTES : 0u), st, c->km, c->tables, p)
#define LH(NR, M) hipLaunchKernelGGL((k_bodyh<NR, M>), dim3(wgs), dim3(AESGCM_BODYH_WG), AESGCM_LDS_BYTES + CYC_LDS_PARK_BYTES, st, c->km, c->tables, p)
    if (half) {
        if (mode == MODE_DEC)    { if (c->nr == 10) LH(10, MODE_DEC); else if (c->nr == 12) LH(12, MODE_DEC); else LH(14, MODE_DEC); }
        else                     { if (c->nr == 10) LH(10, MODE_ENC); else if (c->nr == 12) LH(12, MODE_ENC); else LH(14, MODE_ENC); }
    }
    else if (cyc) {
        if (mode == MODE_DEC)    { if (c->nr == 10) LY(10, MODE_DEC, true); else if (c->nr == 12) LY(12, MODE_DEC, true); else LY(14, MODE_DEC, true); }
        else                     { if (c->nr == 10) LY(10, MODE_ENC, true); else if (c->nr == 12) LY(12, MODE_ENC, true); else LY(14, MODE_ENC, true); }
    }
    else if (mode == MODE_DEC)   { if (c->nr == 10) LY(10, MODE_DEC, false); else if (c->nr == 12) LY(12, MODE_DEC, false); else LY(14, MODE_DEC, false); }
    else if (mode == MODE_PROBE) { if (c->nr == 10) LY(10, MODE_PROBE, false); else if (c->nr == 12) LY(12, MODE_PROBE, false); else LY(14, MODE_PROBE, false); }
    else                         { if (c->nr == 10) LY(10, MODE_ENC, false); else if (c->nr == 12) LY(12, MODE_ENC, false); else LY(14, MODE_ENC, false); }
#undef LY
#undef LH
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) {
        if (c->timing) c->ev_pool.push_back(evp);
        return hip_fail(le, "k_body launch");
    }
    if (!cyc) c->qset ^= 1u;
    if (c->timing) { HIPCHK(hipEventRecord(evp.second, st)); c->ev.push_back(evp); }
    if (c->ev_fused) HIPCHK(hipEventRecord(c->ev_fused, st));
    return AESGCM_OK;
}
// k_body over the planned split + the k_fold levels over its interleaved chunk items
static int enqueue_body(aesgcm_ctx *c, int mode, const uint8_t iv[12], const BodySplit &b, const void *d_in, void *d_out,
                        u64 first_block, hipStream_t st, Partials *po, const FoldClose *close = nullptr) {
    BodyParams p;
    memset(&p, 0, sizeof p);
    int rc = grow_parts(c, (size_t)4 * b.S);
    if (rc) return rc;
    plan_body(p, b, iv, d_in, d_out, first_block, c->parts);
    p.ej0 = c->d_tag + 3; po->ej0 = p.ej0;
    const u32 waves_per_wg = AESGCM_BODY_WG / 64;
    u32 wgs = (p.C + waves_per_wg - 1) / waves_per_wg;
#if AESGCM_T4
    if (wgs > (u32)c->G / 2) wgs = (u32)c->G / 2;                 // one 136 KiB workgroup per CU
#else
    if (wgs > (u32)c->G) wgs = (u32)c->G;
#endif
    plan_queues(p.C, &p.nq, &p.seg);
    p.counter = c->d_counter + 16 * (1 + AESGCM_NQ * c->qset);
    p.counter_zero = c->d_counter + 16 * (1 + AESGCM_NQ * (c->qset ^ 1u));
    if ((rc = launch_body(c, mode, p, wgs, st))) return rc;
    // items 4s + v: phases 64 blocks apart inside a super-chunk, super-chunks 256 T blocks apart
    return enqueue_fold(c, c->parts, p.C, 4, 64, (u64)256 * b.T, st, po, close);
}
// A whole range -- AAD, data from any first block, ragged end -- as ONE k_body launch of cyclic rows (plan_body_cyc) and the k_fold level over its
// 4096 items, when the range is of that size (*took says whether it was).  po describes the items and the partial last row for k_combine.
static bool cyc_capable(const aesgcm_ctx *c) {
#if AESGCM_T4
    return (u32)c->G / 2 * (AESGCM_BODY_WG / 64) == BODY_CYC_WAVES && c->cyc_max_pieces > c->cyc_min_fused;
#else
    return false;
#endif
}
// Is a message of ANOTHER context of this device under way right now?  Every result goes to its context's pinned host slot with the generation number of its
// launch behind it, so "under way" is: the slot does not show the generation last launched.  What the half shape of the cyclic rows is for (two messages
// share every CU); asked once per whole-message launch, a mutex and a few loads.  Contexts register in ctx_create_common and leave in aesgcm_ctx_destroy.
static std::vector<aesgcm_ctx *> &g_ctxs = *new std::vector<aesgcm_ctx *>();          // never destroyed: contexts may be destroyed after this library's static destructors have run
static bool others_in_flight(const aesgcm_ctx *c) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (const aesgcm_ctx *o : g_ctxs) {
        if (o == c || o->device != c->device || !o->h_tag) continue;
        const u64 launched = __atomic_load_n(&o->tag_gen, __ATOMIC_RELAXED);
        if (__atomic_load_n(reinterpret_cast<const u64 *>(o->h_tag + 1), __ATOMIC_RELAXED) != launched) return true;
    }
    return false;
}
static int enqueue_cyc(aesgcm_ctx *c, int mode, const uint8_t iv[12], const void *d_aad, u64 aad_len, const void *d_in, u64 len, void *d_out,
                       u64 first_block, hipStream_t st, Partials *po, bool *took, bool whole_message_tag = false) {
    *took = false;
    const bool fused = whole_message_tag && c->cyc_fuse;
    if (!cyc_capable(c) || len < (fused ? c->cyc_min_fused : c->cyc_min)) return AESGCM_OK;
    if (((uintptr_t)d_in & 15) || ((uintptr_t)d_out & 15)) return AESGCM_OK;     // the caller's other path reports the alignment
    int rc = grow_parts(c, (size_t)BODY_CYC_WAVES + 1);
    if (rc) return rc;
    BodyParams p;
    // a range with pieces around its body (AAD, an odd first block, a ragged end) costs the other paths a launch pair per piece (+45 .. 80 us,
    // profiles/r03c/general_shape.txt): for those the cyclic launch stays ahead for longer
    const bool pieces = aad_len || (first_block & 255) || (len & 1023);
    const u64 lo = fused ? c->cyc_min_fused : c->cyc_min, hi = pieces ? c->cyc_max_pieces : fused ? c->cyc_max_fused : c->cyc_max;
    const bool half = fused && len < c->cyc_half_max && (c->cyc_half == 1 || (c->cyc_half == 2 && others_in_flight(c)));   // two workgroups per CU: for messages in flight beside each other
    if (!plan_body_cyc(p, mode, iv, d_aad, aad_len, d_in, len, d_out, first_block, c->parts, lo, hi, half ? BODY_CYC_WAVES_HALF : BODY_CYC_WAVES)) return AESGCM_OK;
    *took = true;
    *po = Partials();
    p.prio_rows = c->cyc_prio;
    if (fused) {                                                                // the launch closes the tag itself (cyc_close): nothing behind it
        p.fuse = 1; p.aad_len = aad_len; p.ct_len = len; p.acc = c->d_cyc;
        p.tag_out = c->d_tag; p.tag_host = c->h_tag_dev; p.gen = gen_take(c);
        po->done = true;
        c->last_shape = half ? AESGCM_LAUNCH_CYCLIC_HALF : AESGCM_LAUNCH_CYCLIC;
        rc = launch_body(c, mode, p, half ? BODY_CYC_WAVES_HALF / (AESGCM_BODYH_WG / 64) : BODY_CYC_WAVES / (AESGCM_BODY_WG / 64), st);
        if (rc) gen_give_back(c);
        return rc;
    }
    p.ej0 = c->d_tag + 3; po->ej0 = p.ej0;
    if ((rc = launch_body(c, mode, p, BODY_CYC_WAVES / (AESGCM_BODY_WG / 64), st))) return rc;
    if ((rc = enqueue_fold(c, c->parts, BODY_CYC_WAVES, 1, 64, 0, st, po))) return rc;       // always BODY_CYC_WAVES items, 64 blocks apart
    if (p.tb) { po->tail_item = c->parts + (size_t)BODY_CYC_WAVES * 64; po->tail_blocks = p.tb; }
    return AESGCM_OK;
}

// Y' = Y * H^nb ^ P(aad, data) for a whole range, Y in *state (device).  Large ranges go head / k_body / tail,
// each piece folded into the state in order; small ones are a single k_main launch.
static int absorb_range(aesgcm_ctx *c, int mode, const uint8_t iv[12], const void *d_aad, u64 aad_len, const void *d_in, u64 len,
                        void *d_out, u64 first_block, hipStream_t st, uint4 *state, const uint4 **ej0 = nullptr) {
    BodySplit b;
    Partials pp;
    int rc;
    {   // mid-size ranges: the whole range in one k_body launch of cyclic rows
        bool took;
        if ((rc = enqueue_cyc(c, mode, iv, d_aad, aad_len, d_in, len, d_out, first_block, st, &pp, &took))) return rc;
        if (took) {
            if (ej0) *ej0 = pp.ej0;
            const u64 nb = (aad_len + 15) / 16 + (len + 15) / 16;
            return enqueue_combine(c, combine_with_items(plan_combine_carry(pp.ptr, pp.np, pp.kind, state, nb), pp.eA, pp.tail_item, pp.tail_blocks), st);
        }
    }
    if (!ctx_body_split(c, len, first_block, &b)) {
        if ((rc = enqueue_main(c, mode, iv, d_aad, aad_len, d_in, len, d_out, first_block, st, &pp))) return rc;
        const u64 nb = (aad_len + 15) / 16 + (len + 15) / 16;
        return nb ? enqueue_combine(c, combine_with_items(plan_combine_carry(pp.ptr, pp.np, pp.kind, state, nb), pp.eA), st) : AESGCM_OK;
    }
    if (((uintptr_t)d_in & 15) || ((uintptr_t)d_out & 15)) return AESGCM_EALIGN;
    const u64 n_aad = (aad_len + 15) / 16;
    if (n_aad + b.head_blocks) {
        c->timing_mute = true;
        rc = enqueue_main(c, mode, iv, d_aad, aad_len, d_in, 16 * b.head_blocks, d_out, first_block, st, &pp);
        c->timing_mute = false;
        if (rc) return rc;
        if ((rc = enqueue_combine(c, combine_with_items(plan_combine_carry(pp.ptr, pp.np, pp.kind, state, n_aad + b.head_blocks), pp.eA), st))) return rc;
    }
    if ((rc = enqueue_body(c, mode, iv, b, d_in, d_out, first_block, st, &pp))) return rc;
    if (ej0) *ej0 = pp.ej0;                                      // valid until the next launch on this context overwrites the slot: consumed by the caller's final combine
    if ((rc = enqueue_combine(c, combine_with_items(plan_combine_carry(pp.ptr, pp.np, pp.kind, state, b.body_blocks), pp.eA), st))) return rc;
    const u64 done = b.head_blocks + b.body_blocks, tail = len - 16 * done;
    if (tail) {
        c->timing_mute = true;
        rc = enqueue_main(c, mode, iv, nullptr, 0, (const unsigned char *)d_in + 16 * done, tail, (unsigned char *)d_out + 16 * done,
                          first_block + done, st, &pp);
        c->timing_mute = false;
        if (rc) return rc;
        if ((rc = enqueue_combine(c, combine_with_items(plan_combine_carry(pp.ptr, pp.np, pp.kind, state, (tail + 15) / 16), pp.eA), st))) return rc;
    }
    return AESGCM_OK;
}

static int enqueue_combine(aesgcm_ctx *c, const CombineParams &p0, hipStream_t st) {
    CombineParams p = p0;
    const bool to_slot = p.out == c->d_tag;
    if (to_slot) { p.out_host = c->h_tag_dev; p.gen = gen_take(c); }          // results that go to the tag slot are mirrored to the pinned host slot
    if (p.kind == PARTS_ITEM && p.np > 1) {                       // the launch folds the items itself: tables of H^eA, H^(8 eA)
        p.tabA = ptab_ptr(c, p.eA);
        p.tabB = p.np > 4 ? ptab_ptr(c, 4 * p.eA) : nullptr;
        p.tabC = p.np > 16 ? ptab_ptr(c, 16 * p.eA) : nullptr;
        if (p.np > COMBINE_MAX_ITEMS || !p.tabA || (p.np > 4 && !p.tabB) || (p.np > 16 && !p.tabC)) { if (to_slot) gen_give_back(c); snprintf(g_err, sizeof g_err, "k_combine: %u items, spacing %llu not foldable in the launch", p.np, (unsigned long long)p.eA); return AESGCM_EHIP; }
    }
    hipLaunchKernelGGL(k_combine, dim3(1), dim3(COMBINE_THREADS), CMB_LDS_BYTES, st, c->km, c->tables, p);
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) { if (to_slot) gen_give_back(c); return hip_fail(le, "k_combine launch"); }
    return AESGCM_OK;
}

static int check_lengths(u64 aad_len, u64 len) {
    if (len > MAX_DATA) return AESGCM_ETOOLONG;
    if ((aad_len + 15) / 16 + (len + 15) / 16 >= MAX_SEQ_BLOCKS) return AESGCM_ETOOLONG;
    return AESGCM_OK;
}

// whole message on device pointers; leaves the tag in c->d_tag[0]
static int crypt_dev(aesgcm_ctx *c, int dec, const uint8_t iv[12], const void *d_aad, u64 aad_len,
                     const void *d_in, u64 len, void *d_out, hipStream_t st) {
    int rc = check_lengths(aad_len, len);
    if (rc) return rc;
    if (aad_len && !d_aad) return AESGCM_EARG;
    if (len && (!d_in || !d_out)) return AESGCM_EARG;
    HIPCHK(hipSetDevice(c->device));
    {   // mid-size messages: k_body as cyclic rows takes AAD, data and the ragged end in one launch; its items go straight to the tag
        Partials pc;
        bool took;
        if ((rc = enqueue_cyc(c, dec ? MODE_DEC : MODE_ENC, iv, d_aad, aad_len, d_in, len, d_out, 0, st, &pc, &took, true))) return rc;
        if (took && pc.done) return AESGCM_OK;                   // the launch left the tag in d_tag and in the host slot
        if (took) {
            c->last_shape = AESGCM_LAUNCH_CYCLIC;
            CombineParams q = combine_with_items(plan_combine_tag(pc.ptr, pc.np, pc.kind, iv, aad_len, len, c->d_tag), pc.eA, pc.tail_item, pc.tail_blocks);
            q.ej0 = pc.ej0;
            return enqueue_combine(c, q, st);
        }
    }
    BodySplit b;
    if (ctx_body_split(c, len, 0, &b)) {
        c->last_shape = AESGCM_LAUNCH_DEALT;
        if (!aad_len && !b.head_blocks && len == 16 * b.body_blocks) {
            // the whole message is one aligned body (the benchmark's shape): no chaining value to carry, k_body's items go
            // straight to the tag
            Partials pb;
            FoldClose fc = {};
            if (c->fold_close) {                                 // k_fold's first level closes the tag (when there is a k_fold launch at all)
                fc.aad_len = 0; fc.ct_len = len; fc.ej0 = c->d_tag + 3; fc.acc = c->d_cyc;
                fc.tag_out = c->d_tag; fc.tag_host = c->h_tag_dev; fc.gen = gen_now(c) + 1;
            }
            if ((rc = enqueue_body(c, dec ? MODE_DEC : MODE_ENC, iv, b, d_in, d_out, 0, st, &pb, c->fold_close ? &fc : nullptr))) return rc;
            if (pb.done) { gen_take(c); return AESGCM_OK; }
            CombineParams q = combine_with_items(plan_combine_tag(pb.ptr, pb.np, pb.kind, iv, 0, len, c->d_tag), pb.eA);
            q.ej0 = pb.ej0;
            return enqueue_combine(c, q, st);
        }
        // large message: head / k_body / tail folded into a device-side chaining value, then the tag from it
        uint4 *state = c->d_tag + 2;
        HIPCHK(hipMemsetAsync(state, 0, 16, st));
        const uint4 *ej0 = nullptr;
        if ((rc = absorb_range(c, dec ? MODE_DEC : MODE_ENC, iv, d_aad, aad_len, d_in, len, d_out, 0, st, state, &ej0))) return rc;
        CombineParams q = plan_combine_final(state, iv, aad_len, len, c->d_tag);
        q.ej0 = ej0;                                             // left by k_body (every piece of this message writes the same value)
        return enqueue_combine(c, q, st);
    }
    Partials pp;
    c->last_shape = AESGCM_LAUNCH_MAIN;
    rc = enqueue_main(c, dec ? MODE_DEC : MODE_ENC, iv, d_aad, aad_len, d_in, len, d_out, 0, st, &pp, true);
    if (rc) return rc;
    if (pp.done) return AESGCM_OK;                               // single chunk: k_main's tail left the tag in d_tag and in the host slot
    CombineParams q = combine_with_items(plan_combine_tag(pp.ptr, pp.np, pp.kind, iv, aad_len, len, c->d_tag), pp.eA);
    q.ej0 = pp.ej0;                                              // same IV, same stream: k_main left E_K(IV || 1) behind
    return enqueue_combine(c, q, st);
}

// The tag of the last result enqueued for the host slot: the kernel stores it in pinned host memory and then publishes
// the generation number; the host polls that number for a short while (a kernel-completion interrupt costs ~10 us on
// this platform, a poll of coherent host memory well under one) and falls back to a stream synchronisation for long-
// running work or if anything went wrong.
// What has happened when this returns: a tag published from INSIDE a launch (k_body's cyclic rows, cyc_close; k_fold's closing, acc_arrive) is seen while that
// launch is still running, and this function does NOT wait for its end -- the stream is not synchronised.  Every byte of the result is in device memory all the
// same: the rows store through the L2 (global_store ... sc0 sc1, gstore16_wt / gstore*_wt_at, AESGCM_BODY_WT), each workgroup waits for the acknowledgement of
// its own stores (s_waitcnt vmcnt(0)) before it counts itself arrived, and the tag is published by the workgroup that counts the last arrival; the launch
// retires a few microseconds later.  examples/early_read.cpp (tests/test_gpu_cyclic.py) is the standing check: a copy ordered behind nothing reads the whole
// result the moment the tag is there.  Tags that come from k_combine or k_main's tail are published by the last kernel of the call.
static int fetch_tag(aesgcm_ctx *c, hipStream_t st, uint8_t tag[16]) {
    const u64 want = gen_now(c);
    volatile u64 *gen = reinterpret_cast<volatile u64 *>(c->h_tag + 1);
    bool seen = false;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (u32 spin = 0;; ++spin) {                                 // poll for at most ~200 us, then block in the runtime
        if (__atomic_load_n(gen, __ATOMIC_ACQUIRE) == want) { seen = true; break; }
        if ((spin & 63u) == 63u) {
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > c->poll_ns) break;
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    if (!seen) {
        HIPCHK(hipStreamSynchronize(st));
        // the stream the message was enqueued on has drained: its tag is there -- unless `st` is not that stream, or the launch failed after the number was taken
        if (__atomic_load_n(gen, __ATOMIC_ACQUIRE) != want) {
            snprintf(g_err, sizeof g_err, "the host slot shows generation %llu, not %llu: the stream passed is not the one the message was enqueued on", (unsigned long long)__atomic_load_n(gen, __ATOMIC_ACQUIRE), (unsigned long long)want);
            return AESGCM_ESTATE;
        }
    }
    memcpy(tag, c->h_tag, 16);
    return AESGCM_OK;
}

static int ct_compare16(const uint8_t *a, const uint8_t *b) {
    unsigned d = 0;
    for (int i = 0; i < 16; i++) d |= (unsigned)(a[i] ^ b[i]);
    return d == 0;
}

static int grow(unsigned char **p, size_t *cap, size_t need) {
    if (need <= *cap) return AESGCM_OK;
    if (*p) { hipError_t e = hipFree(*p); *p = nullptr; *cap = 0; if (e != hipSuccess) return hip_fail(e, "hipFree"); }
    size_t n = need < 4096 ? 4096 : need;
    hipError_t e = hipMalloc((void **)p, n);
    if (e == hipErrorOutOfMemory) return AESGCM_ENOMEM;
    if (e != hipSuccess) return hip_fail(e, "hipMalloc");
    *cap = n;
    return AESGCM_OK;
}

extern "C" {

int aesgcm_abi_version(void) { return AESGCM_ABI_VERSION; }

const char *aesgcm_strerror(int code) {
    switch (code) {
    case AESGCM_OK: return "ok";
    case AESGCM_EARG: return "invalid argument";
    case AESGCM_EKEYLEN: return "key length must be 16, 24 or 32 bytes";
    case AESGCM_EIVLEN: return "IV must be 12 bytes";
    case AESGCM_ETOOLONG: return "message exceeds the GCM counter space (2^36 - 32 bytes)";
    case AESGCM_EAUTH: return "authentication tag mismatch";
    case AESGCM_EHIP: return "HIP runtime error (see aesgcm_last_error)";
    case AESGCM_ENOMEM: return "out of device memory";
    case AESGCM_ESTATE: return "streaming call out of order";
    case AESGCM_EALIGN: return "device data pointer must be 16-byte aligned";
    case AESGCM_ERCCL: return "RCCL unavailable or a collective failed (see aesgcm_comm_last_error)";
    default: return "unknown error";
    }
}
const char *aesgcm_last_error(void) { return g_err; }

int aesgcm_device_count(int *n) {
    if (!n) return AESGCM_EARG;
    HIPCHK(hipGetDeviceCount(n));
    return AESGCM_OK;
}
int aesgcm_device_name(int device, char *buf, size_t buflen) {
    if (!buf || !buflen) return AESGCM_EARG;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buflen, "%s %s (%d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return AESGCM_OK;
}

// Key material of a context from a key (or a pre-expanded schedule): aes_kexp, H, the H-power tables -- k_setup and k_setup_ptab on the context's stream, waited
// for.  The staging buffer for the key bytes belongs to the context (aesgcm_ctx_rekey comes through here without an allocation) and is wiped behind the kernels.
static int ctx_load_key(aesgcm_ctx *c, const uint8_t *key, size_t key_len, int pre_nr) {
    hipError_t e;
    if (!c->d_keystage && (e = hipMalloc((void **)&c->d_keystage, 256)) != hipSuccess) return hip_fail(e, "hipMalloc");
    const size_t kb = pre_nr ? (size_t)16 * (pre_nr + 1) : key_len;
    e = hipMemcpyAsync(c->d_keystage, key, kb, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_setup, dim3(1), dim3(AESGCM_WG), 0, c->stream, c->km, c->tables, c->d_keystage, (int)key_len, pre_nr, (u32)c->G);
        hipLaunchKernelGGL(k_setup_ptab, dim3(AESGCM_NPTAB + AESGCM_NLTAB), dim3(512), 0, c->stream, c->km);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemsetAsync(c->d_keystage, 0, 256, c->stream);    // do not leave key bytes behind
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return hip_fail(e, "k_setup");
    c->nr = pre_nr ? pre_nr : (int)(key_len / 4 + 6);                               // only now: a load that failed leaves the context's round count with its old key material
    return AESGCM_OK;
}
static int ctx_create_common(aesgcm_ctx **out, int device, const uint8_t *key, size_t key_len, int pre_nr) {
    if (!out || !key) return AESGCM_EARG;
    *out = nullptr;
    DeviceState *ds;
    int rc = device_state(device, &ds);
    if (rc) return rc;
    rc = set_lds_attrs(device, ds);
    if (rc) return rc;
    aesgcm_ctx *c = new (std::nothrow) aesgcm_ctx();
    if (!c) return AESGCM_ENOMEM;
    c->device = device;
    c->tables = ds->tables;
    c->nr = pre_nr ? pre_nr : (int)(key_len / 4 + 6);
    const int per_cu = 2;
    int G = per_cu * ds->n_cu;
    if (G > AESGCM_GMAX) G = AESGCM_GMAX;
    if (G < 1) G = 1;
    c->G = G;
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) { delete c; return hip_fail(e, "hipSetDevice"); }
    {   // a stream a destroyed context left behind, or a new one
        std::lock_guard<std::mutex> lk(g_mu);
        if (!ds->streams.empty()) { c->stream = ds->streams.back(); ds->streams.pop_back(); }
    }
    if (!c->stream && (e = hipStreamCreate(&c->stream)) != hipSuccess) { delete c; return hip_fail(e, "hipStreamCreate"); }
    if ((e = hipMalloc(&c->km, sizeof(KeyMaterial))) != hipSuccess ||
        (e = hipMalloc(&c->fold_a, sizeof(uint4) * 64 * FOLD_A_ITEMS)) != hipSuccess ||
        (e = hipMalloc(&c->fold_b, sizeof(uint4) * 64 * FOLD_B_ITEMS)) != hipSuccess ||
        (e = hipMalloc(&c->d_counter, 64 * (1 + 2 * AESGCM_NQ))) != hipSuccess ||
        (e = hipMemset(c->d_counter, 0, 64 * (1 + 2 * AESGCM_NQ))) != hipSuccess ||
        (e = hipMalloc(&c->d_cyc, 8 * (2 * CYC_ACC_SLOTS + 1))) != hipSuccess ||
        (e = hipMemset(c->d_cyc, 0, 8 * (2 * CYC_ACC_SLOTS + 1))) != hipSuccess ||
        (e = hipMalloc(&c->d_tag, sizeof(uint4) * 4)) != hipSuccess ||
        (e = hipHostMalloc((void **)&c->h_tag, 64, hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess ||
        (e = hipHostGetDevicePointer((void **)&c->h_tag_dev, c->h_tag, 0)) != hipSuccess ||
        (memset(c->h_tag, 0, 64), false) ||
        (e = hipMalloc(&c->d_trace, sizeof(u64) * 4 * AESGCM_GMAX)) != hipSuccess) { aesgcm_ctx_destroy(c); return hip_fail(e, "hipMalloc"); }
    if ((rc = ctx_load_key(c, key, key_len, pre_nr))) { aesgcm_ctx_destroy(c); return rc; }
    { std::lock_guard<std::mutex> lk(g_mu); g_ctxs.push_back(c); }
    *out = c;
    return AESGCM_OK;
}

int aesgcm_ctx_create(aesgcm_ctx **out, int device, const uint8_t *key, size_t key_len) {
    if (key_len != 16 && key_len != 24 && key_len != 32) return AESGCM_EKEYLEN;
    return ctx_create_common(out, device, key, key_len, 0);
}
int aesgcm_ctx_create_preexpanded(aesgcm_ctx **out, int device, const uint8_t *rk, int nr) {
    if (nr != 10 && nr != 12 && nr != 14) return AESGCM_EKEYLEN;
    return ctx_create_common(out, device, rk, (size_t)(4 * (nr - 6)), nr);
}
// A new key for an existing context (the reference core's "load key" between frames, tb/gcm_gctr.py:144-175; H is recomputed only then, src/gcm_gctr.vhd:142-144):
// everything the context owns stays -- stream, scratch, host slot, options -- only the key material is rebuilt.  Waits for the context's queued work first.
int aesgcm_ctx_rekey(aesgcm_ctx *c, const uint8_t *key, size_t key_len) {
    if (!c || !key) return AESGCM_EARG;
    if (key_len != 16 && key_len != 24 && key_len != 32) return AESGCM_EKEYLEN;
    if (c->s_active) return AESGCM_ESTATE;
    HIPCHK(hipSetDevice(c->device));
    // every *_dev entry point takes a caller's stream, so work that reads this context's key material may be queued on any stream of the device: wait for them all
    // (round 4 waited for the context's own stream only -- a message in flight on another stream would have read half-rebuilt tables)
    HIPCHK(hipDeviceSynchronize());
    return ctx_load_key(c, key, key_len, 0);
}
int aesgcm_ctx_destroy(aesgcm_ctx *c) {
    if (!c) return AESGCM_OK;
    { std::lock_guard<std::mutex> lk(g_mu); g_ctxs.erase(std::remove(g_ctxs.begin(), g_ctxs.end(), c), g_ctxs.end()); }
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (auto &e : c->ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (auto &e : c->ev_pool) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    if (c->km) { hipMemset(c->km, 0, sizeof(KeyMaterial)); hipFree(c->km); }
    if (c->d_keystage) hipFree(c->d_keystage);
    if (c->parts) hipFree(c->parts);
    if (c->fold_a) hipFree(c->fold_a);
    if (c->fold_b) hipFree(c->fold_b);
    if (c->d_counter) hipFree(c->d_counter);
    if (c->d_cyc) hipFree(c->d_cyc);
    if (c->d_tag) hipFree(c->d_tag);
    if (c->h_tag) hipHostFree(c->h_tag);
    if (c->h_mtag) hipHostFree(c->h_mtag);
    if (c->d_mtag) hipFree(c->d_mtag);
    if (c->d_trace) hipFree(c->d_trace);
    if (c->rows_buf) hipFree(c->rows_buf);
    pipeline_release(c);
    for (auto &o : c->order) { if (o.perm) hipFree(o.perm); if (o.bins) hipFree(o.bins); if (o.done) hipEventDestroy(o.done); }
    if (c->st_in) hipFree(c->st_in);
    if (c->st_out) hipFree(c->st_out);
    if (c->st_aad) hipFree(c->st_aad);
    if (c->ev_sync) hipEventDestroy(c->ev_sync);
    if (c->ev_fused) hipEventDestroy(c->ev_fused);
    if (c->stream) {                                            // idle by now (synchronised above): kept for the device's next context, up to 64 of them
        std::lock_guard<std::mutex> lk(g_mu);
        if (c->device >= 0 && c->device < (int)g_dev.size() && g_dev[c->device].streams.size() < 64) { g_dev[c->device].streams.push_back(c->stream); c->stream = nullptr; }
    }
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return AESGCM_OK;
}
int aesgcm_ctx_device(const aesgcm_ctx *c) { return c ? c->device : AESGCM_EARG; }
// which launch structure the context's last whole-message call took (AESGCM_LAUNCH_*): the choice between the full and the half shape of the cyclic rows depends on
// what other contexts had under way at the moment of the call, so benches and profiles ask instead of assuming
int aesgcm_ctx_last_launch(const aesgcm_ctx *c, int *shape) {
    if (!c || !shape) return AESGCM_EARG;
    *shape = c->last_shape;
    return AESGCM_OK;
}
// Tunables of ONE context, for tests and profiling scripts (the defaults are the measured best; nothing in the library reads the environment).
// Every value selects between paths that produce the same bytes; the parity tests use them to reach each path at sizes a CPU check finishes in seconds.
int aesgcm_ctx_set_option(aesgcm_ctx *c, const char *key, int64_t value) {
    if (!c || !key || value < 0) return AESGCM_EARG;
    const u64 v = (u64)value;
    if (!strcmp(key, "tw")) c->tw_override = (u32)v;                                   // rows per chunk of the dealt kernels (0 = the library's rule)
    else if (!strcmp(key, "body_min")) {                                              // bytes from which a range's aligned middle goes through k_body
        c->body_min = v;
        if (v >= (1ull << 60)) c->cyc_max = c->cyc_max_pieces = c->cyc_max_fused = 0;  // "never k_body" means the cyclic rows too
    }
    else if (!strcmp(key, "cyc_min")) c->cyc_min = c->cyc_min_fused = v;               // bytes: ranges in [cyc_min, cyc_max) take k_body's cyclic rows; both 0 = never
    else if (!strcmp(key, "cyc_max")) c->cyc_max = c->cyc_max_pieces = c->cyc_max_fused = v;
    else if (!strcmp(key, "cyc_half")) { if (v > 2) return AESGCM_EARG; c->cyc_half = (int)v; }   // whole messages below 80 MiB as k_bodyh (two workgroups per CU): 0 never, 1 always, 2 when another context has a message under way
    else if (!strcmp(key, "cyc_close")) c->cyc_fuse = v != 0;                          // 1: a whole message's cyclic launch closes the tag itself; 0: k_fold + k_combine behind it
    else if (!strcmp(key, "fold_close")) c->fold_close = v != 0;                       // 1: behind the dealt k_body the first (or second) k_fold level closes the tag
    else if (!strcmp(key, "cyc_prio")) c->cyc_prio = (u32)v;                           // rows between rotations of the waves' issue priorities in a cyclic launch (0 = off)
    else if (!strcmp(key, "pkt_order")) c->order_min = (size_t)v;                      // packets from which a launch over packets of mixed length takes them by falling length class (k_len_*); 0 = never
    else if (!strcmp(key, "wipe_on_auth_fail")) c->wipe_on_auth_fail = v != 0;         // decrypt with verification: zero the output of a message / packet whose tag does not match
    else if (!strcmp(key, "rows_min")) c->rows_min = v;                                // bytes per packet from which aesgcm_packets_crypt_dev goes by rows (k_rows); 0 = never
    else if (!strcmp(key, "rows_block")) c->rows_block = (u32)v;                       // units (rows, tails) per dealt block of k_rows; 0 = the library's cut
    else if (!strcmp(key, "poll_us")) c->poll_ns = 1000L * (long)v;                    // how long a tag is polled for in the host slot before the call blocks in the runtime
    else return AESGCM_EARG;
    return AESGCM_OK;
}
int aesgcm_ctx_stream(const aesgcm_ctx *c, void **stream) {
    if (!c || !stream) return AESGCM_EARG;
    *stream = (void *)c->stream;
    return AESGCM_OK;
}
// Everything enqueued from now on on `c`'s own stream starts only after everything enqueued so far on `other`'s own stream
// has completed (one event record + one stream wait; no host synchronisation).  Two contexts of one key on one device
// have separate scratch sets and streams, so consecutive messages can alternate between them and message m+1's fused
// kernel starts while message m's k_fold / k_combine drain; this call orders the step that needs both (the all-gather).
int aesgcm_ctx_wait(aesgcm_ctx *c, aesgcm_ctx *other) {
    if (!c || !other) return AESGCM_EARG;
    if (c == other) return AESGCM_OK;
    if (c->device != other->device) return AESGCM_EARG;
    HIPCHK(hipSetDevice(c->device));
    if (!other->ev_sync) HIPCHK(hipEventCreateWithFlags(&other->ev_sync, hipEventDisableTiming));
    HIPCHK(hipEventRecord(other->ev_sync, other->stream));
    HIPCHK(hipStreamWaitEvent(c->stream, other->ev_sync, 0));
    return AESGCM_OK;
}
// As aesgcm_ctx_wait, but only up to `other`'s most recently enqueued FUSED kernel (k_body / k_main), not its fold / combine
// tail: message m+1's fused kernel (on `c`) then follows message m's (on `other`) back to back, and m's k_fold / k_combine
// launches run beside it.  (Two contexts that simply start together share the CUs -- k_body is one 141 KiB workgroup per CU --
// and finish together: that hides one tail in two; chained, all tails but the last hide.)  The event is recorded from the
// first call on; a wait issued before `other` has launched anything is a no-op.
int aesgcm_ctx_wait_fused(aesgcm_ctx *c, aesgcm_ctx *other) {
    if (!c || !other) return AESGCM_EARG;
    if (c == other) return AESGCM_OK;
    if (c->device != other->device) return AESGCM_EARG;
    HIPCHK(hipSetDevice(c->device));
    if (!other->ev_fused) { HIPCHK(hipEventCreateWithFlags(&other->ev_fused, hipEventDisableTiming)); return AESGCM_OK; }
    HIPCHK(hipStreamWaitEvent(c->stream, other->ev_fused, 0));
    return AESGCM_OK;
}
int aesgcm_ctx_geometry(const aesgcm_ctx *c, int *n_wg, int *wg_lanes, int *lds_bytes) {
    if (!c) return AESGCM_EARG;
    if (n_wg) *n_wg = c->G;
    if (wg_lanes) *wg_lanes = AESGCM_MAIN_WG;
    if (lds_bytes) *lds_bytes = AESGCM_LDS_BYTES;
    return AESGCM_OK;
}

int aesgcm_ctx_body_geometry(const aesgcm_ctx *c, int *n_wg, int *wg_lanes, int *lds_bytes) {
    if (!c) return AESGCM_EARG;
#if AESGCM_T4
    if (n_wg) *n_wg = c->G / 2;
#else
    if (n_wg) *n_wg = c->G;
#endif
    if (wg_lanes) *wg_lanes = AESGCM_BODY_WG;
    if (lds_bytes) *lds_bytes = AESGCM_BODY_LDS;
    return AESGCM_OK;
}

int aesgcm_ctx_split(const aesgcm_ctx *c, size_t len, uint64_t first_block, uint64_t *head_blocks, uint64_t *body_blocks) {
    if (!c) return AESGCM_EARG;
    if (cyc_capable(c)) {                                                  // cyclic rows: the body is every whole row behind the head
        const u64 nfull = len / 16, head = (256 - (first_block & 255)) & 255;
        const u64 R = nfull > head ? (nfull - head) / 64 : 0;
        if (R && R * 1024 >= c->cyc_min && R * 1024 < (((first_block & 255) || (len & 1023)) ? c->cyc_max_pieces : c->cyc_max)) {
            if (head_blocks) *head_blocks = head;
            if (body_blocks) *body_blocks = 64 * R;
            return AESGCM_OK;
        }
    }
    BodySplit b;
    const bool split = ctx_body_split(c, len, first_block, &b);
    if (head_blocks) *head_blocks = split ? b.head_blocks : 0;
    if (body_blocks) *body_blocks = split ? b.body_blocks : 0;
    return AESGCM_OK;
}

// ---------------------------------------------------------------- unit-level
int aesgcm_key_expand(int device, const uint8_t *key, size_t key_len, uint8_t rk[240], int *nr) {
    if (!key || !rk) return AESGCM_EARG;
    if (key_len != 16 && key_len != 24 && key_len != 32) return AESGCM_EKEYLEN;
    aesgcm_ctx *c = nullptr;
    int rc = aesgcm_ctx_create(&c, device, key, key_len);
    if (rc) return rc;
    hipError_t e = hipMemcpy(rk, c->km->rk_bytes, (size_t)16 * (c->nr + 1), hipMemcpyDeviceToHost);
    if (nr) *nr = c->nr;
    aesgcm_ctx_destroy(c);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpy");
    return AESGCM_OK;
}

int aesgcm_get_h(aesgcm_ctx *c, uint8_t h[16]) {
    if (!c || !h) return AESGCM_EARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpy(h, &c->km->h, 16, hipMemcpyDeviceToHost));
    return AESGCM_OK;
}

int aesgcm_gfmul(int device, const uint8_t *h, const uint8_t *x, uint8_t *z, size_t n) {
    if (!h || !x || !z) return AESGCM_EARG;
    if (!n) return AESGCM_OK;
    DeviceState *ds;
    int rc = device_state(device, &ds);
    if (rc) return rc;
    HIPCHK(hipSetDevice(device));
    uint4 *d = nullptr;
    HIPCHK(hipMalloc(&d, 48 * n));
    hipError_t e = hipMemcpy(d, h, 16 * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + n, x, 16 * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_gfmul, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d, d + n, d + 2 * n, n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(z, d + 2 * n, 16 * n, hipMemcpyDeviceToHost);
    hipFree(d);
    if (e != hipSuccess) return hip_fail(e, "aesgcm_gfmul");
    return AESGCM_OK;
}

// ---------------------------------------------------------------- host-pointer wrappers
static int stage_in(aesgcm_ctx *c, const uint8_t *aad, size_t aad_len, const uint8_t *in, size_t len) {
    int rc;
    if ((rc = grow(&c->st_aad, &c->st_aad_cap, aad_len))) return rc;
    if ((rc = grow(&c->st_in, &c->st_in_cap, len))) return rc;
    if ((rc = grow(&c->st_out, &c->st_out_cap, len))) return rc;
    if (aad_len) HIPCHK(hipMemcpyAsync(c->st_aad, aad, aad_len, hipMemcpyHostToDevice, c->stream));
    if (len) HIPCHK(hipMemcpyAsync(c->st_in, in, len, hipMemcpyHostToDevice, c->stream));
    return AESGCM_OK;
}

int aesgcm_encrypt_dev(aesgcm_ctx *c, const uint8_t iv[12], const void *d_aad, size_t aad_len,
                       const void *d_pt, size_t len, void *d_ct, uint8_t tag[16], void *stream) {
    if (!c || !iv) return AESGCM_EARG;
    hipStream_t st = pick_stream(c, stream);
    int rc = crypt_dev(c, 0, iv, d_aad, aad_len, d_pt, len, d_ct, st);
    if (rc) return rc;
    if (tag) return fetch_tag(c, st, tag);
    return AESGCM_OK;
}
int aesgcm_decrypt_dev(aesgcm_ctx *c, const uint8_t iv[12], const void *d_aad, size_t aad_len,
                       const void *d_ct, size_t len, void *d_pt, const uint8_t *expect_tag, uint8_t tag_out[16], void *stream) {
    if (!c || !iv) return AESGCM_EARG;
    hipStream_t st = pick_stream(c, stream);
    int rc = crypt_dev(c, 1, iv, d_aad, aad_len, d_ct, len, d_pt, st);
    if (rc) return rc;
    if (tag_out || expect_tag) {
        uint8_t t[16];
        if ((rc = fetch_tag(c, st, t))) return rc;
        if (tag_out) memcpy(tag_out, t, 16);
        if (expect_tag && !ct_compare16(t, expect_tag)) {
            if (c->wipe_on_auth_fail && len) { HIPCHK(hipMemsetAsync(d_pt, 0, len, st)); HIPCHK(hipStreamSynchronize(st)); }   // nothing unauthenticated is left in the caller's buffer
            return AESGCM_EAUTH;
        }
    }
    return AESGCM_OK;
}
// the tag of the message most recently enqueued with tag = NULL (aesgcm_encrypt_dev / aesgcm_decrypt_dev): through the host slot, as if the call had asked for it
int aesgcm_last_tag(aesgcm_ctx *c, uint8_t tag[16], void *stream) {
    if (!c || !tag) return AESGCM_EARG;
    hipStream_t st = pick_stream(c, stream);
    HIPCHK(hipSetDevice(c->device));
    return fetch_tag(c, st, tag);
}

int aesgcm_encrypt(aesgcm_ctx *c, const uint8_t iv[12], const uint8_t *aad, size_t aad_len,
                   const uint8_t *pt, size_t len, uint8_t *ct, uint8_t tag[16]) {
    if (!c || !iv || !tag || (aad_len && !aad) || (len && (!pt || !ct))) return AESGCM_EARG;
    int rc = check_lengths(aad_len, len);
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    if ((rc = stage_in(c, aad, aad_len, pt, len))) return rc;
    if ((rc = crypt_dev(c, 0, iv, c->st_aad, aad_len, c->st_in, len, c->st_out, c->stream))) return rc;
    // the call returns data synchronously (tb/gcm_model.py:26): the tag's generation number is published by k_combine BEFORE
    // this copy starts, and with a page-locked `ct` (aesgcm_host_alloc) the copy is truly asynchronous -- wait for it.  With
    // len == 0 nothing is copied and the tag alone is polled for.
    if (len) { HIPCHK(hipMemcpyAsync(ct, c->st_out, len, hipMemcpyDeviceToHost, c->stream)); HIPCHK(hipStreamSynchronize(c->stream)); }
    return fetch_tag(c, c->stream, tag);
}
int aesgcm_decrypt(aesgcm_ctx *c, const uint8_t iv[12], const uint8_t *aad, size_t aad_len,
                   const uint8_t *ct, size_t len, uint8_t *pt, const uint8_t *expect_tag, uint8_t tag_out[16]) {
    if (!c || !iv || (aad_len && !aad) || (len && (!ct || !pt))) return AESGCM_EARG;
    int rc = check_lengths(aad_len, len);
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    if ((rc = stage_in(c, aad, aad_len, ct, len))) return rc;
    if ((rc = crypt_dev(c, 1, iv, c->st_aad, aad_len, c->st_in, len, c->st_out, c->stream))) return rc;
    uint8_t t[16];
    const bool hold = expect_tag && c->wipe_on_auth_fail;        // the plaintext leaves the device only once its tag has been checked
    if (len && !hold) { HIPCHK(hipMemcpyAsync(pt, c->st_out, len, hipMemcpyDeviceToHost, c->stream)); HIPCHK(hipStreamSynchronize(c->stream)); }   // as aesgcm_encrypt: never return while `pt` is still landing
    if ((rc = fetch_tag(c, c->stream, t))) return rc;
    if (tag_out) memcpy(tag_out, t, 16);
    if (expect_tag && !ct_compare16(t, expect_tag)) {
        if (hold && len) { memset(pt, 0, len); HIPCHK(hipMemsetAsync(c->st_out, 0, len, c->stream)); HIPCHK(hipStreamSynchronize(c->stream)); }
        return AESGCM_EAUTH;
    }
    if (len && hold) { HIPCHK(hipMemcpyAsync(pt, c->st_out, len, hipMemcpyDeviceToHost, c->stream)); HIPCHK(hipStreamSynchronize(c->stream)); }
    return AESGCM_OK;
}

int aesgcm_ecb_encrypt(aesgcm_ctx *c, const uint8_t *in, size_t nblocks, uint8_t *out) {
    if (!c || (nblocks && (!in || !out))) return AESGCM_EARG;
    if (!nblocks) return AESGCM_OK;
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = stage_in(c, nullptr, 0, in, 16 * nblocks))) return rc;
    if ((rc = enqueue_main(c, MODE_ECB, nullptr, nullptr, 0, c->st_in, 16 * (u64)nblocks, c->st_out, 0, c->stream, nullptr))) return rc;
    HIPCHK(hipMemcpyAsync(out, c->st_out, 16 * nblocks, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return AESGCM_OK;
}

int aesgcm_keystream_dev(aesgcm_ctx *c, const uint8_t iv[12], uint64_t first_block, uint64_t nblocks, void *d_out, void *stream) {
    if (!c || !iv || (nblocks && !d_out)) return AESGCM_EARG;
    if (first_block + nblocks > (((u64)1) << 32) - 2) return AESGCM_ETOOLONG;
    if (!nblocks) return AESGCM_OK;
    HIPCHK(hipSetDevice(c->device));
    return enqueue_main(c, MODE_KS, iv, nullptr, 0, d_out /*unused in*/, 16 * nblocks, d_out, first_block, pick_stream(c, stream), nullptr);
}
int aesgcm_keystream(aesgcm_ctx *c, const uint8_t iv[12], uint64_t first_block, uint64_t nblocks, uint8_t *out) {
    if (!c || !iv || (nblocks && !out)) return AESGCM_EARG;
    if (!nblocks) return AESGCM_OK;
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = grow(&c->st_out, &c->st_out_cap, 16 * nblocks))) return rc;
    if ((rc = aesgcm_keystream_dev(c, iv, first_block, nblocks, c->st_out, nullptr))) return rc;
    HIPCHK(hipMemcpyAsync(out, c->st_out, 16 * nblocks, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return AESGCM_OK;
}

int aesgcm_ghash(aesgcm_ctx *c, const uint8_t *data, size_t len, uint8_t y[16]) {
    if (!c || !y || (len && !data)) return AESGCM_EARG;
    if ((len + 15) / 16 >= MAX_SEQ_BLOCKS) return AESGCM_ETOOLONG;
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = stage_in(c, data, len, nullptr, 0))) return rc;
    Partials pp;
    uint8_t iv0[12] = {0};
    // the data rides in the AAD slot of the GHASH sequence (GHASH only, no AES)
    if ((rc = enqueue_main(c, MODE_ENC, iv0, c->st_aad, len, c->st_in, 0, c->st_out, 0, c->stream, &pp))) return rc;
    if ((rc = enqueue_combine(c, combine_with_items(plan_combine_poly(pp.ptr, pp.np, pp.kind, 1, c->d_tag), pp.eA), c->stream))) return rc;   // Y = P * H
    HIPCHK(hipMemcpyAsync(y, c->d_tag, 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return AESGCM_OK;
}

// ---------------------------------------------------------------- shards
int aesgcm_shard_crypt_dev(aesgcm_ctx *c, int decrypt, const uint8_t iv[12], const void *d_aad, size_t aad_len,
                           const void *d_in, size_t len, void *d_out, uint64_t first_block, uint64_t total_len,
                           void *d_partial, void *stream) {
    if (!c || !iv || !d_partial || (len && (!d_in || !d_out))) return AESGCM_EARG;
    int rc = check_lengths(first_block == 0 ? aad_len : 0, total_len);
    if (rc) return rc;
    const u64 total_blocks = (total_len + 15) / 16;
    const u64 my_blocks = ((u64)len + 15) / 16;
    if (first_block + my_blocks > total_blocks) return AESGCM_EARG;
    if ((len & 15) && first_block + my_blocks != total_blocks) return AESGCM_EARG;   // only the last shard may be ragged
    if (first_block != 0 && aad_len) return AESGCM_EARG;
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = pick_stream(c, stream);
    const u64 after = total_blocks - (first_block + my_blocks);            // blocks of the message behind this shard
    {   // mid-size shards: one k_body launch of cyclic rows, its items straight to the weighted partial W = P H^after
        Partials pc;
        bool took;
        if ((rc = enqueue_cyc(c, decrypt ? MODE_DEC : MODE_ENC, iv, d_aad, aad_len, d_in, len, d_out, first_block, st, &pc, &took))) return rc;
        if (took) return enqueue_combine(c, combine_with_items(plan_combine_poly(pc.ptr, pc.np, pc.kind, after, (uint4 *)d_partial), pc.eA, pc.tail_item, pc.tail_blocks), st);
    }
    BodySplit b;
    if (ctx_body_split(c, len, first_block, &b)) {
        if (!aad_len && !b.head_blocks && len == 16 * b.body_blocks) {
            // the shard is one aligned body (the 8-GPU job's shape: 4 GiB at a multiple of 256 blocks): its items go straight to the
            // weighted partial W = P H^after -- no chaining value, one k_combine instead of memset + carry combine + weighting combine
            Partials pb;
            if ((rc = enqueue_body(c, decrypt ? MODE_DEC : MODE_ENC, iv, b, d_in, d_out, first_block, st, &pb))) return rc;
            return enqueue_combine(c, combine_with_items(plan_combine_poly(pb.ptr, pb.np, pb.kind, after, (uint4 *)d_partial), pb.eA), st);
        }
        uint4 *state = c->d_tag + 2;
        HIPCHK(hipMemsetAsync(state, 0, 16, st));
        if ((rc = absorb_range(c, decrypt ? MODE_DEC : MODE_ENC, iv, d_aad, aad_len, d_in, len, d_out, first_block, st, state))) return rc;
        CombineParams q = plan_combine_poly(nullptr, 0, PARTS_NONE, 0, (uint4 *)d_partial);       // W = Y * H^after
        q.carry = state; q.has_carry = 1; q.e_carry = after;
        return enqueue_combine(c, q, st);
    }
    Partials pp;
    if ((rc = enqueue_main(c, decrypt ? MODE_DEC : MODE_ENC, iv, d_aad, aad_len, d_in, len, d_out, first_block, st, &pp))) return rc;
    return enqueue_combine(c, combine_with_items(plan_combine_poly(pp.ptr, pp.np, pp.kind, after, (uint4 *)d_partial), pp.eA), st);
}
int aesgcm_shard_finalize_strided_dev(aesgcm_ctx *c, const uint8_t iv[12], const void *d_partials, size_t n_partials, size_t stride_bytes,
                                      size_t aad_len, uint64_t total_len, uint8_t tag[16], void *stream) {
    if (!c || !iv || (n_partials && !d_partials) || n_partials > AESGCM_GMAX) return AESGCM_EARG;
    if (stride_bytes < 16 || (stride_bytes & 15) || stride_bytes / 16 > 0xFFFFFFFFull) return AESGCM_EARG;
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = pick_stream(c, stream);
    CombineParams q = plan_combine_tag((const uint4 *)d_partials, (u32)n_partials, PARTS_GATHERED, iv, aad_len, total_len, c->d_tag);
    q.stride = (u32)(stride_bytes / 16);
    int rc = enqueue_combine(c, q, st);
    if (rc) return rc;
    if (tag) return fetch_tag(c, st, tag);
    return AESGCM_OK;
}
// The tags of n_msgs messages in ONE launch (one workgroup per message) and one wait: what a multi-GPU step does after its single
// all-gather.  Per message a k_combine launch costs ~15 us (E_K(IV || 1) bytewise on one lane) plus a host round trip for its tag;
// four of them were ~140 us of a 17 ms rank step (profiles/r03/rank_step_trace.txt).
int aesgcm_shard_finalize_batch_dev(aesgcm_ctx *c, size_t n_msgs, const uint8_t *ivs, const void *d_partials, size_t n_partials,
                                    size_t stride_bytes, size_t msg_stride_bytes, const size_t *aad_lens, const uint64_t *total_lens,
                                    uint8_t *tags, void *stream) {
    if (!c || !ivs || !total_lens || !tags || (n_partials && !d_partials) || n_partials > AESGCM_GMAX) return AESGCM_EARG;
    if (!n_msgs) return AESGCM_OK;
    if (n_msgs > COMBINE_BATCH_MAX) return AESGCM_EARG;
    if (stride_bytes < 16 || (stride_bytes & 15) || stride_bytes / 16 > 0xFFFFFFFFull || (msg_stride_bytes & 15)) return AESGCM_EARG;
    HIPCHK(hipSetDevice(c->device));
    if (!c->h_mtag) {
        HIPCHK(hipHostMalloc((void **)&c->h_mtag, 32 * COMBINE_BATCH_MAX, hipHostMallocMapped | hipHostMallocCoherent));
        memset(c->h_mtag, 0, 32 * COMBINE_BATCH_MAX);
        HIPCHK(hipHostGetDevicePointer((void **)&c->h_mtag_dev, c->h_mtag, 0));
        HIPCHK(hipMalloc(&c->d_mtag, 16 * COMBINE_BATCH_MAX));
    }
    hipStream_t st = pick_stream(c, stream);
    CombineBatch b;
    memset(&b, 0, sizeof b);
    const u64 gen = gen_take(c);
    for (size_t m = 0; m < n_msgs; m++) {
        CombineParams q = plan_combine_tag((const uint4 *)((const unsigned char *)d_partials + m * msg_stride_bytes), (u32)n_partials, PARTS_GATHERED,
                                           ivs + 12 * m, aad_lens ? aad_lens[m] : 0, total_lens[m], c->d_mtag + m);
        q.stride = (u32)(stride_bytes / 16);
        q.out_host = c->h_mtag_dev + 2 * m; q.gen = gen;
        b.p[m] = q;
    }
    hipLaunchKernelGGL(k_combine_batch, dim3((unsigned)n_msgs), dim3(COMBINE_THREADS), CMB_LDS_BYTES, st, c->km, c->tables, b);
    { const hipError_t le = hipGetLastError(); if (le != hipSuccess) { gen_give_back(c); return hip_fail(le, "k_combine_batch launch"); } }
    // every workgroup publishes its own generation word behind its tag: poll them all (short), then fall back to the stream
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    bool seen = false;
    for (u32 spin = 0; !seen; ++spin) {
        seen = true;
        for (size_t m = 0; m < n_msgs; m++)
            if (__atomic_load_n(reinterpret_cast<volatile u64 *>(c->h_mtag + 2 * m + 1), __ATOMIC_ACQUIRE) != gen) { seen = false; break; }
        if (seen) break;
        if ((spin & 63u) == 63u) {
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > c->poll_ns) break;
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    if (!seen) HIPCHK(hipStreamSynchronize(st));
    for (size_t m = 0; m < n_msgs; m++) memcpy(tags + 16 * m, c->h_mtag + 2 * m, 16);
    return AESGCM_OK;
}
int aesgcm_shard_finalize_dev(aesgcm_ctx *c, const uint8_t iv[12], const void *d_partials, size_t n_partials,
                              size_t aad_len, uint64_t total_len, uint8_t tag[16], void *stream) {
    return aesgcm_shard_finalize_strided_dev(c, iv, d_partials, n_partials, 16, aad_len, total_len, tag, stream);
}

// ---------------------------------------------------------------- streaming
// state Y (c->d_tag[1]) = polynomial of everything absorbed so far: sum X_i H^(n-1-i)
int aesgcm_stream_begin(aesgcm_ctx *c, const uint8_t iv[12], int decrypt) {
    if (!c || !iv) return AESGCM_EARG;
    HIPCHK(hipSetDevice(c->device));
    memcpy(c->s_iv, iv, 12);
    c->s_active = true; c->s_data = false; c->s_ragged = false; c->s_dec = decrypt ? 1 : 0;
    c->s_aad_len = 0; c->s_len = 0; c->s_blocks = 0;
    HIPCHK(hipMemsetAsync(c->d_tag + 1, 0, 16, c->stream));
    return AESGCM_OK;
}
static int stream_absorb(aesgcm_ctx *c, const void *d_aad, u64 aad_len, const void *d_in, u64 len, void *d_out, u64 first_block) {
    Partials pp;
    int rc = enqueue_main(c, c->s_dec ? MODE_DEC : MODE_ENC, c->s_iv, d_aad, aad_len, d_in, len, d_out, first_block, c->stream, &pp);
    if (rc) return rc;
    const u64 nb = (aad_len + 15) / 16 + (len + 15) / 16;
    c->s_blocks += nb;
    return enqueue_combine(c, combine_with_items(plan_combine_carry(pp.ptr, pp.np, pp.kind, c->d_tag + 1, nb), pp.eA), c->stream);       // Y' = Y * H^nb ^ P
}
int aesgcm_stream_aad(aesgcm_ctx *c, const uint8_t *aad, size_t len) {
    if (!c || (len && !aad)) return AESGCM_EARG;
    if (!c->s_active || c->s_data || c->s_ragged) return AESGCM_ESTATE;
    if (!len) return AESGCM_OK;
    if (check_lengths(c->s_aad_len + len, 0)) return AESGCM_ETOOLONG;
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = stage_in(c, aad, len, nullptr, 0))) return rc;
    if ((rc = stream_absorb(c, c->st_aad, len, c->st_in, 0, c->st_out, 0))) return rc;
    c->s_aad_len += len;
    if (len & 15) c->s_ragged = true;
    HIPCHK(hipStreamSynchronize(c->stream));
    return AESGCM_OK;
}
int aesgcm_stream_update(aesgcm_ctx *c, const uint8_t *in, size_t len, uint8_t *out) {
    if (!c || (len && (!in || !out))) return AESGCM_EARG;
    if (!c->s_active) return AESGCM_ESTATE;
    if (c->s_data && c->s_ragged) return AESGCM_ESTATE;      // a ragged data chunk must be the last one
    if (!len) return AESGCM_OK;
    if (check_lengths(c->s_aad_len, c->s_len + len)) return AESGCM_ETOOLONG;
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = stage_in(c, nullptr, 0, in, len))) return rc;
    c->s_ragged = false;
    if ((rc = stream_absorb(c, nullptr, 0, c->st_in, len, c->st_out, c->s_len / 16))) return rc;
    c->s_data = true;
    c->s_len += len;
    if (len & 15) c->s_ragged = true;
    HIPCHK(hipMemcpyAsync(out, c->st_out, len, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return AESGCM_OK;
}
int aesgcm_stream_final(aesgcm_ctx *c, uint8_t tag[16]) {
    if (!c || !tag) return AESGCM_EARG;
    if (!c->s_active) return AESGCM_ESTATE;
    HIPCHK(hipSetDevice(c->device));
    int rc = enqueue_combine(c, plan_combine_final(c->d_tag + 1, c->s_iv, c->s_aad_len, c->s_len, c->d_tag), c->stream);
    if (rc) return rc;
    if ((rc = fetch_tag(c, c->stream, tag))) return rc;
    c->s_active = false;
    return AESGCM_OK;
}

// ---------------------------------------------------------------- shapes of the packet kernels
#ifdef AESGCM_DEBUG_KNOBS
// Test / profiling builds only (libaesgcm_hip_dbg.so, -DAESGCM_DEBUG_KNOBS; include/aesgcm_debug.h): force the kernel shape the next launches take, so that every
// shape can be checked on inputs the host's own rule would give to another.  The product library has no such switch and reads no environment.
static struct { int pkt_lanes, pkt_deal, batch_lanes, batch_deal, batch_order, pkt_ilp, pkt_rows; } g_force = {0, 0, 0, 0, 0, 0, 0};
extern "C" __attribute__((visibility("default"))) int aesgcm_debug_force_shape(const char *what, int value) {
    if (!what) return AESGCM_EARG;
    if (!strcmp(what, "pkt_lanes")) { if (value != 0 && value != 1 && value != 4 && value != 8 && value != 16 && value != 64) return AESGCM_EARG; g_force.pkt_lanes = value; }
    else if (!strcmp(what, "pkt_deal")) g_force.pkt_deal = value;
    else if (!strcmp(what, "batch_lanes")) { if (value != 0 && value != 8 && value != 16 && value != 64) return AESGCM_EARG; g_force.batch_lanes = value; }
    else if (!strcmp(what, "batch_deal")) g_force.batch_deal = value;
    else if (!strcmp(what, "pkt_ilp")) { if (value < 0 || value > 2) return AESGCM_EARG; g_force.pkt_ilp = value; }              // k_pktl's ILP form: 0 = the library's rule, 1 = always, 2 = never
    else if (!strcmp(what, "pkt_rows")) { if (value < 0 || value > 2) return AESGCM_EARG; g_force.pkt_rows = value; }            // aesgcm_packets_crypt_dev by rows (k_rows): 0 = the library's rule, 1 = always, 2 = never
    else if (!strcmp(what, "batch_order")) { if (value < 0 || value > 2) return AESGCM_EARG; g_force.batch_order = value; }      // variable-length batches by length class: 0 = the library's rule, 1 = always, 2 = never
    else return AESGCM_EARG;
    return AESGCM_OK;
}
#endif
// Packets under ONE key: how many lanes work on one packet, as log2 (0 = one LANE per packet, k_pktl; 2, 3, 4 = a lane GROUP of 4, 8, 16, k_pktg; 6 = a whole
// wave, k_pktg<.., 6>).  Measured (profiles/r03/packets_sweep_aes256.txt, GiB/s wave / g16 / g8 / g4 / lane): the best shape is the one that just fills the
// resident lanes (256 CUs x 16 waves x 64) -- 65536 x 1 KiB 203 / 232 / 340 / 384 / 194, 16384 x 4 KiB 235 / 367 / 290 / 177 / 53, 4096 x 16 KiB
// 362 / 172 / 95 / 49 / 13 (the one regime where a whole wave per packet is right: at most 4096 packets of at least 4 KiB) -- but never more lanes than an
// eighth of the packet's blocks once the machine is full (closing cost per byte: 16384 x 1 KiB 62 / 128 / 176 / 138 / 50, 16384 x 256 B 16 / 35 / 58 / 72 / 41),
// a quarter when it is not (4096 x 1 KiB 34 / 69 / 57 / 38 / 13).  Lanes win from 131072 packets (2^20 x 1 KiB 303 / 592 / 657 / 742 / 767; 262144 x 4 KiB
// 496 / 656 / 704 / 722 / 724), short packets from 32768 (65536 x 256 B 51 / 61 / 95 / 129 / 148).  With offset arrays the host does not know the lengths: it
// goes by count and assumes 1 KiB.
static int packets_pick_lg(u32 n_cu, size_t n_pkts, size_t pkt_len, bool var, bool ordered = false) {
    const size_t lanes_total = (size_t)n_cu * (AESGCM_PKT_WG / 64) * 64, lanes_l = (size_t)n_cu * AESGCM_PKTL_WG;
    const size_t blocks = var ? 64 : (pkt_len + 15) / 16;
    // One lane per packet once the packets fill k_pktl's resident lanes (256 x 768); frames of up to 1 KiB from three quarters of that, short ones much earlier.
    // Round 4 (profiles/r04/packets_sweep_aes256.txt, after k_pktl's rebuild): 131072 x 4 KiB 553 by lanes against 722 by groups of 4 (196608: 795 / 713),
    // 131072 x 16 KiB 573 / 789, 131072 x 1 KiB 488 / 509 (196608: 677 / 577), 49152 x 256 B 142 / 124, 16384 x 64 B 28 / 23.
    // Offset arrays (the host does not know the lengths): as 1 KiB frames in array order (mixed 64 .. 1514 bytes: 131072 frames 243 by lanes / 277 by groups of
    // 4, 196608: 346 / 296); taken by length class the groups hold on longer (196608: 323 / 394, 262144: 421 / 429, 393216: 584 / 505).
    // k_pktl's ILP form (512-lane workgroups) moves the 1 KiB mark down: 131072 x 1 KiB 592 by lanes against 500 by groups of 4, 98304: 454 / 456.
    const size_t lanes_ilp = (size_t)n_cu * AESGCM_PKTL_WG_ILP;
    if (var ? (ordered ? 3 * n_pkts >= 4 * lanes_l : 4 * n_pkts >= 3 * lanes_l)
            : (n_pkts >= lanes_l || (pkt_len <= 1024 && 8 * n_pkts >= 7 * lanes_ilp) || (pkt_len <= 256 && n_pkts >= 32768) || (pkt_len <= 64 && n_pkts >= 16384))) return 0;
    // Lane groups: the group that just fills the resident lanes.  Packets of 4 KiB and more round the fill UP to a power of two (half again as many lanes as
    // are resident is cheaper than rows twice as long: 49152 x 4 KiB 474 with 4 lanes, 576 with 8; x 16 KiB 542 / 722), shorter ones down (49152 x 1 KiB 325 / 291).
    size_t fill = lanes_total / n_pkts;
    if (!var && pkt_len >= 4096 && (fill & (fill - 1))) { size_t f = 1; while (f < fill) f <<= 1; fill = f; }
    const size_t cap = n_pkts >= 16384 ? blocks / 8 : blocks / 4;
    const size_t g = fill < cap ? fill : cap;
    return g >= 64 ? 6 : g >= 16 ? 4 : g >= 8 ? 3 : 2;
}
// Packets with their OWN key (k_batch3): lanes per packet as log2 (3, 4, 6 = 8 / 16 lanes, a whole wave; the two-pass kernel k_batch of rounds 2 - 3 that
// the numbers below call by name is gone since round 4: k_batch3<.., 6> took its place, 4096 x 1 MiB 443 -> 637 GiB/s).  16 lanes once
// there are packets enough to fill the machine that way (one 1024-lane workgroup per CU = 64 packets per CU) or the packets are short, else one wave per packet.
// Measured, AES-128, GiB/s k_batch / k_batch3 (profiles/r03/batch_sweep_aes128.txt): 4096 x 1 KiB 30 / 56, 4096 x 256 B 7.5 / 17, 1024 x 1 KiB 14 / 16.5; 1024 x 4 KiB
// 45 / 33, 4096 x 4 KiB 108 / 120, 4096 x 16 KiB 286 / 168; from 16384 packets k_batch3 wins at every size (4 KiB 179 / 350).  8 lanes (eight packets per wave
// share what a wave-iteration pays once) when there are packets enough to fill the chip that way and they are not long: 2^20 packets of 64 B 42 -> 74 GiB/s,
// 256 B 163 -> 265, 1 KiB 424 -> 560, 1500 B 484 -> 598, 4 KiB 658 -> 706, 16 KiB 770 -> 736; 16384 packets: 1 KiB 125 -> 155, 4 KiB 352 -> 273
// (profiles/r03c/batch_sweep_lanes8_aes128.txt).  Batches with per-packet lengths (offset arrays on the device: the host does not know the lengths) go by count
// alone and assume frames of MACsec size, where 8 lanes gain most; a batch of frames beyond 8 KiB loses ~5 % by it.
static int batch_pick_lg(int n_cu, size_t n_pkts, size_t pkt_len, bool var) {
    int lg = (n_pkts >= (size_t)64 * n_cu || (!var && pkt_len <= 2048)) ? 4 : 6;
    if (lg == 4 && (var ? n_pkts >= (size_t)64 * n_cu
                        : ((n_pkts >= (size_t)256 * n_cu && pkt_len <= 8192) || (n_pkts >= (size_t)64 * n_cu && pkt_len <= 2048)))) lg = 3;
    return lg;
}

// The order in which a launch takes packets of mixed length: counting sort by falling length class on the launch's stream (k_len_hist, k_len_scan,
// k_len_scatter).  *perm = NULL when it does not pay or is switched off.  Three launches of about 10 us in front of the packet kernel: mixed frames of
// 64 .. 1514 bytes, AES-256, best shape each (profiles/r04/packets_sweep_mixed_*.txt): 16384 frames 101 GiB/s in array order, 79 by class; 65536 223 / 199; 98304
// 256 / 271; 131072 284 / 320; 262144 382 / 429; 2^20 426 / 717 -- the order pays once the machine is full, and the default threshold is there.
static bool packets_ordered(const aesgcm_ctx *c, size_t n_pkts, bool var) { return var && c->order_min && n_pkts >= c->order_min; }
static int order_launch(OrderSlot &o, const u64 *d_off, size_t n_pkts, hipStream_t st, const u32 **perm) {
    if (!o.done) HIPCHK(hipEventCreateWithFlags(&o.done, hipEventDisableTiming));
    else HIPCHK(hipStreamWaitEvent(st, o.done, 0));                                // the slot's previous reader, on whatever stream it ran
    // no memory for the scratch (4 bytes per packet): the launch takes the packets as they come -- slower, never wrong
    if (!o.bins && hipMalloc((void **)&o.bins, LEN_SORT_ENTRIES * sizeof(u32)) != hipSuccess) { o.bins = nullptr; (void)hipGetLastError(); *perm = nullptr; return AESGCM_OK; }
    if (o.cap < n_pkts) {
        if (o.perm) { HIPCHK(hipFree(o.perm)); o.perm = nullptr; o.cap = 0; }     // hipFree waits for the launches that may still read it
        if (hipMalloc((void **)&o.perm, n_pkts * sizeof(u32)) != hipSuccess) { o.perm = nullptr; (void)hipGetLastError(); *perm = nullptr; return AESGCM_OK; }
        o.cap = n_pkts;
    }
    hipLaunchKernelGGL(k_len_hist, dim3(LEN_SORT_WGS), dim3(256), 0, st, d_off, (u32)n_pkts, o.bins);
    hipLaunchKernelGGL(k_len_scan, dim3(1), dim3(1024), 0, st, o.bins);
    hipLaunchKernelGGL(k_len_scatter, dim3(LEN_SORT_WGS), dim3(256), 0, st, d_off, (u32)n_pkts, o.bins, o.perm);
    HIPCHK(hipGetLastError());
    *perm = o.perm;
    return AESGCM_OK;
}
static int packets_order(aesgcm_ctx *c, const u64 *d_off, size_t n_pkts, hipStream_t st, const u32 **perm, OrderSlot **slot) {
    *perm = nullptr; *slot = nullptr;
    if (!packets_ordered(c, n_pkts, true)) return AESGCM_OK;
    *slot = &c->order[c->order_next++ & 3u];
    return order_launch(**slot, d_off, n_pkts, st, perm);
}

// ---------------------------------------------------------------- many messages under the context's key: by rows (aesgcm_rows.h)
// the scratch of the path, carved out of one allocation: per message 16 + 4 + 4 bytes and (offset arrays) the two prefix sums, 32 bytes per record slot.  Zero at rest.
struct RowsScratch { RowsHdr *hdr; u32 *queues; u64 *prefix; u32 *slot_base; RowsRec *rec; unsigned long long *acc; u32 *cnt, *npieces; };
static size_t rows_carve(unsigned char *base, size_t slots, size_t n, RowsScratch *r) {
    size_t o = 0;
    auto take = [&](size_t bytes) { unsigned char *q = base ? base + o : nullptr; o += (bytes + 255) & ~(size_t)255; return q; };
    RowsScratch t;
    t.hdr = (RowsHdr *)take(sizeof(RowsHdr));
    t.queues = (u32 *)take(64 * ROWS_NQ);
    t.prefix = (u64 *)take(8 * (n + 1));
    t.slot_base = (u32 *)take(4 * (n + 1));
    t.rec = (RowsRec *)take(sizeof(RowsRec) * slots);
    t.acc = (unsigned long long *)take(16 * n);
    t.cnt = (u32 *)take(4 * n);
    t.npieces = (u32 *)take(4 * n);
    if (r) *r = t;
    return o;
}
static int rows_scratch(aesgcm_ctx *c, size_t slots, size_t n, hipStream_t st, RowsScratch *r) {
    if (slots > c->rows_cap_slots || n > c->rows_cap_n) {
        if (c->rows_buf) { HIPCHK(hipFree(c->rows_buf)); c->rows_buf = nullptr; c->rows_cap_slots = c->rows_cap_n = 0; }    // hipFree waits for the launches that may still use it
        const size_t cs = slots < 65536 ? 65536 : slots, cn = n < 4096 ? 4096 : n;
        const hipError_t e = hipMalloc((void **)&c->rows_buf, rows_carve(nullptr, cs, cn, nullptr));
        if (e == hipErrorOutOfMemory) return AESGCM_ENOMEM;
        if (e != hipSuccess) return hip_fail(e, "hipMalloc");
        c->rows_cap_slots = cs; c->rows_cap_n = cn; c->rows_dirty = true;
    }
    if (c->rows_dirty) HIPCHK(hipMemsetAsync(c->rows_buf, 0, rows_carve(nullptr, c->rows_cap_slots, c->rows_cap_n, nullptr), st));   // fresh scratch, or a launch failed half way through a call
    rows_carve(c->rows_buf, c->rows_cap_slots, c->rows_cap_n, r);
    return AESGCM_OK;
}
// p: the caller's pointers, counts and lengths; the cut and the scratch are filled in here
static int packets_rows(aesgcm_ctx *c, int decrypt, RowsParams &p, hipStream_t st) {
    const size_t n = p.n_pkts;
    RowsScratch r;
    int rc;
    const bool var = p.data_off != nullptr;
    p.has_aad = (p.aad_off || p.aad_len) ? 1u : 0u;
    u32 wgs = (u32)c->G / 2;                                                 // one 141 KiB workgroup per CU
    size_t slots;
    if (!var) {
        const RowsGeom g = rows_geom(p.pkt_len);
        p.U = rows_units(g, p.has_aad);
        p.G = (u64)n * p.U;
        const u64 need = (p.G + AESGCM_BODY_WG / 64 - 1) / (AESGCM_BODY_WG / 64);     // at least a unit per wave
        if (need < wgs) wgs = (u32)need;
        p.waves = wgs * (AESGCM_BODY_WG / 64);
        rows_cut(p.G, p.waves, c->rows_block, (u64)1 << 30, &p.D, &p.NB, &p.dyn);
        p.SM = rows_nat_count(g, p.has_aad) + (p.U - 1u) / p.D + 1u;
        if ((u64)n * p.SM >= (1ull << 31)) return AESGCM_ETOOLONG;
        slots = n * p.SM;
    } else {
        p.waves = wgs * (AESGCM_BODY_WG / 64);
        slots = 4 * n + ROWS_NB_CAP;                                         // at most 3 natural segments per message (rows, tail, AAD) and one more slot per block boundary inside it
        if (slots >= (1ull << 31)) return AESGCM_ETOOLONG;
    }
    if ((rc = rows_scratch(c, slots, n, st, &r))) return rc;
    p.slot_cap = (u32)slots;
    p.rec = r.rec; p.acc = r.acc; p.cnt = r.cnt; p.npieces = r.npieces; p.queues = r.queues;
    c->rows_dirty = true;                                                    // until both launches are enqueued
    if (var) {
        p.hdr = r.hdr; p.prefix = r.prefix; p.slot_base = r.slot_base;
        hipLaunchKernelGGL(k_rows_plan, dim3(1), dim3(1024), 0, st, p.data_off, p.n_pkts, p.has_aad, p.waves, c->rows_block, (u32)ROWS_NB_CAP, p.slot_cap, r.hdr, r.prefix, r.slot_base);
        HIPCHK(hipGetLastError());
    }
    p.prio_rows = c->cyc_prio;
#define LR(NR, M) hipLaunchKernelGGL((k_rows<NR, M>), dim3(wgs), dim3(AESGCM_BODY_WG), AESGCM_BODY_LDS, st, c->km, c->tables, p)
    if (decrypt) { if (c->nr == 10) LR(10, MODE_DEC); else if (c->nr == 12) LR(12, MODE_DEC); else LR(14, MODE_DEC); }
    else         { if (c->nr == 10) LR(10, MODE_ENC); else if (c->nr == 12) LR(12, MODE_ENC); else LR(14, MODE_ENC); }
#undef LR
    HIPCHK(hipGetLastError());
    const unsigned cw = (unsigned)((p.slot_cap + ROWS_CLOSE_WG - 1) / ROWS_CLOSE_WG);
    if (decrypt) hipLaunchKernelGGL(k_rows_close<1>, dim3(cw), dim3(ROWS_CLOSE_WG), 0, st, c->km, p);
    else hipLaunchKernelGGL(k_rows_close<0>, dim3(cw), dim3(ROWS_CLOSE_WG), 0, st, c->km, p);
    HIPCHK(hipGetLastError());
    c->rows_dirty = false;
    return AESGCM_OK;
}
// does a call go by rows?  Fixed-size records: from rows_min bytes per packet.  Offset arrays: the host does not know the lengths; the caller's pkt_len, otherwise
// unused in that form, is its word for the typical packet (0 = frames: the packet kernels)
static bool packets_by_rows(const aesgcm_ctx *c, size_t pkt_len) {
#ifdef AESGCM_DEBUG_KNOBS
    if (g_force.pkt_rows) return g_force.pkt_rows == 1;
#endif
    return c->rows_min && pkt_len >= c->rows_min;
}

// zero the output of every packet whose d_auth[] entry is 0 (behind the launch that wrote it, on the same stream)
static int wipe_failed(int device, size_t n_pkts, void *d_out, size_t pkt_len, const u64 *d_data_off, const int *d_auth, hipStream_t st) {
    if (!n_pkts || !d_auth || !d_out) return AESGCM_OK;
    HIPCHK(hipSetDevice(device));
    hipLaunchKernelGGL(k_wipe_failed, dim3((unsigned)((n_pkts + 3) / 4)), dim3(256), 0, st, (unsigned char *)d_out, d_auth, d_data_off, (u32)n_pkts, (u32)pkt_len);
    HIPCHK(hipGetLastError());
    return AESGCM_OK;
}
int aesgcm_wipe_failed_dev(int device, size_t n_pkts, void *d_out, size_t pkt_len, const uint64_t *d_data_off, const int *d_auth, void *stream) {
    if (n_pkts >= (((size_t)1) << 31) || pkt_len >= (((size_t)1) << 32)) return AESGCM_ETOOLONG;
    if (n_pkts && (!d_out || !d_auth)) return AESGCM_EARG;
    return wipe_failed(device, n_pkts, d_out, pkt_len, (const u64 *)d_data_off, d_auth, (hipStream_t)stream);
}

// ---------------------------------------------------------------- packets under the context's key
int aesgcm_packets_crypt_dev(aesgcm_ctx *c, int decrypt, size_t n_pkts, const void *d_ivs,
                             const void *d_aad, size_t aad_len, const uint64_t *d_aad_off,
                             const void *d_in, size_t pkt_len, const uint64_t *d_data_off, void *d_out,
                             void *d_tags, const void *d_expect_tags, int *d_auth, void *stream) {
    if (!c) return AESGCM_EARG;
    if (!n_pkts) return AESGCM_OK;
    if (!d_ivs || !d_tags || ((aad_len || d_aad_off) && !d_aad) || ((pkt_len || d_data_off) && (!d_in || !d_out))) return AESGCM_EARG;
    if (n_pkts >= (((size_t)1) << 31) || pkt_len >= (((size_t)1) << 28) || aad_len >= (((size_t)1) << 28)) return AESGCM_ETOOLONG;
    HIPCHK(hipSetDevice(c->device));
    if (packets_by_rows(c, pkt_len)) {                                        // message-sized packets: the rows of all of them through k_body's row loop
        RowsParams r;
        memset(&r, 0, sizeof r);
        r.ivs = (const unsigned char *)d_ivs; r.aad = (const unsigned char *)d_aad; r.in = (const unsigned char *)d_in;
        r.out = (unsigned char *)d_out; r.tags = (unsigned char *)d_tags; r.expect = (const unsigned char *)d_expect_tags; r.auth = d_auth;
        r.data_off = (const u64 *)d_data_off; r.aad_off = (const u64 *)d_aad_off;
        r.n_pkts = (u32)n_pkts; r.pkt_len = (u32)pkt_len; r.aad_len = (u32)aad_len;
        const int rc = packets_rows(c, decrypt, r, pick_stream(c, stream));
        if (!rc && decrypt && c->wipe_on_auth_fail && d_expect_tags) return wipe_failed(c->device, n_pkts, d_out, pkt_len, (const u64 *)d_data_off, d_auth, pick_stream(c, stream));
        return rc;
    }
    PktParams p;
    memset(&p, 0, sizeof p);
    p.ivs = (const unsigned char *)d_ivs; p.aad = (const unsigned char *)d_aad; p.in = (const unsigned char *)d_in;
    p.out = (unsigned char *)d_out; p.tags = (unsigned char *)d_tags; p.expect = (const unsigned char *)d_expect_tags; p.auth = d_auth;
    p.data_off = (const u64 *)d_data_off; p.aad_off = (const u64 *)d_aad_off;
    p.n_pkts = (u32)n_pkts; p.pkt_len = (u32)pkt_len; p.aad_len = (u32)aad_len;
    p.aligned = (((uintptr_t)d_in | (uintptr_t)d_out) & 15) == 0 && (d_data_off || pkt_len % 16 == 0);
    const u32 n_cu = (u32)c->G / 2;                                                 // c->G = two workgroups per CU
    int lg = packets_pick_lg(n_cu, n_pkts, pkt_len, d_data_off != nullptr, packets_ordered(c, n_pkts, d_data_off != nullptr));
#ifdef AESGCM_DEBUG_KNOBS
    if (g_force.pkt_lanes) lg = g_force.pkt_lanes == 1 ? 0 : g_force.pkt_lanes == 64 ? 6 : g_force.pkt_lanes == 16 ? 4 : g_force.pkt_lanes == 8 ? 3 : 2;
#endif
    const int shape = lg == 0 ? 'l' : lg == 6 ? 'w' : 'g';
    hipStream_t st = pick_stream(c, stream);
    p.counter = c->d_counter; p.counter_base = c->counter_base;
    OrderSlot *oslot = nullptr;
    if (d_data_off) { const int rc = packets_order(c, (const u64 *)d_data_off, n_pkts, st, &p.perm, &oslot); if (rc) return rc; }
    if (shape == 'l') {
        const u32 nb = (u32)((n_pkts + 63) / 64);
        // the ILP form (512-lane workgroups, eight independent keystream chains per line) while the packets fit one round of it; its workgroups are spread over
        // all CUs, a wave of 64 packets each first
        // Measured, AES-256, GiB/s 768-lane form / ILP form (profiles/r04/packets_sweep_ilp_aes256.txt): 1 KiB packets 16384 66 / 78, 65536 255 / 306, 131072 481 / 592;
        // 256 B 32768 99 / 95, 98304 245 / 266, 131072 295 / 330; 64 B (no whole line to work on) 16384 27 / 19.
        // Packets shorter than two lines gain from it only once they fill the chip (fewer, fatter waves): 196608 x 256 B 380 / 414, 262144 442 / 460 (2^20: 682 / 642);
        // 64 B 196608 127 / 146, 393216 183 / 201, 2^20 254 / 266.
        bool ilp = n_pkts <= (size_t)n_cu * AESGCM_PKTL_WG_ILP ? (d_data_off || pkt_len >= 512 || (pkt_len >= 256 && n_pkts >= 49152))
                                                                : (!d_data_off && n_pkts >= (size_t)n_cu * AESGCM_PKTL_WG && (pkt_len <= 64 || (pkt_len <= 256 && n_pkts <= 300000)));
#ifdef AESGCM_DEBUG_KNOBS
        if (g_force.pkt_ilp) ilp = g_force.pkt_ilp == 1;
#endif
        const u32 waves_per_wg = (ilp ? AESGCM_PKTL_WG_ILP : AESGCM_PKTL_WG) / 64;
        u32 wgs = ilp ? nb : (nb + waves_per_wg - 1) / waves_per_wg;
        if (wgs > n_cu) wgs = n_cu;                                                  // one workgroup per CU (registers, and with four T-tables the LDS)
        c->counter_base += nb + wgs * waves_per_wg;                                 // every wave ends on one failing fetch
#define LPI(NR, D, I) hipLaunchKernelGGL((k_pktl<NR, D, I>), dim3(wgs), dim3(I ? AESGCM_PKTL_WG_ILP : AESGCM_PKTL_WG), AESGCM_PKTL_LDS, st, c->km, c->tables, p)
#define LP(NR, D) do { if (ilp) LPI(NR, D, 1); else LPI(NR, D, 0); } while (0)
        if (decrypt) { if (c->nr == 10) LP(10, 1); else if (c->nr == 12) LP(12, 1); else LP(14, 1); }
        else         { if (c->nr == 10) LP(10, 0); else if (c->nr == 12) LP(12, 0); else LP(14, 0); }
#undef LPI
#undef LP
    } else {
        const u32 P = 64u >> lg;                                                    // packets per wave-iteration
        p.plain = (lg == 6 || lg == 2) && !d_data_off && !d_aad_off && !aad_len && p.aligned && pkt_len && pkt_len % ((size_t)16 << lg) == 0;
        const u32 waves_per_wg = (u32)PKTG_WG(lg) / 64;
        // deal: about 4 dispenser fetches per resident wave, a multiple of P, at most 64 packets (one E_K(J0) pass per fetch)
        u32 deal = (u32)(n_pkts / ((size_t)n_cu * waves_per_wg * 4));
        deal = deal / P * P;
        deal = deal < P ? P : deal > PKTG_MAX_DEAL ? PKTG_MAX_DEAL : deal;
#ifdef AESGCM_DEBUG_KNOBS
        if (g_force.pkt_deal >= 1 && g_force.pkt_deal <= (int)PKTG_MAX_DEAL) deal = ((u32)g_force.pkt_deal + P - 1) / P * P;
#endif
        p.deal = deal;
        const u32 nb = (u32)((n_pkts + deal - 1) / deal);
        u32 wgs = (nb + waves_per_wg - 1) / waves_per_wg;
        if (wgs > n_cu) wgs = n_cu;                                                  // one workgroup per CU (LDS)
        c->counter_base += nb + wgs * waves_per_wg;                                 // every wave ends on one failing fetch
#define LPG(NR, D, LG) hipLaunchKernelGGL((k_pktg<NR, D, LG>), dim3(wgs), dim3(PKTG_WG(LG)), PKTG_LDS_TOTAL(LG), st, c->km, c->tables, p)
#define LP(NR, D) do { if (lg == 2) LPG(NR, D, 2); else if (lg == 3) LPG(NR, D, 3); else if (lg == 4) LPG(NR, D, 4); else LPG(NR, D, 6); } while (0)
        if (decrypt) { if (c->nr == 10) LP(10, 1); else if (c->nr == 12) LP(12, 1); else LP(14, 1); }
        else         { if (c->nr == 10) LP(10, 0); else if (c->nr == 12) LP(12, 0); else LP(14, 0); }
#undef LP
#undef LPG
    }
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) { c->counter_base = p.counter_base; return hip_fail(le, "k_pkt launch"); }
    if (oslot && p.perm) HIPCHK(hipEventRecord(oslot->done, st));
    if (decrypt && c->wipe_on_auth_fail && d_expect_tags) return wipe_failed(c->device, n_pkts, d_out, pkt_len, (const u64 *)d_data_off, d_auth, st);
    return AESGCM_OK;
}

// ---------------------------------------------------------------- batch (per-packet key and IV)
// Variable-length batches by length class: mixed frames of 64 .. 1514 bytes, GiB/s in array order / by class (profiles/r04/batch_mixed_*.txt): AES-128 65536 packets
// 202 / 178, 262144 313 / 307, 393216 332 / 342, 2^20 362 / 493; AES-256 65536 177 / 167, 98304 205 / 213, 262144 266 / 290, 2^20 301 / 437.
#define BATCH_ORDER_MIN(nr) ((nr) == 10 ? 262144u : 98304u)
static int batch_launch(int device, int decrypt, size_t n_pkts, size_t key_len, BatchParams &p, void *stream) {
    if (key_len != 16 && key_len != 24 && key_len != 32) return AESGCM_EKEYLEN;
    if (n_pkts >= (((size_t)1) << 31)) return AESGCM_ETOOLONG;
    DeviceState *ds;
    int rc = device_state(device, &ds);
    if (rc) return rc;
    if ((rc = set_lds_attrs(device, ds))) return rc;
    HIPCHK(hipSetDevice(device));
    p.n_pkts = (u32)n_pkts;
    u32 wgs = 0;
    {   // a fresh dispenser per launch (zeroed on the launch stream), so launches on different streams may overlap
        std::lock_guard<std::mutex> lk(g_mu);
        p.counter = ds->batch_counter + (ds->batch_slot++ % BATCH_DISPENSERS);
        p.counter_base = 0;
    }
    const int nr = (int)(key_len / 4 + 6);
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(p.counter, 0, 4, st));
    int lg = batch_pick_lg(ds->n_cu, n_pkts, p.pkt_len, p.data_off != nullptr);          // k_batch3 with 8 / 16 / 64 lanes per packet
#ifdef AESGCM_DEBUG_KNOBS
    if (g_force.batch_lanes) lg = g_force.batch_lanes == 8 ? 3 : g_force.batch_lanes == 16 ? 4 : 6;
#endif
    if (lg <= 6) {
        // packets of mixed length: by falling length class once the batch fills the machine several times over (BATCH_ORDER_MIN; as aesgcm_packets_crypt_dev)
        OrderSlot *oslot = nullptr;
        bool ordered = lg < 6 && p.data_off && n_pkts >= BATCH_ORDER_MIN(nr);
#ifdef AESGCM_DEBUG_KNOBS
        if (g_force.batch_order) ordered = lg < 6 && p.data_off && g_force.batch_order == 1;
#endif
        std::unique_lock<std::mutex> order_lock(g_mu, std::defer_lock);             // held from the choice of the slot to the event behind its reader: callers on other threads queue up here
        if (ordered) {
            order_lock.lock();
            oslot = &ds->order[ds->order_next++ & 3u];
            if ((rc = order_launch(*oslot, p.data_off, n_pkts, st, &p.perm))) return rc;
        }
        p.plain = !p.data_off && !p.aad_off && !p.aad_len && p.aligned && p.pkt_len && p.pkt_len % (16u << lg) == 0;
        const u32 waves_per_wg = (u32)BATCH3_LANES(nr) / 64;
        const u32 P = 64u >> lg, per_wg = waves_per_wg * P;
        wgs = (u32)((n_pkts + per_wg - 1) / per_wg);
        if (wgs > (u32)ds->n_cu) wgs = (u32)ds->n_cu;
        u32 deal = (u32)(n_pkts / ((size_t)wgs * waves_per_wg * 16));
        deal = deal < P ? P : deal > 8 * P ? 8 * P : (deal + P - 1) / P * P;
#ifdef AESGCM_DEBUG_KNOBS
        if (g_force.batch_deal >= 1 && g_force.batch_deal <= 4096) deal = ((u32)g_force.batch_deal + P - 1) / P * P;
#endif
        p.deal = deal;
#define LB3(NR, D, LG) hipLaunchKernelGGL((k_batch3<NR, D, LG>), dim3(wgs), dim3(BATCH3_LANES(NR)), BATCH3_LDS_BYTES_LG(LG), st, ds->tables, p)
#define LB3N(D, LG) do { if (nr == 10) LB3(10, D, LG); else if (nr == 12) LB3(12, D, LG); else LB3(14, D, LG); } while (0)
        if (lg == 3) { if (decrypt) LB3N(1, 3); else LB3N(0, 3); }
        else if (lg == 4) { if (decrypt) LB3N(1, 4); else LB3N(0, 4); }
        else { if (decrypt) LB3N(1, 6); else LB3N(0, 6); }
#undef LB3N
#undef LB3
        HIPCHK(hipGetLastError());
        if (oslot && p.perm) HIPCHK(hipEventRecord(oslot->done, st));
        return AESGCM_OK;
    }
    return AESGCM_EARG;                                         // batch_pick_lg gives 3, 4 or 6
}

int aesgcm_batch_crypt_dev(int device, int decrypt, size_t n_pkts, size_t key_len, const void *d_keys, const void *d_ivs,
                           const void *d_aad, size_t aad_len, const void *d_in, size_t pkt_len, void *d_out,
                           void *d_tags, const void *d_expect_tags, int *d_auth, void *stream) {
    if (!n_pkts) return AESGCM_OK;
    if (!d_keys || !d_ivs || !d_tags || (aad_len && !d_aad) || (pkt_len && (!d_in || !d_out))) return AESGCM_EARG;
    if (pkt_len >= (((size_t)1) << 28) || aad_len >= (((size_t)1) << 28)) return AESGCM_ETOOLONG;
    BatchParams p;
    memset(&p, 0, sizeof p);
    p.keys = (const unsigned char *)d_keys; p.ivs = (const unsigned char *)d_ivs; p.aad = (const unsigned char *)d_aad;
    p.in = (const unsigned char *)d_in; p.out = (unsigned char *)d_out; p.tags = (unsigned char *)d_tags;
    p.expect = (const unsigned char *)d_expect_tags; p.auth = d_auth;
    p.pkt_len = (u32)pkt_len; p.aad_len = (u32)aad_len;
    p.aligned = (pkt_len % 16 == 0) && (((uintptr_t)d_in | (uintptr_t)d_out) & 15) == 0;
    return batch_launch(device, decrypt, n_pkts, key_len, p, stream);
}

int aesgcm_batch_crypt_var_dev(int device, int decrypt, size_t n_pkts, size_t key_len, const void *d_keys, const void *d_ivs,
                               const void *d_aad, const uint64_t *d_aad_off, const void *d_in, const uint64_t *d_data_off,
                               void *d_out, void *d_tags, const void *d_expect_tags, int *d_auth, void *stream) {
    if (!n_pkts) return AESGCM_OK;
    if (!d_keys || !d_ivs || !d_tags || !d_data_off || !d_in || !d_out || (d_aad_off && !d_aad)) return AESGCM_EARG;
    BatchParams p;
    memset(&p, 0, sizeof p);
    p.keys = (const unsigned char *)d_keys; p.ivs = (const unsigned char *)d_ivs; p.aad = d_aad_off ? (const unsigned char *)d_aad : nullptr;
    p.in = (const unsigned char *)d_in; p.out = (unsigned char *)d_out; p.tags = (unsigned char *)d_tags;
    p.expect = (const unsigned char *)d_expect_tags; p.auth = d_auth;
    p.data_off = (const u64 *)d_data_off; p.aad_off = (const u64 *)d_aad_off;
    p.aligned = (((uintptr_t)d_in | (uintptr_t)d_out) & 15) == 0;      // per packet: and its offset is a multiple of 16
    return batch_launch(device, decrypt, n_pkts, key_len, p, stream);
}

// Which kernel shape a call with these arguments takes (lanes per packet: 1 = one lane per packet, 4 / 8 / 16 = a lane group, 64 = a whole wave); pkt_len = 0
// with var_len != 0 describes the offset-array forms.  What bench.py and the profiling scripts print beside their numbers.
int aesgcm_batch_shape(int device, size_t n_pkts, size_t pkt_len, int var_len, int *lanes_per_packet) {
    if (!lanes_per_packet || !n_pkts) return AESGCM_EARG;
    DeviceState *ds;
    int rc = device_state(device, &ds);
    if (rc) return rc;
    int lg = batch_pick_lg(ds->n_cu, n_pkts, pkt_len, var_len != 0);
#ifdef AESGCM_DEBUG_KNOBS
    if (g_force.batch_lanes) lg = g_force.batch_lanes == 8 ? 3 : g_force.batch_lanes == 16 ? 4 : 6;
#endif
    *lanes_per_packet = 1 << lg;
    return AESGCM_OK;
}
int aesgcm_packets_shape(const aesgcm_ctx *c, size_t n_pkts, size_t pkt_len, int var_len, int *lanes_per_packet) {
    if (!c || !lanes_per_packet || !n_pkts) return AESGCM_EARG;
    if (packets_by_rows(c, pkt_len)) { *lanes_per_packet = AESGCM_SHAPE_ROWS; return AESGCM_OK; }
    int lg = packets_pick_lg((u32)c->G / 2, n_pkts, pkt_len, var_len != 0, packets_ordered(c, n_pkts, var_len != 0));
#ifdef AESGCM_DEBUG_KNOBS
    if (g_force.pkt_lanes) lg = g_force.pkt_lanes == 1 ? 0 : g_force.pkt_lanes == 64 ? 6 : g_force.pkt_lanes == 16 ? 4 : g_force.pkt_lanes == 8 ? 3 : 2;
#endif
    *lanes_per_packet = 1 << lg;
    return AESGCM_OK;
}

// ---------------------------------------------------------------- pipelined host-buffer path
// H2D of chunk k+1, the fused kernel on chunk k and D2H of chunk k-1 overlap on three streams; the GHASH
// value is carried from chunk to chunk on the device (Y' = Y*H^blocks ^ P, the same combine the beat-by-beat
// interface uses), so the result is bit-identical to one launch over the whole message.
static void pipeline_release(aesgcm_ctx *c) {
    for (int i = 0; i < 2; i++) {
        if (c->pl_buf[i]) { hipFree(c->pl_buf[i]); c->pl_buf[i] = nullptr; }
        if (c->pl_ev_h2d[i]) { hipEventDestroy(c->pl_ev_h2d[i]); c->pl_ev_h2d[i] = nullptr; }
        if (c->pl_ev_k[i]) { hipEventDestroy(c->pl_ev_k[i]); c->pl_ev_k[i] = nullptr; }
        if (c->pl_ev_d2h[i]) { hipEventDestroy(c->pl_ev_d2h[i]); c->pl_ev_d2h[i] = nullptr; }
    }
    if (c->pl_in) { hipStreamDestroy(c->pl_in); c->pl_in = nullptr; }
    if (c->pl_out) { hipStreamDestroy(c->pl_out); c->pl_out = nullptr; }
    c->pl_cap = 0;
}
// all or nothing: either both streams, all six events and both chunk slots of `chunk` bytes exist afterwards, or none
// of them does (pl_in == NULL, pl_cap == 0) and the next call starts from scratch
static int pipeline_prepare(aesgcm_ctx *c, size_t chunk) {
    hipError_t e = hipSuccess;
    if (!c->pl_in) {
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->pl_in, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->pl_out, hipStreamNonBlocking);
        for (int i = 0; i < 2 && e == hipSuccess; i++) {
            e = hipEventCreateWithFlags(&c->pl_ev_h2d[i], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&c->pl_ev_k[i], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&c->pl_ev_d2h[i], hipEventDisableTiming);
        }
        if (e != hipSuccess) { pipeline_release(c); return hip_fail(e, "pipeline streams/events"); }
    }
    if (chunk > c->pl_cap) {
        c->pl_cap = 0;
        for (int i = 0; i < 2 && e == hipSuccess; i++) {
            if (c->pl_buf[i]) { e = hipFree(c->pl_buf[i]); c->pl_buf[i] = nullptr; }
            if (e == hipSuccess) e = hipMalloc((void **)&c->pl_buf[i], chunk);
        }
        if (e != hipSuccess) {
            pipeline_release(c);
            return e == hipErrorOutOfMemory ? AESGCM_ENOMEM : hip_fail(e, "pipeline chunk slots");
        }
        c->pl_cap = chunk;
    }
    return AESGCM_OK;
}

static int crypt_pipelined(aesgcm_ctx *c, int dec, const uint8_t iv[12], const uint8_t *aad, size_t aad_len,
                           const uint8_t *in, size_t len, uint8_t *out, uint8_t tag[16], size_t chunk) {
    int rc = check_lengths(aad_len, len);
    if (rc) return rc;
    // the chunk-to-chunk GHASH value lives in the streaming slot (d_tag[1], s_iv, s_dec): refuse to run inside an open
    // stream_begin .. stream_final session instead of silently corrupting its running GHASH
    if (c->s_active) return AESGCM_ESTATE;
    if (!chunk) chunk = (size_t)64 << 20;
    chunk = (chunk + 1023) / 1024 * 1024;                    // whole rows, 16-byte aligned chunk starts
    if (chunk > len) chunk = (len + 1023) / 1024 * 1024;
    if (!chunk) chunk = 1024;
    HIPCHK(hipSetDevice(c->device));
    if ((rc = pipeline_prepare(c, chunk))) return rc;
    // state Y <- 0, then the AAD (small; through the staging buffer on the compute stream)
    memcpy(c->s_iv, iv, 12);
    c->s_dec = dec ? 1 : 0;
    HIPCHK(hipMemsetAsync(c->d_tag + 1, 0, 16, c->stream));
    if (aad_len) {
        if ((rc = stage_in(c, aad, aad_len, nullptr, 0))) return rc;
        if ((rc = stream_absorb(c, c->st_aad, aad_len, c->st_in, 0, c->st_out, 0))) return rc;
    }
    const size_t n_chunks = (len + chunk - 1) / chunk;
    for (size_t k = 0; k < n_chunks; k++) {
        const int s = (int)(k & 1);
        const size_t off = k * chunk, m = (len - off < chunk) ? len - off : chunk;
        if (k >= 2) HIPCHK(hipStreamWaitEvent(c->pl_in, c->pl_ev_d2h[s], 0));     // slot free again
        HIPCHK(hipMemcpyAsync(c->pl_buf[s], in + off, m, hipMemcpyHostToDevice, c->pl_in));
        HIPCHK(hipEventRecord(c->pl_ev_h2d[s], c->pl_in));
        HIPCHK(hipStreamWaitEvent(c->stream, c->pl_ev_h2d[s], 0));
        if ((rc = stream_absorb(c, nullptr, 0, c->pl_buf[s], m, c->pl_buf[s], off / 16))) return rc;   // in place
        HIPCHK(hipEventRecord(c->pl_ev_k[s], c->stream));
        HIPCHK(hipStreamWaitEvent(c->pl_out, c->pl_ev_k[s], 0));
        HIPCHK(hipMemcpyAsync(out + off, c->pl_buf[s], m, hipMemcpyDeviceToHost, c->pl_out));
        HIPCHK(hipEventRecord(c->pl_ev_d2h[s], c->pl_out));
    }
    if ((rc = enqueue_combine(c, plan_combine_final(c->d_tag + 1, iv, aad_len, len, c->d_tag), c->stream))) return rc;
    if ((rc = fetch_tag(c, c->stream, tag))) return rc;
    HIPCHK(hipStreamSynchronize(c->pl_out));
    return AESGCM_OK;
}

int aesgcm_encrypt_pipelined(aesgcm_ctx *c, const uint8_t iv[12], const uint8_t *aad, size_t aad_len,
                             const uint8_t *pt, size_t len, uint8_t *ct, uint8_t tag[16], size_t chunk_bytes) {
    if (!c || !iv || !tag || (aad_len && !aad) || (len && (!pt || !ct))) return AESGCM_EARG;
    return crypt_pipelined(c, 0, iv, aad, aad_len, pt, len, ct, tag, chunk_bytes);
}
int aesgcm_decrypt_pipelined(aesgcm_ctx *c, const uint8_t iv[12], const uint8_t *aad, size_t aad_len,
                             const uint8_t *ct, size_t len, uint8_t *pt, const uint8_t *expect_tag, uint8_t tag_out[16],
                             size_t chunk_bytes) {
    if (!c || !iv || (aad_len && !aad) || (len && (!ct || !pt))) return AESGCM_EARG;
    uint8_t t[16];
    int rc = crypt_pipelined(c, 1, iv, aad, aad_len, ct, len, pt, t, chunk_bytes);
    if (rc) return rc;
    if (tag_out) memcpy(tag_out, t, 16);
    if (expect_tag && !ct_compare16(t, expect_tag)) {
        if (c->wipe_on_auth_fail && len) {                       // the chunks have landed in `pt` already (that is the pipeline): wipe them, and the device's two chunk slots
            memset(pt, 0, len);
            for (int i = 0; i < 2; i++) if (c->pl_buf[i]) HIPCHK(hipMemsetAsync(c->pl_buf[i], 0, c->pl_cap, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
        }
        return AESGCM_EAUTH;
    }
    return AESGCM_OK;
}
// page-locked host memory, so that the pipelined path's copies are true DMA (pageable buffers work, slower)
int aesgcm_host_alloc(void **p, size_t bytes) {
    if (!p) return AESGCM_EARG;
    hipError_t e = hipHostMalloc(p, bytes ? bytes : 16, hipHostMallocDefault);
    if (e == hipErrorOutOfMemory) return AESGCM_ENOMEM;
    if (e != hipSuccess) return hip_fail(e, "hipHostMalloc");
    return AESGCM_OK;
}
int aesgcm_host_free(void *p) {
    HIPCHK(hipHostFree(p));
    return AESGCM_OK;
}

// ---------------------------------------------------------------- memory helpers
int aesgcm_dev_alloc(int device, void **p, size_t bytes) {
    if (!p) return AESGCM_EARG;
    HIPCHK(hipSetDevice(device));
    hipError_t e = hipMalloc(p, bytes ? bytes : 16);
    if (e == hipErrorOutOfMemory) return AESGCM_ENOMEM;
    if (e != hipSuccess) return hip_fail(e, "hipMalloc");
    return AESGCM_OK;
}
int aesgcm_dev_free(int device, void *p) {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipFree(p));
    return AESGCM_OK;
}
int aesgcm_dev_upload(int device, void *d, const void *h, size_t n) {
    HIPCHK(hipSetDevice(device));
    if (n) HIPCHK(hipMemcpy(d, h, n, hipMemcpyHostToDevice));
    return AESGCM_OK;
}
int aesgcm_dev_download(int device, void *h, const void *d, size_t n) {
    HIPCHK(hipSetDevice(device));
    if (n) HIPCHK(hipMemcpy(h, d, n, hipMemcpyDeviceToHost));
    return AESGCM_OK;
}
int aesgcm_dev_sync(int device) {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipDeviceSynchronize());
    return AESGCM_OK;
}
int aesgcm_dev_copy(int device, void *d_dst, const void *d_src, size_t bytes, void *stream) {
    if (bytes && (!d_dst || !d_src)) return AESGCM_EARG;
    if (((uintptr_t)d_dst | (uintptr_t)d_src | bytes) & 15) return AESGCM_EALIGN;
    if (!bytes) return AESGCM_OK;
    HIPCHK(hipSetDevice(device));
    const u64 n16 = bytes / 16;
    if ((n16 + 255) / 256 > 0x7FFFFFFFull) return AESGCM_ETOOLONG;
    hipLaunchKernelGGL(k_copy16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (uint4 *)d_dst, (const uint4 *)d_src, n16);
    HIPCHK(hipGetLastError());
    return AESGCM_OK;
}
int aesgcm_fill_splitmix64_dev(int device, void *d_buf, size_t len, uint64_t seed, uint64_t first_word, void *stream) {
    if (len && !d_buf) return AESGCM_EARG;
    if ((uintptr_t)d_buf & 7) return AESGCM_EALIGN;
    if (!len) return AESGCM_OK;
    HIPCHK(hipSetDevice(device));
    size_t nw = len / 8;
    size_t blocks = (nw + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_fill_splitmix64, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (u64 *)d_buf, nw, len & 7, seed, first_word);
    HIPCHK(hipGetLastError());
    return AESGCM_OK;
}

// ---------------------------------------------------------------- timing
// a pair of HIP events for callers that time launches of the context-free entry points (batch) on the stream they use
struct aesgcm_timer { int device; hipEvent_t a, b; };
int aesgcm_timer_create(aesgcm_timer **out, int device) {
    if (!out) return AESGCM_EARG;
    *out = nullptr;
    HIPCHK(hipSetDevice(device));
    aesgcm_timer *t = new (std::nothrow) aesgcm_timer();
    if (!t) return AESGCM_ENOMEM;
    t->device = device; t->a = t->b = nullptr;
    hipError_t e = hipEventCreate(&t->a);
    if (e == hipSuccess) e = hipEventCreate(&t->b);
    if (e != hipSuccess) { if (t->a) hipEventDestroy(t->a); delete t; return hip_fail(e, "hipEventCreate"); }
    *out = t;
    return AESGCM_OK;
}
int aesgcm_timer_start(aesgcm_timer *t, void *stream) { if (!t) return AESGCM_EARG; HIPCHK(hipSetDevice(t->device)); HIPCHK(hipEventRecord(t->a, (hipStream_t)stream)); return AESGCM_OK; }
int aesgcm_timer_stop(aesgcm_timer *t, void *stream) { if (!t) return AESGCM_EARG; HIPCHK(hipSetDevice(t->device)); HIPCHK(hipEventRecord(t->b, (hipStream_t)stream)); return AESGCM_OK; }
int aesgcm_timer_ms(aesgcm_timer *t, double *ms) {
    if (!t || !ms) return AESGCM_EARG;
    HIPCHK(hipSetDevice(t->device));
    HIPCHK(hipEventSynchronize(t->b));
    float f = 0;
    HIPCHK(hipEventElapsedTime(&f, t->a, t->b));
    *ms = f;
    return AESGCM_OK;
}
int aesgcm_timer_destroy(aesgcm_timer *t) {
    if (!t) return AESGCM_OK;
    hipSetDevice(t->device);
    hipEventDestroy(t->a); hipEventDestroy(t->b);
    delete t;
    return AESGCM_OK;
}

int aesgcm_ctx_timing_enable(aesgcm_ctx *c, int on) {
    if (!c) return AESGCM_EARG;
    c->timing = on != 0;
    return AESGCM_OK;
}
int aesgcm_ctx_timing_read(aesgcm_ctx *c, uint64_t *n, double *total_ms, int reset) {
    if (!c) return AESGCM_EARG;
    HIPCHK(hipSetDevice(c->device));
    double tot = 0;
    for (auto &e : c->ev) {
        HIPCHK(hipEventSynchronize(e.second));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e.first, e.second));
        tot += ms;
    }
    if (n) *n = c->ev.size();
    if (total_ms) *total_ms = tot;
    if (reset) { for (auto &e : c->ev) c->ev_pool.push_back(e); c->ev.clear(); }
    return AESGCM_OK;
}

// The fused kernel's instruction stream WITHOUT its HBM traffic: k_body<NR, MODE_PROBE> over a virtual range of `nbytes`
// (same chunking, same dispensers, same LDS tables, same scalar loads, same GHASH; no global load, no global store
// except the chunk items).  Its time is the ceiling of the T-table formulation on this chip at this moment's clocks.
int aesgcm_ctx_ceiling_probe(aesgcm_ctx *c, size_t nbytes, double *ms, uint64_t *blocks) {
    if (!c || !ms) return AESGCM_EARG;
    HIPCHK(hipSetDevice(c->device));
    BodySplit b;
    if (!plan_body_split(nbytes, 0, c->tw_override, 0, &b)) return AESGCM_EARG;
    const uint8_t iv[12] = {0};
    Partials pp;
    const bool was = c->timing;
    c->timing = true;
    HIPCHK(hipStreamSynchronize(c->stream));
    const size_t mark = c->ev.size();
    int rc = enqueue_body(c, MODE_PROBE, iv, b, nullptr, nullptr, 0, c->stream, &pp);
    c->timing = was;
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    float t = 0;
    HIPCHK(hipEventElapsedTime(&t, c->ev[mark].first, c->ev[mark].second));
    c->ev_pool.push_back(c->ev[mark]);
    c->ev.erase(c->ev.begin() + (long)mark);
    *ms = t;
    if (blocks) *blocks = b.body_blocks;
    return AESGCM_OK;
}

int aesgcm_ctx_wg_trace(aesgcm_ctx *c, uint64_t *out, size_t max_wgs, size_t *n_wgs) {
    if (!c || !out || !n_wgs) return AESGCM_EARG;
    HIPCHK(hipSetDevice(c->device));
    size_t n = c->last_np < max_wgs ? c->last_np : max_wgs;
    HIPCHK(hipDeviceSynchronize());
    if (n) HIPCHK(hipMemcpy(out, c->d_trace, n * 4 * sizeof(u64), hipMemcpyDeviceToHost));
    *n_wgs = n;
    return AESGCM_OK;
}

}  // extern "C"
