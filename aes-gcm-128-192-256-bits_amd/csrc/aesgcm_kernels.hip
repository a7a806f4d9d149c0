// aesgcm_kernels.hip -- the HIP kernels (gfx950) and their launchers (klaunch_*, at the end of the file; aesgcm_internal.h says how the library is cut).
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
//   k_len_hist, k_len_scan, k_len_scatter (k_len_sort1: small calls, one launch)   a counting sort by falling size class; with a route, the scan DECIDES per message.
//   k_gfmul, k_fill_splitmix64, k_copy16   small utility kernels.
//
// GHASH re-association (DESIGN.md "GHASH as a polynomial"): the GHASH input sequence
// A_0..A_{u-1}, C_0..C_{c-1} (n = u + c blocks) is right-aligned into rows of 64 slots (front padding =
// zeros, which do not change a polynomial) and cut into chunks of Tw rows.  Waves pull chunks from atomic
// dispensers; inside a chunk lane L runs Horner over its column with the per-key constant K = H^64
// (acc = acc*K ^ X) and the wave stores the 64 accumulators as the chunk's item.  k_fold folds the items
// lane-wise (B_L = sum_i item_i[L] * H^(blocks to the end)), k_combine forms P = sum_L B_L * H^(63-L).
// The tag is (P*H ^ L)*H ^ E_K(J0) = P*H^2 ^ L*H ^ E_K(J0).
#include "aesgcm_internal.h"

#include <stddef.h>
#include <stdint.h>

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
// 16 GiB buffers on the same box (profiles/archive/r03/copy_variants.txt: tiles of 4 .. 16 loads in flight per lane, nontemporal
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
// profiles/archive/r02g/cfg5_batch -- for A/B runs; round 4 deleted it.  HISTORY.md.)
// ------------------------------------------------------------------------------------------------
// Lanes per k_batch3 workgroup (one per CU).  The first round-3 build (1024 lanes, 128 registers, round keys in VGPRs) spilled 104 - 124 bytes around its
// packet loop and moved 11.1e9 bytes against 8.64e9 algorithmic; with 768-lane workgroups (160 registers, no scratch) 8.68e9 at the same speed -- the
// extra traffic was scratch (profiles/archive/r03/batch3_wg768_ab.txt).  What was being spilled was bookkeeping, as in k_pktg: the packet's H and E_K(J0) held
// across the block loop (now in the group's LDS slot), the lane's position (lane_id_fresh behind the loop), ds_bpermute index registers (ds_swizzle).
// Without them every instance fits 98 - 115 registers at 1024 lanes with ScratchSize 0.  BATCH3_WG forces another geometry.
// LG = 4: 16 lanes per packet, four packets per wave, two table slots per packet (the closing alternates between them).  LG = 3: 8 lanes per packet, eight
// packets per wave -- what a wave-iteration pays once (key schedule, H and E_K(J0), the H^8 table, the closing) now serves eight packets, and the tree is a
// level shorter; 128 packets per workgroup leave LDS for ONE table slot each, so the closing rebuilds that slot between its multiplies.
// Which lanes are a packet (round 4).  A ds_read_b128 is served in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the
// same + 32 (MI355X_MICROARCH.md, LDS) -- and a packet's Shoup table is a full 256-byte bank row, so lanes of DIFFERENT packets in one service group
// collide whenever they pick the same bank with different entries.  With packets on consecutive lanes a service group holds FOUR packets at 8 lanes per
// packet (lanes 0-3, 12-15, 20-23, 24-27) and two at 16: profiles/archive/r03e/cfg5_n1 counted 9.1e8 conflict cycles in 2.83e9 LDS-array cycles, a table read
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
template <int NR, int DEC, int LG>                       // DEC: 0 encrypt, 1 decrypt, 2 = encrypt WITHOUT the data's loads and stores (aesgcm_batch_ceiling_probe_dev: what the formulation costs by itself)
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
                const uint4 x = DEC == 2 ? make_uint4(l, k, pkt, 0u) : gload16(src);     // (DEC == 2, the PROBE: the same instruction stream without the data's HBM traffic)
                u32 s0, s1, s2, s3;
                ctr_rounds_lds<NR>(bswap32(2u + k * G + l), cc, s0, s1, s2, s3, rk, smem, lb);
                const uint4 y = make_uint4(x.x ^ s0, x.y ^ s1, x.z ^ s2, x.w ^ s3);
                if (act && DEC != 2) gstore16(dst, y);
                const G128 b = mo_to_be(DEC == 1 ? x : y);           // aes_gcm.vhd:207-211
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
                if (DEC == 2) x = make_uint4(l, i, pkt, 0u);
                else if (full) x = aligned ? gload16(in + off) : gload16_any(in + off);
                else x = load_block_bytes(in + off, rem < 16 ? rem : 16);
                u32 s0, s1, s2, s3;
                ctr_rounds_lds<NR>(bswap32(2u + i), cc, s0, s1, s2, s3, rk, smem, lb);
                uint4 y = make_uint4(x.x ^ s0, x.y ^ s1, x.z ^ s2, x.w ^ s3);
                if (rem < 16) y = mask_block(y, rem);
                if (act && DEC != 2) {
                    if (full) { if (aligned) gstore16(out + off, y); else gstore16_any(out + off, y); }
                    else store_block_bytes(out + off, y, rem < 16 ? rem : 16);
                }
                gin = DEC == 1 ? x : y;                          // aes_gcm.vhd:207-211
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
            if (DEC == 1 && p.auth) {
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
__device__ __forceinline__ void len_sort_slice(u32 n, u32 &lo, u32 &hi) {
    const u32 per = (n + LEN_SORT_WGS - 1u) / LEN_SORT_WGS;
    lo = blockIdx.x * per < n ? blockIdx.x * per : n;
    hi = lo + per < n ? lo + per : n;
}
// bad_part != NULL (a routed call): the launch also CHECKS every length it reads -- data or AAD of 2^28 bytes or more, offsets that do not rise (the difference wraps) --
// and leaves the first such message of its slice (or ~0) for k_len_scan, which refuses the call before anything else has run
// (the lengths of LEN_SORT_BATCH messages are requested before the first of them is counted: a thread's sixteen messages of a 2^20-frame call were sixteen
// round trips to memory one behind the other, 23 us of launch; profiles/r06/len_sort_ab.txt)
#define LEN_SORT_BATCH 8u
__global__ __launch_bounds__(256) void k_len_hist(const LenSrc src, u32 n, u32 *__restrict__ counts, unsigned long long *__restrict__ bad_part) {
    __shared__ u32 h[PKT_LEN_CLASSES];
    __shared__ unsigned long long first_bad;
    h[threadIdx.x] = 0;
    if (threadIdx.x == 0) first_bad = ~0ull;
    __syncthreads();
    u32 lo, hi;
    len_sort_slice(n, lo, hi);
    unsigned long long bad = ~0ull;
    for (u32 i0 = lo + threadIdx.x; i0 < hi; i0 += 256u * LEN_SORT_BATCH) {
        u64 dl[LEN_SORT_BATCH], al[LEN_SORT_BATCH];
#pragma unroll
        for (u32 k = 0; k < LEN_SORT_BATCH; ++k) {
            const u32 i = i0 + 256u * k;
            dl[k] = 0; al[k] = 0;
            if (i < hi) { dl[k] = len_src_data(src, i); al[k] = len_src_aad(src, i); }
        }
#pragma unroll
        for (u32 k = 0; k < LEN_SORT_BATCH; ++k) {
            const u32 i = i0 + 256u * k;
            if (i < hi) {
                atomicAdd(&h[pkt_len_class(rows_route_size(dl[k], al[k]))], 1u);
                if (bad_part && bad == ~0ull && (dl[k] >= ROWS_LEN_LIMIT || al[k] >= ROWS_LEN_LIMIT)) bad = i;
            }
        }
    }
    if (bad != ~0ull) atomicMin(&first_bad, bad);
    __syncthreads();
    counts[(PKT_LEN_CLASSES - 1u - threadIdx.x) * LEN_SORT_WGS + blockIdx.x] = h[threadIdx.x];
    if (bad_part && threadIdx.x == 0) bad_part[blockIdx.x] = first_bad;
}
// exclusive prefix sums over the 65536 entries, in place; one workgroup.  With a route (round 6: a call whose lengths are on the device, aesgcm_rows.h RowsHdr) the same
// workgroup then reads off the sums how many messages lie in each size class, and DECIDES: which messages go by rows (route_min), how many are left for the packet
// kernels (n_small), and the packet kernel shape and deal for that many (route_pick_lg, pktg_deal) -- the host launches every shape that count could ask for and
// all but the one named here return at once.  The rule, from profiles/r06/route_sweep.txt (one box, AES-256, short messages only, by rows / by the packet kernels):
//   * the MARK.  Messages between the low mark (2 KiB) and the high one (8 KiB) are slow packets -- a lane, or four, walks 128 .. 512 blocks, a few microseconds each --
//     and fast rows; the packet kernels take them only when there are enough of them to keep every lane of the chip busy that long: U-shaped lengths below 8 KiB (half
//     the messages tiny, half near 8 KiB), 65536 of them 0.69 ms by rows / 0.85 by packets, 131072 1.22 / 1.02, 524288 4.25 / 3.03.  So: the high mark when at least
//     `mid_min` (65536) messages lie between the marks, else the low one.
//   * THE BAND ABOVE.  Ragged messages of 8 .. 16 KiB are the rows' -- 262 144 of them 647 against 606 GiB/s -- until there are enough to fill a lane per packet twice
//     over: 393 216 of them 643 by rows, 681 by the packet kernels, 524 288 647 / 713, 2^20 654 / 736 (profiles/r06/route_band.txt).  From `top_min` messages between
//     the high mark and the last class the sort resolves (16 320 bytes) the mark is that class.  top_min is 458 752, not the crossing of the two curves: what still
//     lies above the mark goes by rows BEHIND the packet launch, not beside it (a 141 KiB workgroup per CU until the end) -- at 393 216 that tail made the routed call
//     622, slower than either pure way; at 524 288 it is 718.  What the classes do not resolve stays by rows: 2^20 U-shaped messages of up to 16 383 bytes -- three in
//     ten of them in the last 64 -- run at 619 routed, 738 by lanes alone.
//   * WORTH IT AT ALL?  What is left below the mark costs the rows' closing launch a lane per block, 12 G blocks/s, and next to nothing per message; the packet kernels
//     do 45 G blocks/s but pay for every message (its E_K(J0), its length block, its closing) and for their own start: fitted to the sweep, ms for n messages of B
//     blocks, rows 0.10 + 0.08 n/10^6 + 0.080 B/10^6, packets 0.12 + 0.17 n/10^6 + 0.022 B/10^6 (and more per message where lane groups, not lanes, take them).
//     Frames of 64 .. 1514 bytes, 16384 of them 0.194 / 0.147 ms, 131072 0.62 / 0.32, 2^20 4.34 / 1.44; messages of 0 .. 128 bytes, 131072 0.161 / 0.242, 2^20
//     0.56 / 0.40; 4096 frames 0.149 / 0.112.  So: the packet kernels when the short messages hold at least `blocks_min` (2^17) + 3.5 per message blocks (counted by size class:
//     a message of class c as 4 c + 2), else everything goes by rows.  profiles/r06/route_sweep.txt: the rule's choice against both, 35 populations.
#define ROUTE_HALF_BLOCKS_PER_MSG 7ull      /* 3.5 blocks per message */
#define ROUTE_SMALL_CALL 4096u
// The DECISION of a routed call, by the workgroup that holds the sorted counts (k_len_scan, or k_len_sort1 for a small call): start_of_class[row] (LDS) = messages of a
// LONGER class than row's (row = 255 - class); my_bad = this thread's candidate for the first message whose length the call cannot take, or ~0.  Every thread of the
// workgroup calls it (at least 256 of them); thread 0 writes the header.
__device__ __forceinline__ void route_decide(const RouteCfg &rc, const u32 *start_of_class, unsigned long long my_bad, volatile u32 *host_status) {
    // messages of class >= c: the position of the first message of class c - 1 (classes are laid out falling: row 255 - class) -- from the scan's LDS, not read back from memory
    auto n_ge = [&](u32 c) { return c == 0u ? rc.n : c >= PKT_LEN_CLASSES ? 0u : start_of_class[PKT_LEN_CLASSES - c]; };
    // blocks of the messages below each mark: thread c < 256 brings its class (a message of class c has 4 c + 2 blocks, give or take two)
    const u32 cls = threadIdx.x;
    const u64 mine_blocks = cls < PKT_LEN_CLASSES ? (u64)(n_ge(cls) - n_ge(cls + 1u)) * (4u * cls + 2u) : 0ull;
    __shared__ unsigned long long blk[4];
    if (threadIdx.x < 2 || threadIdx.x == 3) blk[threadIdx.x] = 0ull;
    if (threadIdx.x == 2) blk[2] = ~0ull;
    __syncthreads();
    if (cls < rc.c_lo && cls < PKT_LEN_CLASSES) atomicAdd(&blk[0], (unsigned long long)mine_blocks);
    else if (cls < rc.c_hi && cls < PKT_LEN_CLASSES) atomicAdd(&blk[1], (unsigned long long)mine_blocks);
    else if (cls < PKT_LEN_CLASSES - 1u) atomicAdd(&blk[3], (unsigned long long)mine_blocks);                   // the band between the high mark and the last class the sort resolves
    if (my_bad != ~0ull) atomicMin(&blk[2], my_bad);                                         // the first length the call cannot take
    __syncthreads();
    if (threadIdx.x == 0) {
        // the verdict on the call's lengths: refused here, before the packet kernels or the row launches (which run side by side behind this launch) have touched anything
        const u64 first_bad = blk[2];
        rc.hdr->bad = first_bad != ~0ull ? 1u : 0u; rc.hdr->status = first_bad != ~0ull ? ROWS_ST_LENGTH : ROWS_ST_OK; rc.hdr->detail = first_bad != ~0ull ? first_bad : 0ull;
        if (first_bad != ~0ull && host_status) { host_status[2] = (u32)first_bad; host_status[3] = (u32)(first_bad >> 32); __threadfence_system(); host_status[0] = ROWS_ST_LENGTH; }
        u32 route_min, n_large;
        if (rc.c_hi >= PKT_LEN_CLASSES) { route_min = ROWS_ROUTE_NEVER; n_large = 0; }           // rows switched off
        else {
            const u32 mid = n_ge(rc.c_lo) - n_ge(rc.c_hi), band = n_ge(rc.c_hi) - n_ge(PKT_LEN_CLASSES - 1u);
            const bool top = rc.top_min && rc.c_hi && rc.c_hi < PKT_LEN_CLASSES - 1u && band >= rc.top_min, high = top || mid >= rc.mid_min;      // (c_hi = 0: everything by rows, forced -- the host has launched no packet kernel)
            const u32 c = top ? PKT_LEN_CLASSES - 1u : high ? rc.c_hi : rc.c_lo;
            const u64 short_blocks = blk[0] + (high ? blk[1] : 0ull) + (top ? blk[3] : 0ull), n_short = rc.n - n_ge(c);
            // (a SMALL call with nothing above the mark is one family's whichever way, and then the packet kernels': 2048 frames 102 us by rows, 73 by lane groups; 512
            // frames 93 / 70 -- the rows' plan, row launch and closing are three dependent launches, the packet kernel is one; profiles/r06/pkt_shape_sweep.txt)
            const bool small_call = n_ge(c) == 0u && n_short <= ROUTE_SMALL_CALL;
            if (rc.blocks_min && !small_call && 2ull * short_blocks < 2ull * rc.blocks_min + ROUTE_HALF_BLOCKS_PER_MSG * n_short) { route_min = 0; n_large = rc.n; }      // not worth a packet launch: everything by rows
            else { route_min = c * 64u; n_large = n_ge(c); }
        }
        const u32 n_small = rc.n - n_large;
        const u32 lg = rc.force_lg != 0xFFu ? rc.force_lg : n_small ? route_pick_lg(rc.n_cu, n_small) : 0u;
        rc.hdr->route_min = route_min; rc.hdr->n_small = n_small; rc.hdr->pkt_lg = lg;
        rc.hdr->pkt_deal = rc.force_deal ? (rc.force_deal + (64u >> lg) - 1u) / (64u >> lg) * (64u >> lg) : pktg_deal(rc.n_cu, n_small, lg);
        rc.hdr->pkt_counter = 0;
        rc.hdr->sc_in = rc.sc_in; rc.hdr->sc_out = rc.sc_out; rc.hdr->sc_aad = rc.sc_aad; rc.hdr->sc_len = rc.sc_len; rc.hdr->sc_alen = rc.sc_alen;
    }
}
__global__ __launch_bounds__(1024) void k_len_scan(u32 *__restrict__ counts, const RouteCfg rc, const unsigned long long *__restrict__ bad_part, volatile u32 *host_status) {
    // Every wave owns 4096 consecutive entries (16 class rows of LEN_SORT_WGS = 256) and holds ALL of them in registers, four consecutive entries per lane and step:
    // one trip to memory for the whole array (the counts were written by other XCDs: every load is a miss), a lane's prefix of four, a wave scan by lane shuffles per
    // step, a carry -- and the same registers are what the exclusive sums are stored from.  (Round 6, first form: a lane per entry, 64 steps in chunks of 16 loads,
    // read once for the totals and again for the scan: eight dependent trips, 26 - 30 us of launch -- a third of what a call of 16384 frames costs; until then a
    // thread owned 64 consecutive entries.  profiles/r06/len_sort_ab.txt.)
    __shared__ u32 wave_base[16];
    __shared__ u32 start_of_class[PKT_LEN_CLASSES];                                        // exclusive prefix at the first entry of every class row: messages of a LONGER class
    constexpr u32 PER_WAVE = LEN_SORT_ENTRIES / 16u, STEPS = PER_WAVE / 256u;
    static_assert(LEN_SORT_ENTRIES % (16u * 256u) == 0 && LEN_SORT_WGS == 256u && STEPS == 16u, "k_len_scan: a step of 64 lanes x 4 entries is one class row");
    const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    uint4 *seg = reinterpret_cast<uint4 *>(counts + w * PER_WAVE);
    uint4 v[STEPS];
#pragma unroll
    for (u32 k = 0; k < STEPS; ++k) v[k] = seg[k * 64u + lane];
    u32 s = 0;
#pragma unroll
    for (u32 k = 0; k < STEPS; ++k) s += v[k].x + v[k].y + v[k].z + v[k].w;
#pragma unroll
    for (u32 off = 32u; off; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) wave_base[w] = s;
    __syncthreads();
    u32 carry = 0;
    for (u32 i = 0; i < w; ++i) carry += wave_base[i];
#pragma unroll
    for (u32 k = 0; k < STEPS; ++k) {
        const u32 mine = v[k].x + v[k].y + v[k].z + v[k].w;
        u32 incl = mine;
#pragma unroll
        for (u32 off = 1; off < 64u; off <<= 1) { const u32 t = __shfl_up(incl, off); if (lane >= off) incl += t; }
        const u32 e0 = carry + incl - mine;
        if (lane == 0) start_of_class[w * STEPS + k] = e0;                                 // (a step is a class row)
        seg[k * 64u + lane] = make_uint4(e0, e0 + v[k].x, e0 + v[k].x + v[k].y, e0 + v[k].x + v[k].y + v[k].z);
        carry += __shfl(incl, 63);
    }
    if (!rc.hdr) return;
    __syncthreads();
    route_decide(rc, start_of_class, bad_part && threadIdx.x < LEN_SORT_WGS ? bad_part[threadIdx.x] : ~0ull, host_status);
}
// small ones FIRST (what the packet kernels take, by falling class: perm[0 .. n_small)), the messages that go by rows behind them (nobody reads those: the row
// launches walk prefix sums) -- without a route n_small is everything
__global__ __launch_bounds__(256) void k_len_scatter(const LenSrc src, u32 n, const u32 *__restrict__ base, u32 *__restrict__ perm, const RowsHdr *__restrict__ hdr, const DescSrc ds) {
    __shared__ u32 cur[PKT_LEN_CLASSES];
    cur[threadIdx.x] = base[(PKT_LEN_CLASSES - 1u - threadIdx.x) * LEN_SORT_WGS + blockIdx.x];
    const u32 n_small = hdr ? hdr->n_small : n, n_large = n - n_small;
    const bool want_desc = ds.desc && hdr && hdr->pkt_lg == 0u && !hdr->bad;               // a lane per packet: the launch reads records, not numbers (aesgcm_pkt.h PktDesc)
    u32 lo, hi;
    len_sort_slice(n, lo, hi);
    __syncthreads();
    for (u32 i0 = lo + threadIdx.x; i0 < hi; i0 += 256u * LEN_SORT_BATCH) {               // (lengths in batches, as k_len_hist)
        u64 dl[LEN_SORT_BATCH], al[LEN_SORT_BATCH];
#pragma unroll
        for (u32 k = 0; k < LEN_SORT_BATCH; ++k) {
            const u32 i = i0 + 256u * k;
            dl[k] = 0; al[k] = 0;
            if (i < hi) { dl[k] = len_src_data(src, i); al[k] = len_src_aad(src, i); }
        }
#pragma unroll
        for (u32 k = 0; k < LEN_SORT_BATCH; ++k) {
            const u32 i = i0 + 256u * k;
            if (i < hi) {
                const u32 pos = atomicAdd(&cur[pkt_len_class(rows_route_size(dl[k], al[k]))], 1u);
                perm[pos >= n_large ? pos - n_large : n_small + pos] = i;
                if (want_desc && pos >= n_large) ds.desc[pos - n_large] = len_src_desc(src, ds, i, dl[k], al[k]);
            }
        }
    }
}

// The whole sort of a SMALL call in one launch, one workgroup (round 6): lengths read once and kept in registers, a histogram in LDS, the scan of its 256 classes, the
// decision, the scatter with LDS cursors.  Three dependent launches cost a call of 4096 frames 20 of its 84 us, one of 16 384 frames 20 of 130
// (profiles/r06/len_sort_ab.txt); the order inside a class is whatever the atomics make it, as in the three-launch form.
#define LEN_SORT1_MAX 16384u
#define LEN_SORT1_PER (LEN_SORT1_MAX / 1024u)
__global__ __launch_bounds__(1024) void k_len_sort1(const LenSrc src, u32 n, u32 *__restrict__ perm, const RouteCfg rc, volatile u32 *host_status, const DescSrc ds) {
    __shared__ u32 h[PKT_LEN_CLASSES], start_of_class[PKT_LEN_CLASSES], cur[PKT_LEN_CLASSES], wave_sum[4];
    const u32 tid = threadIdx.x, lane = tid & 63u;
    if (tid < PKT_LEN_CLASSES) h[tid] = 0;
    __syncthreads();
    u32 sz[LEN_SORT1_PER];
    unsigned long long bad = ~0ull;
#pragma unroll
    for (u32 k = 0; k < LEN_SORT1_PER; ++k) {                                              // every length of the call is in flight at once
        const u32 i = tid + 1024u * k;
        sz[k] = 0;
        if (i < n) {
            const u64 dl = len_src_data(src, i), al = len_src_aad(src, i);
            sz[k] = rows_route_size(dl, al);
            if (rc.hdr && bad == ~0ull && (dl >= ROWS_LEN_LIMIT || al >= ROWS_LEN_LIMIT)) bad = i;
        }
    }
#pragma unroll
    for (u32 k = 0; k < LEN_SORT1_PER; ++k)
        if (tid + 1024u * k < n) atomicAdd(&h[pkt_len_class(sz[k])], 1u);
    __syncthreads();
    // exclusive prefix over the rows (row = 255 - class: longest first): four waves of 64 rows
    u32 v = 0, incl = 0;
    if (tid < PKT_LEN_CLASSES) {
        v = h[PKT_LEN_CLASSES - 1u - tid]; incl = v;
#pragma unroll
        for (u32 off = 1; off < 64u; off <<= 1) { const u32 t = __shfl_up(incl, off); if (lane >= off) incl += t; }
        if (lane == 63u) wave_sum[tid >> 6] = incl;
    }
    __syncthreads();
    if (tid < PKT_LEN_CLASSES) {
        u32 carry = 0;
        for (u32 w = 0; w < (tid >> 6); ++w) carry += wave_sum[w];
        start_of_class[tid] = carry + incl - v;
        cur[PKT_LEN_CLASSES - 1u - tid] = carry + incl - v;                                 // where the class's first message goes
    }
    __syncthreads();
    __shared__ u32 n_small_s, lg_s;
    if (tid == 0) { n_small_s = n; lg_s = 0xFFu; }
    if (rc.hdr) {
        route_decide(rc, start_of_class, bad, host_status);
        if (tid == 0) { n_small_s = rc.hdr->bad ? 0u : rc.hdr->n_small; lg_s = rc.hdr->pkt_lg; }      // (thread 0 reads what it has just written; a refused call scatters nothing anyone reads)
    }
    __syncthreads();
    const u32 n_small = n_small_s;
    const u32 n_large = n - n_small;
    const bool want_desc = ds.desc && rc.hdr && n_small && lg_s == 0u;                     // (a lane per packet in a call this small only when forced; the records all the same)
#pragma unroll
    for (u32 k = 0; k < LEN_SORT1_PER; ++k) {
        const u32 i = tid + 1024u * k;
        if (i < n) {
            const u32 pos = atomicAdd(&cur[pkt_len_class(sz[k])], 1u);
            perm[pos >= n_large ? pos - n_large : n_small + pos] = i;
            if (want_desc && pos >= n_large) ds.desc[pos - n_large] = len_src_desc(src, ds, i, len_src_data(src, i), len_src_aad(src, i));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_pktg: many packets under the context's key, 2^LG lanes per packet (lane bodies: pktg_lane(), pktg_close_lane(),
// pktg_tree_offer(); see "Packets under ONE key" in aesgcm_dev.h).  One 1024-lane workgroup per CU.
// ------------------------------------------------------------------------------------------------
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
// 128-byte requests; with 768-lane workgroups (148 - 165 registers, no scratch) 1.104e9 (profiles/archive/r03/pktg_wg768_ab.txt).  What was being spilled was
// bookkeeping, and it is gone at 1024 lanes too: the index registers of ds_bpermute exchanges (now ds_swizzle, lane_xor), the 64 E_K(J0) values held
// across the packet loop (now in the wave's LDS slot) and the lane's position (recomputed from lane_id_fresh after the loop).  Lane groups therefore
// run 1024-lane workgroups again (4 waves per SIMD: 16 lanes per packet 517 -> 530, 681 -> 700 GiB/s at 1 / 4 KiB against 768 lanes); one packet per
// wave keeps its E_K(J0) values in registers and stays at 768 lanes, where it needs no scratch.  AESGCM_PKTG_WG forces one geometry for all.
template <int NR, int DEC, int LG>
__global__ __launch_bounds__(PKTG_WG(LG), (PKTG_WG(LG) + 255) / 256) void k_pktg(const KeyMaterial *__restrict__ km, const DevTables *__restrict__ tb, const PktParams p) {
#define PKT_SC false
#include "aesgcm_pktg_body.inc"
#undef PKT_SC
}
// ... of messages WHEREVER THEY LIVE (aesgcm_messages_crypt_dev: the short ones of a routed call): the packets' places are addresses from the arrays behind p.route
template <int NR, int DEC, int LG>
__global__ __launch_bounds__(PKTG_WG(LG), (PKTG_WG(LG) + 255) / 256) void k_pktgs(const KeyMaterial *__restrict__ km, const DevTables *__restrict__ tb, const PktParams p) {
#define PKT_SC true
#include "aesgcm_pktg_body.inc"
#undef PKT_SC
}

// ------------------------------------------------------------------------------------------------
// k_pktl: many packets under the context's key, one LANE per packet (lane body: pktl_lane()); waves take
// blocks of 64 consecutive packets from the dispenser.
// ------------------------------------------------------------------------------------------------
// ILP = 1: the same lane code compiled for 512-lane workgroups (two waves per SIMD, 256 registers) with the eight keystream blocks of a line as independent
// chains: for batches that do not fill the chip, where a wave has to hide its own LDS latency (pktl_lane, AESGCM_PKTL_WG_ILP).
template <int NR, int DEC, int ILP, bool SC>
__device__ __forceinline__ void pktl_body(const KeyMaterial *__restrict__ km, const DevTables *__restrict__ tb, const PktParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr u32 WG = ILP ? AESGCM_PKTL_WG_ILP : AESGCM_PKTL_WG;
    const u32 tid = threadIdx.x, lane = tid & 63u;
    u32 n_pkts = p.n_pkts;
    if (p.route) {                                             // a ROUTED call: the count is the device's, and a lane per packet may not be the shape chosen for it (k_pktg above)
        if (p.route->bad || p.route->pkt_lg != 0u || !p.route->n_small) return;
        n_pkts = uniform32(p.route->n_small);
    }
    main_fill_lds(smem, km, tb, tid, true, WG, GH_TAB_H);
#if AESGCM_PKTL_T4
    fill_lds_t4(smem, tb, tid, WG);
#endif
    __syncthreads();
    const u32 nb = (n_pkts + 63u) / 64u;
    for (u32 guard = 0; guard <= nb; ++guard) {                // bounded, as every dispenser loop here
        u32 b = 0;
        if (lane == 0) b = atomicAdd(p.counter, 1u) - p.counter_base;
        b = __builtin_amdgcn_readfirstlane(b);
        if (b >= nb) break;
        const u32 idx = b * 64u + lane;
        if (idx < n_pkts) pktl_lane<NR, DEC, AESGCM_PKTL_T4 != 0, ILP != 0, SC>(km, p, smem, p.desc ? 0u : pkt_map(p, idx), lane, p.desc ? p.desc + idx : nullptr);
    }
}
template <int NR, int DEC, int ILP>
__global__ __launch_bounds__(ILP ? AESGCM_PKTL_WG_ILP : AESGCM_PKTL_WG, ILP ? 2 : AESGCM_PKTL_WAVES) void k_pktl(const KeyMaterial *__restrict__ km, const DevTables *__restrict__ tb, const PktParams p) {
    pktl_body<NR, DEC, ILP, false>(km, tb, p);
}
template <int NR, int DEC>                                     // ... of messages wherever they live (as k_pktgs)
__global__ __launch_bounds__(AESGCM_PKTL_WG, AESGCM_PKTL_WAVES) void k_pktls(const KeyMaterial *__restrict__ km, const DevTables *__restrict__ tb, const PktParams p) {
    pktl_body<NR, DEC, 0, true>(km, tb, p);
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
    if (p.hdr && (p.hdr->bad || !p.hdr->NB)) return;                         // nothing by rows (a routed call whose messages are all the packet kernels'), or a plan that was refused: before the 141 KiB of tables are staged
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
        // The pieces of the block.  Every iteration derives what it needs from (g, m) alone -- lengths, offsets, the IV's constants are loaded again
        // per piece, and the record's header is written BEFORE the rows -- so that almost nothing but g, m and the record's address is live across the row loop (an
        // earlier form that carried the message's geometry through it spilled 114 scalars and 252 bytes of scratch).
        for (u32 guard2 = 0, skipped = 0; g < g_end && guard2 <= 2u * D + 4u && skipped <= p.n_pkts; ++guard2) {
            m = opaque_sgpr(m);
            RowsMsg mq = rows_msg(p, m);
            mq.doff = uniform64(mq.doff); mq.ooff = uniform64(mq.ooff); mq.aoff = uniform64(mq.aoff); mq.len = uniform32(mq.len); mq.alen = uniform32(mq.alen);
            const u32 route_min = uniform32(rows_route_min(p));
            const RowsGeom geo = rows_geom_of(mq, route_min);                        // (a message of the packet kernels counts as empty: no unit -- skipped below like one shorter than a row)
            const u64 g0 = uniform64(rows_unit_base(p, m));
            const u32 U = rows_units(geo, rows_na_of(mq, route_min));
            if (g >= g0 + U) { ++m; if (U == 0) { ++skipped; --guard2; } continue; }     // the next message (one without a unit -- shorter than a row -- does not count against the bound of the walk)
            const RowsPiece pc = rows_piece(geo, uniform32(rows_slot_base(p, m)), g0, (u32)(g - g0), g_end - g, D);
            RowsRec *rr = p.rec + pc.slot;
            if (lane_id_fresh() == 0) { rr->e = pc.e; rr->msg = m; rr->flags = pc.kind == ROWS_TAIL ? ROWS_REC_VALID : (ROWS_REC_VALID | ROWS_REC_WEIGH); }    // (the tail is weighted already: lane terms H^(65 - L))
            // every kind of piece takes the lane's index FRESH (lane_id_fresh: opaque to the compiler), so that nothing lane-dependent of the AAD code -- table
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
                    z = wave_xor(rows_tail_lane<NR, MODE == MODE_DEC>(km, p, mq, smem, cc, lane_id_fresh()));
                }
            }
            if (lane_id_fresh() == 0) rr->w = z;
            g += pc.len;
        }
    }
}

// k_rows_close: the lanes of the launch walk three things -- index i is message i, record slot i, and the blocks i, i + lanes, ... of the smalls axis.  A message's
// lane brings what the message owes once, (length block) H ^ E_K(J0) (rows_msg_term); a record's lane the record times H^e; a smalls block's lane the block
// through the cipher and times its power of H (rows_small_block).  Each XORs into the message's accumulator and counts itself arrived; the lane that counts the
// message's last arrival holds the tag: it stores it and (decrypt) compares.  Memory-side atomics only, as acc_arrive: the XORs have returned before the arrival
// is counted.  Zero at rest: the lane puts its record's flags back to zero, the closing lane the message's accumulator and count, workgroup 0 the dispensers.
template <int DEC>
__device__ __forceinline__ void rows_arrive(const RowsParams &p, u32 m, const G128 &z, u32 count) {           // count: arrivals the caller stands for (a run of smalls blocks folded into one lane)
    const unsigned long long ohi = atomicXor(p.acc + 2u * m, ((unsigned long long)z.w[0] << 32) | z.w[1]);
    const unsigned long long olo = atomicXor(p.acc + 2u * m + 1u, ((unsigned long long)z.w[2] << 32) | z.w[3]);
    u32 dep;
    asm volatile("v_and_b32 %0, 0, %1" : "=v"(dep) : "v"((u32)(ohi ^ olo) | (u32)((ohi ^ olo) >> 32)));
    const u32 arrived = atomicAdd(p.cnt + m, count + dep);
    const RowsMsg mq = rows_msg(p, m);
    if (arrived + count != rows_pieces(rows_geom(mq.len), rows_na(mq.alen), rows_unit_base(p, m), p.hdr ? p.hdr->D : p.D)) return;
    const unsigned long long hi = atomicExch(p.acc + 2u * m, 0ull), lo = atomicExch(p.acc + 2u * m + 1u, 0ull);
    p.cnt[m] = 0;
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
// The lanes of a wave that hold terms of the same thing (`key`; neighbours) fold them -- a segmented reduction by doubling -- and the first lane of every run
// answers true and gets the run's length: it arrives for all.  One address serves 87 M atomics a second (aesgcm_stream.h), and far fewer when the 64 lanes of a
// wave want their old values back from it: an arrival per block made 262 144 x 9000 bytes 325 GiB/s (profiles/r05/rows_ragged_few_before.txt), an arrival per
// record made the closing of 16 messages of 16 MiB -- 256 records each -- 87 us against 14 for 4096 of 64 KiB (rows_few_large_stats.txt).  key 0xFFFFFFFF: the
// lane holds nothing.
__device__ __forceinline__ bool rows_fold(u32 key, G128 &z, u32 lane, u32 *count) {
    const u32 before = __shfl_up(key, 1);
    const bool first = lane == 0 || before != key;
    const unsigned long long firsts = __ballot(first);
    const u32 run = (u32)__builtin_popcountll(firsts & (lane == 63u ? ~0ull : (2ull << lane) - 1ull));     // the lane's run: runs are told apart by their number, not by their key -- a record slot that was not used lies between two runs of ONE message
#pragma unroll
    for (u32 off = 1; off < 64u; off <<= 1) {
        const u32 orun = __shfl_down(run, off);
        const u32 z0 = __shfl_down(z.w[0], off), z1 = __shfl_down(z.w[1], off), z2 = __shfl_down(z.w[2], off), z3 = __shfl_down(z.w[3], off);
        if (lane + off < 64u && orun == run) { z.w[0] ^= z0; z.w[1] ^= z1; z.w[2] ^= z2; z.w[3] ^= z3; }
    }
    const unsigned long long later = lane == 63u ? 0ull : firsts >> (lane + 1u);
    *count = (later ? lane + 1u + (u32)__builtin_ctzll(later) : 64u) - lane;
    return first && key != 0xFFFFFFFFu;
}
static_assert(offsetof(DevTables, te3) - offsetof(DevTables, te0) == 3072 && offsetof(DevTables, te0) % 16 == 0 && ROWS_CLOSE_WG == 256, "rows_close_fill_te: te0 .. te3 as 256 consecutive uint4, one per thread");
template <int DEC>
__global__ __launch_bounds__(ROWS_CLOSE_WG) void k_rows_close(const KeyMaterial *__restrict__ km, const DevTables *__restrict__ tb, const RowsParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4096];
    if (p.hdr && p.hdr->bad) return;                                             // a plan that was refused: nothing ran, nothing to close (the scratch is at rest)
    if (blockIdx.x == 0 && threadIdx.x < ROWS_NQ) p.queues[16u * threadIdx.x] = 0;
    const u32 route_min = rows_route_min(p);
    if (p.hdr && p.hdr->n_small == p.n_pkts) return;                             // every message went to the packet kernels: no unit, no smalls block, no tag of this launch's
    rows_close_fill_te(smem, tb, threadIdx.x);
    __syncthreads();
    const u32 *te = reinterpret_cast<const u32 *>(smem + ROWS_CLOSE_LDS_TE);
    // The launch is a grid of lanes that STRIDE over three things (round 6; until then a lane per record slot the scratch was sized for -- 3 n + 65536 with offset
    // arrays, nearly all of them empty): message i (what it owes once), the blocks of the smalls axis, record slot i up to the slots the plan actually gave out.
    const u64 lanes = (u64)gridDim.x * ROWS_CLOSE_WG, gid = (u64)blockIdx.x * ROWS_CLOSE_WG + threadIdx.x;
    for (u64 i = gid; i < p.n_pkts; i += lanes) {
        const RowsMsg q = rows_msg(p, (u32)i);
        if (!rows_is_small(q.len, q.alen, route_min)) rows_arrive<DEC>(p, (u32)i, rows_msg_term(km, te, p, (u32)i), 1u);     // (a message of the packet kernels has its tag from there)
    }
    // The smalls: the waves of the launch take 64 consecutive blocks of the axis at a time.  The blocks of a segment -- one message's AAD, or its tail -- are
    // neighbours: their lanes fold their terms first (rows_fold) and the first of them pays what is still due of the segment's power of H and arrives for all.
    const u64 total = rows_small_total(p), wave_base = (u64)blockIdx.x * ROWS_CLOSE_WG + (threadIdx.x & ~63u);
    const u32 lane = threadIdx.x & 63u;
    for (u64 base = wave_base, guard = 0; base < total && guard <= total / lanes + 1u; base += lanes, ++guard) {
        const u64 t = base + lane;
        const bool active = t < total;
        G128 z; z.w[0] = z.w[1] = z.w[2] = z.w[3] = 0;
        u32 m = 0xFFFFFFFFu;
        u64 e_run = 0;
        if (active) m = rows_small_block<DEC>(km, te, p, t, &z, &e_run);
        u32 key = active ? 2u * m + (e_run ? 1u : 0u) : 0xFFFFFFFFu, count;            // a segment: one message's AAD, or its tail
        if (rows_fold(key, z, lane, &count)) rows_arrive<DEC>(p, m, rows_small_due(km, z, e_run), count);
    }
    // The records: those of one message have neighbouring slots; they fold the same way.
    const u64 slots = p.slot_base ? p.slot_base[p.n_pkts] : p.slot_cap;
    for (u64 base = wave_base, guard = 0; base < slots && guard <= slots / lanes + 1u; base += lanes, ++guard) {
        const u64 i = base + lane;
        RowsRec r;
        r.flags = 0; r.msg = 0;
        if (i < slots) r = p.rec[i];
        const bool valid = (r.flags & ROWS_REC_VALID) != 0;
        G128 z; z.w[0] = z.w[1] = z.w[2] = z.w[3] = 0;
        if (valid) { p.rec[i].flags = 0; z = rows_weigh(km, r); }
        u32 count;
        if (rows_fold(valid ? r.msg : 0xFFFFFFFFu, z, lane, &count)) rows_arrive<DEC>(p, r.msg, z, count);
    }
}

// The cut of a call with offset arrays, on the device (the host does not know the lengths).  Units per message -> prefix[0 .. n] and G; the cut (rows_cut);
// record slots per message -> slot_base[0 .. n]; the header.  Round 6: a ROUTED call (a.routed: k_len_scan left route_min in the header) counts the messages
// below the mark as nothing -- they are the packet kernels' --, and every length is CHECKED here, where the round-5 code truncated it: a length or an
// offset difference of 2^28 or more (offsets that do not rise are one: the difference wraps) refuses the whole call -- hdr->bad, nothing runs, outputs untouched --
// and says so in the context's pinned host slot (status, detail = the first such message; aesgcm_ctx_status), as the RTL raises its flag when the counter cannot
// go on (src/aes_icb.vhd:65,98,114,119).
struct RowsPlan {
    const u64 *off, *aoff; const u32 *len_arr, *alen_arr;
    u32 pkt_len, aad_len, n, waves, force_d, nb_cap, slot_cap, nwg, routed;
    RowsHdr *hdr; u64 *prefix, *sprefix; u32 *slot_base;
    u64 *part;                                                               // 4 x nwg: units, smalls blocks, slots, first bad message per workgroup -- then what lies in front of each
    volatile u32 *host_status;                                               // the context's pinned host slot: {code, 0, detail lo, detail hi}, or NULL
};
// exclusive prefix of `mine` over the 1024 threads; part[1023] = the total afterwards.  Wave scans by lane shuffles and one scan of the 16 wave totals (round 6: the
// Hillis-Steele form over LDS -- ten steps, two barriers each -- made the three scans of a plan launch 17 - 20 us for 2^20 messages)
__device__ __forceinline__ u64 block_scan_u64(unsigned long long *part, u64 mine, u32 tid) {
    const u32 lane = tid & 63u, w = tid >> 6;
    unsigned long long v = mine;
#pragma unroll
    for (u32 off = 1; off < 64u; off <<= 1) { const unsigned long long t = __shfl_up(v, off); if (lane >= off) v += t; }
    __syncthreads();                                                         // (whoever still reads part[] from an earlier call)
    if (lane == 63u) part[w] = v;
    __syncthreads();
    if (w == 0) {
        unsigned long long x = lane < 16u ? part[lane] : 0ull;
#pragma unroll
        for (u32 off = 1; off < 16u; off <<= 1) { const unsigned long long t = __shfl_up(x, off); if (lane >= off) x += t; }
        if (lane < 16u) part[16u + lane] = x;
        if (lane == 15u) part[1023] = x;
    }
    __syncthreads();
    return (w ? part[16u + w - 1u] : 0ull) + v - mine;
}
__device__ __forceinline__ u64 block_min_u64(unsigned long long *part, u64 mine, u32 tid) {       // the minimum of `mine` over the 1024 threads (everybody gets it)
    __syncthreads();
    part[tid] = mine;
    __syncthreads();
    for (u32 d = 512u; d; d >>= 1) { if (tid < d && part[tid + d] < part[tid]) part[tid] = part[tid + d]; __syncthreads(); }
    const u64 r = part[0];
    __syncthreads();
    return r;
}
__device__ __forceinline__ u64 plan_len(const RowsPlan &a, u32 m) { return a.len_arr ? (u64)a.len_arr[m] : a.off ? a.off[m + 1] - a.off[m] : (u64)a.pkt_len; }
__device__ __forceinline__ u64 plan_alen(const RowsPlan &a, u32 m) { return a.len_arr ? (a.alen_arr ? (u64)a.alen_arr[m] : 0ull) : a.aoff ? a.aoff[m + 1] - a.aoff[m] : (u64)a.aad_len; }
__device__ __forceinline__ bool plan_bad_len(const RowsPlan &a, u32 m) { return plan_len(a, m) >= ROWS_LEN_LIMIT || plan_alen(a, m) >= ROWS_LEN_LIMIT; }
// (lengths beyond the limit count as nothing: the call is refused anyway, and the sums stay in range)
__device__ __forceinline__ RowsGeom plan_geom(const RowsPlan &a, u32 m, u32 route_min) { return plan_bad_len(a, m) ? rows_geom(0) : rows_geom_routed(plan_len(a, m), plan_alen(a, m), route_min); }
__device__ __forceinline__ u32 plan_na(const RowsPlan &a, u32 m, u32 route_min) { return plan_bad_len(a, m) ? 0u : rows_na_routed(plan_len(a, m), plan_alen(a, m), route_min); }
__device__ __forceinline__ u32 plan_route_min(const RowsPlan &a) { return a.routed ? a.hdr->route_min : 0u; }
__device__ __forceinline__ void plan_refuse(const RowsPlan &a, u32 status, u64 detail) {          // one thread
    a.hdr->G = 0; a.hdr->NB = 0; a.hdr->bad = 1; a.hdr->status = status; a.hdr->detail = detail;
    if (a.host_status) { a.host_status[2] = (u32)detail; a.host_status[3] = (u32)(detail >> 32); __threadfence_system(); a.host_status[0] = status; }
}
// up to ROWS_PLAN_ONE_WG messages: ONE workgroup
__global__ __launch_bounds__(1024) void k_rows_plan(const RowsPlan a) {
    __shared__ unsigned long long part[1024];
    const u32 tid = threadIdx.x, n = a.n, per = (n + 1023u) / 1024u;
    const u32 lo = tid * per < n ? tid * per : n, hi = lo + per < n ? lo + per : n;
    const u32 route_min = plan_route_min(a);
    u64 s = 0, ss = 0, bad = ~0ull;
    for (u32 m = lo; m < hi; ++m) {
        const RowsGeom g = plan_geom(a, m, route_min); const u32 na = plan_na(a, m, route_min);
        s += rows_units(g, na); ss += rows_smalls(g, na);
        if (bad == ~0ull && plan_bad_len(a, m)) bad = m;
    }
    u64 run = block_scan_u64(part, s, tid);                                  // row units in front of the thread's messages
    const u64 GR = part[1023];
    __syncthreads();
    u64 srun = block_scan_u64(part, ss, tid);                                // smalls blocks in front of them
    const u64 ST = part[1023];
    __syncthreads();
    const u64 first_bad = block_min_u64(part, bad, tid);
    const u64 G = GR;
    u32 D, NB, dyn;
    const bool cut_ok = rows_cut(G, a.waves, a.force_d, a.nb_cap, &D, &NB, &dyn);
    u64 t = 0;
    for (u32 m = lo; m < hi; ++m) {
        const RowsGeom g = plan_geom(a, m, route_min); const u32 na = plan_na(a, m, route_min);
        a.prefix[m] = run; a.sprefix[m] = srun;
        t += rows_slots(g, na, run, D);
        run += rows_units(g, na); srun += rows_smalls(g, na);
    }
    u64 slot = block_scan_u64(part, t, tid);
    const u64 slots = part[1023];
    run = lo < n ? a.prefix[lo] : 0;
    for (u32 m = lo; m < hi; ++m) {
        const RowsGeom g = plan_geom(a, m, route_min); const u32 na = plan_na(a, m, route_min);
        a.slot_base[m] = (u32)slot;
        slot += rows_slots(g, na, run, D);
        run += rows_units(g, na);
    }
    if (tid == 0) {
        a.prefix[n] = GR; a.sprefix[n] = ST; a.slot_base[n] = (u32)(slots <= a.slot_cap ? slots : 0);
        a.hdr->G = G; a.hdr->D = D; a.hdr->NB = NB; a.hdr->dyn = dyn;
        if (!a.routed) { a.hdr->bad = 0; a.hdr->status = ROWS_ST_OK; a.hdr->detail = 0; a.hdr->route_min = 0; a.hdr->n_small = 0; }           // (not routed: everything by rows -- what the row launches read the header for; a routed call has its verdict on the lengths from k_len_scan already)
        if (a.hdr->bad) { a.hdr->G = 0; a.hdr->NB = 0; }
        else if (first_bad != ~0ull) plan_refuse(a, ROWS_ST_LENGTH, first_bad);
        else if (!cut_ok) plan_refuse(a, ROWS_ST_UNITS, G);
        else if (slots > a.slot_cap) plan_refuse(a, ROWS_ST_PLAN_FIT, slots);   // (the host sizes the scratch for the worst case; a cut that does not fit would be its bug: then nothing runs)
    }
}

// The same plan for MANY messages: the one workgroup above takes 4 ns a message -- 2.1 ms in front of a 5.1 ms row launch for 524 288 messages of 8 KiB
// (profiles/r05/rows_var_stats_before.txt).  Five small launches instead, a thread per message and a workgroup per 1024 of them: sums per workgroup; their scan
// and the cut (one workgroup); the first two prefix sums and the slot counts; the scan of those; the third prefix sum.
__global__ __launch_bounds__(1024) void k_rows_plan_sums(const RowsPlan a) {
    __shared__ unsigned long long part[1024];
    const u32 tid = threadIdx.x, m = blockIdx.x * 1024u + tid;
    const u32 route_min = plan_route_min(a);
    u64 u = 0, s = 0, bad = ~0ull;
    if (m < a.n) { const RowsGeom g = plan_geom(a, m, route_min); const u32 na = plan_na(a, m, route_min); u = rows_units(g, na); s = rows_smalls(g, na); if (plan_bad_len(a, m)) bad = m; }
    block_scan_u64(part, u, tid);
    const u64 U = part[1023];
    __syncthreads();
    block_scan_u64(part, s, tid);
    const u64 S = part[1023];
    const u64 first_bad = block_min_u64(part, bad, tid);
    if (tid == 0) { a.part[blockIdx.x] = U; a.part[a.nwg + blockIdx.x] = S; a.part[3u * a.nwg + blockIdx.x] = first_bad; }
}
// exclusive scan of v[0 .. n) in place by ONE workgroup (tiles of 1024 with a carry); returns the total
__device__ __forceinline__ u64 plan_scan_in_place(unsigned long long *part, u64 *v, u32 n, u32 tid) {
    u64 carry = 0;
    for (u32 base = 0; base < n; base += 1024u) {
        const u32 i = base + tid;
        const u64 mine = i < n ? v[i] : 0ull;
        const u64 before = block_scan_u64(part, mine, tid);
        const u64 total = part[1023];
        if (i < n) v[i] = carry + before;
        carry += total;
        __syncthreads();
    }
    return carry;
}
__global__ __launch_bounds__(1024) void k_rows_plan_cut(const RowsPlan a) {
    __shared__ unsigned long long part[1024];
    const u32 tid = threadIdx.x;
    const u64 GR = plan_scan_in_place(part, a.part, a.nwg, tid), ST = plan_scan_in_place(part, a.part + a.nwg, a.nwg, tid);
    u64 bad = ~0ull;
    for (u32 i = tid; i < a.nwg; i += 1024u) { const u64 b = a.part[3u * a.nwg + i]; if (b < bad) bad = b; }
    const u64 first_bad = block_min_u64(part, bad, tid);
    if (tid == 0) {
        u32 D, NB, dyn;
        const bool cut_ok = rows_cut(GR, a.waves, a.force_d, a.nb_cap, &D, &NB, &dyn);
        a.hdr->G = GR; a.hdr->D = D; a.hdr->NB = NB; a.hdr->dyn = dyn;
        if (!a.routed) { a.hdr->bad = 0; a.hdr->status = ROWS_ST_OK; a.hdr->detail = 0; a.hdr->route_min = 0; a.hdr->n_small = 0; }
        a.prefix[a.n] = GR; a.sprefix[a.n] = ST;
        if (a.hdr->bad) { a.hdr->G = 0; a.hdr->NB = 0; }
        else if (first_bad != ~0ull) plan_refuse(a, ROWS_ST_LENGTH, first_bad);
        else if (!cut_ok) plan_refuse(a, ROWS_ST_UNITS, GR);
    }
}
__global__ __launch_bounds__(1024) void k_rows_plan_place(const RowsPlan a) {
    __shared__ unsigned long long part[1024];
    const u32 tid = threadIdx.x, m = blockIdx.x * 1024u + tid;
    const u32 route_min = plan_route_min(a);
    RowsGeom g = rows_geom(0);
    u32 na = 0;
    u64 u = 0, s = 0;
    if (m < a.n) { g = plan_geom(a, m, route_min); na = plan_na(a, m, route_min); u = rows_units(g, na); s = rows_smalls(g, na); }
    const u64 g0 = a.part[blockIdx.x] + block_scan_u64(part, u, tid);
    __syncthreads();
    const u64 s0 = a.part[a.nwg + blockIdx.x] + block_scan_u64(part, s, tid);
    __syncthreads();
    u64 t = 0;
    if (m < a.n) { a.prefix[m] = g0; a.sprefix[m] = s0; t = rows_slots(g, na, g0, a.hdr->D); a.slot_base[m] = (u32)t; }
    block_scan_u64(part, t, tid);
    if (tid == 0) a.part[2u * a.nwg + blockIdx.x] = part[1023];
}
__global__ __launch_bounds__(1024) void k_rows_plan_slots(const RowsPlan a) {
    __shared__ unsigned long long part[1024];
    const u32 tid = threadIdx.x;
    const u64 slots = plan_scan_in_place(part, a.part + 2u * a.nwg, a.nwg, tid);
    if (tid == 0) {
        a.slot_base[a.n] = (u32)(slots <= a.slot_cap ? slots : 0);
        if (slots > a.slot_cap && !a.hdr->bad) plan_refuse(a, ROWS_ST_PLAN_FIT, slots);      // (the host sizes the scratch for the worst case; a cut that does not fit would be its bug: then nothing runs)
    }
}
__global__ __launch_bounds__(1024) void k_rows_plan_base(const RowsPlan a) {
    __shared__ unsigned long long part[1024];
    const u32 tid = threadIdx.x, m = blockIdx.x * 1024u + tid;
    const u64 t = m < a.n ? a.slot_base[m] : 0u;
    const u64 before = block_scan_u64(part, t, tid);
    if (m < a.n) a.slot_base[m] = (u32)(a.part[2u * a.nwg + blockIdx.x] + before);
}

// k_wipe_failed: a wave per packet; packets whose auth[] says 0 get their output bytes zeroed (context option "wipe_on_auth_fail", aesgcm_wipe_failed_dev)
__global__ __launch_bounds__(256) void k_wipe_failed(unsigned char *out, const int *auth, const u64 *data_off, u32 n_pkts, u32 pkt_len, const u64 *out_ptr, const u32 *len_arr) {
    const u32 pkt = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (pkt >= n_pkts || auth[pkt]) return;
    u64 lo = data_off ? data_off[pkt] : (u64)pkt * pkt_len, hi = data_off ? data_off[pkt + 1] : lo + pkt_len;
    if (len_arr) { lo = out_ptr[pkt]; hi = lo + len_arr[pkt]; }                 // the scattered form: addresses (out is NULL)
    unsigned char *p = reinterpret_cast<unsigned char *>((uintptr_t)out + lo);
    const u64 len = hi - lo, head = len < 16 ? len : ((16u - ((uintptr_t)p & 15u)) & 15u);
    if (lane < head) p[lane] = 0;
    const u64 nvec = (len - head) / 16;
    for (u64 i = lane; i < nvec; i += 64) gstore16(p + head + 16 * i, make_uint4(0, 0, 0, 0));
    const u64 done = head + 16 * nvec;
    if (done + lane < len) p[done + lane] = 0;
}

// ================================================================================================
// launchers: the only code that names a kernel (aesgcm_internal.h).  Each picks the template instance by round count / mode / shape and returns hipGetLastError().
// ================================================================================================
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
hipError_t klaunch_main(int mode, int nr, unsigned wgs, hipStream_t st, const KeyMaterial *km, const DevTables *tb, const MainParams &p) {
    switch (mode) {
    case MODE_ENC: return launch_main_nr<MODE_ENC>(nr, dim3(wgs), st, km, tb, p);
    case MODE_DEC: return launch_main_nr<MODE_DEC>(nr, dim3(wgs), st, km, tb, p);
    case MODE_KS:  return launch_main_nr<MODE_KS>(nr, dim3(wgs), st, km, tb, p);
    default:       return launch_main_nr<MODE_ECB>(nr, dim3(wgs), st, km, tb, p);
    }
}

hipError_t klaunch_set_attributes() {
#define ATTRCHK(call) do { const hipError_t _e = (call); if (_e != hipSuccess) return _e; } while (0)
#define SETATTR(NR, MODE) ATTRCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_main<NR, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, AESGCM_LDS_BYTES))
    SETATTR(10, MODE_ENC); SETATTR(12, MODE_ENC); SETATTR(14, MODE_ENC);
    SETATTR(10, MODE_DEC); SETATTR(12, MODE_DEC); SETATTR(14, MODE_DEC);
    SETATTR(10, MODE_KS);  SETATTR(12, MODE_KS);  SETATTR(14, MODE_KS);
    SETATTR(10, MODE_ECB); SETATTR(12, MODE_ECB); SETATTR(14, MODE_ECB);
#undef SETATTR
#define SETATTRY(NR, MODE, CYC) ATTRCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_body<NR, MODE, CYC>), hipFuncAttributeMaxDynamicSharedMemorySize, AESGCM_BODY_LDS + (CYC ? CYC_LDS_PARK_BYTES : 0u)))
    SETATTRY(10, MODE_ENC, false); SETATTRY(12, MODE_ENC, false); SETATTRY(14, MODE_ENC, false); SETATTRY(10, MODE_DEC, false); SETATTRY(12, MODE_DEC, false); SETATTRY(14, MODE_DEC, false);
    SETATTRY(10, MODE_ENC, true); SETATTRY(12, MODE_ENC, true); SETATTRY(14, MODE_ENC, true); SETATTRY(10, MODE_DEC, true); SETATTRY(12, MODE_DEC, true); SETATTRY(14, MODE_DEC, true);
    SETATTRY(10, MODE_PROBE, false); SETATTRY(12, MODE_PROBE, false); SETATTRY(14, MODE_PROBE, false);
#undef SETATTRY
#define SETATTRH(NR, MODE) ATTRCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bodyh<NR, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, AESGCM_LDS_BYTES + CYC_LDS_PARK_BYTES))
    SETATTRH(10, MODE_ENC); SETATTRH(12, MODE_ENC); SETATTRH(14, MODE_ENC); SETATTRH(10, MODE_DEC); SETATTRH(12, MODE_DEC); SETATTRH(14, MODE_DEC);
#undef SETATTRH
#define SETATTRR(NR, MODE) ATTRCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rows<NR, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, AESGCM_BODY_LDS))
    SETATTRR(10, MODE_ENC); SETATTRR(12, MODE_ENC); SETATTRR(14, MODE_ENC); SETATTRR(10, MODE_DEC); SETATTRR(12, MODE_DEC); SETATTRR(14, MODE_DEC);
#undef SETATTRR
    ATTRCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fold), hipFuncAttributeMaxDynamicSharedMemorySize, FOLD_LDS_CLOSE_BYTES));
#define SETATTRB(NR, D) ATTRCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pktg<NR, D, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, PKTG_LDS_TOTAL(2))); \
    ATTRCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pktg<NR, D, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, PKTG_LDS_TOTAL(3))); \
    ATTRCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pktg<NR, D, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, PKTG_LDS_TOTAL(4))); \
    ATTRCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pktg<NR, D, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, PKTG_LDS_TOTAL(6))); \
    ATTRCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pktl<NR, D, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, AESGCM_PKTL_LDS)); \
    ATTRCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pktl<NR, D, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, AESGCM_PKTL_LDS))
    SETATTRB(10, 0); SETATTRB(12, 0); SETATTRB(14, 0); SETATTRB(10, 1); SETATTRB(12, 1); SETATTRB(14, 1);
#undef SETATTRB
#define SETATTRS(NR, D) ATTRCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pktgs<NR, D, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, PKTG_LDS_TOTAL(2))); \
    ATTRCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pktgs<NR, D, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, PKTG_LDS_TOTAL(3))); \
    ATTRCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pktgs<NR, D, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, PKTG_LDS_TOTAL(4))); \
    ATTRCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pktls<NR, D>), hipFuncAttributeMaxDynamicSharedMemorySize, AESGCM_PKTL_LDS))
    SETATTRS(10, 0); SETATTRS(12, 0); SETATTRS(14, 0); SETATTRS(10, 1); SETATTRS(12, 1); SETATTRS(14, 1);
#undef SETATTRS
#define SETATTRP(NR) ATTRCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pktg<NR, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, PKTG_LDS_TOTAL(2))); \
    ATTRCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pktg<NR, 2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, PKTG_LDS_TOTAL(3))); \
    ATTRCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pktg<NR, 2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, PKTG_LDS_TOTAL(4))); \
    ATTRCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pktl<NR, 2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, AESGCM_PKTL_LDS))
    SETATTRP(10); SETATTRP(12); SETATTRP(14);                     // the probes of the packet kernels (aesgcm_frames_ceiling_probe_dev)
#undef SETATTRP
    ATTRCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_combine), hipFuncAttributeMaxDynamicSharedMemorySize, CMB_LDS_BYTES));
    ATTRCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_combine_batch), hipFuncAttributeMaxDynamicSharedMemorySize, CMB_LDS_BYTES));
#define SETATTRB3(NR, D) ATTRCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_batch3<NR, D, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, BATCH3_LDS_BYTES_LG(6))); \
                         ATTRCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_batch3<NR, D, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, BATCH3_LDS_BYTES_LG(4))); \
                         ATTRCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_batch3<NR, D, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, BATCH3_LDS_BYTES_LG(3)))
    SETATTRB3(10, 0); SETATTRB3(12, 0); SETATTRB3(14, 0); SETATTRB3(10, 1); SETATTRB3(12, 1); SETATTRB3(14, 1);
#undef SETATTRB3
#define SETATTRB3P(NR) ATTRCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_batch3<NR, 2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, BATCH3_LDS_BYTES_LG(3)))
    SETATTRB3P(10); SETATTRB3P(12); SETATTRB3P(14);
#undef SETATTRB3P
#undef ATTRCHK
    return hipSuccess;
}
hipError_t klaunch_init_tables(DevTables *t) { hipLaunchKernelGGL(k_init_tables, dim3(1), dim3(256), 0, 0, t); return hipGetLastError(); }
hipError_t klaunch_setup(hipStream_t st, KeyMaterial *km, const DevTables *tb, const uint8_t *d_key, int key_len, int pre_nr, u32 G) {
    hipLaunchKernelGGL(k_setup, dim3(1), dim3(AESGCM_WG), 0, st, km, tb, d_key, key_len, pre_nr, G);
    hipLaunchKernelGGL(k_setup_ptab, dim3(AESGCM_NPTAB + AESGCM_NLTAB), dim3(512), 0, st, km);
    return hipGetLastError();
}
hipError_t klaunch_gfmul(const uint4 *h, const uint4 *x, uint4 *z, size_t n) { hipLaunchKernelGGL(k_gfmul, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, h, x, z, n); return hipGetLastError(); }
hipError_t klaunch_copy16(hipStream_t st, uint4 *dst, const uint4 *src, u64 n16) { hipLaunchKernelGGL(k_copy16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, st, dst, src, n16); return hipGetLastError(); }
hipError_t klaunch_fill_splitmix64(hipStream_t st, unsigned blocks, u64 *buf, size_t n_words, size_t tail_bytes, u64 seed, u64 first_word) {
    hipLaunchKernelGGL(k_fill_splitmix64, dim3(blocks), dim3(256), 0, st, buf, n_words, tail_bytes, seed, first_word);
    return hipGetLastError();
}
hipError_t klaunch_body(int mode, int nr, bool cyc, bool half, unsigned wgs, hipStream_t st, const KeyMaterial *km, const DevTables *tb, const BodyParams &p) {
#define LY(NR, M, CYC) hipLaunchKernelGGL((k_body<NR, M, CYC>), dim3(wgs), dim3(AESGCM_BODY_WG), AESGCM_BODY_LDS + (CYC ? CYC_LDS_PARK_BYTES : 0u), st, km, tb, p)
#define LH(NR, M) hipLaunchKernelGGL((k_bodyh<NR, M>), dim3(wgs), dim3(AESGCM_BODYH_WG), AESGCM_LDS_BYTES + CYC_LDS_PARK_BYTES, st, km, tb, p)
    if (half) {
        if (mode == MODE_DEC)    { if (nr == 10) LH(10, MODE_DEC); else if (nr == 12) LH(12, MODE_DEC); else LH(14, MODE_DEC); }
        else                     { if (nr == 10) LH(10, MODE_ENC); else if (nr == 12) LH(12, MODE_ENC); else LH(14, MODE_ENC); }
    }
    else if (cyc) {
        if (mode == MODE_DEC)    { if (nr == 10) LY(10, MODE_DEC, true); else if (nr == 12) LY(12, MODE_DEC, true); else LY(14, MODE_DEC, true); }
        else                     { if (nr == 10) LY(10, MODE_ENC, true); else if (nr == 12) LY(12, MODE_ENC, true); else LY(14, MODE_ENC, true); }
    }
    else if (mode == MODE_DEC)   { if (nr == 10) LY(10, MODE_DEC, false); else if (nr == 12) LY(12, MODE_DEC, false); else LY(14, MODE_DEC, false); }
    else if (mode == MODE_PROBE) { if (nr == 10) LY(10, MODE_PROBE, false); else if (nr == 12) LY(12, MODE_PROBE, false); else LY(14, MODE_PROBE, false); }
    else                         { if (nr == 10) LY(10, MODE_ENC, false); else if (nr == 12) LY(12, MODE_ENC, false); else LY(14, MODE_ENC, false); }
#undef LY
#undef LH
    return hipGetLastError();
}
hipError_t klaunch_fold(unsigned wgs, bool closing, hipStream_t st, const KeyMaterial *km, const FoldParams &p) {
    hipLaunchKernelGGL(k_fold, dim3(wgs), dim3(FOLD_WG), closing ? FOLD_LDS_CLOSE_BYTES : FOLD_LDS_BYTES, st, km, p);
    return hipGetLastError();
}
hipError_t klaunch_combine(hipStream_t st, const KeyMaterial *km, const DevTables *tb, const CombineParams &p) {
    hipLaunchKernelGGL(k_combine, dim3(1), dim3(COMBINE_THREADS), CMB_LDS_BYTES, st, km, tb, p);
    return hipGetLastError();
}
hipError_t klaunch_combine_batch(unsigned n, hipStream_t st, const KeyMaterial *km, const DevTables *tb, const CombineBatch &b) {
    hipLaunchKernelGGL(k_combine_batch, dim3(n), dim3(COMBINE_THREADS), CMB_LDS_BYTES, st, km, tb, b);
    return hipGetLastError();
}
hipError_t klaunch_pktl(int nr, int dec, bool ilp, unsigned wgs, hipStream_t st, const KeyMaterial *km, const DevTables *tb, const PktParams &p) {
    if (p.scattered) {                                          // messages wherever they live: k_pktls (no ILP form: a routed call takes a lane per packet only when the packets fill the chip)
#define LS(NR, D) hipLaunchKernelGGL((k_pktls<NR, D>), dim3(wgs), dim3(AESGCM_PKTL_WG), AESGCM_PKTL_LDS, st, km, tb, p)
        if (dec) { if (nr == 10) LS(10, 1); else if (nr == 12) LS(12, 1); else LS(14, 1); }
        else     { if (nr == 10) LS(10, 0); else if (nr == 12) LS(12, 0); else LS(14, 0); }
#undef LS
        return hipGetLastError();
    }
#define LPI(NR, D, I) hipLaunchKernelGGL((k_pktl<NR, D, I>), dim3(wgs), dim3(I ? AESGCM_PKTL_WG_ILP : AESGCM_PKTL_WG), AESGCM_PKTL_LDS, st, km, tb, p)
#define LP(NR, D) do { if (ilp) LPI(NR, D, 1); else LPI(NR, D, 0); } while (0)
    if (dec == 2) { if (ilp) return hipErrorInvalidValue; if (nr == 10) LPI(10, 2, 0); else if (nr == 12) LPI(12, 2, 0); else LPI(14, 2, 0); }      // the probe: the 768-lane form
    else if (dec) { if (nr == 10) LP(10, 1); else if (nr == 12) LP(12, 1); else LP(14, 1); }
    else     { if (nr == 10) LP(10, 0); else if (nr == 12) LP(12, 0); else LP(14, 0); }
#undef LPI
#undef LP
    return hipGetLastError();
}
hipError_t klaunch_pktg(int nr, int dec, int lg, unsigned wgs, hipStream_t st, const KeyMaterial *km, const DevTables *tb, const PktParams &p) {
    if (p.scattered) {                                          // messages wherever they live: k_pktgs, lane groups of 4 / 8 / 16
        if (lg != 2 && lg != 3 && lg != 4) return hipErrorInvalidValue;
#define LSG(NR, D, LG) hipLaunchKernelGGL((k_pktgs<NR, D, LG>), dim3(wgs), dim3(PKTG_WG(LG)), PKTG_LDS_TOTAL(LG), st, km, tb, p)
#define LS(NR, D) do { if (lg == 2) LSG(NR, D, 2); else if (lg == 3) LSG(NR, D, 3); else LSG(NR, D, 4); } while (0)
        if (dec) { if (nr == 10) LS(10, 1); else if (nr == 12) LS(12, 1); else LS(14, 1); }
        else     { if (nr == 10) LS(10, 0); else if (nr == 12) LS(12, 0); else LS(14, 0); }
#undef LS
#undef LSG
        return hipGetLastError();
    }
#define LPG(NR, D, LG) hipLaunchKernelGGL((k_pktg<NR, D, LG>), dim3(wgs), dim3(PKTG_WG(LG)), PKTG_LDS_TOTAL(LG), st, km, tb, p)
#define LP(NR, D) do { if (lg == 2) LPG(NR, D, 2); else if (lg == 3) LPG(NR, D, 3); else if (lg == 4) LPG(NR, D, 4); else LPG(NR, D, 6); } while (0)
#define LPP(NR) do { if (lg == 2) LPG(NR, 2, 2); else if (lg == 3) LPG(NR, 2, 3); else LPG(NR, 2, 4); } while (0)
    if (dec == 2) { if (lg != 2 && lg != 3 && lg != 4) return hipErrorInvalidValue; if (nr == 10) LPP(10); else if (nr == 12) LPP(12); else LPP(14); }      // the probe: lane groups of 4 / 8 / 16
    else if (dec) { if (nr == 10) LP(10, 1); else if (nr == 12) LP(12, 1); else LP(14, 1); }
    else     { if (nr == 10) LP(10, 0); else if (nr == 12) LP(12, 0); else LP(14, 0); }
#undef LP
#undef LPP
#undef LPG
    return hipGetLastError();
}
hipError_t klaunch_batch3(int nr, int dec, int lg, unsigned wgs, hipStream_t st, const DevTables *tb, const BatchParams &p) {
#define LB3(NR, D, LG) hipLaunchKernelGGL((k_batch3<NR, D, LG>), dim3(wgs), dim3(BATCH3_LANES(NR)), BATCH3_LDS_BYTES_LG(LG), st, tb, p)
#define LB3N(D, LG) do { if (nr == 10) LB3(10, D, LG); else if (nr == 12) LB3(12, D, LG); else LB3(14, D, LG); } while (0)
    if (dec == 2) { if (lg != 3) return hipErrorInvalidValue; LB3N(2, 3); }   // the probe exists in the shape of BASELINE config 5
    else if (lg == 3) { if (dec) LB3N(1, 3); else LB3N(0, 3); }
    else if (lg == 4) { if (dec) LB3N(1, 4); else LB3N(0, 4); }
    else { if (dec) LB3N(1, 6); else LB3N(0, 6); }
#undef LB3N
#undef LB3
    return hipGetLastError();
}
hipError_t klaunch_len_sort(hipStream_t st, const LenSrc &src, u32 n, u32 *bins, u32 *perm, const RouteCfg &rc, u64 *bad_part, u32 *host_status, const DescSrc &ds) {
    if (n <= LEN_SORT1_MAX) {                                                                // a small call: the whole sort in one launch
        hipLaunchKernelGGL(k_len_sort1, dim3(1), dim3(1024), 0, st, src, n, perm, rc, (volatile u32 *)host_status, ds);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_len_hist, dim3(LEN_SORT_WGS), dim3(256), 0, st, src, n, bins, (unsigned long long *)bad_part);
    hipLaunchKernelGGL(k_len_scan, dim3(1), dim3(1024), 0, st, bins, rc, (const unsigned long long *)bad_part, (volatile u32 *)host_status);
    hipLaunchKernelGGL(k_len_scatter, dim3(LEN_SORT_WGS), dim3(256), 0, st, src, n, bins, perm, (const RowsHdr *)rc.hdr, ds);
    return hipGetLastError();
}
hipError_t klaunch_rows_plan(hipStream_t st, const RowsParams &p, bool routed, u32 force_d, u32 nb_cap, u64 *part, u32 *host_status) {
    RowsPlan a;
    a.off = p.data_off; a.aoff = p.aad_off; a.len_arr = p.len_arr; a.alen_arr = p.alen_arr; a.pkt_len = p.pkt_len; a.aad_len = p.aad_len; a.n = p.n_pkts; a.waves = p.waves; a.force_d = force_d;
    a.nb_cap = nb_cap; a.slot_cap = p.slot_cap; a.nwg = (p.n_pkts + 1023u) / 1024u; a.routed = routed ? 1u : 0u;
    a.hdr = const_cast<RowsHdr *>(p.hdr); a.prefix = const_cast<u64 *>(p.prefix); a.sprefix = const_cast<u64 *>(p.sprefix); a.slot_base = const_cast<u32 *>(p.slot_base); a.part = part;
    a.host_status = host_status;
    if (a.n <= ROWS_PLAN_ONE_WG) {
        hipLaunchKernelGGL(k_rows_plan, dim3(1), dim3(1024), 0, st, a);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_rows_plan_sums, dim3(a.nwg), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(k_rows_plan_cut, dim3(1), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(k_rows_plan_place, dim3(a.nwg), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(k_rows_plan_slots, dim3(1), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(k_rows_plan_base, dim3(a.nwg), dim3(1024), 0, st, a);
    return hipGetLastError();
}
hipError_t klaunch_rows(int nr, int dec, unsigned wgs, hipStream_t st, const KeyMaterial *km, const DevTables *tb, const RowsParams &p) {
#define LR(NR, M) hipLaunchKernelGGL((k_rows<NR, M>), dim3(wgs), dim3(AESGCM_BODY_WG), AESGCM_BODY_LDS, st, km, tb, p)
    if (dec) { if (nr == 10) LR(10, MODE_DEC); else if (nr == 12) LR(12, MODE_DEC); else LR(14, MODE_DEC); }
    else     { if (nr == 10) LR(10, MODE_ENC); else if (nr == 12) LR(12, MODE_ENC); else LR(14, MODE_ENC); }
#undef LR
    return hipGetLastError();
}
hipError_t klaunch_rows_close(int dec, unsigned wgs, hipStream_t st, const KeyMaterial *km, const DevTables *tb, const RowsParams &p) {
    if (dec) hipLaunchKernelGGL(k_rows_close<1>, dim3(wgs), dim3(ROWS_CLOSE_WG), 0, st, km, tb, p);
    else hipLaunchKernelGGL(k_rows_close<0>, dim3(wgs), dim3(ROWS_CLOSE_WG), 0, st, km, tb, p);
    return hipGetLastError();
}
hipError_t klaunch_wipe_failed(hipStream_t st, unsigned char *out, const int *auth, const u64 *data_off, u32 n_pkts, u32 pkt_len, const u64 *out_ptr, const u32 *len_arr) {
    hipLaunchKernelGGL(k_wipe_failed, dim3((n_pkts + 3u) / 4u), dim3(256), 0, st, out, auth, data_off, n_pkts, pkt_len, out_ptr, len_arr);
    return hipGetLastError();
}
