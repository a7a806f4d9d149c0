// CPU harness: runs the SAME per-lane code the HIP kernels run (csrc/aesgcm_dev.h is __host__
// __device__) over an emulated launch -- LDS image, workgroup/lane geometry, front padding, Horner
// with K, per-lane tail powers, workgroup fold, k_combine fold -- and compares ciphertext and tag with
// the oracle (oracle/aesgcm_oracle.c, linked in).  It exists because the build container has no GPU:
// it pins the arithmetic and the index algebra before any GPU minute is spent.  Test infrastructure only.
#include "../../aes-gcm-128-192-256-bits_amd/csrc/aesgcm_dev.h"
#include "../../aes-gcm-128-192-256-bits_amd/csrc/aesgcm_rows.h"
#include "aesgcm_bs.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

extern "C" {
int orc_key_expand(const uint8_t *key, size_t key_len, uint8_t *rk, int *nr);
int orc_aes_encrypt_block(const uint8_t *rk, int nr, const uint8_t in[16], uint8_t out[16]);
void orc_gfmul(const uint8_t h[16], const uint8_t x[16], uint8_t z[16]);
int orc_gcm_crypt(int dec, const uint8_t *key, size_t key_len, const uint8_t iv[12], const uint8_t *aad, size_t aad_len,
                  const uint8_t *in, size_t len, uint8_t *out, uint8_t tag[16]);
void orc_ghash_update(const uint8_t h[16], uint8_t y[16], const uint8_t *data, size_t len);
void orc_fill_splitmix64(uint8_t *buf, size_t len, uint64_t seed, uint64_t first_word);
uint8_t orc_sbox(uint8_t x);
}

static int g_fail = 0;
#define CHECK(cond, ...) do { if (!(cond)) { g_fail++; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

static DevTables g_tb;
static void init_tables() { for (u32 x = 0; x < 256; x++) { u32 s = sbox_calc(x); g_tb.sbox[x] = (uint8_t)s; g_tb.te0[x] = te0_calc(s); g_tb.te1[x] = rotl32(te0_calc(s), 8); g_tb.te2[x] = rotl32(te0_calc(s), 16); g_tb.te3[x] = rotl32(te0_calc(s), 24); } }

// emulated k_setup (same barrier structure: compute every lane's product, then commit)
static void emu_setup(KeyMaterial *km, const uint8_t *key, int key_len, int pre_nr, u32 G) {
    static uint4 tab[AESGCM_NPW];
    setup_lane0(km, g_tb.sbox, key, key_len, pre_nr, G, tab);
    for (int d = 0; d < 4; d++) {
        for (int j = 0; j < AESGCM_LOG_WG; j++) {
            static uint4 prod[AESGCM_WG]; static bool act[AESGCM_WG];
            for (int tid = 0; tid < AESGCM_WG; tid++) act[tid] = setup_level(tab, j, tid, &prod[tid]);
            for (int tid = 0; tid < AESGCM_WG; tid++) if (act[tid]) tab[(1 << j) + tid] = prod[tid];
        }
        for (int k = 0; k < AESGCM_NPW; k++) km->pw[d][k] = tab[k];
        if (d == 1) for (int tid = 0; tid < AESGCM_WG; tid++) setup_beta_lane(km, tab, tid);
        if (d < 3) { uint4 next = tab[AESGCM_WG]; tab[0] = gf_one_mo(); tab[1] = next; }
    }
    for (u32 k = 0; k < AESGCM_NPTAB; k++) for (u32 tid = 0; tid < 512; tid++) setup_ptab_lane(km, k, tid);
    for (u32 e = 0; e < AESGCM_NLTAB; e++) for (u32 tid = 0; tid < 32; tid++) setup_ltab_lane(km, e, tid);
}

static void xor_g(G128 &a, const G128 &b) { for (int k = 0; k < 4; k++) a.w[k] ^= b.w[k]; }

template <int NR, int MODE>
static void emu_main_nr(const KeyMaterial *km, const MainParams &p) {
    static unsigned char smem[AESGCM_LDS_BYTES] __attribute__((aligned(16)));
    constexpr bool GH = (MODE == MODE_ENC || MODE == MODE_DEC);
    for (u32 tid = 0; tid < AESGCM_MAIN_WG; tid++) main_fill_lds(smem, km, &g_tb, tid, GH);   // image is workgroup independent
    // the dispenser hands chunk c to whichever wave asks next; results do not depend on who that is, so the
    // harness walks the chunks in a scrambled order to make any accidental order dependence visible
    for (u32 k = 0; k < p.C; k++) {
        const u32 c = (p.C - 1) - k;      // reverse order
        for (u32 lane = 0; lane < 64; lane++) {
            const CtrConsts cc = main_lane_consts<MODE>(km, p, smem, lane);
            uint4 acc = main_chunk_lane<NR, MODE>(km, p, smem, cc, c, lane);
            if (GH) p.parts[(size_t)c * 64 + lane] = acc;
        }
    }
}
static void emu_main(int mode, const KeyMaterial *km, const MainParams &p) {
#define D(NR) switch (mode) { case MODE_ENC: emu_main_nr<NR, MODE_ENC>(km, p); break; case MODE_DEC: emu_main_nr<NR, MODE_DEC>(km, p); break; \
                              case MODE_KS: emu_main_nr<NR, MODE_KS>(km, p); break; default: emu_main_nr<NR, MODE_ECB>(km, p); }
    if (km->nr == 10) { D(10) } else if (km->nr == 12) { D(12) } else { D(14) }
#undef D
}
// emulated cyc_close(): the fused closing of a cyclic launch, workgroup by workgroup with the same lane pieces (tree levels, lane terms from the staged
// tables, the weight through a two-table Shoup form, the terms that occur once in workgroup 0); the accumulator slots are plain XORs here
static void emu_cyc_close(const KeyMaterial *km, const BodyParams &p, const uint4 *items, const uint4 *last, uint4 ej0) {
    static unsigned char smem[CYC_LDS_END] __attribute__((aligned(16)));
    const u32 wgs = BODY_CYC_WAVES / 16;
    G128 slots[CYC_ACC_SLOTS];
    memset(slots, 0, sizeof slots);
    u32 arrived = 0;
    for (u32 gg = 0; gg < wgs; gg++) {
        const u32 g = (gg * 37u + 5u) % wgs;                               // any arrival order
        for (u32 tid = 0; tid < 1024; tid++) {
            reinterpret_cast<uint4 *>(smem + CYC_LDS_TREE_TAB)[tid] = (&km->ptab[0][0])[tid];
            reinterpret_cast<uint4 *>(smem + CYC_LDS_TREE_TAB)[1024u + tid] = (&km->ptab[0][0])[1024u + tid];
            for (u32 k = tid; k < 2048; k += 1024) {
                *reinterpret_cast<uint4 *>(smem + cyc_ltab_off(k)) = cyc_ltab_entry(km, k, p.tb);
                if (g == 0 && p.tb) *reinterpret_cast<uint4 *>(smem + cyc_ltab_off(k, CYC_LDS_LTAB0)) = cyc_ltab_entry(km, k, 0u);
            }
            *reinterpret_cast<uint4 *>(smem + cyc_stage_off(0) + (tid >> 6) * 1024u + (tid & 63u) * 16u) = items[(size_t)(16 * g + (tid >> 6)) * 64 + (tid & 63u)];
        }
        uint4 y[64];
        for (u32 level = 0; level < 4; level++)
            for (u32 wv = 0; wv < (8u >> level); wv++) for (u32 lane = 0; lane < 64; lane++) {
                const uint4 v = cyc_tree_lane(smem, level, wv, lane);
                if (level < 3) *reinterpret_cast<uint4 *>(smem + cyc_stage_off(level + 1) + wv * 1024u + lane * 16u) = v; else y[lane] = v;
            }
        G128 z = {{0, 0, 0, 0}};
        for (u32 lane = 0; lane < 64; lane++) xor_g(z, cyc_lane_term_lds(smem, y[lane], lane));
        if (g + 1 != wgs) {
            for (u32 lane = 0; lane < 32; lane++) *reinterpret_cast<uint4 *>(smem + CYC_LDS_WTAB + 16u * lane) = shoup2_entry(mo_to_be(km->pw[1][wgs - 1 - g]), lane);
            z = shoup2_gmul_lds(z, reinterpret_cast<const uint4 *>(smem + CYC_LDS_WTAB));
        }
        if (g == 0) {
            if (p.tb) for (u32 lane = 0; lane < 64; lane++) xor_g(z, cyc_lane_term_lds(smem, last[lane], lane, CYC_LDS_LTAB0));
            xor_g(z, tag_len_term(km, p.aad_len, p.ct_len));
            xor_g(z, mo_to_be(ej0));
        }
        xor_g(slots[g & (CYC_ACC_SLOTS - 1u)], z);
        ++arrived;
    }
    CHECK(arrived == wgs, "cyc_close arrivals");
    G128 t = {{0, 0, 0, 0}};
    for (u32 k = 0; k < CYC_ACC_SLOTS; k++) xor_g(t, slots[k]);
    *p.tag_out = be_to_mo(t);
}
template <int NR, int MODE>
static void emu_body_nr(const KeyMaterial *km, const BodyParams &p) {
    static unsigned char smem[AESGCM_LDS_BYTES_T4] __attribute__((aligned(16)));
    for (u32 tid = 0; tid < AESGCM_MAIN_WG; tid++) main_fill_lds(smem, km, &g_tb, tid, true, AESGCM_MAIN_WG, p.cyc ? GH_TAB_K2P18 : GH_TAB_K256);
#if AESGCM_T4
    for (u32 tid = 0; tid < AESGCM_MAIN_WG; tid++) fill_lds_t4(smem, &g_tb, tid, AESGCM_MAIN_WG);     // second table region (T1 | T3)
#endif
    if (p.cyc) {                                                        // cyclic rows: one strand and one item per wave (k_body<.., true>)
        for (u32 k = 0; k < BODY_CYC_WAVES; k++) {
            const u32 w = (k * 2741u + 17u) % BODY_CYC_WAVES;           // scrambled order (2741 is odd: a permutation)
            for (u32 lane = 0; lane < 64; lane++) {
                const CtrConsts cc = ctr_round1_consts(p.iv0, p.iv1, p.iv2, km->rk, smem, (lane & 31u) << 2);
                p.parts[(size_t)w * 64 + lane] = body_cyc_lane<NR, MODE>(km, &g_tb, p, smem, cc, w, lane);
                if (w == 0 && p.tb) p.parts[(size_t)BODY_CYC_WAVES * 64 + lane] = body_cyc_last_lane<NR, MODE>(km, p, smem, cc, lane);
            }
        }
        if (p.fuse) {                                                   // the launch closes the tag itself
            const CtrConsts cc = ctr_round1_consts(p.iv0, p.iv1, p.iv2, km->rk, smem, 0);
            u32 s0, s1, s2, s3;
            ctr_rounds_lds<NR>(bswap32(1u), cc, s0, s1, s2, s3, km->rk, smem, 0);
            emu_cyc_close(km, p, p.parts, p.parts + (size_t)BODY_CYC_WAVES * 64, make_uint4(s0, s1, s2, s3));
        }
        return;
    }
    for (u32 k = 0; k < p.C; k++) {
        const u32 c = (k * 7 + 3) % p.C == k ? k : (p.C - 1) - k;      // scrambled order (any permutation will do)
        for (u32 lane = 0; lane < 64; lane++) {
            const CtrConsts cc = ctr_round1_consts(p.iv0, p.iv1, p.iv2, km->rk, smem, (lane & 31u) << 2);
            p.parts[(size_t)c * 64 + lane] = body_chunk_lane<NR, MODE>(km, &g_tb, p, smem, cc, c, lane);
        }
    }
}
static void emu_body(int mode, const KeyMaterial *km, const BodyParams &p) {
#define D(NR) if (mode == MODE_DEC) emu_body_nr<NR, MODE_DEC>(km, p); else emu_body_nr<NR, MODE_ENC>(km, p);
    if (km->nr == 10) { D(10) } else if (km->nr == 12) { D(12) } else { D(14) }
#undef D
}
static G128 emu_pow_h(const KeyMaterial *km, u64 e) { return gf_pow_h_serial(km, e); }
// emulated k_combine: the same lane pieces (in-launch item fold, one level of per-lane table multiplies through km->ltab)
static void emu_combine(const KeyMaterial *km, const CombineParams &p0) {
    static unsigned char smem[CMB_LDS_BYTES] __attribute__((aligned(16)));
    CombineParams p = p0;
    const bool tag = p.want_tag != 0, items = p.kind == PARTS_ITEM;
    auto tp = [&](u64 e) -> const uint4 * { int k = ptab_index(e); return k < 0 ? nullptr : km->ptab[k]; };
    if (items && p.np > 1) { p.tabA = tp(p.eA); p.tabB = p.np > 4 ? tp(4 * p.eA) : nullptr; p.tabC = p.np > 16 ? tp(16 * p.eA) : nullptr; }
    CHECK(!(items && p.np > 1) || (p.np <= COMBINE_MAX_ITEMS && p.tabA && (p.np <= 4 || p.tabB) && (p.np <= 16 || p.tabC)), "emu_combine: %u items not foldable", p.np);
    const u32 J1 = items ? fold4_groups(p.np) : 0, J2 = items ? fold4_groups(J1) : 0;
    if (items && p.np > 1) memcpy(smem + CMB_LDS_TABA, p.tabA, 8192);
    if (J1 > 1) memcpy(smem + CMB_LDS_TABB, p.tabB, 8192);
    if (J2 > 1) memcpy(smem + CMB_LDS_TABC, p.tabC, 8192);
    memcpy(smem + CMB_LDS_SBOX, g_tb.sbox, 256);
    for (u32 w = 0; w < J1; w++) for (u32 lane = 0; lane < 64; lane++)
        *reinterpret_cast<uint4 *>(smem + CMB_LDS_STAGE1 + w * 1024u + lane * 16u) = combine_fold_items(combine_fold_load(p, w, lane), smem, CMB_LDS_TABA);
    for (u32 w = 0; w < J2; w++) for (u32 lane = 0; lane < 64; lane++)
        *reinterpret_cast<uint4 *>(smem + CMB_LDS_STAGE2 + w * 1024u + lane * 16u) = combine_fold_staged(smem, CMB_LDS_STAGE1, J1, w, CMB_LDS_TABB, lane);
    G128 acc = {{0, 0, 0, 0}};
    if (items) {
        for (u32 lane = 0; lane < 64; lane++) xor_g(acc, shoup2_gmul(mo_to_be(combine_fold_staged(smem, CMB_LDS_STAGE2, J2, 0, CMB_LDS_TABC, lane)), km->ltab[(tag ? 65u : 63u) - lane + (p.tail_item ? p.tail_blocks : 0u)]));
        if (p.tail_item) for (u32 lane = 0; lane < 64; lane++) xor_g(acc, shoup2_gmul(mo_to_be(p.tail_item[lane]), km->ltab[(tag ? 65u : 63u) - lane]));
    } else if (p.kind == PARTS_GATHERED) {
        for (u32 tid = 0; tid < p.np; tid++) {
            G128 z = mo_to_be(p.parts[(size_t)tid * (p.stride ? p.stride : 1u)]);
            if (tag) z = shoup2_gmul(z, km->ltab[2]);
            xor_g(acc, z);
        }
    }
    if (tag) {
        xor_g(acc, tag_len_term(km, p.aad_len, p.ct_len));
        xor_g(acc, p.ej0 ? mo_to_be(*p.ej0) : combine_ej0_bytes(km, smem + CMB_LDS_SBOX, p));
        if (p.has_carry && !p.e_carry) xor_g(acc, shoup2_gmul(mo_to_be(*p.carry), km->ltab[2]));
    }
    if (!tag && p.e) acc = gf_mul(acc, emu_pow_h(km, p.e));
    if (p.has_carry && (!tag || p.e_carry)) {
        G128 c = mo_to_be(*p.carry);
        if (p.e_carry) c = gf_mul(c, emu_pow_h(km, p.e_carry));
        if (tag) c = gf_mul(c, mo_to_be(km->pw[0][2]));
        xor_g(acc, c);
    }
    *p.out = be_to_mo(acc);
}

struct Parts { const uint4 *ptr; u32 np; u32 gathered; u64 eA; const uint4 *tail_item = nullptr; u32 tail_blocks = 0; };   // gathered = PARTS_* kind; eA: item spacing when k_combine folds the items itself
struct Emu {
    KeyMaterial km; u32 tw; std::vector<uint4> parts, fold_a, fold_b;
    u32 n_cyc = 0;                  // launches that went through run_cyc()
    u32 n_fold_close = 0;           // messages whose tag came out of k_fold's closing
    bool fuse = true;               // whole messages: the cyclic launch closes the tag itself (cyc_close)
    bool cyc = false; u64 cyc_min = 1024;   // ranges with at least one whole body row as cyclic rows of k_body (every size) instead of the pieces
    Emu(const uint8_t *key, int key_len, u32 tw_) : tw(tw_), parts(1 << 16), fold_a(1 << 16), fold_b(1 << 12) { emu_setup(&km, key, key_len, 0, 512); }
    // mirrors enqueue_fold(): k_fold launches until one item is left
    // mirrors k_fold's closing (FoldClose): the workgroups of the first level turn their output items into the tag
    bool fold_close = true;
    void fold_closing(const FoldParams &f, u32 G, u64 aad_len, u64 ct_len, uint4 ej0, uint4 *tag_out) {
        static unsigned char lds[FOLD_LDS_CLOSE_BYTES] __attribute__((aligned(16)));
        for (u32 k = 0; k < 2048; k++) *reinterpret_cast<uint4 *>(lds + cyc_ltab_off(k, FOLD_LDS_LTAB)) = cyc_ltab_entry(&km, k, 0u);
        const u64 step = fold_out_step(f);
        G128 slots[CYC_ACC_SLOTS];
        memset(slots, 0, sizeof slots);
        for (u32 gg = 0; gg < G; gg++) {
            const u32 g = (gg * 29u + 3u) % G == gg ? gg : G - 1 - gg;      // any arrival order
            G128 z = {{0, 0, 0, 0}};
            for (u32 lane = 0; lane < 64; lane++) xor_g(z, cyc_lane_term_lds(lds, f.out[(size_t)g * 64 + lane], lane, FOLD_LDS_LTAB));
            const u64 e = step * (u64)(G - 1 - g);
            for (u32 d = 0; d < 4; d++) {
                const u32 dig = (u32)(e >> (AESGCM_LOG_WG * d)) & (u32)(AESGCM_WG - 1);
                if (!dig) continue;
                for (u32 lane = 0; lane < 32; lane++) *reinterpret_cast<uint4 *>(lds + FOLD_LDS_WTAB + 16u * lane) = shoup2_entry(mo_to_be(km.pw[d][dig]), lane);
                z = shoup2_gmul_lds(z, reinterpret_cast<const uint4 *>(lds + FOLD_LDS_WTAB));
            }
            if (g + 1 == G) { xor_g(z, tag_len_term(&km, aad_len, ct_len)); xor_g(z, mo_to_be(ej0)); }
            xor_g(slots[g & (CYC_ACC_SLOTS - 1u)], z);
        }
        G128 t = {{0, 0, 0, 0}};
        for (u32 k = 0; k < CYC_ACC_SLOTS; k++) xor_g(t, slots[k]);
        *tag_out = be_to_mo(t);
    }
    Parts fold(const uint4 *items, u32 n, u32 period, u64 eA, u64 eB, const CombineParams *close = nullptr, uint4 *close_tag = nullptr) {
        static unsigned char smem[FOLD_LDS_BYTES] __attribute__((aligned(16)));
        const uint4 *cur = items; int which = 0;
        bool first = true;
        while (n > 1) {
            {   // mirrors enqueue_fold(): the last level(s) belong to k_combine when the spacing has tables
                auto has = [&](u64 e) { return ptab_index(e) >= 0; };
                if (period <= 1 && n <= COMBINE_MAX_ITEMS && has(eA) && (n <= 4 || has(4 * eA)) && (n <= 16 || has(16 * eA))) { Parts q = {cur, n, PARTS_ITEM, eA}; return q; }
            }
            std::vector<uint4> &ob = which ? fold_b : fold_a;
            const u32 G = fold_wgs(n, fold_group(n, period));
            if (ob.size() < (size_t)G * 64) ob.resize((size_t)G * 64);
            FoldParams f;
            plan_fold(f, cur, ob.data(), n, period, eA, eB);
            auto tp = [&](u64 e) -> const uint4 * { int k = ptab_index(e); return k < 0 ? nullptr : km.ptab[k]; };
            f.tabA = tp(f.eA); f.tabB = tp(f.eB); f.tabC = tp(f.eC);
            for (u32 tid = 0; tid < FOLD_WG; tid++) {
                fold_fill_lds(smem, &km, f.tabA, f.eA, 0u, tid, FOLD_WG);
                if (period > 1) fold_fill_lds(smem, &km, f.tabB, f.eB, 8192u, tid, FOLD_WG);
                fold_fill_lds(smem, &km, f.tabC, f.eC, 16384u, tid, FOLD_WG);
            }
            for (u32 g = 0; g < G; g++) {
                u32 start, end;
                const u32 J = fold_wg_range(n, f.group, g, &start, &end);
                for (u32 w = 0; w < J; w++) for (u32 lane = 0; lane < 64; lane++)
                    *reinterpret_cast<uint4 *>(smem + FOLD_LDS_TAB + w * 1024u + lane * 16u) = fold_wave_lane(f, smem, start, end, J, w, lane);
                for (u32 lane = 0; lane < 64; lane++) f.out[(size_t)g * 64 + lane] = fold_wg_lane(smem, J, lane);
            }
            if (close && fold_close && G <= 512) {      // (first is kept for the bookkeeping below)      // mirrors enqueue_fold(): the first level closes a whole message itself
                CHECK(close->ej0 != nullptr, "fold closing without E_K(J0)");
                fold_closing(f, G, close->aad_len, close->ct_len, *close->ej0, close_tag);
                Parts q = {nullptr, 0, PARTS_NONE, 0};
                return q;
            }
            first = false;
            eA = fold_out_step(f); eB = 0; period = 1; cur = f.out; n = G; which ^= 1;
        }
        Parts r = {cur, 1, PARTS_ITEM, 0};
        return r;
    }
    // mirrors enqueue_main(): launch + k_fold levels
    Parts run(int mode, const uint8_t *iv, const void *aad, u64 aad_len, const void *in, u64 len, void *out, u64 first_block) {
        MainParams p; memset(&p, 0, sizeof p);
        u32 C = plan_main(p, mode, tw, iv, aad, aad_len, in, len, out, first_block, nullptr);
        Parts r = {nullptr, 0, PARTS_NONE, 0};
        if (!C) return r;
        if (parts.size() < (size_t)C * 64) parts.resize((size_t)C * 64);
        p.parts = parts.data();
        emu_main(mode, &km, p);
        if (mode != MODE_ENC && mode != MODE_DEC) return r;
        // mirrors enqueue_main(): few chunks with table-backed spacing go to k_combine unfolded
        const u64 eA = (u64)64 * p.Tw;
        auto has = [&](u64 e) { return ptab_index(e) >= 0; };
        if (C <= COMBINE_MAX_ITEMS && (C == 1 || (has(eA) && (C <= 4 || has(4 * eA)) && (C <= 16 || has(16 * eA))))) { Parts q = {parts.data(), C, PARTS_ITEM, eA}; return q; }
        return fold(parts.data(), C, 1, eA, 0);
    }
    // mirrors enqueue_body(): k_body + k_fold with the interleaved first level
    Parts run_body(int mode, const uint8_t *iv, const BodySplit &b, const void *in, void *out, u64 first_block, const CombineParams *close = nullptr, uint4 *close_tag = nullptr) {
        BodyParams p; memset(&p, 0, sizeof p);
        if (parts.size() < 256 * (size_t)b.S) parts.resize(256 * (size_t)b.S);
        plan_body(p, b, iv, in, out, first_block, parts.data());
        emu_body(mode, &km, p);
        return fold(parts.data(), p.C, 4, 64, (u64)256 * b.T, close, close_tag);
    }
    // mirrors enqueue_cyc(): the whole range as cyclic rows of k_body + one k_fold level; false = the range is not of that size
    bool run_cyc(int mode, const uint8_t *iv, const void *aad, u64 aad_len, const void *in, u64 len, void *out, u64 first_block, Parts *po, uint4 *fused_tag = nullptr) {
        if (!cyc) return false;
        if (parts.size() < (size_t)64 * (BODY_CYC_WAVES + 1)) parts.resize((size_t)64 * (BODY_CYC_WAVES + 1));
        BodyParams p;
        if (!plan_body_cyc(p, mode, iv, aad, aad_len, in, len, out, first_block, parts.data(), cyc_min, ~0ull)) return false;
        if (fused_tag && fuse) { p.fuse = 1; p.aad_len = aad_len; p.ct_len = len; p.tag_out = fused_tag; }
        emu_body(mode, &km, p);
        ++n_cyc;
        if (p.fuse) { po->np = 0; return true; }
        *po = fold(parts.data(), BODY_CYC_WAVES, 1, 64, 0);
        if (p.tb) { po->tail_item = parts.data() + (size_t)BODY_CYC_WAVES * 64; po->tail_blocks = p.tb; }
        return true;
    }
    // mirrors absorb_range()
    bool absorb(int mode, const uint8_t *iv, const uint8_t *aad, u64 aad_len, const uint8_t *in, u64 len, uint8_t *out, u64 first_block, u64 body_min, uint4 *Y) {
        BodySplit b;
        {   // mirrors absorb_range(): cyclic rows first
            Parts pc = {nullptr, 0, PARTS_NONE, 0};
            if (run_cyc(mode, iv, aad, aad_len, in, len, out, first_block, &pc)) {
                const u64 nb = (aad_len + 15) / 16 + (len + 15) / 16;
                emu_combine(&km, combine_with_items(plan_combine_carry(pc.ptr, pc.np, pc.gathered, Y, nb), pc.eA, pc.tail_item, pc.tail_blocks));
                return true;
            }
        }
        if (!plan_body_split(len, first_block, tw, body_min, &b)) {
            Parts pp = run(mode, iv, aad, aad_len, in, len, out, first_block);
            const u64 nb = (aad_len + 15) / 16 + (len + 15) / 16;
            if (nb) emu_combine(&km, combine_with_items(plan_combine_carry(pp.ptr, pp.np, pp.gathered, Y, nb), pp.eA));
            return false;
        }
        const u64 n_aad = (aad_len + 15) / 16;
        if (n_aad + b.head_blocks) {
            Parts pp = run(mode, iv, aad, aad_len, in, 16 * b.head_blocks, out, first_block);
            emu_combine(&km, combine_with_items(plan_combine_carry(pp.ptr, pp.np, pp.gathered, Y, n_aad + b.head_blocks), pp.eA));
        }
        Parts pb = run_body(mode, iv, b, in, out, first_block);
        emu_combine(&km, combine_with_items(plan_combine_carry(pb.ptr, pb.np, pb.gathered, Y, b.body_blocks), pb.eA));
        const u64 done = b.head_blocks + b.body_blocks, tail = len - 16 * done;
        if (tail) {
            Parts pp = run(mode, iv, nullptr, 0, in + 16 * done, tail, out + 16 * done, first_block + done);
            emu_combine(&km, combine_with_items(plan_combine_carry(pp.ptr, pp.np, pp.gathered, Y, (tail + 15) / 16), pp.eA));
        }
        return true;
    }
    // mirrors crypt_dev() for a message that takes the split; returns whether the split applied
    bool crypt_split(int dec, const uint8_t iv[12], const uint8_t *aad, u64 aad_len, const uint8_t *in, u64 len, uint8_t *out, uint8_t tag[16], u64 body_min) {
        uint4 Y = make_uint4(0, 0, 0, 0), t;
        {   // mirrors crypt_dev(): cyclic rows first, items straight to the tag
            Parts pc = {nullptr, 0, PARTS_NONE, 0};
            if (run_cyc(dec ? MODE_DEC : MODE_ENC, iv, aad, aad_len, in, len, out, 0, &pc, &t)) {
                if (!fuse) emu_combine(&km, combine_with_items(plan_combine_tag(pc.ptr, pc.np, pc.gathered, iv, aad_len, len, &t), pc.eA, pc.tail_item, pc.tail_blocks));
                memcpy(tag, &t, 16);
                return true;
            }
        }
        BodySplit b0;
        if (plan_body_split(len, 0, tw, body_min, &b0) && !aad_len && !b0.head_blocks && len == 16 * b0.body_blocks) {
            // the whole message is one aligned body: k_body's items go straight to the tag (no chaining value)
            CombineParams q0 = plan_combine_tag(nullptr, 0, PARTS_NONE, iv, 0, len, &t);
            const uint4 ej0 = be_to_mo(combine_ej0_bytes(&km, g_tb.sbox, q0));      // what k_body's chunk-0 wave leaves behind
            q0.ej0 = &ej0;
            Parts pb = run_body(dec ? MODE_DEC : MODE_ENC, iv, b0, in, out, 0, &q0, &t);
            if (pb.gathered != PARTS_NONE || pb.np) emu_combine(&km, combine_with_items(plan_combine_tag(pb.ptr, pb.np, pb.gathered, iv, 0, len, &t), pb.eA));
            else ++n_fold_close;
            memcpy(tag, &t, 16);
            return true;
        }
        const bool split = absorb(dec ? MODE_DEC : MODE_ENC, iv, aad, aad_len, in, len, out, 0, body_min, &Y);
        emu_combine(&km, plan_combine_final(&Y, iv, aad_len, len, &t));
        memcpy(tag, &t, 16);
        return split;
    }
    void crypt(int dec, const uint8_t iv[12], const uint8_t *aad, u64 aad_len, const uint8_t *in, u64 len, uint8_t *out, uint8_t tag[16]) {
        Parts pp = run(dec ? MODE_DEC : MODE_ENC, iv, aad, aad_len, in, len, out, 0);
        uint4 t;
        if (pp.np == 1 && pp.gathered == PARTS_ITEM) {
            // mirrors k_main's tail (single-chunk message: no k_combine launch): every term one table multiply deep
            G128 P = {{0, 0, 0, 0}};
            for (u32 lane = 0; lane < 64; lane++) xor_g(P, tag_lane_term(&km, pp.ptr[lane], lane));
            xor_g(P, tag_len_term(&km, aad_len, len));
            CombineParams q = plan_combine_tag(nullptr, 0, PARTS_NONE, iv, aad_len, len, &t);
            xor_g(P, combine_ej0_bytes(&km, g_tb.sbox, q));
            t = be_to_mo(P);
        } else {
            emu_combine(&km, combine_with_items(plan_combine_tag(pp.ptr, pp.np, pp.gathered, iv, aad_len, len, &t), pp.eA));
        }
        memcpy(tag, &t, 16);
    }
};

static std::vector<uint8_t> rnd(size_t n, u64 seed) {
    std::vector<uint8_t> v(n + 16);     // slack keeps .data() valid for n = 0
    orc_fill_splitmix64(v.data(), n, seed, 0);
    v.resize(n);
    return v;
}
// 16-byte aligned copy (the device data path requires aligned in/out)
struct ABuf { uint8_t *p; size_t n; ABuf(size_t n_) : n(n_) { p = (uint8_t *)aligned_alloc(16, (n + 31) / 16 * 16 + 16); memset(p, 0xA5, (n + 31) / 16 * 16 + 16); } ~ABuf() { free(p); } };

static void test_units() {
    for (u32 x = 0; x < 256; x++) CHECK(g_tb.sbox[x] == orc_sbox((uint8_t)x), "sbox[%u]", x);
    // k_fold level plan for every item count and both first-level periods: each level's output fits the buffer it is written to
    // (A, B, A, ...), the walk ends at <= COMBINE_MAX_ITEMS items, and the group is a power of two and a multiple of the period
    for (u32 period : {1u, 4u}) for (u32 n0 = COMBINE_MAX_ITEMS + 1; n0 <= AESGCM_MAX_CHUNKS; n0 += (n0 < 70000 ? 1 : 4099)) {
        u32 n = n0, per = period, which = 0, levels = 0;
        while (n > COMBINE_MAX_ITEMS || per > 1) {
            const u32 g = fold_group(n, per), G = fold_wgs(n, g);
            CHECK(g >= 1 && g <= FOLD_GROUP && (g & (g - 1)) == 0 && g % per == 0, "fold_group(%u, %u) = %u", n, per, g);
            CHECK(G <= (which ? FOLD_B_ITEMS : FOLD_A_ITEMS), "fold level %u of n0 = %u leaves %u items", levels, n0, G);
            CHECK(G < n || n == 1, "fold level does not shrink: %u -> %u", n, G);
            n = G; per = 1; which ^= 1; if (++levels > 4) break;
        }
        CHECK(levels <= 3, "n0 = %u needs %u k_fold levels", n0, levels);
    }
    // gf_mul vs oracle
    for (int i = 0; i < 200; i++) {
        auto a = rnd(16, 100 + i), b = rnd(16, 900 + i);
        if (i == 0) { memset(a.data(), 0, 16); a[0] = 0x80; }
        uint4 am, bm; memcpy(&am, a.data(), 16); memcpy(&bm, b.data(), 16);
        uint4 z = gf_mul_mo(am, bm);
        uint8_t zo[16]; orc_gfmul(b.data(), a.data(), zo);
        CHECK(memcmp(&z, zo, 16) == 0, "gf_mul %d", i);
    }
}

// bitsliced AES (tests/host_emul/aesgcm_bs.h: the round-2 experiment that measured 0.7 x; kept under test because its S-box circuit is tools/sbox_lut3.py's output): the LUT3-mapped S-box on all 256 inputs, the transpose, and whole blocks (32 per
// "lane") for the three key sizes against the oracle's literal cipher
static void test_bitslice() {
    {   // S-box: byte values 0..255 in 8 lanes-worth of 32 slots
        for (u32 base = 0; base < 256; base += 32) {
            u32 x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (u32 i = 0; i < 32; i++) for (u32 b = 0; b < 8; b++) x[b] |= (((base + i) >> b) & 1u) << i;
            bs_sbox(x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]);
            for (u32 i = 0; i < 32; i++) {
                u32 y = 0;
                for (u32 b = 0; b < 8; b++) y |= ((x[b] >> i) & 1u) << b;
                CHECK(y == orc_sbox((uint8_t)(base + i)), "bs_sbox %u", base + i);
            }
        }
    }
    {   // transpose
        u32 w[32], ref[32];
        auto r = rnd(128, 4242);
        memcpy(w, r.data(), 128); memcpy(ref, r.data(), 128);
        bs_transpose32(w);
        for (u32 p = 0; p < 32; p++) for (u32 i = 0; i < 32; i++) CHECK(((w[i] >> p) & 1u) == ((ref[p] >> i) & 1u), "transpose %u %u", p, i);
    }
    for (int kl : {16, 24, 32}) {
        auto key = rnd((size_t)kl, 5000 + kl);
        uint8_t rk[240]; int nr = 0;
        orc_key_expand(key.data(), (size_t)kl, rk, &nr);
        static u32 rkm[15 * 128];
        bs_key_masks(rk, nr, rkm);
        auto in = rnd(32 * 16, 6000 + kl);
        u32 st[128];
        for (int q = 0; q < 128; q++) st[q] = 0;
        for (u32 i = 0; i < 32; i++) for (u32 j = 0; j < 16; j++) for (u32 b = 0; b < 8; b++) st[8 * j + b] |= (u32)((in[16 * i + j] >> b) & 1u) << i;
        for (int q = 0; q < 128; q++) st[q] ^= rkm[q];                      // AddRoundKey 0
        bs_rounds(st, rkm, nr);
        for (u32 i = 0; i < 32; i++) {
            uint8_t want[16], got[16];
            orc_aes_encrypt_block(rk, nr, in.data() + 16 * i, want);
            for (u32 j = 0; j < 16; j++) { got[j] = 0; for (u32 b = 0; b < 8; b++) got[j] |= (uint8_t)(((st[8 * j + b] >> i) & 1u) << b); }
            CHECK(memcmp(want, got, 16) == 0, "bitsliced AES-%d block %u", 8 * kl, i);
        }
    }
}

// batch-path pieces: Shoup 4-bit multiply against the bit-serial multiply, in-kernel key schedule against the oracle
static void test_batch_pieces() {
    static unsigned char smem[BATCH_LDS_BYTES] __attribute__((aligned(16)));
    for (u32 tid = 0; tid < AESGCM_WG; tid++) main_fill_lds(smem, nullptr, &g_tb, tid, false, AESGCM_WG);
    for (u32 v = 0; v < 16; v++) *reinterpret_cast<u32 *>(smem + BATCH_LDS_RTAB_OFF + 4 * v) = shoup_rem_calc(v);
    for (int it = 0; it < 50; it++) {
        auto cb = rnd(16, 7000 + it), yb = rnd(16, 8000 + it);
        if (it == 1) memset(yb.data(), 0xFF, 16);
        if (it == 2) { memset(yb.data(), 0, 16); yb[15] = 1; }
        uint4 cm, ym; memcpy(&cm, cb.data(), 16); memcpy(&ym, yb.data(), 16);
        const G128 c = mo_to_be(cm), y = mo_to_be(ym);
        const u32 tab = 256 * (it % 8);
        for (u32 v = 0; v < 16; v++) { G128 e = shoup_entry(c, v); *reinterpret_cast<uint4 *>(smem + tab + 16 * v) = make_uint4(e.w[0], e.w[1], e.w[2], e.w[3]); }
        const G128 z = shoup_mul(y, smem, tab), want = gf_mul(y, c);
        CHECK(memcmp(&z, &want, 16) == 0, "shoup_mul %d", it);
        {   // byte-wise variant with the two tables Th, Tl (k_batch3)
            static unsigned char t2[512] __attribute__((aligned(16)));
            for (u32 v = 0; v < 16; v++) {
                const G128 e = shoup_entry(c, v), el = gf_mulx4(e);
                *reinterpret_cast<uint4 *>(t2 + 16 * v) = make_uint4(e.w[0], e.w[1], e.w[2], e.w[3]);
                *reinterpret_cast<uint4 *>(t2 + 256 + 16 * v) = make_uint4(el.w[0], el.w[1], el.w[2], el.w[3]);
            }
            const G128 z2 = shoup2_mul(y, t2, 0u);
            CHECK(memcmp(&z2, &want, 16) == 0, "shoup2_mul %d", it);
            const G128 z3 = shoup2_mul_dr(y, t2, 0u);                  // round 4: the same product with the reduction delayed
            CHECK(memcmp(&z3, &want, 16) == 0, "shoup2_mul_dr %d", it);
            u32 Vo[6], Vh[6];                                           // ... and split over an owner and a helper lane (k_batch3 at 8 lanes per packet)
            shoup2_half_dr(y.w[0], y.w[1], t2, 0u, Vo);
            shoup2_half_dr(y.w[2], y.w[3], t2, 0u, Vh);
            const G128 z4 = shoup2_pair_join(Vo, Vh);
            CHECK(memcmp(&z4, &want, 16) == 0, "shoup2_half_dr / pair_join %d", it);
        }
        const G128 s1 = gf_sqr(y), s2 = gf_mul(y, y), s3 = gf_sqr(c), s4 = gf_mul(c, c);
        CHECK(memcmp(&s1, &s2, 16) == 0 && memcmp(&s3, &s4, 16) == 0, "gf_sqr %d", it);
    }
    for (int klen : {16, 24, 32}) {
        auto key = rnd(klen, 9000 + klen);
        uint8_t rkb[240]; int nr; orc_key_expand(key.data(), klen, rkb, &nr);
        u32 rk[60];
        for (u32 lane : {0u, 17u, 63u}) {
            const u32 lb = (lane & 31u) << 2;
            if (klen == 16) batch_key_expand<10>(key.data(), rk, smem, lb);
            else if (klen == 24) batch_key_expand<12>(key.data(), rk, smem, lb);
            else batch_key_expand<14>(key.data(), rk, smem, lb);
            CHECK(memcmp(rk, rkb, 16 * (nr + 1)) == 0, "batch_key_expand %d lane %u", klen, lane);
        }
    }
}

// packets under one key, 2^LG lanes per packet (k_pktg): per-packet IV/AAD/length; lane accumulators, the H^2 step with the
// length block in lane G-2, the cross-lane tree with wave-uniform constants, E_K(J0) from the one-lane-per-packet pass
template <int NR, int DEC, int LG>
static uint4 emu_pktg_packet(const KeyMaterial *km, const PktParams &p, const unsigned char *smem, u32 pkt, u32 extra_iters, u32 grp) {
    constexpr u32 G = 1u << LG;
    const PktInfo q = pkt_info(p, pkt);
    const u32 iters = pktg_iters(q, G) + extra_iters;             // a wave runs to the longest packet of its groups
    uint4 acc[G];
    for (u32 l = 0; l < G; l++) acc[l] = pktg_close_lane<LG>(pktg_lane<NR, DEC, LG>(km, p, q, smem, l, grp * G + l, iters, true), q, smem, l);
    for (int j = 0; j < LG; j++) {
        uint4 o[G];
        for (u32 l = 0; l < G; l++) o[l] = pktg_tree_offer(acc[l], smem, j);
        for (u32 l = 0; l < G; l++) if (l & (1u << j)) acc[l] = xor4(acc[l], o[l ^ (1u << j)]);
    }
    return xor4(acc[G - 1], pktg_ej0_lane<NR>(km, p, smem, pkt, (pkt * 5 + 3) % 64));
}
template <int LG>
static uint4 emu_pktg(const KeyMaterial *km, int dec, const PktParams &p, const unsigned char *smem, u32 pkt, u32 extra, u32 grp) {
    if (km->nr == 10) return dec ? emu_pktg_packet<10, 1, LG>(km, p, smem, pkt, extra, grp) : emu_pktg_packet<10, 0, LG>(km, p, smem, pkt, extra, grp);
    if (km->nr == 12) return dec ? emu_pktg_packet<12, 1, LG>(km, p, smem, pkt, extra, grp) : emu_pktg_packet<12, 0, LG>(km, p, smem, pkt, extra, grp);
    return dec ? emu_pktg_packet<14, 1, LG>(km, p, smem, pkt, extra, grp) : emu_pktg_packet<14, 0, LG>(km, p, smem, pkt, extra, grp);
}
static void test_packets(int key_len, u64 seed) {
    auto key = rnd(key_len, seed);
    Emu E(key.data(), key_len, 0);
    const u32 lens[] = {0, 1, 15, 16, 17, 48, 1000, 1008, 1024, 4096, 4100, 70000, 0, 256, 240};
    const u32 aads[] = {0, 20, 28, 16, 0, 33, 68, 0, 5, 0, 12, 64, 16, 0, 16};
    const int n = 15;
    std::vector<u64> doff(n + 1, 0), aoff(n + 1, 0);
    for (int i = 0; i < n; i++) { doff[i + 1] = doff[i] + lens[i]; aoff[i + 1] = aoff[i] + aads[i]; }
    ABuf in(doff[n]), out(doff[n]);
    auto aad = rnd(aoff[n], seed + 1), ivs = rnd(12 * n, seed + 2);
    orc_fill_splitmix64(in.p, doff[n], seed + 3, 0);
    std::vector<uint8_t> tags(16 * n);
    PktParams p; memset(&p, 0, sizeof p);
    p.ivs = ivs.data(); p.aad = aad.data(); p.in = in.p; p.out = out.p; p.tags = tags.data();
    p.data_off = doff.data(); p.aad_off = aoff.data(); p.n_pkts = n; p.aligned = 1;
    static unsigned char smem[PKTG_LDS_BYTES(6)] __attribute__((aligned(16)));
    for (int lg : {2, 3, 4, 6}) {
        memset(smem, 0xEE, sizeof smem);
        for (u32 tid = 0; tid < AESGCM_PKT_WG; tid++) pktg_fill_lds(smem, &E.km, &g_tb, tid, AESGCM_PKT_WG, lg);
        for (int dec = 0; dec < 2; dec++) {
            p.in = dec ? out.p : in.p; p.out = out.p;           // decrypt in place
            for (u32 pkt = 0; pkt < (u32)n; pkt++) {
                const u32 extra = (pkt % 3 == 1) ? 2 : 0, grp = pkt & ((64u >> lg) - 1u);
                const uint4 t = lg == 2 ? emu_pktg<2>(&E.km, dec, p, smem, pkt, extra, grp) : lg == 3 ? emu_pktg<3>(&E.km, dec, p, smem, pkt, extra, grp)
                              : lg == 4 ? emu_pktg<4>(&E.km, dec, p, smem, pkt, extra, grp) : emu_pktg<6>(&E.km, dec, p, smem, pkt, extra, grp);
                std::vector<uint8_t> ref(lens[pkt] + 16); uint8_t rtag[16];
                if (!dec) {
                    orc_gcm_crypt(0, key.data(), key_len, ivs.data() + 12 * pkt, aad.data() + aoff[pkt], aads[pkt], in.p + doff[pkt], lens[pkt], ref.data(), rtag);
                    CHECK(memcmp(ref.data(), out.p + doff[pkt], lens[pkt]) == 0, "pktg ct %u lg %d", pkt, lg);
                    memcpy(tags.data() + 16 * pkt, &t, 16);
                } else {
                    CHECK(memcmp(in.p + doff[pkt], out.p + doff[pkt], lens[pkt]) == 0, "pktg dec %u lg %d", pkt, lg);
                    memcpy(rtag, tags.data() + 16 * pkt, 16);
                }
                CHECK(memcmp(&t, rtag, 16) == 0, "pktg tag %u len %u aad %u dec %d lg %d", pkt, lens[pkt], aads[pkt], dec, lg);
            }
        }
    }
    // the same packets, one lane per packet (k_pktl): tags must equal the ones above, decrypt restores the input
    static unsigned char smem_h[AESGCM_LDS_BYTES] __attribute__((aligned(16)));
    for (u32 tid = 0; tid < AESGCM_MAIN_WG; tid++) main_fill_lds(smem_h, &E.km, &g_tb, tid, true, AESGCM_MAIN_WG, GH_TAB_H);
    ABuf out2(doff[n]);
    std::vector<uint8_t> tags2(16 * n + 16), tags3(16 * n + 16);
    std::vector<int> auth(n, -1);
    for (int dec = 0; dec < 2; dec++) {
        p.in = dec ? out2.p : in.p; p.out = out2.p;
        p.tags = dec ? tags3.data() + 1 : tags2.data();        // an unaligned tag array takes the bytewise store
        p.expect = dec ? tags2.data() : nullptr; p.auth = dec ? auth.data() : nullptr;
        for (u32 pkt = 0; pkt < (u32)n; pkt++) {
            const u32 lane = (pkt * 7) % 64;
            if (E.km.nr == 10) { if (dec) pktl_lane<10, 1>(&E.km, p, smem_h, pkt, lane); else pktl_lane<10, 0>(&E.km, p, smem_h, pkt, lane); }
            else if (E.km.nr == 12) { if (dec) pktl_lane<12, 1>(&E.km, p, smem_h, pkt, lane); else pktl_lane<12, 0>(&E.km, p, smem_h, pkt, lane); }
            else { if (dec) pktl_lane<14, 1>(&E.km, p, smem_h, pkt, lane); else pktl_lane<14, 0>(&E.km, p, smem_h, pkt, lane); }
        }
        if (!dec) {
            CHECK(memcmp(tags2.data(), tags.data(), 16 * n) == 0, "pktl tags");
            std::vector<uint8_t> ref(doff[n] + 16); uint8_t rtag[16];
            for (int k = 0; k < n; k++) {
                orc_gcm_crypt(0, key.data(), key_len, ivs.data() + 12 * k, aad.data() + aoff[k], aads[k], in.p + doff[k], lens[k], ref.data(), rtag);
                CHECK(memcmp(ref.data(), out2.p + doff[k], lens[k]) == 0, "pktl ct %d", k);
            }
        } else {
            CHECK(memcmp(tags3.data() + 1, tags.data(), 16 * n) == 0, "pktl dec tags");
            CHECK(memcmp(out2.p, in.p, doff[n]) == 0, "pktl dec data");
            for (int k = 0; k < n; k++) CHECK(auth[k] == 1, "pktl auth %d", k);
        }
    }
}

static void test_key(int key_len, u32 G /* rows per chunk override, 0 = production rule */, u64 seed, const std::vector<std::pair<u64, u64>> &sizes) {
    auto key = rnd(key_len, seed);
    Emu E(key.data(), key_len, G);
    uint8_t rk[240]; int nr;
    orc_key_expand(key.data(), key_len, rk, &nr);
    CHECK((int)E.km.nr == nr && memcmp(E.km.rk_bytes, rk, 16 * (nr + 1)) == 0, "key schedule len %d", key_len);
    uint8_t zero[16] = {0}, h[16];
    orc_aes_encrypt_block(rk, nr, zero, h);
    CHECK(memcmp(&E.km.h, h, 16) == 0, "H");
    // power tables spot checks: pw[0][k] = H^k via oracle repeated multiply
    {
        uint8_t acc[16] = {0x80};
        for (int k = 0; k <= AESGCM_WG; k++) {
            if (k == 0 || k == 1 || k == 2 || k == 3 || k == 255 || k == 256 || k == 257 || k == 511 || k == 512 || k == AESGCM_WG - 1 || k == AESGCM_WG)
                CHECK(memcmp(&E.km.pw[0][k], acc, 16) == 0, "pw[0][%d]", k);
            orc_gfmul(h, acc, acc);
        }
        CHECK(memcmp(&E.km.pw[1][1], &E.km.pw[0][AESGCM_WG], 16) == 0, "beta");
        uint4 b2 = gf_mul_mo(E.km.pw[1][1], E.km.pw[1][1]);
        CHECK(memcmp(&E.km.pw[1][2], &b2, 16) == 0, "beta^2");
        uint4 g3 = gf_mul_mo(gf_mul_mo(E.km.pw[2][1], E.km.pw[2][1]), E.km.pw[2][1]);
        CHECK(memcmp(&E.km.pw[2][3], &g3, 16) == 0, "gamma^3");
        CHECK(memcmp(&E.km.pw[3][1], &E.km.pw[2][AESGCM_WG], 16) == 0, "delta");
    }
    // the five-bit GHASH tables of the three fixed Horner constants: Y * c through ghash_mul_const_lds against the bit-serial
    // product, on every single-bit value (each of the 26 groups, the three that straddle a dword included) and random ones
    for (int which = 0; which < 3; which++) {
        static unsigned char smem[AESGCM_LDS_BYTES] __attribute__((aligned(16)));
        for (u32 tid = 0; tid < AESGCM_MAIN_WG; tid++) main_fill_lds(smem, &E.km, &g_tb, tid, true, AESGCM_MAIN_WG, which);
        const uint4 c = which == GH_TAB_K64 ? E.km.pw[0][64] : which == GH_TAB_H ? E.km.h : E.km.pw[0][256];
        for (int i = 0; i < 128 + 64; i++) {
            uint4 y;
            if (i < 128) { u32 w[4] = {0, 0, 0, 0}; w[i >> 5] = 1u << (i & 31); y = make_uint4(w[0], w[1], w[2], w[3]); }
            else { auto r = rnd(16, seed * 131 + i); memcpy(&y, r.data(), 16); }
            const uint4 a = ghash_mul_const_lds(y, smem), b = gf_mul_mo(y, c);
            CHECK(memcmp(&a, &b, 16) == 0, "ghash_mul_const_lds table %d case %d", which, i);
        }
    }
    // ECB through the LDS round code
    {
        const size_t nb = 700;
        ABuf in(16 * nb), out(16 * nb);
        orc_fill_splitmix64(in.p, 16 * nb, seed + 5, 0);
        E.run(MODE_ECB, nullptr, nullptr, 0, in.p, 16 * nb, out.p, 0);
        for (size_t i = 0; i < nb; i++) { uint8_t o[16]; orc_aes_encrypt_block(rk, nr, in.p + 16 * i, o); CHECK(memcmp(o, out.p + 16 * i, 16) == 0, "ecb block %zu", i); }
    }
    for (auto &sz : sizes) {
        const u64 al = sz.first, n = sz.second;
        auto iv = rnd(12, seed + 11 + n), aad = rnd(al, seed + 12 + al);
        ABuf pt(n), ct(n), ref(n), back(n);
        orc_fill_splitmix64(pt.p, n, seed + 13, 0);
        uint8_t tag[16], rtag[16], dtag[16];
        // misalign the AAD on purpose for odd sizes
        std::vector<uint8_t> aad_buf(al + 32);
        uint8_t *aadp = aad_buf.data() + ((al & 1) ? 3 : 0);
        if (al) memcpy(aadp, aad.data(), al);
        E.crypt(0, iv.data(), aadp, al, pt.p, n, ct.p, tag);
        orc_gcm_crypt(0, key.data(), key_len, iv.data(), aad.data(), al, pt.p, n, ref.p, rtag);
        CHECK(memcmp(ct.p, ref.p, n) == 0, "ct  key %d G %u aad %llu len %llu", key_len, G, (unsigned long long)al, (unsigned long long)n);
        CHECK(memcmp(tag, rtag, 16) == 0, "tag key %d G %u aad %llu len %llu", key_len, G, (unsigned long long)al, (unsigned long long)n);
        CHECK(ct.p[n] == 0xA5, "overrun past ragged tail");
        E.crypt(1, iv.data(), aadp, al, ct.p, n, back.p, dtag);
        CHECK(memcmp(back.p, pt.p, n) == 0 && memcmp(dtag, rtag, 16) == 0, "dec key %d G %u aad %llu len %llu", key_len, G, (unsigned long long)al, (unsigned long long)n);
    }
}

// shards: split one message over R ranks (each with its own emulated context), gather, finalize
static void test_shards(int key_len, u32 G, u64 al, u64 n, int R, u64 seed) {
    auto key = rnd(key_len, seed), iv = rnd(12, seed + 1), aad = rnd(al, seed + 2);
    ABuf pt(n), ct(n), ref(n);
    orc_fill_splitmix64(pt.p, n, seed + 3, 0);
    uint8_t rtag[16];
    orc_gcm_crypt(0, key.data(), key_len, iv.data(), aad.data(), al, pt.p, n, ref.p, rtag);
    Emu E(key.data(), key_len, G);
    const u64 total_blocks = (n + 15) / 16;
    std::vector<uint4> gathered(R);
    u64 first = 0;
    for (int r = 0; r < R; r++) {
        u64 blocks = total_blocks / R + ((u64)r < total_blocks % R ? 1 : 0);
        u64 end = first + blocks;
        u64 len = (end == total_blocks ? n : 16 * end) - 16 * first;
        Parts pp = E.run(MODE_ENC, iv.data(), r == 0 ? aad.data() : nullptr, r == 0 ? al : 0, pt.p + 16 * first, len, ct.p + 16 * first, first);
        emu_combine(&E.km, combine_with_items(plan_combine_poly(pp.ptr, pp.np, pp.gathered, total_blocks - end, &gathered[r]), pp.eA));
        first = end;
    }
    uint4 t;
    emu_combine(&E.km, plan_combine_tag(gathered.data(), R, PARTS_GATHERED, iv.data(), al, n, &t));
    CHECK(memcmp(ct.p, ref.p, n) == 0, "shard ct R %d", R);
    CHECK(memcmp(&t, rtag, 16) == 0, "shard tag key %d G %u aad %llu len %llu R %d", key_len, G, (unsigned long long)al, (unsigned long long)n, R);
}

// head / k_body / tail split (absorb_range): whole messages and shards with arbitrary first blocks
static void test_body(int key_len, u32 G, u64 al, u64 n, u64 seed, bool cyc = false) {
    auto key = rnd(key_len, seed), iv = rnd(12, seed + 1), aad = rnd(al, seed + 2);
    ABuf pt(n), ct(n), ref(n), back(n);
    orc_fill_splitmix64(pt.p, n, seed + 3, 0);
    uint8_t rtag[16], tag[16], dtag[16];
    orc_gcm_crypt(0, key.data(), key_len, iv.data(), aad.data(), al, pt.p, n, ref.p, rtag);
    Emu E(key.data(), key_len, G);
    E.cyc = cyc;
    if (cyc) {                                                          // first with k_fold / k_combine behind the launch, then (below) with the fused closing
        uint8_t utag[16];
        E.fuse = false;
        E.crypt_split(0, iv.data(), aad.data(), al, pt.p, n, ct.p, utag, 4096);
        CHECK(memcmp(utag, rtag, 16) == 0, "cyclic unfused tag key %d aad %llu len %llu", key_len, (unsigned long long)al, (unsigned long long)n);
        E.fuse = true; E.n_cyc = 0;
        memset(ct.p, 0xA5, n);
    }
    const bool split = E.crypt_split(0, iv.data(), aad.data(), al, pt.p, n, ct.p, tag, 4096);
    CHECK(split, "body split did not apply: len %llu G %u", (unsigned long long)n, G);
    CHECK(!cyc || E.n_cyc == 1, "cyclic rows did not take the message: len %llu", (unsigned long long)n);
    CHECK(cyc || al || (n % 4096) || n < 16 * 256 * 40 || E.n_fold_close == 1, "k_fold did not close the message: len %llu", (unsigned long long)n);
    CHECK(memcmp(ct.p, ref.p, n) == 0, "body ct key %d G %u aad %llu len %llu", key_len, G, (unsigned long long)al, (unsigned long long)n);
    CHECK(memcmp(tag, rtag, 16) == 0, "body tag key %d G %u aad %llu len %llu", key_len, G, (unsigned long long)al, (unsigned long long)n);
    CHECK(ct.p[n] == 0xA5, "body overrun");
    E.crypt_split(1, iv.data(), aad.data(), al, ct.p, n, back.p, dtag, 4096);
    CHECK(memcmp(back.p, pt.p, n) == 0 && memcmp(dtag, rtag, 16) == 0, "body dec key %d G %u", key_len, G);
    // the same message as R shards, every shard through absorb() from its own first block
    for (int R : {2, 3}) {
        ABuf ct2(n);
        const u64 total_blocks = (n + 15) / 16;
        std::vector<uint4> gathered(R);
        u64 first = 0; int nsplit = 0;
        for (int r = 0; r < R; r++) {
            u64 blocks = total_blocks / R + ((u64)r < total_blocks % R ? 1 : 0), end = first + blocks;
            u64 len = (end == total_blocks ? n : 16 * end) - 16 * first;
            uint4 Y = make_uint4(0, 0, 0, 0);
            nsplit += E.absorb(MODE_ENC, iv.data(), r == 0 ? aad.data() : nullptr, r == 0 ? al : 0, pt.p + 16 * first, len, ct2.p + 16 * first, first, 4096, &Y);
            CombineParams q = plan_combine_poly(nullptr, 0, PARTS_NONE, 0, &gathered[r]);
            q.carry = &Y; q.has_carry = 1; q.e_carry = total_blocks - end;
            emu_combine(&E.km, q);
            first = end;
        }
        uint4 t;
        emu_combine(&E.km, plan_combine_tag(gathered.data(), R, PARTS_GATHERED, iv.data(), al, n, &t));
        CHECK(nsplit > 0, "no shard took the split");
        CHECK(memcmp(ct2.p, ref.p, n) == 0 && memcmp(&t, rtag, 16) == 0, "body shards R %d key %d G %u", R, key_len, G);
    }
}

// streaming: AAD in chunks, data in chunks, running state carried through combine
static void test_stream(int key_len, u32 G, u64 al, u64 n, u64 chunk, u64 seed) {
    auto key = rnd(key_len, seed), iv = rnd(12, seed + 1), aad = rnd(al, seed + 2);
    ABuf pt(n), ct(n), ref(n);
    orc_fill_splitmix64(pt.p, n, seed + 3, 0);
    uint8_t rtag[16];
    orc_gcm_crypt(0, key.data(), key_len, iv.data(), aad.data(), al, pt.p, n, ref.p, rtag);
    Emu E(key.data(), key_len, G);
    uint4 Y = make_uint4(0, 0, 0, 0);
    for (u64 off = 0; off < al; off += chunk) {
        u64 m = al - off < chunk ? al - off : chunk;
        Parts pp = E.run(MODE_ENC, iv.data(), aad.data() + off, m, nullptr, 0, nullptr, 0);
        emu_combine(&E.km, combine_with_items(plan_combine_carry(pp.ptr, pp.np, pp.gathered, &Y, (m + 15) / 16), pp.eA));
    }
    for (u64 off = 0; off < n; off += chunk) {
        u64 m = n - off < chunk ? n - off : chunk;
        Parts pp = E.run(MODE_ENC, iv.data(), nullptr, 0, pt.p + off, m, ct.p + off, off / 16);
        emu_combine(&E.km, combine_with_items(plan_combine_carry(pp.ptr, pp.np, pp.gathered, &Y, (m + 15) / 16), pp.eA));
    }
    uint4 t;
    emu_combine(&E.km, plan_combine_final(&Y, iv.data(), al, n, &t));
    CHECK(memcmp(ct.p, ref.p, n) == 0, "stream ct");
    CHECK(memcmp(&t, rtag, 16) == 0, "stream tag key %d aad %llu len %llu chunk %llu", key_len, (unsigned long long)al, (unsigned long long)n, (unsigned long long)chunk);
}

static void test_keystream_and_ghash(u64 seed) {
    auto key = rnd(32, seed), iv = rnd(12, seed + 1);
    Emu E(key.data(), 32, 2);
    uint8_t rk[240]; int nr; orc_key_expand(key.data(), 32, rk, &nr);
    const u64 first = 0x01FFFF00ull - 2, nb = 600;      // counter crosses a 2^8, 2^16, 2^24 carry boundary
    ABuf out(16 * nb);
    E.run(MODE_KS, iv.data(), nullptr, 0, out.p, 16 * nb, out.p, first);
    for (u64 i = 0; i < nb; i++) {
        uint8_t cb[16], o[16]; memcpy(cb, iv.data(), 12);
        u32 c = (u32)(2 + first + i); cb[12] = c >> 24; cb[13] = c >> 16; cb[14] = c >> 8; cb[15] = c;
        orc_aes_encrypt_block(rk, nr, cb, o);
        CHECK(memcmp(o, out.p + 16 * i, 16) == 0, "keystream block %llu", (unsigned long long)i);
    }
    // GHASH chaining value Y = P*H
    for (u64 n : {1ull, 16ull, 17ull, 8191ull, 20000ull}) {
        auto d = rnd(n, seed + n);
        Parts pp = E.run(MODE_ENC, iv.data(), d.data(), n, nullptr, 0, nullptr, 0);
        uint4 y; emu_combine(&E.km, combine_with_items(plan_combine_poly(pp.ptr, pp.np, pp.gathered, 1, &y), pp.eA));
        uint8_t yo[16] = {0}; orc_ghash_update((const uint8_t *)&E.km.h, yo, d.data(), n);
        CHECK(memcmp(&y, yo, 16) == 0, "ghash len %llu", (unsigned long long)n);
    }
}

// Many messages under one key by rows (k_rows / k_rows_close, csrc/aesgcm_rows.h): the cut (fixed-size records: arithmetic; offset arrays: the planner's two
// prefix sums), every block in a scrambled order with the same piece walk and lane code as k_rows, the record slots of the closing in a scrambled order with plain
// XORs for the atomics; ciphertext and tags against the oracle, and the zero-at-rest rule of the shared scratch.
// g_route_min != 0 (offset arrays / scattered only): the call is ROUTED per message as k_len_scan + k_rows_plan do on the device (round 6) -- messages whose data + AAD
// lie below the mark are the packet kernels' (here: pktl_lane, a lane each), the row launches must see them as nothing: no unit, no smalls block, no slot, no arrival
static u32 g_route_min = 0;
template <int NR>
static void emu_rows_nr(const KeyMaterial *km, int dec, RowsParams &p, u32 waves, u32 force_d) {
    static unsigned char smem[AESGCM_LDS_BYTES_T4] __attribute__((aligned(16)));
    for (u32 tid = 0; tid < AESGCM_MAIN_WG; tid++) main_fill_lds(smem, km, &g_tb, tid, true, AESGCM_MAIN_WG, GH_TAB_K64);
    for (u32 tid = 0; tid < AESGCM_MAIN_WG; tid++) fill_lds_t4(smem, &g_tb, tid, AESGCM_MAIN_WG);
    const u32 n = p.n_pkts;
    p.waves = waves;
    std::vector<u64> prefix(n + 1, 0), sprefix(n + 1, 0);
    std::vector<u32> slot_base(n + 1, 0);
    RowsHdr hdr;
    memset(&hdr, 0, sizeof hdr);
    size_t slots;
    const bool var = p.data_off || p.aad_off || p.len_arr;
    const u32 route_min = var ? g_route_min : 0u;
    u32 n_small = 0;
    auto geom_m = [&](u32 m) { return rows_geom_of(rows_msg(p, m), route_min); };
    auto na_m = [&](u32 m) { return rows_na_of(rows_msg(p, m), route_min); };
    auto small_m = [&](u32 m) { const RowsMsg q = rows_msg(p, m); return rows_is_small(q.len, q.alen, route_min); };
    if (var) {                                                           // k_rows_plan
        hdr.route_min = route_min;
        p.hdr = &hdr;
        CHECK(rows_route_min(p) == route_min, "rows: route_min");
        for (u32 m = 0; m < n; m++) {
            if (small_m(m)) { ++n_small; CHECK(rows_units(geom_m(m), na_m(m)) == 0 && rows_smalls(geom_m(m), na_m(m)) == 0 && rows_slots(geom_m(m), na_m(m), prefix[m], 7) == 0, "rows: a routed message counts"); }
            prefix[m + 1] = prefix[m] + rows_units(geom_m(m), na_m(m));
            sprefix[m + 1] = sprefix[m] + rows_smalls(geom_m(m), na_m(m));
        }
        hdr.G = prefix[n]; hdr.n_small = n_small;
        rows_cut(hdr.G, waves, force_d, ROWS_NB_CAP, &hdr.D, &hdr.NB, &hdr.dyn);
        for (u32 m = 0; m < n; m++) slot_base[m + 1] = slot_base[m] + rows_slots(geom_m(m), na_m(m), prefix[m], hdr.D);
        slots = ROWS_SLOTS_PER_MSG * (size_t)n + ROWS_NB_CAP;
        CHECK(slot_base[n] <= slots, "rows: %u slots planned, %zu held", slot_base[n], slots);
        p.prefix = prefix.data(); p.sprefix = sprefix.data(); p.slot_base = slot_base.data();
    } else {
        const RowsGeom g = rows_geom(p.pkt_len);
        const u32 na = rows_na(p.aad_len);
        p.U = rows_units(g, na); p.S = rows_smalls(g, na); p.G = (u64)n * p.U;
        rows_cut(p.G, waves, force_d, (u64)1 << 30, &p.D, &p.NB, &p.dyn);
        p.SM = p.U ? rows_nat_count(g, na) + (p.U - 1u) / p.D + 1u : 0u;
        slots = (size_t)n * p.SM;
    }
    p.slot_cap = (u32)slots;
    std::vector<RowsRec> rec(slots + 1);
    memset(rec.data(), 0, rec.size() * sizeof(RowsRec));
    std::vector<unsigned long long> acc(2 * (size_t)n + 2, 0);
    std::vector<u32> cnt(n + 1, 0), made_of(n + 1, 0);
    p.rec = rec.data(); p.acc = acc.data(); p.cnt = cnt.data();
    const u64 G = p.hdr ? p.hdr->G : p.G;
    const u32 D = p.hdr ? p.hdr->D : p.D, NB = p.hdr ? p.hdr->NB : p.NB, dyn = p.hdr ? p.hdr->dyn : p.dyn;
    CHECK(dyn || NB <= waves, "rows: %u blocks for %u waves", NB, waves);
    auto put = [&](u32 slot, const G128 &z, u64 e, u32 m, u32 flags) {
        CHECK(slot < slots && !rec[slot].flags, "rows: slot %u of %zu taken twice (message %u)", slot, slots, m);
        RowsRec r; r.w = z; r.e = e; r.msg = m; r.flags = flags;
        rec[slot] = r;
        ++made_of[m];
    };
    for (u32 kb = 0; kb < NB; kb++) {
        const u32 b = NB - 1 - kb;                                       // last block first
        u64 g = (u64)b * D;
        const u64 g_end = g + D < G ? g + D : G;
        u32 m = rows_find_msg(p, g);
        while (g < g_end) {
            const RowsMsg mq = rows_msg(p, m);
            const RowsGeom geo = geom_m(m);
            const u64 g0 = rows_unit_base(p, m);
            const u32 U = rows_units(geo, na_m(m)), sbase = rows_slot_base(p, m);
            if (U == 0) { ++m; continue; }                                 // a message shorter than a row has no unit: the closing alone sees it
            CHECK(g >= g0 && g < g0 + U, "rows: unit %llu outside message %u", (unsigned long long)g, m);
            while (g < g_end && g < g0 + U) {
                const RowsPiece pc = rows_piece(geo, sbase, g0, (u32)(g - g0), g_end - g, D);
                G128 z = {{0, 0, 0, 0}};
                for (u32 lane = 0; lane < 64; lane++) {
                    const CtrConsts cc = ctr_round1_consts(load_le32(p.ivs + 12 * m), load_le32(p.ivs + 12 * m + 4), load_le32(p.ivs + 12 * m + 8), km->rk, smem, (lane & 31u) << 2);
                    if (pc.kind == ROWS_RUN) {
                        const uint4 a = dec ? rows_run_lane<NR, MODE_DEC>(km, &g_tb, p, mq, pc, smem, cc, lane, 0, 0) : rows_run_lane<NR, MODE_ENC>(km, &g_tb, p, mq, pc, smem, cc, lane, 0, 0);
                        xor_g(z, rows_run_term(km, a, lane));
                    } else if (pc.kind == ROWS_TAIL) {
                        xor_g(z, dec ? rows_tail_lane<NR, 1>(km, p, mq, smem, cc, lane) : rows_tail_lane<NR, 0>(km, p, mq, smem, cc, lane));
                    } else {
                        xor_g(z, rows_aad_lane(km, p, mq, smem, lane));
                    }
                }
                put(pc.slot, z, pc.e, m, pc.kind == ROWS_TAIL ? ROWS_REC_VALID : (ROWS_REC_VALID | ROWS_REC_WEIGH));
                g += pc.len;
            }
            ++m;
        }
    }
    // k_rows_close: the lanes last first; lane i is message i (its length block and E_K(J0)), slot i, and the blocks i, i + lanes, ... of the smalls axis
    u32 finals = 0;
    auto due = [&](u32 m) { const RowsMsg mq = rows_msg(p, m); return rows_pieces(geom_m(m), na_m(m), rows_unit_base(p, m), D); };
    auto arrive = [&](u32 m, const G128 &z) {
        acc[2 * m] ^= ((unsigned long long)z.w[0] << 32) | z.w[1]; acc[2 * m + 1] ^= ((unsigned long long)z.w[2] << 32) | z.w[3];
        if (++cnt[m] == due(m)) {
            G128 t; t.w[0] = (u32)(acc[2 * m] >> 32); t.w[1] = (u32)acc[2 * m]; t.w[2] = (u32)(acc[2 * m + 1] >> 32); t.w[3] = (u32)acc[2 * m + 1];
            acc[2 * m] = acc[2 * m + 1] = 0; cnt[m] = 0;
            store_block_bytes(p.tags + (size_t)m * 16, be_to_mo(t), 16);
            ++finals;
        }
    };
    // the messages of the packet kernels: a lane each (k_pktl), reading the same arrays
    if (n_small) {
        static unsigned char smem_h[AESGCM_LDS_BYTES] __attribute__((aligned(16)));
        for (u32 tid = 0; tid < AESGCM_MAIN_WG; tid++) main_fill_lds(smem_h, km, &g_tb, tid, true, AESGCM_MAIN_WG, GH_TAB_H);
        PktParams k; memset(&k, 0, sizeof k);
        k.ivs = p.ivs; k.aad = p.aad; k.in = p.in; k.out = p.out; k.tags = p.tags; k.data_off = p.data_off; k.aad_off = p.aad_off; k.n_pkts = n; k.aad_len = p.aad_len; k.aligned = 1;
        hdr.sc_in = (u64)(uintptr_t)p.in_ptr; hdr.sc_out = (u64)(uintptr_t)p.out_ptr; hdr.sc_aad = (u64)(uintptr_t)p.aad_ptr; hdr.sc_len = (u64)(uintptr_t)p.len_arr; hdr.sc_alen = (u64)(uintptr_t)p.alen_arr;
        k.route = &hdr; k.scattered = p.len_arr ? 1u : 0u;
        for (u32 m = 0; m < n; m++) {
            if (!small_m(m)) continue;
            if (k.scattered) { if (dec) pktl_lane<NR, 1, false, false, true>(km, k, smem_h, m, (m * 5u) % 64u); else pktl_lane<NR, 0, false, false, true>(km, k, smem_h, m, (m * 5u) % 64u); }
            else if (dec) pktl_lane<NR, 1>(km, k, smem_h, m, (m * 5u) % 64u); else pktl_lane<NR, 0>(km, k, smem_h, m, (m * 5u) % 64u);
        }
    }
    const size_t lanes = slots > n ? slots : n;
    for (size_t k = 0; k < lanes; k++) {
        const size_t i = lanes - 1 - k;
        if (i < n && !small_m((u32)i)) arrive((u32)i, rows_msg_term(km, g_tb.te0, p, (u32)i));
        for (u64 t = i; t < rows_small_total(p); t += lanes) {
            G128 z;
            u64 e_run = 0;
            const u32 m = dec ? rows_small_block<1>(km, g_tb.te0, p, t, &z, &e_run) : rows_small_block<0>(km, g_tb.te0, p, t, &z, &e_run);
            ++made_of[m];
            arrive(m, rows_small_due(km, z, e_run));                       // (the kernel XORs the blocks of a segment first and pays once: the multiply is linear)
        }
        if (i >= slots) continue;
        const RowsRec r = rec[i];
        if (!(r.flags & ROWS_REC_VALID)) continue;
        rec[i].flags = 0;
        arrive(r.msg, rows_weigh(km, r));
    }
    CHECK(finals == n - n_small, "rows: %u of %u messages closed", finals, n - n_small);
    for (u32 m = 0; m < n; m++) {
        CHECK(!acc[2 * m] && !acc[2 * m + 1] && !cnt[m], "rows: message %u not zero at rest", m);
        if (small_m(m)) CHECK(made_of[m] == 0, "rows: message %u is the packet kernels' and fell into %u pieces here", m, made_of[m]);
        else CHECK(made_of[m] + 1u == due(m), "rows: message %u fell into %u pieces", m, made_of[m]);
    }
}
// var: 0 fixed-size records, 1 offset arrays, 2 messages wherever they live (arrays of addresses and lengths: every message with a gap of its own in front, in and
// out at different spacings)
static void test_rows(int key_len, u64 seed, u32 waves, u32 force_d, int var, const std::vector<u32> &lens, const std::vector<u32> &aads, u32 misalign = 0) {
    auto key = rnd(key_len, seed);
    Emu E(key.data(), key_len, 0);
    const u32 n = (u32)lens.size();
    const u64 gi = var == 2 ? 48 : 0, go = var == 2 ? 80 : 0;                // gaps between the messages of the scattered form (input, output)
    std::vector<u64> doff(n + 1, misalign), aoff(n + 1, 0);
    for (u32 i = 0; i < n; i++) { doff[i + 1] = doff[i] + lens[i]; aoff[i + 1] = aoff[i] + aads[i]; }
    ABuf in(doff[n] + gi * n), out(doff[n] + go * n), back(doff[n] + gi * n);
    auto aad = rnd(aoff[n], seed + 1), ivs = rnd(12 * n, seed + 2);
    orc_fill_splitmix64(in.p, doff[n] + gi * n, seed + 3, 0);
    std::vector<uint8_t> tags(16 * n + 16), tags2(16 * n + 16);
    std::vector<u64> in_ptr(n), out_ptr(n), aad_ptr(n);
    std::vector<u32> alens(aads);
    for (int dec = 0; dec < 2; dec++) {
        RowsParams p; memset(&p, 0, sizeof p);
        p.ivs = ivs.data(); p.tags = dec ? tags2.data() : tags.data();
        p.n_pkts = n;
        if (var == 2) {                                                  // encrypt in -> out, decrypt out -> back
            for (u32 k = 0; k < n; k++) {
                in_ptr[k] = (u64)(uintptr_t)((dec ? out.p : in.p) + doff[k] + (dec ? go : gi) * k);
                out_ptr[k] = (u64)(uintptr_t)((dec ? back.p : out.p) + doff[k] + (dec ? gi : go) * k);
                aad_ptr[k] = (u64)(uintptr_t)(aad.data() + aoff[k]);
            }
            p.in_ptr = in_ptr.data(); p.out_ptr = out_ptr.data(); p.len_arr = lens.data();
            if (aoff[n]) { p.aad_ptr = aad_ptr.data(); p.alen_arr = alens.data(); }
        } else {
            p.aad = aad.data(); p.in = dec ? out.p : in.p; p.out = out.p;
            if (var) { p.data_off = doff.data(); if (aoff[n]) p.aad_off = aoff.data(); }        // (no AAD anywhere: no AAD array -- empty messages then have no unit at all)
            else { p.pkt_len = lens[0]; p.aad_len = aads[0]; }
        }
        if (E.km.nr == 10) emu_rows_nr<10>(&E.km, dec, p, waves, force_d); else if (E.km.nr == 12) emu_rows_nr<12>(&E.km, dec, p, waves, force_d); else emu_rows_nr<14>(&E.km, dec, p, waves, force_d);
        if (!dec) {
            for (u32 k = 0; k < n; k++) {
                std::vector<uint8_t> ref(lens[k] + 16); uint8_t rtag[16];
                orc_gcm_crypt(0, key.data(), key_len, ivs.data() + 12 * k, aad.data() + aoff[k], aads[k], in.p + doff[k] + gi * k, lens[k], ref.data(), rtag);
                CHECK(memcmp(ref.data(), out.p + doff[k] + go * k, lens[k]) == 0, "rows ct %u len %u waves %u D %u var %d", k, lens[k], waves, force_d, var);
                CHECK(memcmp(rtag, tags.data() + 16 * k, 16) == 0, "rows tag %u len %u aad %u waves %u D %u var %d", k, lens[k], aads[k], waves, force_d, var);
            }
        } else {
            if (var == 2) { for (u32 k = 0; k < n; k++) CHECK(memcmp(back.p + doff[k] + gi * k, in.p + doff[k] + gi * k, lens[k]) == 0, "rows dec data %u (scattered) waves %u D %u", k, waves, force_d); }
            else CHECK(memcmp(out.p + misalign, in.p + misalign, doff[n] - misalign) == 0, "rows dec data waves %u D %u var %d", waves, force_d, var);
            CHECK(memcmp(tags2.data(), tags.data(), 16 * n) == 0, "rows dec tags waves %u D %u var %d", waves, force_d, var);
        }
    }
}

int main(int argc, char **argv) {
    int level = argc > 1 ? atoi(argv[1]) : 1;
    init_tables();
    test_units();
    test_bitslice();
    const u64 W = AESGCM_WG;     // blocks per chunk at the production Tw = 16
    const std::vector<std::pair<u64, u64>> small = {{0, 0}, {0, 1}, {0, 15}, {0, 16}, {0, 17}, {1, 0}, {20, 48}, {28, 48}, {68, 0}, {16, 63 * 16}, {17, 64 * 16}, {0, 65 * 16 + 5},
                                                    {16, (W - 1) * 16}, {17, W * 16}, {0, (W + 1) * 16 + 5}, {4095, 4097}};
    test_key(16, 0, 1, small);          // production chunking rule
    test_key(24, 1, 2, small);          // Tw = 1: every row its own chunk -> many items, in-kernel fold tables where 64*Tw*16^l is not in ptab
    test_key(32, 3, 3, small);          // Tw = 3: ragged first chunk
    // several rows per chunk (Horner with K = H^64), ragged tails, front padding, > GMAX chunks
    test_key(16, 0, 4, {{0, 16 * W * 3}, {5, 16 * W * 2 + 7}, {33, 16 * (W * 3 - 36) + 1}});
    test_key(32, 2, 5, {{0, 16 * 2048 * 2}, {16, 16 * 2048 * 2 - 16}, {40, 16 * 2048 * 3 + 13}, {1000 * 16, 16 * 6144}});
    test_key(24, 5, 6, {{7, 16 * 8192 + 9}});
    test_key(16, 1, 10, {{0, 16 * 64 * 1500 + 3}, {5, 16 * 64 * 3000}, {16, 16 * 64 * 5000 + 9}});   // 1500 / 3000 / 5000 items: k_fold with 2, 4, 8 items per wave (fold_group)
    if (level >= 3) test_key(32, 1, 11, {{0, 16ull * 64 * 65535}});
    test_key(32, 1, 9, {{3, 16 * 64 * 700 + 11}});      // 700+ chunks > GMAX -> two-stage fold with > 1 stage-1 lanes... one stage-1 workgroup
    test_shards(32, 2, 37, 203 * 16 + 5, 8, 77);
    test_shards(16, 0, 0, 16 * (W * 5 + 100) + 3, 3, 78);
    test_shards(24, 1, 20, 16 * 7, 8, 79);      // fewer blocks than ranks: some shards are empty
    test_stream(32, 2, 40, 1000, 16, 90);
    test_stream(16, 0, 0, 16 * (W * 4 + 52) + 9, 16 * (W + 188), 91);
    test_stream(24, 3, 16 * 40 + 3, 33, 16 * 8, 92);
    test_keystream_and_ghash(55);
    // k_body: T = 1, 2, 3 rows per chunk; lengths chosen so head, tail and ragged end are all non-trivial
    test_body(16, 1, 0, 16 * 3000 + 5, 101);
    test_body(24, 2, 20, 16 * (254 + 2048 * 3 + 777) + 11, 102);
    test_body(32, 3, 37, 16 * 9000, 103);
    test_body(32, 1, 16, 16 * 254 + 16 * 1024 * 2, 104);          // head = 254 blocks exactly, empty tail
    test_body(16, 2, 0, 16 * 256 * 2 * 6, 106);                     // no AAD, no head, no tail: the whole message is one body (direct tag path)
    test_body(32, 1, 0, 16 * 256 * 300, 107);                       // ... with enough items for a k_fold level: its ten workgroups close the tag themselves (FoldClose)
    test_body(24, 3, 0, 16 * 256 * 3 * 40, 116);                    // ... three rows per chunk: the weight of the first workgroup is H^(24576), digit 24 of radix 1024
    test_body(16, 1, 0, 16 * 256 * 7, 108, true);                   // cyclic rows: 28 rows, the other strands leave zero items; the whole message is the body
    test_body(32, 2, 21, 16 * (100 + 256 * 40 + 70) + 3, 109, true); // ... with AAD (a front row), a ragged partial last row, and as shards from odd first blocks (head blocks in the front rows)
    test_body(24, 1, 16 * 70 + 3, 16 * 64 * 9 + 1023, 112, true);   // two front rows of AAD, the longest last row there is (63 blocks + 15 bytes = 64 blocks)
    test_body(16, 1, 16 * 64, 16 * 64 * 5 + 16, 113, true);         // AAD fills its row exactly (no front padding), one whole block behind the body
    test_body(32, 1, 1, 16 * 64 * 3 + 1, 114, true);                 // one byte of AAD, one byte behind the body
    test_batch_pieces();
    test_packets(16, 61); test_packets(24, 62); test_packets(32, 63);
    // many messages by rows: offset arrays with every kind of length (empty, shorter than a block, tails of 63 blocks + 15 bytes = two tail rows, whole super-rows,
    // 1 .. 3 rows behind them), AAD of none / a ragged block / more than a row; fixed-size records; one block per wave for few and for many waves (cuts in the
    // middle of strands), dealt blocks of 1 / 2 / 3 / 7 units
    const std::vector<u32> rl = {0, 1, 15, 16, 1023, 1024, 1040, 4096, 4097, 5 * 1024 + 1008 + 15, 3 * 4096 + 2 * 1024 + 17, 9 * 4096, 7 * 4096 + 3 * 1024 + 1023};
    const std::vector<u32> ra = {0, 20, 0, 16, 1, 0, 33, 0, 13, 1024 + 7, 0, 8, 2048};
    test_rows(16, 201, 5, 0, true, rl, ra);
    test_rows(24, 208, 64, 0, true, rl, ra);
    test_rows(32, 209, 4096, 1, true, rl, ra);
    test_rows(32, 202, 16, 3, true, {10 * 4096 + 5, 0, 4096 * 7, 3 * 1024, 1024 * 6 + 100}, {0, 0, 12, 5, 0});
    test_rows(24, 203, 7, 0, true, {65536, 65536 + 1024 + 3, 20000, 131072 + 17}, {20, 0, 28, 0});
    test_rows(32, 204, 3, 0, false, {65536, 65536, 65536}, {0, 0, 0});            // 65 units per message, one message per wave
    test_rows(16, 210, 13, 0, false, {65536, 65536, 65536}, {0, 0, 0});           // 15 units per wave: cuts inside the strands
    test_rows(16, 205, 16, 2, false, {4096 * 5 + 2048 + 9, 4096 * 5 + 2048 + 9, 4096 * 5 + 2048 + 9, 4096 * 5 + 2048 + 9, 4096 * 5 + 2048 + 9}, {13, 13, 13, 13, 13});
    test_rows(24, 211, 16, 7, false, {4096 * 5 + 2048 + 9, 4096 * 5 + 2048 + 9, 4096 * 5 + 2048 + 9}, {0, 0, 0});
    test_rows(24, 206, 16, 0, false, {700, 700}, {0, 0});                       // records shorter than a row: no strand at all, tails only
    test_rows(32, 207, 4, 2, true, {4096 * 3 + 5, 9000, 1024 * 5}, {7, 0, 16}, 5);   // packed back to back from an odd byte address
    test_rows(32, 212, 3, 0, true, {0, 0, 0, 2048, 0, 1024, 1024 + 1009, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0, 0});   // messages without a unit (empty, no AAD) first, in a row and last; a tail of 64 blocks
    test_rows(16, 213, 5, 1, true, {0, 0, 3000, 0}, {0, 0, 0, 0});
    test_rows(24, 214, 4, 0, false, {0, 0, 0}, {0, 0, 0});                       // fixed-size records of no bytes: no unit in the whole call, the closing makes the tags
    test_rows(24, 215, 4, 0, false, {0, 0, 0}, {5, 5, 5});                       // ... with AAD: an AAD unit each
    test_rows(32, 216, 4, 0, false, {1024, 1024, 1024, 1024, 1024}, {0, 0, 0, 0, 0});   // one row and nothing else per message
    test_rows(16, 217, 2, 0, false, {2033, 2033, 2033}, {0, 0, 0});              // a tail of 64 blocks (the last one byte long)
    test_rows(32, 218, 6, 0, true, {16400, 16400, 100, 16384 + 1023, 0, 5000, 16, 40 * 16 + 3, 2048 + 700}, {13, 13, 1024, 1025, 20, 0, 1023, 600, 16});   // short AADs and tails share units and straddle them; an AAD of exactly 64 blocks; one of 65 (a unit of its own)
    test_rows(24, 219, 5, 2, false, {16400, 16400, 16400, 16400, 16400, 16400, 16400}, {13, 13, 13, 13, 13, 13, 13});      // TLS-shaped records: 2 + 1 blocks of smalls each
    test_rows(16, 220, 9, 0, false, {1024 + 40 * 16, 1024 + 40 * 16, 1024 + 40 * 16, 1024 + 40 * 16, 1024 + 40 * 16}, {30 * 16 + 1, 30 * 16 + 1, 30 * 16 + 1, 30 * 16 + 1, 30 * 16 + 1});   // 31 + 40 blocks per message: every unit boundary inside a segment
    test_rows(32, 221, 3, 1, false, {700, 700, 700}, {2000, 2000, 2000});          // a long AAD (125 blocks) and a tail, no row
    test_rows(32, 222, 6, 0, 2, rl, ra);                                         // messages wherever they live: the lengths of the first case, every message in buffers of its own
    test_rows(16, 223, 4, 2, 2, {16400, 0, 100, 16384 + 1023, 5000, 16, 2048 + 700, 65536 + 1}, {13, 0, 1024, 1025, 0, 20, 16, 0}, 7);
    test_rows(24, 224, 5, 0, 2, {4096, 0, 1024, 3000}, {0, 0, 0, 0});              // ... without AAD arrays
    // ROUTED calls (round 6): the messages below the mark (data + AAD) are the packet kernels', the others go by rows -- the same lengths at three marks: a few
    // small ones, most of them, all of them (no unit in the whole call)
    for (u32 mark : {1024u, 4160u, 200000u}) {
        g_route_min = mark;
        test_rows(16, 230, 5, 0, true, rl, ra);
        test_rows(32, 231, 4, 2, 2, {16400, 0, 100, 16384 + 1023, 5000, 16, 2048 + 700, 65536 + 1}, {13, 0, 1024, 1025, 0, 20, 16, 0}, 7);
        test_rows(24, 232, 3, 0, true, {0, 0, 0, 2048, 0, 1024, 1024 + 1009, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0, 0});
        g_route_min = 0;
    }
    for (u64 nn = 1, prev = 4; nn < (1u << 22); nn += 1 + nn / 7) {               // the packet kernel shape a routed call takes for nn short messages: never more lanes per packet as they get more
        const u32 lg = route_pick_lg(256, nn);
        CHECK((lg == 0 || lg == 2 || lg == 3 || lg == 4) && lg <= prev, "route_pick_lg(%llu) = %u after %llu", (unsigned long long)nn, lg, (unsigned long long)prev);
        const u32 d = pktg_deal(256, nn, lg ? lg : 2);
        CHECK(d >= (64u >> (lg ? lg : 2)) && d <= 64u && d % (64u >> (lg ? lg : 2)) == 0, "pktg_deal(%llu, %u) = %u", (unsigned long long)nn, lg, d);
        prev = lg;
    }
    if (level > 1) {
        test_key(32, 1, 7, {{123, 16 * 64 * 1100 + 11}});     // 1100 chunks: two stage-1 workgroups
        test_key(16, 0, 8, {{0, 16 * W * 600}});              // production rule, > GMAX chunks of Tw = 16
        test_key(24, 1, 10, {{5, 16 * 64 * 4200 + 3}});        // 4200 one-row chunks: three k_fold levels
        test_body(32, 1, 9, 16 * (200 + 256 * 1100) + 7, 105);  // 4400 body items: interleaved first level + two more
        test_body(24, 1, 0, 16 * 64 * 4096, 110, true);        // cyclic rows: every strand exactly one row (4 MiB)
        test_body(32, 1, 5, 16 * (31 + 64 * 5333 + 50) + 9, 111, true);     // 5333 body rows + a front row: strands of two and of one, residues rotated
        test_body(16, 1, 16 * 64 * 3 + 7, 16 * 64 * 4200, 115, true);       // four front rows, the first strands start with them and go on with body rows
    }
    printf(g_fail ? "EMUL FAILED (%d)\n" : "EMUL OK\n", g_fail);
    return g_fail ? 1 : 0;
}
