// CPU harness: runs the SAME per-lane code the HIP kernels run (csrc/aesgcm_dev.h is __host__
// __device__) over an emulated launch -- LDS image, workgroup/lane geometry, front padding, Horner
// with K, per-lane tail powers, workgroup fold, k_combine fold -- and compares ciphertext and tag with
// the oracle (oracle/aesgcm_oracle.c, linked in).  It exists because the build container has no GPU:
// it pins the arithmetic and the index algebra before any GPU minute is spent.  Test infrastructure only.
#include "../../aes-gcm-128-192-256-bits_amd/csrc/aesgcm_dev.h"
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
static void init_tables() { for (u32 x = 0; x < 256; x++) { u32 s = sbox_calc(x); g_tb.sbox[x] = (uint8_t)s; g_tb.te0[x] = te0_calc(s); } }

// emulated k_setup (same barrier structure: compute every lane's product, then commit)
static void emu_setup(KeyMaterial *km, const uint8_t *key, int key_len, int pre_nr, u32 G) {
    static uint4 tab[513];
    setup_lane0(km, g_tb.sbox, key, key_len, pre_nr, G, tab);
    for (int d = 0; d < 4; d++) {
        for (int j = 0; j < 9; j++) {
            static uint4 prod[512]; static bool act[512];
            for (int tid = 0; tid < 512; tid++) act[tid] = setup_level(tab, j, tid, &prod[tid]);
            for (int tid = 0; tid < 512; tid++) if (act[tid]) tab[(1 << j) + tid] = prod[tid];
        }
        for (int k = 0; k < 513; k++) km->pw[d][k] = tab[k];
        if (d == 1) for (int tid = 0; tid < 512; tid++) setup_beta_lane(km, tab, tid);
        if (d < 3) { uint4 next = tab[512]; tab[0] = gf_one_mo(); tab[1] = next; }
    }
}

static void xor_g(G128 &a, const G128 &b) { for (int k = 0; k < 4; k++) a.w[k] ^= b.w[k]; }

template <int NR, int MODE>
static void emu_main_nr(const KeyMaterial *km, const MainParams &p, u32 Gp) {
    static unsigned char smem[AESGCM_LDS_BYTES] __attribute__((aligned(16)));
    constexpr bool GH = (MODE == MODE_ENC || MODE == MODE_DEC);
    for (u32 tid = 0; tid < AESGCM_WG; tid++) main_fill_lds(smem, km, &g_tb, tid, GH);   // image is workgroup independent
    for (u32 wg = 0; wg < Gp; wg++) {
        G128 fold = {{0, 0, 0, 0}};
        for (u32 tid = 0; tid < AESGCM_WG; tid++) {
            uint4 acc = main_lane<NR, MODE>(km, p, smem, wg, tid);
            if (GH) xor_g(fold, main_lane_tail(km, acc, tid));
        }
        if (GH) p.parts[wg] = be_to_mo(fold);
    }
}
static void emu_main(int mode, const KeyMaterial *km, const MainParams &p, u32 Gp) {
#define D(NR) switch (mode) { case MODE_ENC: emu_main_nr<NR, MODE_ENC>(km, p, Gp); break; case MODE_DEC: emu_main_nr<NR, MODE_DEC>(km, p, Gp); break; \
                              case MODE_KS: emu_main_nr<NR, MODE_KS>(km, p, Gp); break; default: emu_main_nr<NR, MODE_ECB>(km, p, Gp); }
    if (km->nr == 10) { D(10) } else if (km->nr == 12) { D(12) } else { D(14) }
#undef D
}
static G128 emu_pow_h(const KeyMaterial *km, u64 e) {
    G128 v = gf_pow_h_digit(km, e, 0);
    for (u32 d = 1; d < 4; d++) v = gf_mul(v, gf_pow_h_digit(km, e, d));
    return v;
}
static void emu_combine(const KeyMaterial *km, const CombineParams &p) {
    G128 acc = {{0, 0, 0, 0}};
    for (u32 tid = 0; tid < COMBINE_THREADS; tid++) xor_g(acc, combine_lane(km, g_tb.sbox, p, tid));
    const bool tag = p.want_tag != 0;
    if (!tag && p.e) acc = gf_mul(acc, emu_pow_h(km, p.e));
    if (p.has_carry) {
        G128 c = mo_to_be(*p.carry);
        if (p.e_carry) c = gf_mul(c, emu_pow_h(km, p.e_carry));
        if (tag) c = gf_mul(c, mo_to_be(km->pw[0][2]));
        xor_g(acc, c);
    }
    *p.out = be_to_mo(acc);
}

struct Emu {
    KeyMaterial km; u32 G; std::vector<uint4> parts;
    Emu(const uint8_t *key, int key_len, u32 G_) : G(G_), parts(AESGCM_GMAX) { emu_setup(&km, key, key_len, 0, G); }
    void crypt(int dec, const uint8_t iv[12], const uint8_t *aad, u64 aad_len, const uint8_t *in, u64 len, uint8_t *out, uint8_t tag[16]) {
        MainParams p; memset(&p, 0, sizeof p);
        u32 Gp = plan_main(p, dec ? MODE_DEC : MODE_ENC, G, iv, aad, aad_len, in, len, out, 0, parts.data());
        if (Gp) emu_main(dec ? MODE_DEC : MODE_ENC, &km, p, Gp);
        uint4 t;
        emu_combine(&km, plan_combine_tag(parts.data(), Gp, false, iv, aad_len, len, &t));
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

static void test_key(int key_len, u32 G, u64 seed, const std::vector<std::pair<u64, u64>> &sizes) {
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
        for (int k = 0; k <= 512; k++) {
            if (k == 0 || k == 1 || k == 2 || k == 3 || k == 255 || k == 256 || k == 257 || k == 511 || k == 512)
                CHECK(memcmp(&E.km.pw[0][k], acc, 16) == 0, "pw[0][%d]", k);
            orc_gfmul(h, acc, acc);
        }
        CHECK(memcmp(&E.km.pw[1][1], &E.km.pw[0][512], 16) == 0, "beta");
        uint4 b2 = gf_mul_mo(E.km.pw[1][1], E.km.pw[1][1]);
        CHECK(memcmp(&E.km.pw[1][2], &b2, 16) == 0, "beta^2");
        uint4 g3 = gf_mul_mo(gf_mul_mo(E.km.pw[2][1], E.km.pw[2][1]), E.km.pw[2][1]);
        CHECK(memcmp(&E.km.pw[2][3], &g3, 16) == 0, "gamma^3");
        CHECK(memcmp(&E.km.pw[3][1], &E.km.pw[2][512], 16) == 0, "delta");
    }
    // ECB through the LDS round code
    {
        const size_t nb = 700;
        ABuf in(16 * nb), out(16 * nb);
        orc_fill_splitmix64(in.p, 16 * nb, seed + 5, 0);
        MainParams p; memset(&p, 0, sizeof p);
        u32 Gp = plan_main(p, MODE_ECB, G, nullptr, nullptr, 0, in.p, 16 * nb, out.p, 0, E.parts.data());
        emu_main(MODE_ECB, &E.km, p, Gp);
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
        MainParams p; memset(&p, 0, sizeof p);
        u32 Gp = plan_main(p, MODE_ENC, G, iv.data(), r == 0 ? aad.data() : nullptr, r == 0 ? al : 0, pt.p + 16 * first, len, ct.p + 16 * first, first, E.parts.data());
        if (Gp) emu_main(MODE_ENC, &E.km, p, Gp);
        emu_combine(&E.km, plan_combine_poly(E.parts.data(), Gp, total_blocks - end, &gathered[r]));
        first = end;
    }
    uint4 t;
    emu_combine(&E.km, plan_combine_tag(gathered.data(), R, true, iv.data(), al, n, &t));
    CHECK(memcmp(ct.p, ref.p, n) == 0, "shard ct R %d", R);
    CHECK(memcmp(&t, rtag, 16) == 0, "shard tag key %d G %u aad %llu len %llu R %d", key_len, G, (unsigned long long)al, (unsigned long long)n, R);
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
        MainParams p; memset(&p, 0, sizeof p);
        u32 Gp = plan_main(p, MODE_ENC, G, iv.data(), aad.data() + off, m, nullptr, 0, nullptr, 0, E.parts.data());
        emu_main(MODE_ENC, &E.km, p, Gp);
        emu_combine(&E.km, plan_combine_carry(E.parts.data(), Gp, &Y, (m + 15) / 16));
    }
    for (u64 off = 0; off < n; off += chunk) {
        u64 m = n - off < chunk ? n - off : chunk;
        MainParams p; memset(&p, 0, sizeof p);
        u32 Gp = plan_main(p, MODE_ENC, G, iv.data(), nullptr, 0, pt.p + off, m, ct.p + off, off / 16, E.parts.data());
        emu_main(MODE_ENC, &E.km, p, Gp);
        emu_combine(&E.km, plan_combine_carry(E.parts.data(), Gp, &Y, (m + 15) / 16));
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
    MainParams p; memset(&p, 0, sizeof p);
    u32 Gp = plan_main(p, MODE_KS, E.G, iv.data(), nullptr, 0, out.p, 16 * nb, out.p, first, E.parts.data());
    emu_main(MODE_KS, &E.km, p, Gp);
    for (u64 i = 0; i < nb; i++) {
        uint8_t cb[16], o[16]; memcpy(cb, iv.data(), 12);
        u32 c = (u32)(2 + first + i); cb[12] = c >> 24; cb[13] = c >> 16; cb[14] = c >> 8; cb[15] = c;
        orc_aes_encrypt_block(rk, nr, cb, o);
        CHECK(memcmp(o, out.p + 16 * i, 16) == 0, "keystream block %llu", (unsigned long long)i);
    }
    // GHASH chaining value Y = P*H
    for (u64 n : {1ull, 16ull, 17ull, 8191ull, 20000ull}) {
        auto d = rnd(n, seed + n);
        memset(&p, 0, sizeof p);
        Gp = plan_main(p, MODE_ENC, E.G, iv.data(), d.data(), n, nullptr, 0, nullptr, 0, E.parts.data());
        emu_main(MODE_ENC, &E.km, p, Gp);
        uint4 y; emu_combine(&E.km, plan_combine_poly(E.parts.data(), Gp, 1, &y));
        uint8_t yo[16] = {0}; orc_ghash_update((const uint8_t *)&E.km.h, yo, d.data(), n);
        CHECK(memcmp(&y, yo, 16) == 0, "ghash len %llu", (unsigned long long)n);
    }
}

int main(int argc, char **argv) {
    int level = argc > 1 ? atoi(argv[1]) : 1;
    init_tables();
    test_units();
    const std::vector<std::pair<u64, u64>> small = {{0, 0}, {0, 1}, {0, 15}, {0, 16}, {0, 17}, {1, 0}, {20, 48}, {28, 48}, {68, 0}, {16, 511 * 16}, {17, 512 * 16}, {0, 513 * 16 + 5}, {4095, 4097}};
    test_key(16, 1, 1, small);
    test_key(24, 2, 2, small);
    test_key(32, 3, 3, small);
    // T > 1 (Horner with K = H^(G*512)) incl. ragged tails and front padding
    test_key(16, 1, 4, {{0, 16 * 512 * 3}, {5, 16 * 512 * 2 + 7}, {33, 16 * 1500 + 1}});
    test_key(32, 2, 5, {{0, 16 * 1024 * 2}, {16, 16 * 1024 * 2 - 16}, {40, 16 * 1024 * 3 + 13}, {1000 * 16, 16 * 3000}});
    test_key(24, 4, 6, {{7, 16 * 2048 * 2 + 9}});
    test_shards(32, 2, 37, 203 * 16 + 5, 8, 77);
    test_shards(16, 1, 0, 16 * 5000 + 3, 3, 78);
    test_shards(24, 2, 20, 16 * 7, 8, 79);      // fewer blocks than ranks: some shards are empty
    test_stream(32, 2, 40, 1000, 16, 90);
    test_stream(16, 1, 0, 16 * 2100 + 9, 16 * 700, 91);
    test_stream(24, 3, 16 * 40 + 3, 33, 16 * 8, 92);
    test_keystream_and_ghash(55);
    if (level > 1) {
        test_key(32, 8, 7, {{123, 16 * 4096 * 5 + 11}});
        test_key(16, 512, 8, {{0, 16 * 3000}});              // production G with T = 1
    }
    printf(g_fail ? "EMUL FAILED (%d)\n" : "EMUL OK\n", g_fail);
    return g_fail ? 1 : 0;
}
