/* Plain-C caller of libaesgcm_hip.so: MACsec-shaped traffic under one key -- the reference's deployment (README.md:236-257: frames with a 20 / 28-byte SecTAG
 * header as AAD, frame after frame under one SAK, tb/gcm_test.py:76-85) -- as ONE device call per batch: aesgcm_packets_crypt_dev with offset arrays, the
 * frames packed back to back at whatever byte the previous one ended on.  Frame 0 is the reference's README vector (README.md:251), the rest are synthetic;
 * every frame is decrypted again in place and authenticated, and one forged tag must be reported.
 *
 *   make -C examples frames && examples/frames [n_frames]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "aesgcm.h"

static int unhex(const char *h, unsigned char *out) {
    int n = 0;
    for (; h[0] && h[1]; h += 2, n++) { unsigned v; sscanf(h, "%2x", &v); out[n] = (unsigned char)v; }
    return n;
}
#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d (%s; %s)\n", #call, rc_, aesgcm_strerror(rc_), aesgcm_last_error()); return 1; } } while (0)

int main(int argc, char **argv) {
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 200000;
    unsigned char key[16], iv0[12], aad0[64], pt0[64], want_ct[64], want_tag[16];
    unhex("AD7A2BD03EAC835A6F620FDCB506B345", key);
    unhex("12153524C0895E81B2C28465", iv0);
    const size_t al0 = (size_t)unhex("D609B1F056637A0D46DF998D88E52E00B2C2846512153524C0895E81", aad0);
    const size_t n0 = (size_t)unhex("08000F101112131415161718191A1B1C1D1E1F202122232425262728292A2B2C2D2E2F303132333435363738393A0002", pt0);
    unhex("701AFA1CC039C0D765128A665DAB69243899BF7318CCDC81C9931DA17FBE8EDD7D17CB8B4C26FC81E3284F2B7FBA713D", want_ct);
    unhex("4F8D55E7D3F06FD5A13C0C29B9D5B880", want_tag);

    /* frame lengths 46 .. 1514, header 20 or 28 bytes; offsets as they come (nothing is padded to 16 bytes) */
    uint64_t *doff = malloc((n + 1) * sizeof *doff), *aoff = malloc((n + 1) * sizeof *aoff);
    uint32_t s = 12345;
    doff[0] = aoff[0] = 0;
    for (size_t p = 0; p < n; p++) {
        s = s * 1664525u + 1013904223u;
        const size_t len = p == 0 ? n0 : 46 + (s >> 8) % 1469, al = p == 0 ? al0 : ((s >> 3) & 1 ? 28 : 20);
        doff[p + 1] = doff[p] + len; aoff[p + 1] = aoff[p] + al;
    }
    const size_t nd = (size_t)doff[n], na = (size_t)aoff[n];
    unsigned char *pt = malloc(nd + 16), *aad = malloc(na + 16), *ivs = malloc(12 * n), *ct = malloc(nd + 16), *tags = malloc(16 * n);
    int *auth = malloc(n * sizeof *auth);
    for (size_t i = 0; i < nd; i++) pt[i] = (unsigned char)(i * 131 + (i >> 9));
    for (size_t i = 0; i < na; i++) aad[i] = (unsigned char)(i * 17 + 3);
    for (size_t p = 0; p < n; p++) { memcpy(ivs + 12 * p, iv0, 12); ivs[12 * p + 8] ^= (unsigned char)(p >> 24); ivs[12 * p + 9] ^= (unsigned char)(p >> 16); ivs[12 * p + 10] ^= (unsigned char)(p >> 8); ivs[12 * p + 11] ^= (unsigned char)p; }
    memcpy(pt, pt0, n0); memcpy(aad, aad0, al0);

    aesgcm_ctx *ctx = NULL;
    void *d_buf, *d_aad, *d_ivs, *d_tags, *d_doff, *d_aoff, *d_auth, *d_exp;
    CHECK(aesgcm_ctx_create(&ctx, 0, key, sizeof key));
    CHECK(aesgcm_dev_alloc(0, &d_buf, nd + 16)); CHECK(aesgcm_dev_alloc(0, &d_aad, na + 16)); CHECK(aesgcm_dev_alloc(0, &d_ivs, 12 * n));
    CHECK(aesgcm_dev_alloc(0, &d_tags, 16 * n)); CHECK(aesgcm_dev_alloc(0, &d_exp, 16 * n)); CHECK(aesgcm_dev_alloc(0, &d_auth, sizeof(int) * n));
    CHECK(aesgcm_dev_alloc(0, &d_doff, 8 * (n + 1))); CHECK(aesgcm_dev_alloc(0, &d_aoff, 8 * (n + 1)));
    CHECK(aesgcm_dev_upload(0, d_buf, pt, nd)); CHECK(aesgcm_dev_upload(0, d_aad, aad, na)); CHECK(aesgcm_dev_upload(0, d_ivs, ivs, 12 * n));
    CHECK(aesgcm_dev_upload(0, d_doff, doff, 8 * (n + 1))); CHECK(aesgcm_dev_upload(0, d_aoff, aoff, 8 * (n + 1)));

    /* encrypt in place: one call for all frames */
    CHECK(aesgcm_packets_crypt_dev(ctx, 0, n, d_ivs, d_aad, 0, (const uint64_t *)d_aoff, d_buf, 0, (const uint64_t *)d_doff, d_buf, d_tags, NULL, NULL, NULL));
    CHECK(aesgcm_dev_sync(0));
    CHECK(aesgcm_dev_download(0, ct, d_buf, nd)); CHECK(aesgcm_dev_download(0, tags, d_tags, 16 * n));
    if (memcmp(ct, want_ct, n0) || memcmp(tags, want_tag, 16)) { fprintf(stderr, "frame 0 is not the reference's README vector\n"); return 1; }
    /* a frame from the middle against the one-shot entry point */
    {
        const size_t p = n / 2, len = (size_t)(doff[p + 1] - doff[p]), al = (size_t)(aoff[p + 1] - aoff[p]);
        unsigned char one_ct[1600], one_tag[16];
        CHECK(aesgcm_encrypt(ctx, ivs + 12 * p, aad + aoff[p], al, pt + doff[p], len, one_ct, one_tag));
        if (memcmp(one_ct, ct + doff[p], len) || memcmp(one_tag, tags + 16 * p, 16)) { fprintf(stderr, "frame %zu differs from aesgcm_encrypt\n", p); return 1; }
    }
    /* decrypt in place and authenticate; the tag of the last frame is forged */
    tags[16 * (n - 1) + 5] ^= 0x40;
    CHECK(aesgcm_dev_upload(0, d_exp, tags, 16 * n));
    CHECK(aesgcm_packets_crypt_dev(ctx, 1, n, d_ivs, d_aad, 0, (const uint64_t *)d_aoff, d_buf, 0, (const uint64_t *)d_doff, d_buf, d_tags, d_exp, (int *)d_auth, NULL));
    CHECK(aesgcm_dev_sync(0));
    CHECK(aesgcm_dev_download(0, ct, d_buf, nd)); CHECK(aesgcm_dev_download(0, auth, d_auth, sizeof(int) * n));
    if (memcmp(ct, pt, nd)) { fprintf(stderr, "decrypt does not give the plaintext back\n"); return 1; }
    size_t bad = 0;
    for (size_t p = 0; p < n; p++) bad += auth[p] ? 0 : 1;
    if (bad != 1 || auth[n - 1]) { fprintf(stderr, "%zu frames failed authentication (expected: the forged last one)\n", bad); return 1; }
    int lanes = 0;
    CHECK(aesgcm_packets_shape(ctx, n, 0, 1, &lanes));
    printf("FRAMES OK (%zu frames, %zu bytes, %d lane(s) per frame)\n", n, nd, lanes);
    aesgcm_dev_free(0, d_buf); aesgcm_dev_free(0, d_aad); aesgcm_dev_free(0, d_ivs); aesgcm_dev_free(0, d_tags); aesgcm_dev_free(0, d_exp); aesgcm_dev_free(0, d_auth);
    aesgcm_dev_free(0, d_doff); aesgcm_dev_free(0, d_aoff);
    CHECK(aesgcm_ctx_destroy(ctx));
    return 0;
}
