/* Plain-C caller of libaesgcm_hip.so: the README / IEEE 802.1AE vector of the reference (README.md:251)
 * through the one-shot host entry points, then the same frame through the beat-by-beat streaming calls in
 * the order the reference harness uses them (tb/gcm_test.py:76-85).
 *
 *   gcc -std=c99 -Iinclude examples/kat.c -o examples/kat -L aes-gcm-128-192-256-bits_amd -laesgcm_hip \
 *       -Wl,-rpath,'$ORIGIN/../aes-gcm-128-192-256-bits_amd'
 */
#include <stdio.h>
#include <string.h>
#include "aesgcm.h"

static int unhex(const char *h, unsigned char *out) {
    int n = 0;
    for (; h[0] && h[1]; h += 2, n++) { unsigned v; sscanf(h, "%2x", &v); out[n] = (unsigned char)v; }
    return n;
}
#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d (%s; %s)\n", #call, rc_, aesgcm_strerror(rc_), aesgcm_last_error()); return 1; } } while (0)

int main(void) {
    unsigned char key[16], iv[12], aad[64], pt[64], want_ct[64], want_tag[16], ct[64], back[64], tag[16], tag2[16];
    unhex("AD7A2BD03EAC835A6F620FDCB506B345", key);
    unhex("12153524C0895E81B2C28465", iv);
    const int al = unhex("D609B1F056637A0D46DF998D88E52E00B2C2846512153524C0895E81", aad);
    const int n = unhex("08000F101112131415161718191A1B1C1D1E1F202122232425262728292A2B2C2D2E2F303132333435363738393A0002", pt);
    unhex("701AFA1CC039C0D765128A665DAB69243899BF7318CCDC81C9931DA17FBE8EDD7D17CB8B4C26FC81E3284F2B7FBA713D", want_ct);
    unhex("4F8D55E7D3F06FD5A13C0C29B9D5B880", want_tag);

    aesgcm_ctx *ctx = NULL;
    CHECK(aesgcm_ctx_create(&ctx, 0, key, sizeof key));
    CHECK(aesgcm_encrypt(ctx, iv, aad, (size_t)al, pt, (size_t)n, ct, tag));
    if (memcmp(ct, want_ct, (size_t)n) || memcmp(tag, want_tag, 16)) { fprintf(stderr, "one-shot mismatch\n"); return 1; }
    CHECK(aesgcm_decrypt(ctx, iv, aad, (size_t)al, ct, (size_t)n, back, tag, tag2));
    if (memcmp(back, pt, (size_t)n)) { fprintf(stderr, "decrypt mismatch\n"); return 1; }
    tag[3] ^= 1;
    if (aesgcm_decrypt(ctx, iv, aad, (size_t)al, ct, (size_t)n, back, tag, tag2) != AESGCM_EAUTH) { fprintf(stderr, "tamper not reported\n"); return 1; }

    /* beat by beat: AAD in 16-byte beats, then data in 16-byte beats, then the tag */
    CHECK(aesgcm_stream_begin(ctx, iv, 0));
    for (int o = 0; o < al; o += 16) CHECK(aesgcm_stream_aad(ctx, aad + o, (size_t)(al - o < 16 ? al - o : 16)));
    for (int o = 0; o < n; o += 16) CHECK(aesgcm_stream_update(ctx, pt + o, (size_t)(n - o < 16 ? n - o : 16), ct + o));
    CHECK(aesgcm_stream_final(ctx, tag2));
    if (memcmp(ct, want_ct, (size_t)n) || memcmp(tag2, want_tag, 16)) { fprintf(stderr, "streaming mismatch\n"); return 1; }
    CHECK(aesgcm_ctx_destroy(ctx));
    printf("KAT OK\n");
    return 0;
}
