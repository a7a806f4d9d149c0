/* Plain-C caller of libaesgcm_hip.so: MANY MESSAGES under one key -- the reference's deployment (frame after frame under one key, tb/gcm_test.py:76-85; H kept
 * while no key is loaded, src/gcm_gctr.vhd:142-144) at message size -- as ONE device call: aesgcm_packets_crypt_dev over fixed-size records, which from 8 KiB
 * per packet goes by rows (aesgcm_packets_shape says AESGCM_SHAPE_ROWS; csrc/aesgcm_rows.h).  A sample of the messages is encrypted once more through the
 * single-message entry point (aesgcm_encrypt_dev) and must give the same tag; all are decrypted in place and authenticated, one forged tag must be reported
 * and -- with the context option wipe_on_auth_fail -- its message come back as zeros.  The same messages also go through aesgcm_messages_crypt_dev (arrays of
 * device addresses and lengths: messages wherever they live) and must give the same tags.
 *
 *   make -C examples messages && examples/messages [n_messages] [bytes_per_message]
 */
#define _POSIX_C_SOURCE 200809L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "aesgcm.h"

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d (%s; %s)\n", #call, rc_, aesgcm_strerror(rc_), aesgcm_last_error()); return 1; } } while (0)
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv) {
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 1024, size = argc > 2 ? (size_t)atol(argv[2]) : (size_t)1 << 20;
    unsigned char key[32];
    for (int i = 0; i < 32; i++) key[i] = (unsigned char)(7 * i + 1);
    aesgcm_ctx *ctx, *one;
    CHECK(aesgcm_ctx_create(&ctx, 0, key, sizeof key));
    CHECK(aesgcm_ctx_create(&one, 0, key, sizeof key));
    CHECK(aesgcm_ctx_set_option(ctx, "wipe_on_auth_fail", 1));
    int shape = 0;
    CHECK(aesgcm_packets_shape(ctx, n, size, 0, &shape));
    void *d_pt, *d_ct, *d_ivs, *d_tags, *d_exp, *d_one;
    int *d_auth;
    CHECK(aesgcm_dev_alloc(0, &d_pt, n * size)); CHECK(aesgcm_dev_alloc(0, &d_ct, n * size)); CHECK(aesgcm_dev_alloc(0, &d_one, size));
    CHECK(aesgcm_dev_alloc(0, &d_ivs, 12 * n + 16)); CHECK(aesgcm_dev_alloc(0, &d_tags, 16 * n)); CHECK(aesgcm_dev_alloc(0, &d_exp, 16 * n)); CHECK(aesgcm_dev_alloc(0, (void **)&d_auth, 4 * n));
    CHECK(aesgcm_fill_splitmix64_dev(0, d_pt, n * size, 0xAE5C0055ull, 0, NULL));
    unsigned char *ivs = malloc(12 * n), *tags = malloc(16 * n), t1[16];
    for (size_t p = 0; p < n; p++) { memset(ivs + 12 * p, 0, 12); memcpy(ivs + 12 * p, &p, sizeof p); ivs[12 * p + 11] = 0xA5; }     /* a distinct IV per message */
    CHECK(aesgcm_dev_upload(0, d_ivs, ivs, 12 * n));
    CHECK(aesgcm_dev_sync(0));

    CHECK(aesgcm_packets_crypt_dev(ctx, 0, n, d_ivs, NULL, 0, NULL, d_pt, size, NULL, d_ct, d_tags, NULL, NULL, NULL));    /* warm */
    CHECK(aesgcm_dev_sync(0));
    const int reps = 8;
    const double t0 = now();
    for (int r = 0; r < reps; r++) CHECK(aesgcm_packets_crypt_dev(ctx, 0, n, d_ivs, NULL, 0, NULL, d_pt, size, NULL, d_ct, d_tags, NULL, NULL, NULL));
    CHECK(aesgcm_dev_sync(0));
    const double dt = (now() - t0) / reps;
    CHECK(aesgcm_dev_download(0, tags, d_tags, 16 * n));
    for (size_t p = 0; p < n; p += (n / 5 ? n / 5 : 1)) {                         /* a sample through the single-message path (16-byte aligned starts only) */
        if ((p * size) % 16) continue;
        CHECK(aesgcm_encrypt_dev(one, ivs + 12 * p, NULL, 0, (const char *)d_pt + p * size, size, d_one, t1, NULL));
        if (memcmp(t1, tags + 16 * p, 16)) { fprintf(stderr, "message %zu: the packets call and aesgcm_encrypt_dev disagree\n", p); return 1; }
    }
    /* the same messages "wherever they live": arrays of device addresses and lengths (aesgcm_messages_crypt_dev) -- here they point into the same buffers, in
     * reverse order; the tags must be the same ones */
    {
        uint64_t *ip = malloc(8 * n), *op = malloc(8 * n);
        uint32_t *ln = malloc(4 * n);
        unsigned char *ivr = malloc(12 * n), *tr = malloc(16 * n);
        void *d_ip, *d_op, *d_ln, *d_ivr, *d_tr;
        for (size_t p = 0; p < n; p++) {
            const size_t q = n - 1 - p;
            ip[p] = (uint64_t)(uintptr_t)((const char *)d_pt + q * size); op[p] = (uint64_t)(uintptr_t)((char *)d_ct + q * size); ln[p] = (uint32_t)size;
            memcpy(ivr + 12 * p, ivs + 12 * q, 12);
        }
        CHECK(aesgcm_dev_alloc(0, &d_ip, 8 * n)); CHECK(aesgcm_dev_alloc(0, &d_op, 8 * n)); CHECK(aesgcm_dev_alloc(0, &d_ln, 4 * n));
        CHECK(aesgcm_dev_alloc(0, &d_ivr, 12 * n + 16)); CHECK(aesgcm_dev_alloc(0, &d_tr, 16 * n));
        CHECK(aesgcm_dev_upload(0, d_ip, ip, 8 * n)); CHECK(aesgcm_dev_upload(0, d_op, op, 8 * n)); CHECK(aesgcm_dev_upload(0, d_ln, ln, 4 * n)); CHECK(aesgcm_dev_upload(0, d_ivr, ivr, 12 * n));
        CHECK(aesgcm_messages_crypt_dev(ctx, 0, n, d_ivr, NULL, NULL, (const uint64_t *)d_ip, (const uint32_t *)d_ln, (const uint64_t *)d_op, d_tr, NULL, NULL, NULL));
        CHECK(aesgcm_dev_sync(0));
        CHECK(aesgcm_dev_download(0, tr, d_tr, 16 * n));
        for (size_t p = 0; p < n; p++)
            if (memcmp(tr + 16 * p, tags + 16 * (n - 1 - p), 16)) { fprintf(stderr, "message %zu: aesgcm_messages_crypt_dev and aesgcm_packets_crypt_dev disagree\n", n - 1 - p); return 1; }
        free(ip); free(op); free(ln); free(ivr); free(tr);
        aesgcm_dev_free(0, d_ip); aesgcm_dev_free(0, d_op); aesgcm_dev_free(0, d_ln); aesgcm_dev_free(0, d_ivr); aesgcm_dev_free(0, d_tr);
    }
    /* decrypt in place with verification; the tag of message 1 forged */
    const size_t bad = n > 1 ? 1 : 0;
    tags[16 * bad + 5] ^= 0x40;
    CHECK(aesgcm_dev_upload(0, d_exp, tags, 16 * n));
    CHECK(aesgcm_packets_crypt_dev(ctx, 1, n, d_ivs, NULL, 0, NULL, d_ct, size, NULL, d_ct, d_tags, d_exp, d_auth, NULL));
    CHECK(aesgcm_dev_sync(0));
    int *auth = malloc(4 * n);
    CHECK(aesgcm_dev_download(0, auth, d_auth, 4 * n));
    size_t failed = 0;
    for (size_t p = 0; p < n; p++) failed += !auth[p];
    unsigned char *back = malloc(size), *want = malloc(size);
    const size_t probe[2] = {bad, n - 1};
    for (int k = 0; k < 2; k++) {
        CHECK(aesgcm_dev_download(0, back, (const char *)d_ct + probe[k] * size, size));
        CHECK(aesgcm_dev_download(0, want, (const char *)d_pt + probe[k] * size, size));
        if (probe[k] == bad) memset(want, 0, size);                                /* wipe_on_auth_fail: zeros, not unauthenticated plaintext */
        if (memcmp(back, want, size)) { fprintf(stderr, "message %zu: wrong bytes after decrypt\n", probe[k]); return 1; }
    }
    if (failed != 1 || auth[bad]) { fprintf(stderr, "%zu messages failed authentication, expected exactly message %zu\n", failed, bad); return 1; }
    printf("%zu messages of %zu bytes under one key, one call: %s, %.3f ms per call, %.1f GiB/s\nMESSAGES OK\n", n, size,
           shape == AESGCM_SHAPE_ROWS ? "by rows" : "packet kernels", dt * 1e3, (double)n * size / dt / (1024.0 * 1024 * 1024));
    aesgcm_ctx_destroy(ctx); aesgcm_ctx_destroy(one);
    return 0;
}
