/* Plain-C latency probe of the device-pointer entry point: microseconds per aesgcm_encrypt_dev call (launches, the tag
 * landing in the pinned host slot, stream synchronisation) for a range of message sizes -- what a C caller of the ABI
 * sees, without the Python/ctypes call overhead that profiles/latency.py includes.
 *   make -C examples latency && examples/latency [calls per size]                                                  */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "aesgcm.h"

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d (%s; %s)\n", #call, rc_, aesgcm_strerror(rc_), aesgcm_last_error()); return 1; } } while (0)
static double now_us(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }
static int cmp(const void *a, const void *b) { double x = *(const double *)a, y = *(const double *)b; return x < y ? -1 : x > y; }

int main(int argc, char **argv) {
    const int n_calls = argc > 1 ? atoi(argv[1]) : 500;
    unsigned char key[32], iv[12], tag[16];
    for (int i = 0; i < 32; i++) key[i] = (unsigned char)i;
    for (int i = 0; i < 12; i++) iv[i] = (unsigned char)i;
    aesgcm_ctx *ctx = NULL;
    void *d_in = NULL, *d_out = NULL;
    CHECK(aesgcm_ctx_create(&ctx, 0, key, sizeof key));
    CHECK(aesgcm_dev_alloc(0, &d_in, 4u << 20));
    CHECK(aesgcm_dev_alloc(0, &d_out, 4u << 20));
    CHECK(aesgcm_fill_splitmix64_dev(0, d_in, 4u << 20, 1, 0, NULL));
    CHECK(aesgcm_dev_sync(0));
    const size_t sizes[] = {48, 1024, 4096, 16384, 65536, 262144, 1048576, 4194304};
    double *t = (double *)malloc(sizeof(double) * (size_t)n_calls);
    printf("%10s %10s %10s   (AES-256-GCM, device-resident, tag to host; us per aesgcm_encrypt_dev call from C)\n", "bytes", "median", "best");
    for (unsigned s = 0; s < sizeof sizes / sizeof sizes[0]; s++) {
        for (int i = 0; i < 20; i++) CHECK(aesgcm_encrypt_dev(ctx, iv, NULL, 0, d_in, sizes[s], d_out, tag, NULL));
        for (int i = 0; i < n_calls; i++) {
            const double t0 = now_us();
            CHECK(aesgcm_encrypt_dev(ctx, iv, NULL, 0, d_in, sizes[s], d_out, tag, NULL));
            t[i] = now_us() - t0;
        }
        qsort(t, (size_t)n_calls, sizeof(double), cmp);
        printf("%10zu %10.1f %10.1f\n", sizes[s], t[n_calls / 2], t[0]);
    }
    /* the host-buffer entry point (aesgcm_encrypt: H2D, kernels, D2H, tag), both buffers pageable, and both
     * page-locked (aesgcm_host_alloc): since round 3 the call waits for the D2H copy in both cases (with a
     * page-locked buffer the copy is truly asynchronous and the call used to return on the tag alone) */
    {
        const size_t hs[] = {65536, 4194304};
        unsigned char *pt_pageable = (unsigned char *)malloc(4u << 20), *ct_pageable = (unsigned char *)malloc(4u << 20);
        void *ct_pinned = NULL, *pt_pinned = NULL;
        CHECK(aesgcm_host_alloc(&ct_pinned, 4u << 20));
        CHECK(aesgcm_host_alloc(&pt_pinned, 4u << 20));
        memset(pt_pageable, 0x5a, 4u << 20); memset(pt_pinned, 0x5a, 4u << 20);
        printf("%10s %10s %10s %10s %10s   (aesgcm_encrypt from host memory: plaintext and ciphertext both pageable / both page-locked; median, best)\n",
               "bytes", "pageable", "best", "pinned", "best");
        for (unsigned s = 0; s < sizeof hs / sizeof hs[0]; s++) {
            double med[2], best[2];
            for (int which = 0; which < 2; which++) {
                unsigned char *ct = which ? (unsigned char *)ct_pinned : ct_pageable;
                const unsigned char *pt = which ? (const unsigned char *)pt_pinned : pt_pageable;
                for (int i = 0; i < 20; i++) CHECK(aesgcm_encrypt(ctx, iv, NULL, 0, pt, hs[s], ct, tag));
                for (int i = 0; i < n_calls; i++) {
                    const double t0 = now_us();
                    CHECK(aesgcm_encrypt(ctx, iv, NULL, 0, pt, hs[s], ct, tag));
                    t[i] = now_us() - t0;
                }
                qsort(t, (size_t)n_calls, sizeof(double), cmp);
                med[which] = t[n_calls / 2]; best[which] = t[0];
            }
            printf("%10zu %10.1f %10.1f %10.1f %10.1f\n", hs[s], med[0], best[0], med[1], best[1]);
        }
        free(pt_pageable); free(ct_pageable);
        aesgcm_host_free(ct_pinned); aesgcm_host_free(pt_pinned);
    }
    free(t);
    aesgcm_dev_free(0, d_in); aesgcm_dev_free(0, d_out);
    aesgcm_ctx_destroy(ctx);
    return 0;
}
