/* Several host threads, a few contexts each, every context with one message in flight (enqueued with tag = NULL, its tag collected through aesgcm_last_tag when
 * the context comes round again): what the device sustains on messages so small that ONE thread's cost per launch (about 6 us) is the limit.  All messages
 * under one key; tags are compared with a waited reference call.
 *   make -C examples mt_stream && examples/mt_stream [message KiB] [threads] [contexts per thread] [messages per thread]                                  */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "aesgcm.h"

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d (%s; %s)\n", #call, rc_, aesgcm_strerror(rc_), aesgcm_last_error()); exit(1); } } while (0)
static size_t g_size; static int g_ctxs, g_msgs, g_ring;
static unsigned char g_key[32];
static void *g_pt, *g_ct;
static unsigned char (*g_want)[16];
static pthread_barrier_t g_bar;
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static void *worker(void *arg) {
    const int t = (int)(intptr_t)arg;
    aesgcm_ctx **ctx = malloc(sizeof *ctx * g_ctxs);
    int *pending = malloc(sizeof *pending * g_ctxs);
    for (int k = 0; k < g_ctxs; k++) { CHECK(aesgcm_ctx_create(&ctx[k], 0, g_key, sizeof g_key)); pending[k] = -1; }
    long bad = 0;
    pthread_barrier_wait(&g_bar);
    for (int i = 0; i < g_msgs; i++) {
        const int k = i % g_ctxs, r = (i * 7 + t * 3) % g_ring;
        unsigned char iv[12] = {0}, tag[16];
        if (pending[k] >= 0) { CHECK(aesgcm_last_tag(ctx[k], tag, NULL)); bad += memcmp(tag, g_want[pending[k]], 16) != 0; }
        iv[11] = (unsigned char)r;
        CHECK(aesgcm_encrypt_dev(ctx[k], iv, NULL, 0, (char *)g_pt + (size_t)r * g_size, g_size, (char *)g_ct + ((size_t)t * g_ring + r) * g_size, NULL, NULL));
        pending[k] = r;
    }
    for (int k = 0; k < g_ctxs; k++) if (pending[k] >= 0) { unsigned char tag[16]; CHECK(aesgcm_last_tag(ctx[k], tag, NULL)); bad += memcmp(tag, g_want[pending[k]], 16) != 0; }
    pthread_barrier_wait(&g_bar);
    for (int k = 0; k < g_ctxs; k++) aesgcm_ctx_destroy(ctx[k]);
    free(ctx); free(pending);
    return (void *)(intptr_t)bad;
}

int main(int argc, char **argv) {
    g_size = (size_t)(argc > 1 ? atol(argv[1]) : 1024) << 10;
    const int T = argc > 2 ? atoi(argv[2]) : 4;
    g_ctxs = argc > 3 ? atoi(argv[3]) : 3;
    g_msgs = argc > 4 ? atoi(argv[4]) : 20000;
    g_ring = 16;
    for (int i = 0; i < 32; i++) g_key[i] = (unsigned char)(i * 5 + 1);
    CHECK(aesgcm_dev_alloc(0, &g_pt, g_size * g_ring)); CHECK(aesgcm_dev_alloc(0, &g_ct, g_size * g_ring * (size_t)(T + 1)));
    CHECK(aesgcm_fill_splitmix64_dev(0, g_pt, g_size * g_ring, 9, 0, NULL));
    g_want = malloc(16 * (size_t)g_ring);
    aesgcm_ctx *ref; CHECK(aesgcm_ctx_create(&ref, 0, g_key, sizeof g_key));
    for (int r = 0; r < g_ring; r++) { unsigned char iv[12] = {0}; iv[11] = (unsigned char)r; CHECK(aesgcm_encrypt_dev(ref, iv, NULL, 0, (char *)g_pt + (size_t)r * g_size, g_size, (char *)g_ct + (size_t)T * g_ring * g_size, g_want[r], NULL)); }
    aesgcm_ctx_destroy(ref);
    pthread_barrier_init(&g_bar, NULL, (unsigned)T + 1);
    pthread_t *th = malloc(sizeof *th * T);
    for (int t = 0; t < T; t++) pthread_create(&th[t], NULL, worker, (void *)(intptr_t)t);
    pthread_barrier_wait(&g_bar);
    const double t0 = now();
    pthread_barrier_wait(&g_bar);
    const double dt = now() - t0;
    long bad = 0;
    for (int t = 0; t < T; t++) { void *r; pthread_join(th[t], &r); bad += (long)(intptr_t)r; }
    printf("%8zu KiB  threads %2d x contexts %d: %9.0f messages/s  %7.1f GiB/s  %6.2f us per message  tags %s\n", g_size >> 10, T, g_ctxs,
           (double)T * g_msgs / dt, (double)T * g_msgs * g_size / dt / (1 << 30), dt / ((double)T * g_msgs) * 1e6, bad ? "BAD" : "ok");
    return bad != 0;
}
