// mt_drive.cpp -- the HOST side of libaesgcm_hip.so under the address / undefined-behaviour / thread sanitizers (test infrastructure only; round 6).
//
// csrc/aesgcm_host.hip, aesgcm_abi.hip and aesgcm_comm.hip hold registries (contexts, per-device state), a pool of streams, a pinned host slot polled from whatever
// thread asks for a tag, scratch that grows, a side stream per context -- the part of the library that takes threads (examples/mt_stream.c).  Linked here against the
// fake HIP runtime of this directory (fakehip.cpp: no GPU, launches launch nothing) and driven through the C ABI by eight threads over four fake devices, every
// thread for itself: create, messages queued with tag = NULL and collected with aesgcm_last_tag, packets (fixed-size records by rows and through the packet kernels,
// offset arrays: the routed call with its fork to the side stream), messages wherever they live, a streaming session exported and imported into a second context,
// rekey, status, destroy; one thread more drives the four-device object.  asan.mk / tsan.mk build it; tests/test_fake_hip.py runs both.  The fake records device
// discipline as always: its violation count must be zero at the end.
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../../include/aesgcm.h"

extern "C" int fake_violations(char *buf, size_t n);

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "thread %d: %s -> %d (%s; %s)\n", t, #call, rc_, aesgcm_strerror(rc_), aesgcm_last_error()); exit(1); } } while (0)
static const size_t MB = 1 << 20;
static int g_rounds = 6;

static void *worker(void *arg) {
    const int t = (int)(intptr_t)arg, dev = t % 4;
    unsigned char key[32], iv[12] = {0}, tag[16], blob[AESGCM_STREAM_STATE_BYTES];
    for (int i = 0; i < 32; i++) key[i] = (unsigned char)(i * 7 + t);
    for (int r = 0; r < g_rounds; r++) {
        aesgcm_ctx *a = nullptr, *b = nullptr;
        CHECK(aesgcm_ctx_create(&a, dev, key, 16 + 8 * (size_t)((r + t) % 3)));
        CHECK(aesgcm_ctx_create(&b, dev, key, 16 + 8 * (size_t)((r + t) % 3)));
        void *d_in = nullptr, *d_out = nullptr, *d_ivs = nullptr, *d_tags = nullptr, *d_off = nullptr, *d_ptr = nullptr, *d_len = nullptr, *d_auth = nullptr;
        const size_t n = 3000;
        CHECK(aesgcm_dev_alloc(dev, &d_in, 8 * MB)); CHECK(aesgcm_dev_alloc(dev, &d_out, 8 * MB)); CHECK(aesgcm_dev_alloc(dev, &d_ivs, 12 * n)); CHECK(aesgcm_dev_alloc(dev, &d_tags, 16 * n));
        CHECK(aesgcm_dev_alloc(dev, &d_off, 8 * (n + 1))); CHECK(aesgcm_dev_alloc(dev, &d_ptr, 8 * n)); CHECK(aesgcm_dev_alloc(dev, &d_len, 4 * n)); CHECK(aesgcm_dev_alloc(dev, &d_auth, 4 * n));
        std::vector<uint64_t> off(n + 1), ptr(n);
        std::vector<uint32_t> len(n, 2048);
        for (size_t i = 0; i <= n; i++) off[i] = 2048 * i;
        for (size_t i = 0; i < n; i++) ptr[i] = (uint64_t)(uintptr_t)d_in + 2048 * i;
        CHECK(aesgcm_dev_upload(dev, d_off, off.data(), 8 * (n + 1))); CHECK(aesgcm_dev_upload(dev, d_ptr, ptr.data(), 8 * n)); CHECK(aesgcm_dev_upload(dev, d_len, len.data(), 4 * n));
        // whole messages: queued on two contexts, collected when the context comes round again; a waited call; host buffers
        for (int m = 0; m < 6; m++) {
            aesgcm_ctx *c = (m & 1) ? b : a;
            if (m >= 2) CHECK(aesgcm_last_tag(c, tag, nullptr));
            iv[11] = (unsigned char)m;
            CHECK(aesgcm_encrypt_dev(c, iv, nullptr, 0, d_in, (size_t)(m + 1) * 300000, d_out, nullptr, nullptr));
        }
        CHECK(aesgcm_last_tag(a, tag, nullptr)); CHECK(aesgcm_last_tag(b, tag, nullptr));
        CHECK(aesgcm_ctx_wait(a, b));
        CHECK(aesgcm_decrypt_dev(a, iv, nullptr, 0, d_out, 5 * MB, d_in, nullptr, tag, nullptr));
        { std::vector<unsigned char> h(70000), o(70000); CHECK(aesgcm_encrypt(b, iv, key, 20, h.data(), h.size(), o.data(), tag)); }
        // packets: fixed-size records (packet kernels, rows), offset arrays (routed: the fork to the side stream), messages wherever they live, decrypt with verdicts and the wipe
        CHECK(aesgcm_ctx_set_option(a, "wipe_on_auth_fail", 1));
        CHECK(aesgcm_packets_crypt_dev(a, 0, n, d_ivs, nullptr, 0, nullptr, d_in, 1024, nullptr, d_out, d_tags, nullptr, nullptr, nullptr));
        CHECK(aesgcm_packets_crypt_dev(a, 0, 100, d_ivs, nullptr, 0, nullptr, d_in, 65536, nullptr, d_out, d_tags, nullptr, nullptr, nullptr));
        CHECK(aesgcm_packets_crypt_dev(a, 1, n, d_ivs, nullptr, 0, nullptr, d_in, 0, (const uint64_t *)d_off, d_out, d_tags, d_tags, (int *)d_auth, nullptr));
        CHECK(aesgcm_messages_crypt_dev(b, 0, n, d_ivs, nullptr, nullptr, (const uint64_t *)d_ptr, (const uint32_t *)d_len, (const uint64_t *)d_ptr, d_tags, nullptr, nullptr, nullptr));
        { int code = -1; uint64_t detail = 0; CHECK(aesgcm_ctx_status(a, &code, &detail)); CHECK(aesgcm_ctx_status(b, &code, &detail)); }
        { uint64_t route[4]; CHECK(aesgcm_ctx_last_route(a, route)); }
        // a streaming session that changes contexts
        CHECK(aesgcm_stream_begin(a, iv, 0)); CHECK(aesgcm_stream_aad(a, key, 16)); CHECK(aesgcm_stream_update_dev(a, d_in, 2 * MB, d_out, nullptr));
        CHECK(aesgcm_stream_export(a, blob)); CHECK(aesgcm_stream_import(b, blob)); CHECK(aesgcm_stream_final(a, tag));
        CHECK(aesgcm_stream_update_dev(b, d_in, 100, d_out, nullptr)); CHECK(aesgcm_stream_final(b, tag));
        // a new key, a batch with a key per packet on the context's stream, teardown
        CHECK(aesgcm_ctx_rekey(a, key, 32));
        { void *st = nullptr; CHECK(aesgcm_ctx_stream(b, &st)); CHECK(aesgcm_batch_crypt_dev(dev, 0, 500, 16, d_in, d_ivs, nullptr, 0, d_in, 1024, d_out, d_tags, nullptr, nullptr, st)); }
        CHECK(aesgcm_ctx_destroy(a)); CHECK(aesgcm_ctx_destroy(b));
        void *bufs[] = {d_in, d_out, d_ivs, d_tags, d_off, d_ptr, d_len, d_auth};
        for (void *p : bufs) CHECK(aesgcm_dev_free(dev, p));
    }
    return nullptr;
}

static void *mgpu_worker(void *arg) {
    const int t = (int)(intptr_t)arg;
    unsigned char key[32] = {9}, iv[12] = {0}, tags[16 * 8];
    const int devs[4] = {0, 1, 2, 3};
    for (int r = 0; r < g_rounds; r++) {
        aesgcm_mgpu *m = nullptr;
        CHECK(aesgcm_mgpu_create(&m, 4, devs, key, 32));
        void *in[4], *out[4];
        size_t len[4] = {MB, MB, MB, MB - 3};
        for (int g = 0; g < 4; g++) { CHECK(aesgcm_dev_alloc(g, &in[g], MB)); CHECK(aesgcm_dev_alloc(g, &out[g], MB)); }
        for (int k = 0; k < 5; k++) CHECK(aesgcm_mgpu_crypt_dev(m, 0, iv, nullptr, 0, in, len, out, nullptr));
        CHECK(aesgcm_mgpu_last_tags(m, 2, tags)); CHECK(aesgcm_mgpu_last_tags(m, 3, tags));
        CHECK(aesgcm_mgpu_crypt_dev(m, 1, iv, nullptr, 0, in, len, out, tags));
        CHECK(aesgcm_mgpu_sync(m));
        for (int g = 0; g < 4; g++) { CHECK(aesgcm_dev_free(g, in[g])); CHECK(aesgcm_dev_free(g, out[g])); }
        CHECK(aesgcm_mgpu_destroy(m));
    }
    return nullptr;
}

int main(int argc, char **argv) {
    if (argc > 1) g_rounds = atoi(argv[1]);
    const int T = 8;
    pthread_t th[T + 1];
    for (int t = 0; t < T; t++) pthread_create(&th[t], nullptr, worker, (void *)(intptr_t)t);
    pthread_create(&th[T], nullptr, mgpu_worker, (void *)(intptr_t)T);
    for (int t = 0; t <= T; t++) pthread_join(th[t], nullptr);
    static char buf[1 << 16];
    const int bad = fake_violations(buf, sizeof buf);
    if (bad) { fprintf(stderr, "%d violations of device discipline\n%s", bad, buf); return 1; }
    printf("MT DRIVE OK (%d threads x %d rounds over 4 fake devices)\n", T + 1, g_rounds);
    return 0;
}
