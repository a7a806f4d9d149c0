// A routed packets call captured into a hipGraph and replayed: examples/graph_replay [n_frames] [replays] [max_len]
//
// With offset arrays every decision of aesgcm_packets_crypt_dev is taken ON THE DEVICE (k_len_scan: which messages go by rows, which to the packet kernels, in which
// shape), so the host's side of the call is a fixed sequence of launches for given pointers and count -- the sort, every candidate packet shape, the row launches on the
// context's side stream, the join.  That is what a graph is for: a gateway that encrypts batch after batch out of the same ring of buffers captures the call once and
// replays it; the frames' lengths, offsets, IVs and bytes may change between replays, the pointers and the count may not.
// MEASURED (MI355X, ROCm 7.2, profiles/r06/graph_replay.txt): the replay is correct and NOT faster -- 4096 frames 84 us per direct call, 102 us per replay; 16384 frames
// 130 / 149; 2^20 frames 1323 / 1310; the host spends 40 - 60 us per call either way.  The example stays as the proof of the property, not as a recommendation.
//
// The library promises for it (include/aesgcm.h, "capture"): after ONE ordinary call with the same context and sizes (scratch and side stream exist then) the enqueue
// path of aesgcm_packets_crypt_dev / aesgcm_messages_crypt_dev makes no allocation, no synchronisation and no host-side read of device results.
//
// The program encrypts n frames (64 .. max_len bytes, default 1514 -- beyond 2 KiB some go by rows, some to the packet kernels --, 28 bytes of AAD, byte-packed) directly and through the graph, with DIFFERENT lengths in the replays than at capture
// time, checks that tags and ciphertext agree, and prints the microseconds per call of both ways.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "aesgcm.h"

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d (%s; %s)\n", #call, rc_, aesgcm_strerror(rc_), aesgcm_last_error()); return 1; } } while (0)
#define HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)

static void draw(std::vector<uint64_t> &doff, uint32_t seed, size_t max_len) {            // lengths 64 .. max_len
    uint32_t s = seed;
    doff[0] = 0;
    for (size_t p = 0; p + 1 < doff.size(); p++) { s = s * 1664525u + 1013904223u; doff[p + 1] = doff[p] + 64 + (s >> 8) % (max_len - 63); }
}

int main(int argc, char **argv) {
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 16384;
    const int replays = argc > 2 ? atoi(argv[2]) : 200;
    const size_t max_len = argc > 3 ? (size_t)atol(argv[3]) : 1514;
    if (max_len < 64 || n * max_len > ((size_t)8 << 30)) { fprintf(stderr, "max_len: 64 .. 8 GiB / n\n"); return 2; }
    const size_t al = 28, cap = n * max_len + 64;
    unsigned char key[32];
    for (int i = 0; i < 32; i++) key[i] = (unsigned char)(i * 7 + 1);
    std::vector<uint64_t> doff(n + 1), aoff(n + 1);
    for (size_t p = 0; p <= n; p++) aoff[p] = p * al;
    std::vector<unsigned char> pt(cap), aad(al * n + 16), ivs(12 * n);
    for (size_t i = 0; i < cap; i++) pt[i] = (unsigned char)(i * 131 + (i >> 9));
    for (size_t i = 0; i < aad.size(); i++) aad[i] = (unsigned char)(i * 17 + 3);
    for (size_t i = 0; i < ivs.size(); i++) ivs[i] = (unsigned char)(i * 29 + (i >> 7));

    aesgcm_ctx *ctx = nullptr;
    void *d_in, *d_out, *d_out2, *d_aad, *d_ivs, *d_tags, *d_tags2, *d_doff, *d_aoff;
    CHECK(aesgcm_ctx_create(&ctx, 0, key, sizeof key));
    CHECK(aesgcm_dev_alloc(0, &d_in, cap)); CHECK(aesgcm_dev_alloc(0, &d_out, cap)); CHECK(aesgcm_dev_alloc(0, &d_out2, cap));
    CHECK(aesgcm_dev_alloc(0, &d_aad, aad.size())); CHECK(aesgcm_dev_alloc(0, &d_ivs, ivs.size()));
    CHECK(aesgcm_dev_alloc(0, &d_tags, 16 * n)); CHECK(aesgcm_dev_alloc(0, &d_tags2, 16 * n));
    CHECK(aesgcm_dev_alloc(0, &d_doff, 8 * (n + 1))); CHECK(aesgcm_dev_alloc(0, &d_aoff, 8 * (n + 1)));
    CHECK(aesgcm_dev_upload(0, d_in, pt.data(), cap)); CHECK(aesgcm_dev_upload(0, d_aad, aad.data(), aad.size())); CHECK(aesgcm_dev_upload(0, d_ivs, ivs.data(), ivs.size()));
    CHECK(aesgcm_dev_upload(0, d_aoff, aoff.data(), 8 * (n + 1)));
    draw(doff, 12345, max_len);
    CHECK(aesgcm_dev_upload(0, d_doff, doff.data(), 8 * (n + 1)));

    hipStream_t st;
    HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto call = [&](void *out, void *tags) {
        return aesgcm_packets_crypt_dev(ctx, 0, n, d_ivs, d_aad, 0, (const uint64_t *)d_aoff, d_in, 0, (const uint64_t *)d_doff, out, tags, nullptr, nullptr, st);
    };
    // one ordinary call: the context's scratch and side stream exist from here on
    CHECK(call(d_out2, d_tags2));
    HIP(hipStreamSynchronize(st));

    // capture the same call (into the other output buffers), instantiate
    hipGraph_t graph; hipGraphExec_t exec;
    HIP(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    const int rc_cap = call(d_out, d_tags);
    HIP(hipStreamEndCapture(st, &graph));
    if (rc_cap) { fprintf(stderr, "call under capture -> %d (%s)\n", rc_cap, aesgcm_last_error()); return 1; }
    HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    size_t nodes = 0;
    HIP(hipGraphGetNodes(graph, nullptr, &nodes));

    // other lengths than at capture time: the graph holds launches, not decisions
    draw(doff, 777, max_len);
    CHECK(aesgcm_dev_upload(0, d_doff, doff.data(), 8 * (n + 1)));
    const size_t total = (size_t)doff[n];
    HIP(hipGraphLaunch(exec, st));
    CHECK(call(d_out2, d_tags2));
    HIP(hipStreamSynchronize(st));
    int code = 0; uint64_t detail = 0;
    CHECK(aesgcm_ctx_status(ctx, &code, &detail));
    if (code) { fprintf(stderr, "status %d detail %llu\n", code, (unsigned long long)detail); return 1; }
    std::vector<unsigned char> a(total), b(total), ta(16 * n), tb(16 * n);
    CHECK(aesgcm_dev_download(0, a.data(), d_out, total)); CHECK(aesgcm_dev_download(0, b.data(), d_out2, total));
    CHECK(aesgcm_dev_download(0, ta.data(), d_tags, 16 * n)); CHECK(aesgcm_dev_download(0, tb.data(), d_tags2, 16 * n));
    if (memcmp(a.data(), b.data(), total) || memcmp(ta.data(), tb.data(), 16 * n)) { fprintf(stderr, "graph replay and direct call DIFFER\n"); return 1; }

    // time: back-to-back calls, one synchronisation at the end
    auto now = [] { return std::chrono::steady_clock::now(); };
    for (int i = 0; i < 10; i++) CHECK(call(d_out2, d_tags2));
    HIP(hipStreamSynchronize(st));
    auto t0 = now();
    for (int i = 0; i < replays; i++) CHECK(call(d_out2, d_tags2));
    auto t0h = now();
    HIP(hipStreamSynchronize(st));
    auto t1 = now();
    for (int i = 0; i < 10; i++) HIP(hipGraphLaunch(exec, st));
    HIP(hipStreamSynchronize(st));
    auto t2 = now();
    for (int i = 0; i < replays; i++) HIP(hipGraphLaunch(exec, st));
    auto t2h = now();
    HIP(hipStreamSynchronize(st));
    auto t3 = now();
    auto us = [](auto d) { return std::chrono::duration<double, std::micro>(d).count(); };
    const double direct = us(t1 - t0) / replays, direct_host = us(t0h - t0) / replays, replay = us(t3 - t2) / replays, replay_host = us(t2h - t2) / replays;
    printf("{\"n_frames\": %zu, \"bytes\": %zu, \"graph_nodes\": %zu, \"replays\": %d, \"direct_us_per_call\": %.1f, \"direct_host_us_per_call\": %.1f, "
           "\"graph_us_per_call\": %.1f, \"graph_host_us_per_call\": %.1f, \"direct_gib_per_s\": %.1f, \"graph_gib_per_s\": %.1f, \"equal\": true}\n",
           n, total, nodes, replays, direct, direct_host, replay, replay_host, total / direct * 1e6 / 1073741824.0, total / replay * 1e6 / 1073741824.0);
    printf("GRAPH REPLAY OK\n");
    (void)hipGraphExecDestroy(exec); (void)hipGraphDestroy(graph); (void)hipStreamDestroy(st);
    aesgcm_ctx_destroy(ctx);
    return 0;
}
