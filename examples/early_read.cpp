/* Is the ciphertext in memory when aesgcm_encrypt_dev returns with the tag?  For messages of 64 KiB .. 512 MiB the tag is shown by the one launch
 * that encrypts the message, from inside (k_body's cyclic rows, cyc_close), while that launch is still running; the header promises the whole result
 * in device memory at that moment.  This program reads the ciphertext back through a copy that is ordered behind NOTHING -- hipMemcpyAsync on a
 * non-blocking stream of its own, issued the moment the call returns -- and compares it with the ciphertext of the same message read after a device
 * synchronisation.  Consecutive calls use different IVs, so bytes that were still on their way would show the previous call's ciphertext.
 *   make -C examples early_read && examples/early_read [calls per size]                                                                      */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "aesgcm.h"

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d (%s; %s)\n", #call, rc_, aesgcm_strerror(rc_), aesgcm_last_error()); return 1; } } while (0)
#define HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
    const int n_calls = argc > 1 ? atoi(argv[1]) : 60;
    unsigned char key[32], iv[2][12], tag[16], tag_ref[2][16];
    for (int i = 0; i < 32; i++) key[i] = (unsigned char)(i * 7 + 1);
    for (int i = 0; i < 12; i++) { iv[0][i] = (unsigned char)i; iv[1][i] = (unsigned char)(0xA0 + i); }
    const size_t nmax = (size_t)64 << 20;
    aesgcm_ctx *ctx = NULL;
    void *d_in = NULL, *d_out = NULL;
    unsigned char *h_early = NULL;
    CHECK(aesgcm_ctx_create(&ctx, 0, key, sizeof key));
    CHECK(aesgcm_dev_alloc(0, &d_in, nmax + 64));
    CHECK(aesgcm_dev_alloc(0, &d_out, nmax + 64));
    CHECK(aesgcm_fill_splitmix64_dev(0, d_in, nmax, 1, 0, NULL));
    CHECK(aesgcm_dev_sync(0));
    HIP(hipHostMalloc((void **)&h_early, nmax + 64, hipHostMallocDefault));
    hipStream_t side;
    HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    const size_t sizes[] = {(size_t)64 << 10, ((size_t)1 << 20) + 5, (size_t)16 << 20, ((size_t)64 << 20) - 1008};
    int bad_total = 0;
    for (unsigned s = 0; s < sizeof sizes / sizeof sizes[0]; s++) {
        const size_t n = sizes[s];
        std::vector<unsigned char> ref[2];
        for (int k = 0; k < 2; k++) {                                   // the two reference ciphertexts, read after a full synchronisation
            CHECK(aesgcm_encrypt_dev(ctx, iv[k], NULL, 0, d_in, n, d_out, tag_ref[k], NULL));
            HIP(hipDeviceSynchronize());
            ref[k].resize(n);
            HIP(hipMemcpy(ref[k].data(), d_out, n, hipMemcpyDeviceToHost));
        }
        int bad = 0;
        for (int i = 0; i < n_calls; i++) {
            const int k = i & 1;
            CHECK(aesgcm_encrypt_dev(ctx, iv[k], NULL, 0, d_in, n, d_out, tag, NULL));
            HIP(hipMemcpyAsync(h_early, d_out, n, hipMemcpyDeviceToHost, side));       // ordered behind nothing
            HIP(hipStreamSynchronize(side));
            if (memcmp(tag, tag_ref[k], 16) || memcmp(h_early, ref[k].data(), n)) {
                size_t first = 0;
                while (first < n && h_early[first] == ref[k][first]) first++;
                if (!bad) fprintf(stderr, "size %zu call %d: early read differs from byte %zu on\n", n, i, first);
                bad++;
            }
            HIP(hipDeviceSynchronize());                                               // the next call overwrites d_out
        }
        printf("%10zu bytes: %d of %d early reads differ\n", n, bad, n_calls);
        bad_total += bad;
    }
    (void)hipStreamDestroy(side);
    (void)hipHostFree(h_early);
    aesgcm_dev_free(0, d_in); aesgcm_dev_free(0, d_out);
    aesgcm_ctx_destroy(ctx);
    if (bad_total) { printf("EARLY READ FAILED\n"); return 1; }
    printf("EARLY READ OK\n");
    return 0;
}
