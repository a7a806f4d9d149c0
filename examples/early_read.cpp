/* Is the result in memory when aesgcm_encrypt_dev / aesgcm_decrypt_dev return with the tag?  For messages of 64 KiB .. 1 GiB (1.25 GiB with AAD or a
 * ragged end) the tag is shown by the one launch that processes the message, from inside (k_body's cyclic rows, cyc_close), while that launch is still
 * running; the header promises the whole result in device memory at that moment, and what stands behind the promise is not a release fence but
 * through-the-L2 stores plus s_waitcnt (aesgcm_kernels.hip, fetch_tag).  This program reads the result back through a copy that is ordered behind NOTHING
 * -- hipMemcpyAsync on a non-blocking stream of its own, issued the moment the call returns -- and compares it with the result of the same call read
 * after a device synchronisation.  Consecutive calls alternate between two IVs, so bytes that were still on their way would show the other call's
 * output.  Shapes: encrypt and decrypt, out of place and in place, with AAD in front (front rows through the general row code, byte stores for the
 * ragged end: gstore1_wt_at) and without, sizes from the bottom to the top of the cyclic range.  For big messages only the region that is written
 * LAST is read early (the strands of a cyclic launch end on the last 4096 rows = the last 4 MiB), because a whole-message copy takes milliseconds
 * and would hide a late store behind its own duration.
 *   make -C examples early_read && examples/early_read [scale]          (scale multiplies the call counts; 1 = about 3000 calls)                     */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "aesgcm.h"

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d (%s; %s)\n", #call, rc_, aesgcm_strerror(rc_), aesgcm_last_error()); return 1; } } while (0)
#define HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)

struct Shape { size_t n, aad; int dec, inplace, calls, half; };      /* half: the context option cyc_half (k_bodyh, two workgroups per CU; messages < 80 MiB) */

int main(int argc, char **argv) {
    const double scale = argc > 1 ? atof(argv[1]) : 1.0;
    unsigned char key[32], iv[2][12], tag[16], tag_ref[2][16], aad[4096];
    for (int i = 0; i < 32; i++) key[i] = (unsigned char)(i * 7 + 1);
    for (int i = 0; i < 12; i++) { iv[0][i] = (unsigned char)i; iv[1][i] = (unsigned char)(0xA0 + i); }
    for (int i = 0; i < 4096; i++) aad[i] = (unsigned char)(i * 13 + 5);
    const size_t MiB = (size_t)1 << 20, KiB = 1024;
    const Shape shapes[] = {
        {64 * KiB, 0, 0, 0, 600}, {64 * KiB + 1, 20, 0, 1, 300}, {96 * KiB + 1013, 4095, 1, 0, 300}, {MiB + 5, 0, 0, 0, 300}, {MiB + 5, 16, 1, 1, 300},
        {4 * MiB - 1008, 1000, 0, 1, 200}, {16 * MiB, 0, 0, 0, 200}, {16 * MiB + 17, 20, 1, 0, 200}, {16 * MiB - 15, 0, 1, 1, 100},
        {64 * MiB - 1008, 0, 0, 0, 100}, {64 * MiB + 3, 68, 0, 1, 60}, {64 * MiB + 1023, 1, 1, 1, 60},
        {256 * MiB + 16, 20, 0, 0, 40}, {256 * MiB, 0, 1, 1, 30},
        {1008 * MiB + 5, 0, 0, 0, 20}, {1023 * MiB + 1019, 28, 1, 0, 20}, {1200 * MiB + 7, 20, 0, 1, 16},          /* the top of the cyclic range, with and without pieces */
        /* the half shape of the launch: the same promise from k_bodyh's closing */
        {64 * KiB, 0, 0, 0, 300, 1}, {96 * KiB + 1013, 4095, 1, 1, 200, 1}, {MiB + 5, 16, 0, 1, 200, 1}, {4 * MiB - 1008, 1000, 1, 0, 150, 1},
        {16 * MiB + 17, 20, 0, 0, 150, 1}, {16 * MiB - 15, 0, 1, 1, 100, 1}, {64 * MiB + 1023, 1, 0, 1, 60, 1}, {79 * MiB + 5, 68, 1, 0, 40, 1},
    };
    size_t nmax = 0;
    for (const Shape &s : shapes) nmax = s.n > nmax ? s.n : nmax;
    const size_t window = 8 * MiB;                                     /* what is read early of a big message: its end */
    aesgcm_ctx *ctx = NULL, *ctx_full = NULL, *ctx_half = NULL;
    void *d_pt = NULL, *d_ct[2] = {NULL, NULL}, *d_work = NULL, *d_out = NULL, *d_aad = NULL;
    unsigned char *h_early = NULL;
    CHECK(aesgcm_ctx_create(&ctx_full, 0, key, sizeof key));
    CHECK(aesgcm_ctx_create(&ctx_half, 0, key, sizeof key));
    CHECK(aesgcm_ctx_set_option(ctx_full, "cyc_half", 0));            /* never / always: the library's own rule (half when another context has a message under way) is not what is tested here */
    CHECK(aesgcm_ctx_set_option(ctx_half, "cyc_half", 1));
    CHECK(aesgcm_dev_alloc(0, &d_pt, nmax + 64));
    CHECK(aesgcm_dev_alloc(0, &d_ct[0], nmax + 64));
    CHECK(aesgcm_dev_alloc(0, &d_ct[1], nmax + 64));
    CHECK(aesgcm_dev_alloc(0, &d_work, nmax + 64));
    CHECK(aesgcm_dev_alloc(0, &d_out, nmax + 64));
    CHECK(aesgcm_dev_alloc(0, &d_aad, sizeof aad));
    CHECK(aesgcm_dev_upload(0, d_aad, aad, sizeof aad));
    CHECK(aesgcm_fill_splitmix64_dev(0, d_pt, (nmax + 7) / 8 * 8, 1, 0, NULL));
    CHECK(aesgcm_dev_sync(0));
    HIP(hipHostMalloc((void **)&h_early, window + 64, hipHostMallocDefault));
    hipStream_t side;
    HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    int bad_total = 0;
    long calls_total = 0;
    std::vector<unsigned char> ref[2];
    for (const Shape &s : shapes) {
        const size_t n = s.n, w = n < window ? n : window, w0 = n - w;              /* early-read region [w0, n) */
        const void *a = s.aad ? d_aad : NULL;
        ctx = s.half ? ctx_half : ctx_full;
        /* reference results, read after a full synchronisation: ciphertext k for encrypt, the plaintext for decrypt */
        for (int k = 0; k < 2; k++) {
            CHECK(aesgcm_encrypt_dev(ctx, iv[k], a, s.aad, d_pt, n, d_ct[k], tag_ref[k], NULL));
            HIP(hipDeviceSynchronize());
            ref[k].resize(w);
            HIP(hipMemcpy(ref[k].data(), (const unsigned char *)(s.dec ? d_pt : d_ct[k]) + w0, w, hipMemcpyDeviceToHost));
        }
        const int n_calls = (int)(s.calls * scale) < 4 ? 4 : (int)(s.calls * scale);
        int bad = 0;
        for (int i = 0; i < n_calls; i++) {
            const int k = i & 1;
            const void *src = s.dec ? d_ct[k] : d_pt;
            void *dst = d_out;
            if (s.inplace) {                                                         /* the input is overwritten: a fresh copy of it for every call */
                HIP(hipMemcpy(d_work, src, n, hipMemcpyDeviceToDevice));
                src = dst = d_work;
            }
            HIP(hipDeviceSynchronize());
            if (s.dec) CHECK(aesgcm_decrypt_dev(ctx, iv[k], a, s.aad, src, n, dst, tag_ref[k], tag, NULL));
            else CHECK(aesgcm_encrypt_dev(ctx, iv[k], a, s.aad, src, n, dst, tag, NULL));
            HIP(hipMemcpyAsync(h_early, (const unsigned char *)dst + w0, w, hipMemcpyDeviceToHost, side));       /* ordered behind nothing */
            HIP(hipStreamSynchronize(side));
            if (memcmp(tag, tag_ref[k], 16) || memcmp(h_early, ref[k].data(), w)) {
                size_t first = 0;
                while (first < w && h_early[first] == ref[k][first]) first++;
                if (!bad) fprintf(stderr, "n %zu aad %zu dec %d inplace %d call %d: early read differs from byte %zu on\n", n, s.aad, s.dec, s.inplace, i, w0 + first);
                bad++;
            }
        }
        HIP(hipDeviceSynchronize());
        printf("%11zu bytes, aad %4zu, %s, %s, %s shape: %d of %d early reads differ\n", n, s.aad, s.dec ? "decrypt" : "encrypt", s.inplace ? "in place    " : "out of place",
               s.half ? "half" : "full", bad, n_calls);
        bad_total += bad;
        calls_total += n_calls;
    }
    (void)hipStreamDestroy(side);
    (void)hipHostFree(h_early);
    aesgcm_dev_free(0, d_pt); aesgcm_dev_free(0, d_ct[0]); aesgcm_dev_free(0, d_ct[1]); aesgcm_dev_free(0, d_work); aesgcm_dev_free(0, d_out); aesgcm_dev_free(0, d_aad);
    aesgcm_ctx_destroy(ctx_full); aesgcm_ctx_destroy(ctx_half);
    if (bad_total) { printf("EARLY READ FAILED (%d of %ld)\n", bad_total, calls_total); return 1; }
    printf("EARLY READ OK (%ld calls)\n", calls_total);
    return 0;
}
