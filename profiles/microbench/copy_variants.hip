// copy_variants.hip -- which plain copy kernel reaches the guide's ~6.3 TB/s (MI355X_MICROARCH.md: float4 copy, 79 % of 8 TB/s)?
// The yardstick bench.py prints beside the fused kernel (roofline.measured_copy_kernel) was k_copy16 = variant A, 4.8-5.1 TB/s.
//   hipcc -O3 --offload-arch=gfx950 -o copy_variants copy_variants.hip && ./copy_variants [GiB]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t v4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// A: the round-2 kernel: grid-stride, one 16-byte element per iteration
__global__ __launch_bounds__(256) void kA(v4 *__restrict__ d, const v4 *__restrict__ s, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) d[i] = s[i];
}
// B: one element per thread, no loop
__global__ __launch_bounds__(256) void kB(v4 *__restrict__ d, const v4 *__restrict__ s, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = s[i];
}
// C<U,NT>: a block copies a contiguous tile of 256*U elements: U loads in flight per lane, then U stores; grid-stride over tiles
template <int U, bool NT>
__global__ __launch_bounds__(256) void kC(v4 *__restrict__ d, const v4 *__restrict__ s, uint64_t n) {
    const uint64_t tiles = n / (256 * U);
    for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const v4 *sp = s + t * 256 * U + threadIdx.x;
        v4 *dp = d + t * 256 * U + threadIdx.x;
        v4 r[U];
#pragma unroll
        for (int k = 0; k < U; k++) r[k] = NT ? __builtin_nontemporal_load(sp + 256 * k) : sp[256 * k];
#pragma unroll
        for (int k = 0; k < U; k++) { if (NT) __builtin_nontemporal_store(r[k], dp + 256 * k); else dp[256 * k] = r[k]; }
    }
}
template <typename F> static double run(F launch, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(a, 0)); launch(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    return best;
}
int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 16.0;
    const uint64_t bytes = (uint64_t)(gib * (1ull << 30)) / 32768 * 32768, n = bytes / 16;
    v4 *s, *d; CK(hipMalloc(&s, bytes)); CK(hipMalloc(&d, bytes));
    CK(hipMemset(s, 0x5a, bytes)); CK(hipMemset(d, 0, bytes));
    auto rep = [&](const char *name, double ms) { printf("%-44s %8.3f ms  %7.1f GB/s read+write\n", name, ms, 2.0 * bytes / ms / 1e6); fflush(stdout); };
    rep("A grid-stride 8192x256, 1 x 16 B", run([&] { hipLaunchKernelGGL(kA, dim3(8192), dim3(256), 0, 0, d, s, n); }));
    rep("A grid-stride 2048x256, 1 x 16 B", run([&] { hipLaunchKernelGGL(kA, dim3(2048), dim3(256), 0, 0, d, s, n); }));
    rep("B one element per thread", run([&] { hipLaunchKernelGGL(kB, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d, s, n); }));
    for (int g : {1024, 2048, 4096, 16384}) {
        char nm[96];
        snprintf(nm, sizeof nm, "C U=4 tiles, grid %d", g); rep(nm, run([&] { hipLaunchKernelGGL((kC<4, false>), dim3(g), dim3(256), 0, 0, d, s, n); }));
        snprintf(nm, sizeof nm, "C U=8 tiles, grid %d", g); rep(nm, run([&] { hipLaunchKernelGGL((kC<8, false>), dim3(g), dim3(256), 0, 0, d, s, n); }));
        snprintf(nm, sizeof nm, "C U=8 tiles, nontemporal, grid %d", g); rep(nm, run([&] { hipLaunchKernelGGL((kC<8, true>), dim3(g), dim3(256), 0, 0, d, s, n); }));
        snprintf(nm, sizeof nm, "C U=16 tiles, grid %d", g); rep(nm, run([&] { hipLaunchKernelGGL((kC<16, false>), dim3(g), dim3(256), 0, 0, d, s, n); }));
    }
    rep("C U=8 one tile per block", run([&] { hipLaunchKernelGGL((kC<8, false>), dim3((unsigned)(n / 2048)), dim3(256), 0, 0, d, s, n); }));
    rep("C U=4 one tile per block", run([&] { hipLaunchKernelGGL((kC<4, false>), dim3((unsigned)(n / 1024)), dim3(256), 0, 0, d, s, n); }));
    rep("hipMemcpyAsync device to device", run([&] { CK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0)); }));
    return 0;
}
