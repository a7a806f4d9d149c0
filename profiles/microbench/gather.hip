// Microbenchmark (round 2): can the vector-memory path (L1-resident table, per-lane gather) take AES lookups off the LDS?
// The fused kernel is LDS-array bound (32 lookup addresses per clock per CU).  A 1 KiB T-table read with global_load_dword
// gathers never leaves the CU's 32 KiB vector L1; if the texture path sustains its own rate beside a busy LDS, moving one
// AES round of fourteen there would relieve the LDS by 1/14.
//   k<NL,NG>: one 1024-lane workgroup per CU (k_body's shape), every wave iterates "blocks": NL rounds of 16 conflict-free
//   ds_read_b32 lookups (addresses from the previous round: 1 v_perm each) followed by NG rounds of 16 global gathers
//   (address = table + 4 * byte).  Prints ms and the shader clock.
//   hipcc --offload-arch=gfx950 -O3 -o gather gather.hip && ./gather
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32;
#define LDS32(off) (*(const __attribute__((address_space(3))) u32 *)(uintptr_t)(off))

template <int NL, int NG>
__global__ __launch_bounds__(1024, 1) void k(u32 *out, const u32 *__restrict__ tab, int iters, unsigned long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (u32 i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((u32 *)smem)[i] = i * 2654435761u;
    __syncthreads();
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lb = (lane & 31) << 2;
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    u32 s[4] = {lane * 7 + 1, lane * 13 + 5, lane * 29 + 3, lane + 11 + blockIdx.x};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < NL; r++) {
            u32 acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const u32 addr = __builtin_amdgcn_perm(s[j & 3], lb, 0x0c0c0000u | ((4u + ((j >> 2) & 3)) << 8));
                acc[j & 3] ^= LDS32(addr);
            }
            s[0] ^= acc[0]; s[1] ^= acc[1]; s[2] ^= acc[2]; s[3] ^= acc[3];
        }
#pragma unroll
        for (int r = 0; r < NG; r++) {
            u32 acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const u32 b = (s[j & 3] >> (8 * ((j >> 2) & 3))) & 0xffu;
                acc[j & 3] ^= tab[b + 256 * ((j >> 2) & 3)];
            }
            s[0] ^= acc[0]; s[1] ^= acc[1]; s[2] ^= acc[2]; s[3] ^= acc[3];
        }
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    if (lane == 0) { cyc[2 * (blockIdx.x * 16 + wave)] = t1 - t0; cyc[2 * (blockIdx.x * 16 + wave) + 1] = w1 - w0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] ^ s[1] ^ s[2] ^ s[3];
}

template <int NL, int NG>
static float run(int iters, u32 *out, const u32 *tab, unsigned long long *cyc, int n_cu) {
    hipFuncSetAttribute((const void *)&k<NL, NG>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NL, NG>), dim3(n_cu), dim3(1024), 65536, 0, out, tab, iters / 50 + 1, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NL, NG>), dim3(n_cu), dim3(1024), 65536, 0, out, tab, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[2 * 16 * 512];
    const int waves = 16 * n_cu;
    hipMemcpy(h, cyc, 16 * waves, hipMemcpyDeviceToHost);
    double cs = 0, ws = 0; for (int i = 0; i < waves; i++) { cs += (double)h[2 * i]; ws += (double)h[2 * i + 1]; }
    const double mhz = cs / ws * 100.0;
    // CU cycles per wave-"block" (one iteration of one wave), 16 waves per CU
    printf("%2d LDS rounds + %d gather rounds: %8.3f ms  %5.0f MHz  %7.1f CU-cycles per wave-iteration\n", NL, NG, ms, mhz,
           ms * 1e-3 * mhz * 1e6 / (16.0 * iters));
    return ms;
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    u32 *out, *tab; unsigned long long *cyc;
    hipMalloc(&out, 4 * 1024 * n_cu); hipMalloc(&cyc, 16 * 16 * n_cu); hipMalloc(&tab, 4096);
    u32 ht[1024]; for (int i = 0; i < 1024; i++) ht[i] = (u32)i * 2246822519u + 12345u;
    hipMemcpy(tab, ht, 4096, hipMemcpyHostToDevice);
    printf("%s, %d CUs, one 1024-lane workgroup per CU, 64 KiB LDS\n", prop.name, n_cu);
    const int IT = 400;
    run<14, 0>(IT, out, tab, cyc, n_cu);
    run<13, 0>(IT, out, tab, cyc, n_cu);
    run<13, 1>(IT, out, tab, cyc, n_cu);
    run<12, 0>(IT, out, tab, cyc, n_cu);
    run<12, 2>(IT, out, tab, cyc, n_cu);
    run<0, 1>(IT * 4, out, tab, cyc, n_cu);
    run<0, 2>(IT * 4, out, tab, cyc, n_cu);
    run<14, 0>(IT, out, tab, cyc, n_cu);
    return 0;
}
