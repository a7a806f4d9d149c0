// Microbenchmark: what do ds_read_b32 / b64 / b128 and VALU instructions cost on MI355X when 32 waves per
// CU issue them the way the AES-GCM kernel does (batches of independent reads whose addresses come from the
// previous batch, conflict-free layouts)?  Prints shader cycles per loop iteration per CU-wave-slot.
//   hipcc --offload-arch=gfx950 -O3 -o issue_cost issue_cost.hip && ./issue_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32;
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define LDS32(off) (*(const __attribute__((address_space(3))) u32 *)(uintptr_t)(off))
#define LDS64(off) (*(const __attribute__((address_space(3))) u32x2 *)(uintptr_t)(off))
#define LDS128(off) (*(const __attribute__((address_space(3))) u32x4 *)(uintptr_t)(off))

// NB32 reads of the 32-replica byte table (address = one v_perm), NB64 / NB128 reads of one-row tables,
// NV extra independent VALU ops, per iteration.
template <int NB32, int NB64, int NB128, int NV>
__global__ __launch_bounds__(1024, 8) void k(u32 *out, int iters, unsigned long long *cycles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (u32 i = threadIdx.x; i < 73728 / 4; i += 1024) ((u32 *)smem)[i] = i * 2654435761u;
    __syncthreads();
    const u32 lane = threadIdx.x & 63, lb = (lane & 31) << 2;
    u32 s[4] = {lane * 7 + 1, lane * 13 + 5, lane * 29 + 3, lane + 11};
    u32 d0 = lane, d1 = lane + 1, d2 = lane + 2, d3 = lane + 3;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        u32 acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < NB32; j++) {
            const u32 addr = __builtin_amdgcn_perm(s[j & 3], lb, 0x0c0c0000u | ((4u + ((j >> 2) & 3)) << 8));
            acc[j & 3] ^= LDS32(addr + 8192u);
        }
#pragma unroll
        for (int j = 0; j < NB64; j++) {
            const u32 off = ((s[j & 3] >> (2 * (j >> 2))) & 0xF8u) + (u32)(j & 15) * 256u;
            const u32x2 v = LDS64(off);
            acc[j & 3] ^= v.x; acc[(j + 1) & 3] ^= v.y;
        }
#pragma unroll
        for (int j = 0; j < NB128; j++) {
            const u32 off = ((s[j & 3] >> (2 * (j >> 2))) & 0xF0u) + (u32)(j & 31) * 256u;
            const u32x4 v = LDS128(off);
            acc[0] ^= v.x; acc[1] ^= v.y; acc[2] ^= v.z; acc[3] ^= v.w;
        }
#pragma unroll
        for (int j = 0; j < NV; j += 4) {
            d0 = __builtin_amdgcn_bitop3_b32(d0, d1, d2, 0x96); d1 = __builtin_amdgcn_bitop3_b32(d1, d2, d3, 0x96);
            d2 = __builtin_amdgcn_bitop3_b32(d2, d3, d0, 0x96); d3 = __builtin_amdgcn_bitop3_b32(d3, d0, d1, 0x96);
        }
        s[0] ^= acc[0]; s[1] ^= acc[1]; s[2] ^= acc[2]; s[3] ^= acc[3];
    }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 1024 + threadIdx.x] = s[0] ^ s[1] ^ s[2] ^ s[3] ^ d0 ^ d1 ^ d2 ^ d3;
}

template <int NB32, int NB64, int NB128, int NV>
static void run(const char *name, u32 *out, unsigned long long *cyc, int n_cu) {
    const int iters = 2000, wgs = 2 * n_cu;
    hipFuncSetAttribute((const void *)&k<NB32, NB64, NB128, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NB32, NB64, NB128, NV>), dim3(wgs), dim3(1024), 73728, 0, out, 10, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NB32, NB64, NB128, NV>), dim3(wgs), dim3(1024), 73728, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[1024]; hipMemcpy(h, cyc, sizeof(unsigned long long) * wgs, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < wgs; i++) avg += (double)h[i]; avg /= wgs;
    // per CU: 32 waves each doing `iters` iterations in avg cycles -> CU-cycles per wave-iteration
    const double per = avg / iters / 32.0;
    printf("%-34s %7.2f ms  clock %4.0f MHz  %8.2f CU-cycles per wave-iteration", name, ms, avg / (ms * 1e3), per);
    const double units = NB32 + NB64 + NB128 + NV;
    printf("   (%.3f per op)\n", per / (units > 0 ? units : 1));
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    u32 *out; unsigned long long *cyc;
    hipMalloc(&out, 4 * 1024 * 2 * n_cu); hipMalloc(&cyc, 8 * 2 * n_cu);
    printf("%s, %d CUs, 2 x 1024-thread workgroups per CU (32 waves/CU)\n", prop.name, n_cu);
    run<16, 0, 0, 0>("16 b32 (+16 perm +20 xor)", out, cyc, n_cu);
    run<32, 0, 0, 0>("32 b32", out, cyc, n_cu);
    run<0, 16, 0, 0>("16 b64", out, cyc, n_cu);
    run<0, 0, 16, 0>("16 b128", out, cyc, n_cu);
    run<0, 0, 32, 0>("32 b128", out, cyc, n_cu);
    run<0, 0, 0, 64>("64 valu", out, cyc, n_cu);
    run<0, 0, 0, 256>("256 valu", out, cyc, n_cu);
    run<16, 0, 0, 32>("16 b32 + 32 valu", out, cyc, n_cu);
    run<16, 0, 0, 64>("16 b32 + 64 valu", out, cyc, n_cu);
    run<16, 0, 0, 128>("16 b32 + 128 valu", out, cyc, n_cu);
    run<16, 0, 8, 0>("16 b32 + 8 b128", out, cyc, n_cu);
    run<16, 16, 0, 0>("16 b32 + 16 b64", out, cyc, n_cu);
    run<16, 0, 8, 64>("16 b32 + 8 b128 + 64 valu", out, cyc, n_cu);
    return 0;
}
