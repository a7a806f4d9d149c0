// Microbenchmark (round 2): do LDS-lookup waves and pure-VALU waves on the same CU overlap, and what does the VALU
// really sustain?  Roles are assigned per wave so that one launch can mix them:
//   role L : the AES T-table pattern -- 16 conflict-free ds_read_b32 per iteration, addresses from the previous batch
//            (1 v_perm per address, xor tree), exactly the inner loop shape of k_main
//   role V : 256 independent-ish v_bitop3_b32 per iteration on 16 registers (the shape of a bitsliced S-box)
// mode 0: every wave L.  mode 1: every wave V.  mode 2: role by (wave & 1) -> L on SIMDs {0,1}, V on {2,3} (a workgroup's
// waves are dealt to SIMDs 0,2,1,3).  mode 3: role by ((wave >> 2) & 1) -> both roles on every SIMD.
// No overlap at all  => t(mix) = (t(L) + t(V)) / 2;   perfect overlap => t(mix) = max(t(L), t(V)) / 2.
// Also prints the shader clock each body sustains (s_memtime ticks / s_memrealtime 100 MHz ticks).
//   hipcc --offload-arch=gfx950 -O3 -o overlap overlap.hip && ./overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32;
#define LDS32(off) (*(const __attribute__((address_space(3))) u32 *)(uintptr_t)(off))

template <int WPS>   // waves per SIMD the launch bound asks for (register budget)
__global__ __launch_bounds__(WPS * 128, WPS) void k(u32 *out, int iters_l, int iters_v, int mode, unsigned long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (u32 i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((u32 *)smem)[i] = i * 2654435761u;
    __syncthreads();
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lb = (lane & 31) << 2;
    const bool role_v = mode == 1 || (mode == 2 && (wave & 1)) || (mode == 3 && ((wave >> 2) & 1));
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    u32 res = 0;
    if (!role_v) {
        u32 s[4] = {lane * 7 + 1, lane * 13 + 5, lane * 29 + 3, lane + 11};
        for (int it = 0; it < iters_l; it++) {
            u32 acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const u32 addr = __builtin_amdgcn_perm(s[j & 3], lb, 0x0c0c0000u | ((4u + ((j >> 2) & 3)) << 8));
                acc[j & 3] ^= LDS32(addr);
            }
            s[0] ^= acc[0]; s[1] ^= acc[1]; s[2] ^= acc[2]; s[3] ^= acc[3];
        }
        res = s[0] ^ s[1] ^ s[2] ^ s[3];
    } else {
        u32 r[16];
#pragma unroll
        for (int i = 0; i < 16; i++) r[i] = lane * (2 * i + 3) + i;
        for (int it = 0; it < iters_v; it++) {
#pragma unroll
            for (int j = 0; j < 256; j += 2) {
                r[j & 15] = __builtin_amdgcn_bitop3_b32(r[(j + 1) & 15], r[(j + 5) & 15], r[(j + 10) & 15], 0x96);
                r[(j + 1) & 15] = __builtin_amdgcn_bitop3_b32(r[(j + 2) & 15], r[(j + 6) & 15], r[(j + 11) & 15], 0x78);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; i++) res ^= r[i];
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    if (lane == 0) { cyc[2 * (blockIdx.x * (blockDim.x >> 6) + wave)] = t1 - t0; cyc[2 * (blockIdx.x * (blockDim.x >> 6) + wave) + 1] = w1 - w0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
}

template <int WPS>
static float run(const char *name, int mode, int il, int iv, u32 *out, unsigned long long *cyc, int n_cu) {
    const int wgs = 2 * n_cu, threads = WPS * 128, waves = wgs * threads / 64;
    hipFuncSetAttribute((const void *)&k<WPS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<WPS>), dim3(wgs), dim3(threads), 65536, 0, out, il / 50 + 1, iv / 50 + 1, mode, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<WPS>), dim3(wgs), dim3(threads), 65536, 0, out, il, iv, mode, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[2 * 8192 * 2];
    hipMemcpy(h, cyc, 16 * waves, hipMemcpyDeviceToHost);
    double cs = 0, ws = 0; for (int i = 0; i < waves; i++) { cs += (double)h[2 * i]; ws += (double)h[2 * i + 1]; }
    printf("%-44s %8.3f ms   shader clock %5.0f MHz (s_memtime / s_memrealtime@100MHz)\n", name, ms, cs / ws * 100.0);
    return ms;
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    u32 *out; unsigned long long *cyc;
    hipMalloc(&out, 4 * 1024 * 2 * n_cu); hipMalloc(&cyc, 16 * 16 * 2 * n_cu);
    printf("%s, %d CUs, 2 workgroups per CU\n", prop.name, n_cu);
    const int IL = 4000, IV = 1000;
    {   // 8 waves per SIMD (the product kernel's occupancy)
        float tl = run<8>("8 w/SIMD: all L (4000 x 16 lookups)", 0, IL, IV, out, cyc, n_cu);
        float tv = run<8>("8 w/SIMD: all V (1000 x 256 bitop3)", 1, IL, IV, out, cyc, n_cu);
        float t2 = run<8>("8 w/SIMD: L on SIMD{0,1}, V on SIMD{2,3}", 2, IL, IV, out, cyc, n_cu);
        float t3 = run<8>("8 w/SIMD: L and V on every SIMD", 3, IL, IV, out, cyc, n_cu);
        printf("   no-overlap prediction %.3f ms, perfect-overlap prediction %.3f ms; measured split-SIMD %.3f, shared-SIMD %.3f\n",
               (tl + tv) / 2, (tl > tv ? tl : tv) / 2, t2, t3);
        const double lk = 2.0 * n_cu * 16 * IL * 16.0, vo = 2.0 * n_cu * 16 * IV * 256.0;      // wave-instructions, whole chip
        printf("   L: %.3f ns per wave-lookup per CU (perm + read + xor share);  V: %.3f ns per wave-bitop3 per CU = %.2f wave-instr/ns/CU\n",
               tl * 1e6 / (lk / n_cu), tv * 1e6 / (vo / n_cu), (vo / n_cu) / (tv * 1e6));
    }
    // VALU alone at lower occupancy (a bitsliced kernel needs ~200 VGPRs: 2 waves per SIMD)
    { float t = run<4>("4 w/SIMD: all V", 1, IL, IV, out, cyc, n_cu); printf("   %.2f wave-instr/ns/CU\n", 2.0 * 8 * IV * 256.0 / (t * 1e6)); }
    { float t = run<2>("2 w/SIMD: all V", 1, IL, IV, out, cyc, n_cu); printf("   %.2f wave-instr/ns/CU\n", 2.0 * 4 * IV * 256.0 / (t * 1e6)); }
    { float t = run<1>("1 w/SIMD: all V", 1, IL, IV, out, cyc, n_cu); printf("   %.2f wave-instr/ns/CU\n", 2.0 * 2 * IV * 256.0 / (t * 1e6)); }
    return 0;
}
