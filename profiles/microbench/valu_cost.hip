// Microbenchmark: throughput of the individual VALU instructions the AES-GCM kernel uses, at 32 waves/CU.
//   hipcc --offload-arch=gfx950 -O3 -o valu_cost valu_cost.hip && ./valu_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32;
enum { OP_XOR, OP_BITOP3, OP_PERM, OP_ALIGNBIT, OP_ANDOR, OP_LSHR_AND, OP_BFE, OP_ADD, OP_MOV_SGPRXOR, OP_SDWA_MOV, OP_MOV, OP_ROT };
template <int OP>
__device__ __forceinline__ u32 op(u32 a, u32 b, u32 c, u32 sk) {
    if (OP == OP_XOR) return a ^ b;
    if (OP == OP_BITOP3) return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
    if (OP == OP_PERM) return __builtin_amdgcn_perm(a, b, 0x0c0c0500u);
    if (OP == OP_ALIGNBIT) return __builtin_amdgcn_alignbit(a, a, 24) ^ 0;   // rotate
    if (OP == OP_ANDOR) return (a & 0xF0u) | b;
    if (OP == OP_LSHR_AND) return (a >> 12) & 0xF0u;                          // 2 instructions
    if (OP == OP_BFE) return __builtin_amdgcn_ubfe(a, 8, 8);
    if (OP == OP_ADD) return a + b;
    if (OP == OP_SDWA_MOV) { u32 d = a; asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(d) : "v"(b)); return d; }
    if (OP == OP_MOV) { u32 d; asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"(a)); return d; }
    if (OP == OP_ROT) { u32 d; asm volatile("v_alignbit_b32 %0, %1, %1, 24" : "=v"(d) : "v"(a)); return d; }
    return a ^ sk;                                                            // xor with a scalar operand
}
template <int OP>
__global__ __launch_bounds__(1024, 8) void k(u32 *out, int iters, unsigned long long *cycles, u32 sk) {
    const u32 lane = threadIdx.x;
    u32 r[8];
    for (int i = 0; i < 8; i++) r[i] = lane * (2 * i + 3) + i;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 128; j++) {
            const int d = j & 7;
            r[d] = op<OP>(r[(d + 1) & 7], r[(d + 3) & 7], r[(d + 5) & 7], sk);   // 8 independent chains
        }
    }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    u32 x = 0; for (int i = 0; i < 8; i++) x ^= r[i];
    out[blockIdx.x * 1024 + threadIdx.x] = x;
}
template <int OP>
static void run(const char *name, u32 *out, unsigned long long *cyc, int n_cu, int n_instr) {
    const int iters = 20000, wgs = 2 * n_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP>), dim3(wgs), dim3(1024), 0, 0, out, 100, cyc, 0x12345u);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP>), dim3(wgs), dim3(1024), 0, 0, out, iters, cyc, 0x12345u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[1024]; hipMemcpy(h, cyc, sizeof(unsigned long long) * wgs, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < wgs; i++) avg += (double)h[i]; avg /= wgs;
    const double per = avg / iters / 32.0 / (128.0 * n_instr);
    printf("%-24s %8.2f ms  clock %4.0f MHz  %.3f CU-cycles per wave-instruction  (%.1f wave-instr/clk/CU)\n", name, ms, avg / (ms * 1e3), per, 1.0 / per);
}
int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    u32 *out; unsigned long long *cyc;
    hipMalloc(&out, 4 * 1024 * 2 * n_cu); hipMalloc(&cyc, 8 * 2 * n_cu);
    printf("%d CUs, 32 waves/CU, 8 independent dependency chains per lane\n", n_cu);
    run<OP_XOR>("v_xor_b32", out, cyc, n_cu, 1);
    run<OP_BITOP3>("v_bitop3_b32", out, cyc, n_cu, 1);
    run<OP_PERM>("v_perm_b32", out, cyc, n_cu, 1);
    run<OP_ALIGNBIT>("v_alignbit_b32", out, cyc, n_cu, 1);
    run<OP_ANDOR>("v_and_or_b32", out, cyc, n_cu, 1);
    run<OP_LSHR_AND>("v_lshrrev+v_and", out, cyc, n_cu, 2);
    run<OP_BFE>("v_bfe_u32", out, cyc, n_cu, 1);
    run<OP_ADD>("v_add_u32", out, cyc, n_cu, 1);
    run<OP_SDWA_MOV>("v_mov_b32_sdwa byte->byte", out, cyc, n_cu, 1);
    run<OP_MOV>("v_mov_b32", out, cyc, n_cu, 1);
    run<OP_ROT>("v_alignbit_b32 (rotate)", out, cyc, n_cu, 1);
    return 0;
}
