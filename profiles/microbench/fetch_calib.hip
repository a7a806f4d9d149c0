// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on a KNOWN byte count in k_main's and k_body's access patterns
// (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern").  Both kernels copy the
// same 4 GiB with 16 bytes per lane and 1 KiB per wave instruction; they differ only in which rows a wave takes:
//   k_rows_contiguous  a wave walks T consecutive rows (k_main's chunk)
//   k_rows_phase       a wave walks rows 4q + v, q = sT .. sT+T-1 (k_body's chunk: every fourth row)
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE -d out -- ./fetch_calib ; rocprofv3 --pmc WRITE_SIZE -d out2 -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned long long u64;
#define T 64
__global__ __launch_bounds__(1024) void k_rows_contiguous(const uint4 *__restrict__ src, uint4 *__restrict__ dst, u64 chunks) {
    const u64 wave = (u64)blockIdx.x * 16 + (threadIdx.x >> 6), nw = (u64)gridDim.x * 16, lane = threadIdx.x & 63;
    for (u64 c = wave; c < chunks; c += nw)
        for (u64 i = 0; i < T; i++) { const u64 row = c * T + i; dst[row * 64 + lane] = src[row * 64 + lane]; }
}
__global__ __launch_bounds__(1024) void k_rows_phase(const uint4 *__restrict__ src, uint4 *__restrict__ dst, u64 chunks) {
    const u64 wave = (u64)blockIdx.x * 16 + (threadIdx.x >> 6), nw = (u64)gridDim.x * 16, lane = threadIdx.x & 63;
    for (u64 c = wave; c < chunks; c += nw) {
        const u64 s = c >> 2, v = c & 3;
        for (u64 i = 0; i < T; i++) { const u64 row = 4 * (s * T + i) + v; dst[row * 64 + lane] = src[row * 64 + lane]; }
    }
}
// the same two patterns at the fused kernels' pace (one row per ~8 us per wave) and with k_body's 2 x 1024-lane workgroups
// per CU: do requests change size when the rows of a neighbourhood are touched far apart in time?
template <int PHASE>
__global__ __launch_bounds__(1024) void k_rows_slow(const uint4 *__restrict__ src, uint4 *__restrict__ dst, u64 chunks) {
    const u64 wave = (u64)blockIdx.x * 16 + (threadIdx.x >> 6), nw = (u64)gridDim.x * 16, lane = threadIdx.x & 63;
    for (u64 c = wave; c < chunks; c += nw) {
        const u64 s = c >> 2, v = c & 3;
        for (u64 i = 0; i < T; i++) {
            const u64 row = PHASE ? 4 * (s * T + i) + v : c * T + i;
            const uint4 x = src[row * 64 + lane];
            __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127);
            dst[row * 64 + lane] = x;
        }
    }
}
int main() {
    const u64 bytes = 4ull << 30, rows = bytes / 1024, chunks = rows / T;
    uint4 *a, *b;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(a, 0x5a, bytes); hipMemset(b, 0, bytes);
    for (int it = 0; it < 3; it++) {
        hipLaunchKernelGGL(k_rows_contiguous, dim3(512), dim3(1024), 0, 0, a, b, chunks);
        hipLaunchKernelGGL(k_rows_phase, dim3(512), dim3(1024), 0, 0, a, b, chunks);
    }
    hipLaunchKernelGGL(k_rows_slow<0>, dim3(512), dim3(1024), 0, 0, a, b, chunks);
    hipLaunchKernelGGL(k_rows_slow<1>, dim3(512), dim3(1024), 0, 0, a, b, chunks);
    hipDeviceSynchronize();
    printf("copied %llu bytes per launch (read) + the same written\n", bytes);
    return 0;
}
