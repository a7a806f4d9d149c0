/* What the PCIe link gives the pipelined host path (aesgcm_encrypt_pipelined): page-locked host memory to device and back, one direction alone and both at
 * once on two streams, per transfer size.  hipcc -O2 pcie_probe.cpp -o pcie_probe && ./pcie_probe       (GPU box; prints GiB/s)                          */
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t total = (size_t)1 << 30;
    void *h_in, *h_out, *d_a, *d_b;
    HIP(hipHostMalloc(&h_in, total, hipHostMallocDefault)); HIP(hipHostMalloc(&h_out, total, hipHostMallocDefault));
    HIP(hipMalloc(&d_a, total)); HIP(hipMalloc(&d_b, total));
    hipStream_t s1, s2;
    HIP(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); HIP(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    printf("%10s %12s %12s %24s\n", "chunk MiB", "H2D alone", "D2H alone", "both at once (each way)");
    for (size_t chunk = (size_t)1 << 20; chunk <= total; chunk <<= 2) {
        const size_t n = total / chunk;
        double t[3];
        for (int mode = 0; mode < 3; mode++) {
            HIP(hipDeviceSynchronize());
            const double t0 = now();
            for (size_t k = 0; k < n; k++) {
                if (mode != 1) HIP(hipMemcpyAsync((char *)d_a + k * chunk, (char *)h_in + k * chunk, chunk, hipMemcpyHostToDevice, s1));
                if (mode != 0) HIP(hipMemcpyAsync((char *)h_out + k * chunk, (char *)d_b + k * chunk, chunk, hipMemcpyDeviceToHost, s2));
            }
            HIP(hipStreamSynchronize(s1)); HIP(hipStreamSynchronize(s2));
            t[mode] = now() - t0;
        }
        printf("%10zu %12.1f %12.1f %24.1f\n", chunk >> 20, 1.0 / t[0], 1.0 / t[1], 1.0 / t[2]);
    }
    return 0;
}
