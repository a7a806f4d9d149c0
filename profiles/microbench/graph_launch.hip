// Microbenchmark (round 2): is a pre-instantiated two-node hipGraph a faster way to issue the two dependent launches of a
// small message (k_main + k_combine) than two plain launches?  Both variants end the same way the library
// does: the second kernel stores a generation number into pinned host memory and the host polls it.
//   (a) two hipLaunchKernelGGL on a stream
//   (b) hipGraphLaunch of a graph with the two kernel nodes; parameters unchanged between launches (best case)
//   (c) the same, with hipGraphExecKernelNodeSetParams on both nodes before every launch (what a library call would need:
//       pointers, lengths and the IV differ per message)
//   hipcc --offload-arch=gfx950 -O3 -o graph_launch graph_launch.hip && ./graph_launch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <time.h>
#include <algorithm>
#include <vector>

__global__ void k_a(uint32_t *buf, uint32_t v) { if (threadIdx.x == 0 && blockIdx.x == 0) buf[0] = v; }
__global__ void k_b(const uint32_t *buf, volatile uint32_t *host, uint32_t gen) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { __threadfence_system(); host[0] = buf[0] + gen; __threadfence_system(); host[1] = gen; }
}
static double now_us() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }
static void wait_gen(volatile uint32_t *h, uint32_t gen) { while (h[1] != gen) __builtin_ia32_pause(); }
static void report(const char *name, std::vector<double> &v) {
    std::sort(v.begin(), v.end());
    printf("%-64s median %6.2f us   best %6.2f   p90 %6.2f\n", name, v[v.size() / 2], v[0], v[v.size() * 9 / 10]);
}

int main() {
    hipStream_t st; hipStreamCreate(&st);
    uint32_t *d; hipMalloc(&d, 64);
    uint32_t *h; hipHostMalloc(&h, 64, hipHostMallocMapped); h[0] = h[1] = 0;
    uint32_t *hd; hipHostGetDevicePointer((void **)&hd, h, 0);
    const int N = 2000;
    uint32_t gen = 0;
    std::vector<double> t;
    for (int i = 0; i < N + 200; i++) {                       // (a)
        const double t0 = now_us();
        ++gen;
        hipLaunchKernelGGL(k_a, dim3(64), dim3(256), 0, st, d, gen);
        hipLaunchKernelGGL(k_b, dim3(1), dim3(256), 0, st, d, hd, gen);
        wait_gen(h, gen);
        if (i >= 200) t.push_back(now_us() - t0);
    }
    report("(a) two plain launches + host poll", t);

    hipGraph_t g; hipGraphCreate(&g, 0);
    uint32_t va = 0, vb = 0;
    void *pa[] = {&d, &va}, *pb[] = {&d, &hd, &vb};
    hipKernelNodeParams na = {}, nb = {};
    na.func = (void *)k_a; na.gridDim = dim3(64); na.blockDim = dim3(256); na.kernelParams = pa;
    nb.func = (void *)k_b; nb.gridDim = dim3(1); nb.blockDim = dim3(256); nb.kernelParams = pb;
    hipGraphNode_t a, b;
    if (hipGraphAddKernelNode(&a, g, nullptr, 0, &na) != hipSuccess || hipGraphAddKernelNode(&b, g, &a, 1, &nb) != hipSuccess) { printf("graph build failed\n"); return 1; }
    hipGraphExec_t ge;
    if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("instantiate failed\n"); return 1; }
    t.clear();
    for (int i = 0; i < N + 200; i++) {                       // (b): k_b's gen stays 0 -> poll the data word instead
        h[0] = 0xFFFFFFFFu;
        const double t0 = now_us();
        hipGraphLaunch(ge, st);
        while (((volatile uint32_t *)h)[0] == 0xFFFFFFFFu) __builtin_ia32_pause();
        if (i >= 200) t.push_back(now_us() - t0);
        hipStreamSynchronize(st);
    }
    report("(b) hipGraphLaunch, parameters fixed + host poll", t);
    t.clear();
    for (int i = 0; i < N + 200; i++) {                       // (c)
        const double t0 = now_us();
        ++gen; va = gen; vb = gen;
        hipGraphExecKernelNodeSetParams(ge, a, &na);
        hipGraphExecKernelNodeSetParams(ge, b, &nb);
        hipGraphLaunch(ge, st);
        wait_gen(h, gen);
        if (i >= 200) t.push_back(now_us() - t0);
    }
    report("(c) SetParams on both nodes + hipGraphLaunch + host poll", t);
    return 0;
}
