/* What the runtime calls of aesgcm_ctx_create / aesgcm_ctx_destroy cost on this box (microseconds, median of 20):  hipcc -O2 runtime_costs.cpp -o runtime_costs */
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class F> static double med(F f) { std::vector<double> t; for (int i = 0; i < 20; i++) { double a = now(); f(); t.push_back(now() - a); } std::sort(t.begin(), t.end()); return t[10]; }
int main() {
    (void)hipFree(0);
    void *p = nullptr; hipStream_t s; hipEvent_t e;
    printf("hipMalloc 64 KiB        %8.1f us\n", med([&] { (void)hipMalloc(&p, 65536); }));
    printf("hipFree                 %8.1f us\n", med([&] { (void)hipMalloc(&p, 65536); double a = now(); (void)a; (void)hipFree(p); }));
    std::vector<void *> v;
    double tm = med([&] { (void)hipMalloc(&p, 4 << 20); v.push_back(p); });
    printf("hipMalloc 4 MiB         %8.1f us\n", tm);
    printf("hipFree 4 MiB           %8.1f us\n", med([&] { (void)hipFree(v.back()); v.pop_back(); }));
    printf("hipHostMalloc 64 B map  %8.1f us\n", med([&] { (void)hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocCoherent); }));
    printf("hipHostFree             %8.1f us\n", med([&] { (void)hipHostMalloc(&p, 64, hipHostMallocMapped); (void)hipHostFree(p); }));
    printf("hipStreamCreate         %8.1f us\n", med([&] { (void)hipStreamCreate(&s); }));
    printf("hipStreamDestroy        %8.1f us\n", med([&] { (void)hipStreamCreate(&s); (void)hipStreamDestroy(s); }));
    printf("hipEventCreate          %8.1f us\n", med([&] { (void)hipEventCreate(&e); }));
    (void)hipMalloc(&p, 65536);
    printf("hipMemset 64 KiB (sync) %8.1f us\n", med([&] { (void)hipMemset(p, 0, 65536); }));
    (void)hipStreamCreate(&s);
    printf("hipMemsetAsync + sync   %8.1f us\n", med([&] { (void)hipMemsetAsync(p, 0, 256, s); (void)hipStreamSynchronize(s); }));
    return 0;
}
