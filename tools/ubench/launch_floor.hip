// What does a launch cost on this stack when the kernel does (almost) nothing?  Back-to-back launches on one stream,
// average time per launch by HIP events: empty kernel / with 23 KB of static LDS / with a 400-byte argument struct /
// 4220 workgroups instead of 1045.  (vd_short's floor is 4.8 us per launch: where does it come from?)
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { double x[50]; int *p; };
__global__ void k_empty(int *p) { if (threadIdx.x == 0 && blockIdx.x == 1 << 30) *p = 1; }
__global__ void k_lds(int *p) {
    __shared__ double s[2900];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (s[255 - threadIdx.x] < 0) *p = 1;
}
__global__ void k_big(Big b) { if (threadIdx.x == 0 && blockIdx.x == 1 << 30) *b.p = int(b.x[3]); }
__global__ void k_store(int8_t *y) { if (threadIdx.x == 0) y[blockIdx.x * 2048] = 1; }
template <class F>
static double timeit(F f, int iters = 2000) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int i = 0; i < 200; ++i) f();
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) f();
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1e3 / iters;
}
int main() {
    int *p;
    int8_t *y;
    (void)hipMalloc(&p, 4);
    (void)hipMalloc(&y, 4220 * 2048);
    Big b{};
    b.p = p;
    for (int rep = 0; rep < 2; ++rep) {
        printf("empty, 1045 x 256:          %.2f us\n", timeit([&] { k_empty<<<1045, 256>>>(p); }));
        printf("empty, 4220 x 256:          %.2f us\n", timeit([&] { k_empty<<<4220, 256>>>(p); }));
        printf("empty, 64 x 256:            %.2f us\n", timeit([&] { k_empty<<<64, 256>>>(p); }));
        printf("23 KB LDS + barrier, 1045:  %.2f us\n", timeit([&] { k_lds<<<1045, 256>>>(p); }));
        printf("400-byte argument, 1045:    %.2f us\n", timeit([&] { k_big<<<1045, 256>>>(b); }));
        printf("one store per WG, 1045:     %.2f us\n", timeit([&] { k_store<<<1045, 256>>>(y); }));
    }
    return 0;
}
