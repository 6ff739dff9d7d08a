// What does a launch cost on this stack when the kernel does (almost) nothing?  Back-to-back launches on one stream,
// average time per launch by HIP events: empty kernel / with 23 KB of static LDS / with a 400-byte argument struct /
// 4220 workgroups instead of 1045.  (vd_short's floor is 4.8 us per launch: where does it come from?)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
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
// latency form: one launch, then wait for it -- what a synchronous one-shot call cannot go below
template <class F>
static void sync_latency(const char *what, F f, int iters = 2000) {
    hipStream_t st;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t ev;
    (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    std::vector<double> a(iters), b(iters);
    for (int pass = 0; pass < 2; ++pass)
        for (int i = 0; i < iters; ++i) {
            auto t0 = std::chrono::steady_clock::now();
            f(st);
            if (pass == 0) {
                (void)hipStreamSynchronize(st);
            } else {
                (void)hipEventRecord(ev, st);
                (void)hipEventSynchronize(ev);
            }
            (pass ? b : a)[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        }
    std::sort(a.begin(), a.end());
    std::sort(b.begin(), b.end());
    printf("%-38s launch + hipStreamSynchronize: median %.2f us (p10 %.2f, p90 %.2f); launch + event record + hipEventSynchronize: median %.2f us\n",
           what, a[iters / 2], a[iters / 10], a[iters * 9 / 10], b[iters / 2]);
    (void)hipEventDestroy(ev);
    (void)hipStreamDestroy(st);
}
__global__ void k_host_rw(const int *in, double *out) { out[threadIdx.x] = double(in[threadIdx.x]) + 1.0; }
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
    sync_latency("empty kernel, 1 x 256:", [&](hipStream_t st) { k_empty<<<1, 256, 0, st>>>(p); });
    sync_latency("empty kernel, 1045 x 256:", [&](hipStream_t st) { k_empty<<<1045, 256, 0, st>>>(p); });
    {   // a kernel that reads its input from and writes its output to pinned host memory (no copy commands at all)
        int *hin;
        double *hout;
        (void)hipHostMalloc(&hin, 1024, hipHostMallocMapped);
        (void)hipHostMalloc(&hout, 2048, hipHostMallocMapped);
        for (int i = 0; i < 256; ++i) hin[i] = i;
        sync_latency("256 loads + stores in pinned host memory:", [&](hipStream_t st) { k_host_rw<<<1, 256, 0, st>>>(hin, hout); });
        // the same result by copy commands: H2D copy, kernel on device memory, D2H copy, one stream
        int *din;
        double *dout;
        (void)hipMalloc(&din, 1024);
        (void)hipMalloc(&dout, 2048);
        sync_latency("H2D 1 KB + kernel + D2H 2 KB, one stream:", [&](hipStream_t st) {
            (void)hipMemcpyAsync(din, hin, 1024, hipMemcpyHostToDevice, st);
            k_host_rw<<<1, 256, 0, st>>>(din, dout);
            (void)hipMemcpyAsync(hout, dout, 2048, hipMemcpyDeviceToHost, st);
        });
    }
    return 0;
}
