// Host<->device copy rates the batch driver is designed against: pinned vs pageable, one direction
// and both at once (two streams), in the chunk sizes the driver uses.  hipcc --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t N = 64u << 20;
    char *d0, *d1, *hp0, *hp1;
    CK(hipMalloc((void **)&d0, N)); CK(hipMalloc((void **)&d1, N));
    CK(hipHostMalloc((void **)&hp0, N, hipHostMallocDefault)); CK(hipHostMalloc((void **)&hp1, N, hipHostMallocDefault));
    std::vector<char> pg(N, 1);
    memset(hp0, 1, N); memset(hp1, 2, N);
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    for (size_t sz : {size_t(64) << 10, size_t(256) << 10, size_t(1) << 20, size_t(4) << 20, size_t(16) << 20, size_t(64) << 20}) {
        const int reps = sz >= (16u << 20) ? 10 : 50;
        auto timeit = [&](auto fn) -> double { fn(); (void)hipDeviceSynchronize(); double t = now(); for (int i = 0; i < reps; ++i) fn(); (void)hipDeviceSynchronize(); return (now() - t) / reps; };
        double h2d = timeit([&] { (void)hipMemcpyAsync(d0, hp0, sz, hipMemcpyHostToDevice, s0); return 0; });
        double d2h = timeit([&] { (void)hipMemcpyAsync(hp1, d1, sz, hipMemcpyDeviceToHost, s1); return 0; });
        double both = timeit([&] { (void)hipMemcpyAsync(d0, hp0, sz, hipMemcpyHostToDevice, s0); (void)hipMemcpyAsync(hp1, d1, sz, hipMemcpyDeviceToHost, s1); return 0; });
        double pgh2d = timeit([&] { (void)hipMemcpy(d0, pg.data(), sz, hipMemcpyHostToDevice); return 0; });
        double pgd2h = timeit([&] { (void)hipMemcpy(pg.data(), d1, sz, hipMemcpyDeviceToHost); return 0; });
        printf("%8zu KiB  pinned H2D %6.1f GB/s (%7.1f us)  D2H %6.1f GB/s  both %6.1f+%6.1f GB/s  pageable H2D %5.1f D2H %5.1f GB/s\n", sz >> 10,
               sz / h2d / 1e9, h2d * 1e6, sz / d2h / 1e9, sz / both / 1e9, sz / both / 1e9, sz / pgh2d / 1e9, sz / pgd2h / 1e9);
    }
    // host memcpy rate into pinned memory (staging of pageable caller buffers), one thread
    { double t = now(); for (int i = 0; i < 5; ++i) memcpy(hp0, pg.data(), N); double dt = (now() - t) / 5; printf("host memcpy pageable->pinned %.1f GB/s\n", N / dt / 1e9); }
    // hipHostRegister cost
    { std::vector<char> big(32u << 20, 3); double t = now(); hipError_t e = hipHostRegister(big.data(), big.size(), hipHostRegisterDefault); double dt = now() - t;
      printf("hipHostRegister 32 MiB: %s, %.2f ms\n", hipGetErrorString(e), dt * 1e3);
      if (e == hipSuccess) { t = now(); (void)hipHostUnregister(big.data()); printf("hipHostUnregister: %.2f ms\n", (now() - t) * 1e3); } }
    // launch + sync latency, event record cost
    { hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); double t = now(); for (int i = 0; i < 1000; ++i) { (void)hipEventRecord(ev, s0); } CK(hipDeviceSynchronize()); printf("hipEventRecord %.2f us each\n", (now() - t) * 1e3);
      t = now(); for (int i = 0; i < 200; ++i) { (void)hipMemcpyAsync(d0, hp0, 4096, hipMemcpyHostToDevice, s0); (void)hipStreamSynchronize(s0); } printf("4 KiB H2D + sync %.2f us\n", (now() - t) / 200 * 1e6);
      t = now(); for (int i = 0; i < 50; ++i) { void *p; (void)hipMalloc(&p, 8u << 20); (void)hipFree(p); } printf("hipMalloc+hipFree 8 MiB %.1f us\n", (now() - t) / 50 * 1e6);
      t = now(); for (int i = 0; i < 20; ++i) { void *p; (void)hipHostMalloc(&p, 8u << 20, 0); (void)hipHostFree(p); } printf("hipHostMalloc+Free 8 MiB %.1f us\n", (now() - t) / 20 * 1e6); }
    return 0;
}
