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
// a kernel whose only job is to write `n` doubles into (pinned host) memory the way the window tiles do: 8 bytes per lane, write-through
__global__ void k_store(double *out, size_t n) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) __hip_atomic_store(reinterpret_cast<unsigned long long *>(out + i), 0x3ff0000000000000ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
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
    // C3's one-shot call in bytes: 19.3 MB up, 16 MB down -- one after the other, both at once by the copy engine(s), and with the
    // download done by a kernel's stores into pinned host memory while the copy engine uploads
    {
        const size_t up = 19300000, down = 16000000;
        auto timeit = [&](auto fn) -> double { fn(); (void)hipDeviceSynchronize(); double t = now(); for (int i = 0; i < 20; ++i) fn(); (void)hipDeviceSynchronize(); return (now() - t) / 20; };
        void *hpd = nullptr;
        CK(hipHostGetDevicePointer(&hpd, hp1, 0));
        const double a = timeit([&] { (void)hipMemcpyAsync(d0, hp0, up, hipMemcpyHostToDevice, s0); return 0; });
        const double b = timeit([&] { (void)hipMemcpyAsync(hp1, d1, down, hipMemcpyDeviceToHost, s1); return 0; });
        const double c = timeit([&] { (void)hipMemcpyAsync(d0, hp0, up, hipMemcpyHostToDevice, s0); (void)hipMemcpyAsync(hp1, d1, down, hipMemcpyDeviceToHost, s1); return 0; });
        const double d = timeit([&] { k_store<<<unsigned((down / 8 + 255) / 256), 256, 0, s1>>>(static_cast<double *>(hpd), down / 8); return 0; });
        const double e = timeit([&] { (void)hipMemcpyAsync(d0, hp0, up, hipMemcpyHostToDevice, s0); k_store<<<unsigned((down / 8 + 255) / 256), 256, 0, s1>>>(static_cast<double *>(hpd), down / 8); return 0; });
        // the same in four chunks each, alternating (what the batch driver issues)
        const double f = timeit([&] { for (int k = 0; k < 4; ++k) { (void)hipMemcpyAsync(d0 + k * (up / 4), hp0 + k * (up / 4), up / 4, hipMemcpyHostToDevice, s0); (void)hipMemcpyAsync(hp1 + k * (down / 4), d1 + k * (down / 4), down / 4, hipMemcpyDeviceToHost, s1); } return 0; });
        printf("C3 wire: H2D 19.3 MB alone %.0f us (%.1f GB/s); D2H 16 MB alone %.0f us (%.1f GB/s); both at once %.0f us; kernel stores 16 MB to host alone %.0f us (%.1f GB/s); "
               "H2D + kernel stores at once %.0f us; both in 4 chunks each %.0f us\n",
               a * 1e6, up / a / 1e9, b * 1e6, down / b / 1e9, c * 1e6, d * 1e6, down / d / 1e9, e * 1e6, f * 1e6);
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
