// What overlaps with what across HIP streams on this runtime (round 4, for the batch driver):
//  * kernels on up to 4 streams overlap, beyond that streams share hardware queues;
//  * a download (device -> pinned host) done BY A KERNEL -- ours, or the runtime's own `__amd_rocclr_copyBuffer`, which
//    hipMemcpyAsync turns into under the HIP runtime PyTorch bundles -- starves every other launch of the device for as
//    long as it runs (a tiny kernel issued 30 us into a 90 us download completes when the download does): posted PCIe writes
//    fill the upstream link the command processor fetches its packets through.  The same download through the copy engine
//    (the system runtime's hipMemcpyAsync) starves nothing, but shares ONE engine with uploads: a 4 MB upload and a 4 MB
//    download take 168 us, not 85;
//  * an upload done by a kernel that reads pinned host memory runs beside a copy-engine download at full rate.
// The batch driver therefore cannot overlap a chunk's download with the next chunk's launches under PyTorch's runtime,
// whatever it does on its side (kernel uploads, narrow download kernels and stream priorities were tried: 0.64-0.72 ms
// per decode call against 0.63).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/stream_overlap.hip -o tools/ubench/stream_overlap.bin && tools/ubench/stream_overlap.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(long long cycles, int *sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink && threadIdx.x == 1024) *sink = 1;
}
__global__ void to_host(const uint64_t *src, uint64_t *dst, size_t n) {
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) dst[i] = src[i];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const long long cyc = 10000;  // 100 MHz wall clock: 100 us
    for (int nstreams : {1, 2, 3, 4, 6, 8}) {
        std::vector<hipStream_t> st(nstreams);
        for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        for (auto &s : st) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 10LL, nullptr);
        hipDeviceSynchronize();
        const double t0 = now();
        for (auto &s : st) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, cyc, nullptr);
        hipDeviceSynchronize();
        std::printf("%d streams, one 100 us kernel each: %.0f us\n", nstreams, (now() - t0) * 1e6);
        for (auto &s : st) hipStreamDestroy(s);
    }
    // a kernel that writes pinned host memory on one stream, a spin kernel on another, an H2D copy on a third
    hipStream_t a, b, c;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&c, hipStreamNonBlocking);
    const size_t bytes = 4 << 20;
    uint64_t *d = nullptr, *h = nullptr, *d2 = nullptr, *h2 = nullptr;
    hipMalloc(&d, bytes); hipMalloc(&d2, bytes);
    hipHostMalloc(&h, bytes, hipHostMallocPortable); hipHostMalloc(&h2, bytes, hipHostMallocPortable);
    for (int rep = 0; rep < 3; ++rep) {
        hipDeviceSynchronize();
        double t0 = now();
        hipLaunchKernelGGL(to_host, dim3(64), dim3(256), 0, a, d, h, bytes / 8);
        hipStreamSynchronize(a);
        const double t_copy = (now() - t0) * 1e6;
        t0 = now();
        hipLaunchKernelGGL(to_host, dim3(64), dim3(256), 0, a, d, h, bytes / 8);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, cyc, nullptr);
        hipMemcpyAsync(d2, h2, bytes, hipMemcpyHostToDevice, c);
        hipDeviceSynchronize();
        const double t_all = (now() - t0) * 1e6;
        t0 = now();
        hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, a);
        hipStreamSynchronize(a);
        const double t_memcpy = (now() - t0) * 1e6;
        t0 = now();
        hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, a);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, cyc, nullptr);
        hipMemcpyAsync(d2, h2, bytes, hipMemcpyHostToDevice, c);
        hipDeviceSynchronize();
        const double t_all2 = (now() - t0) * 1e6;
        std::printf("4 MB to host by a 64-workgroup kernel: %.0f us; with a 100 us kernel and a 4 MB upload on other streams: %.0f us | "
                    "by hipMemcpyAsync: %.0f us; with the same company: %.0f us\n", t_copy, t_all, t_memcpy, t_all2);
    }
    // a tiny launch issued WHILE a download saturates the link: when does it complete?
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipDeviceSynchronize();
            const double t0 = now();
            if (mode == 0) hipLaunchKernelGGL(to_host, dim3(64), dim3(256), 0, c, d, h, bytes / 8);
            else if (mode == 1) hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c);
            else hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, c, 9000LL, nullptr);
            while (now() - t0 < 30e-6) {}
            const double t1 = now();
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, 10LL, nullptr);
            hipStreamSynchronize(b);
            const double t2 = now();
            hipStreamSynchronize(c);
            std::printf("%s on one stream; 30 us later a tiny kernel on another: done %.0f us after its launch (the long one: %.0f us)\n",
                        mode == 0 ? "4 MB to host by a kernel" : mode == 1 ? "4 MB to host by hipMemcpyAsync" : "a 90 us spin kernel",
                        (t2 - t1) * 1e6, (now() - t0) * 1e6);
        }
    }
    // when does hipMemcpyAsync to pinned host memory go through the copy engine, when through a shader copy (which starves
    // launches on every other stream for as long as it runs)?
    {
        hipEvent_t ev;
        hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        const char *what[] = {"idle stream", "behind hipStreamWaitEvent on a completed event", "behind a kernel on the same stream",
                              "behind hipStreamWaitEvent on a running kernel of another stream", "while an upload of 4 MB runs on another stream",
                              "behind another 4 MB download on the same stream", "while an upload BY A KERNEL runs on another stream"};
        for (int mode = 0; mode < 7; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                hipDeviceSynchronize();
                if (mode == 1) {
                    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, 10LL, nullptr);
                    hipEventRecord(ev, a);
                    hipStreamSynchronize(a);
                }
                const double t0 = now();
                if (mode == 1) hipStreamWaitEvent(c, ev, 0);
                if (mode == 2) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, c, 500LL, nullptr);
                if (mode == 3) {
                    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, 500LL, nullptr);
                    hipEventRecord(ev, a);
                    hipStreamWaitEvent(c, ev, 0);
                }
                if (mode == 4) hipMemcpyAsync(d2, h2, bytes, hipMemcpyHostToDevice, a);
                if (mode == 5) hipMemcpyAsync(h2, d2, bytes, hipMemcpyDeviceToHost, c);
                if (mode == 6) hipLaunchKernelGGL(to_host, dim3(64), dim3(256), 0, a, h2, d2, bytes / 8);  // (reads pinned host memory)
                hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c);
                while (now() - t0 < 40e-6) {}
                const double t1 = now();
                hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, 10LL, nullptr);
                hipStreamSynchronize(b);
                const double t2 = now();
                hipStreamSynchronize(c);
                std::printf("hipMemcpyAsync 4 MB to host, %s: a tiny kernel launched 40 us in is done after %.0f us (copy done at %.0f us)\n",
                            what[mode], (t2 - t1) * 1e6, (now() - t0) * 1e6);
            }
        }
    }
    // the batch driver's pattern: per chunk an upload (stream a), a kernel behind it (stream b), a download behind that (stream c)
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t up[4], comp[4], done[4];
        for (int i = 0; i < 4; ++i) {
            hipEventCreateWithFlags(&up[i], hipEventDisableTiming);
            hipEventCreateWithFlags(&comp[i], hipEventDisableTiming);
            hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
        }
        for (int rep = 0; rep < 3; ++rep) {
            hipDeviceSynchronize();
            const double t0 = now();
            for (int i = 0; i < 4; ++i) {
                hipMemcpyAsync(d2, h2, bytes / 2, hipMemcpyHostToDevice, a);
                hipEventRecord(up[i], a);
                hipStreamWaitEvent(b, up[i], 0);
                hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, 2000LL, nullptr);  // 20 us
                hipEventRecord(comp[i], b);
                hipStreamWaitEvent(c, comp[i], 0);
                if (mode == 0) hipLaunchKernelGGL(to_host, dim3(64), dim3(256), 0, c, d, h, bytes / 8);
                else hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c);
                hipEventRecord(done[i], c);
            }
            const double t1 = now();
            for (int i = 0; i < 4; ++i) hipEventSynchronize(done[i]);
            std::printf("driver pattern, 4 chunks (2 MB up, 20 us kernel, 4 MB down by %s): issued in %.0f us, done in %.0f us\n",
                        mode ? "hipMemcpyAsync" : "a 64-workgroup kernel", (t1 - t0) * 1e6, (now() - t0) * 1e6);
        }
    }
    return 0;
}
