// Which submission pattern overlaps upload / kernel / download of consecutive chunks on this runtime?
// Emulates the batch driver's data flow (9.6 B/gene in, 8 B/gene out, 2 M genes) with a trivial kernel.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void work(const int *__restrict__ in, double *__restrict__ out, int n) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const int *p = in + size_t(g) * 12 / 5;  // ~2.4 ints per gene
    double s = double(p[0]) + double(p[1]);
    for (int k = 0; k < 40; ++k) s = s * 1.0000001 + 0.5;
    out[g] = s;
}
__global__ void copy16(const int4 *__restrict__ src, int4 *__restrict__ dst, size_t n16) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += size_t(gridDim.x) * blockDim.x) dst[i] = src[i];
}
int main(int argc, char **argv) {
    const int N = 2000000, IN_INTS = N * 12 / 5 + 8;
    int *h_in; double *h_out;
    hipHostMalloc((void **)&h_in, size_t(IN_INTS) * 4, hipHostMallocPortable);
    hipHostMalloc((void **)&h_out, size_t(N) * 8, hipHostMallocPortable);
    for (int i = 0; i < IN_INTS; ++i) h_in[i] = i & 1023;
    int *d_in[4]; double *d_out[4]; hipStream_t st[4], su, sd; hipEvent_t ev[4], evu[8], evk[8];
    for (int i = 0; i < 4; ++i) { hipMalloc((void **)&d_in[i], size_t(IN_INTS) * 4); hipMalloc((void **)&d_out[i], size_t(N) * 8); hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking); hipEventCreateWithFlags(&ev[i], hipEventDisableTiming); }
    hipStreamCreateWithFlags(&su, hipStreamNonBlocking); hipStreamCreateWithFlags(&sd, hipStreamNonBlocking);
    for (int i = 0; i < 8; ++i) { hipEventCreateWithFlags(&evu[i], hipEventDisableTiming); hipEventCreateWithFlags(&evk[i], hipEventDisableTiming); }
    for (int chunks : {1, 2, 4, 8}) {
        const int per = N / chunks;
        auto U = [&](int k, hipStream_t s, int slot) { hipMemcpyAsync(d_in[slot], h_in + size_t(k) * per * 12 / 5, size_t(per) * 12 / 5 * 4 + 16, hipMemcpyHostToDevice, s); };
        auto K = [&](int k, hipStream_t s, int slot, double *dst) { hipLaunchKernelGGL(work, dim3((per + 255) / 256), dim3(256), 0, s, d_in[slot], dst, per); };
        auto D = [&](int k, hipStream_t s, int slot) { hipMemcpyAsync(h_out + size_t(k) * per, d_out[slot], size_t(per) * 8, hipMemcpyDeviceToHost, s); };
        for (int mode = 0; mode < 9; ++mode) {
            double best = 1e9;
            for (int rep = 0; rep < 8; ++rep) {
                hipDeviceSynchronize();
                double t = now();
                if (mode == 0) {  // depth-first, one stream per chunk (ring of 3)
                    for (int k = 0; k < chunks; ++k) { int s = k % 3; if (k >= 3) hipEventSynchronize(ev[s]); U(k, st[s], s); K(k, st[s], s, d_out[s]); D(k, st[s], s); hipEventRecord(ev[s], st[s]); }
                } else if (mode == 1) {  // staggered: upload+kernel of k+1 is enqueued before the download of k
                    for (int k = 0; k <= chunks; ++k) {
                        if (k < chunks) { int s = k % 4; if (k >= 4) hipEventSynchronize(ev[s]); U(k, st[s], s); K(k, st[s], s, d_out[s]); }
                        if (k >= 1) { int s = (k - 1) % 4; D(k - 1, st[s], s); hipEventRecord(ev[s], st[s]); }
                    }
                } else if (mode == 2) {  // one stream per direction + compute, events in between
                    for (int k = 0; k < chunks; ++k) { int s = k % 4; if (k >= 4) hipEventSynchronize(ev[s]); U(k, su, s); hipEventRecord(evu[k % 8], su); hipStreamWaitEvent(st[0], evu[k % 8], 0); K(k, st[0], s, d_out[s]); hipEventRecord(evk[k % 8], st[0]); hipStreamWaitEvent(sd, evk[k % 8], 0); D(k, sd, s); hipEventRecord(ev[s], sd); }
                } else if (mode == 3) {  // kernel stores straight into pinned host memory (no download copies)
                    for (int k = 0; k < chunks; ++k) { int s = k % 3; if (k >= 3) hipEventSynchronize(ev[s]); U(k, st[s], s); K(k, st[s], s, h_out + size_t(k) * per); hipEventRecord(ev[s], st[s]); }
                } else if (mode >= 5) {
                    // 5: uploads by a copy kernel (stream su), compute (st[0]), downloads by SDMA (sd)
                    // 6: uploads by SDMA, downloads by a copy kernel; 7: both by copy kernels; 8: like 2 with two upload streams
                    auto KU = [&](int k, hipStream_t s, int slot) { size_t n16 = (size_t(per) * 12 / 5 * 4 + 16) / 16; hipLaunchKernelGGL(copy16, dim3(128), dim3(256), 0, s, (const int4 *)(h_in + size_t(k) * per * 12 / 5 / 4 * 4), (int4 *)d_in[slot], n16); };
                    auto KD = [&](int k, hipStream_t s, int slot) { size_t n16 = size_t(per) * 8 / 16; hipLaunchKernelGGL(copy16, dim3(128), dim3(256), 0, s, (const int4 *)d_out[slot], (int4 *)(h_out + size_t(k) * per), n16); };
                    for (int k = 0; k < chunks; ++k) {
                        int s = k % 4;
                        if (k >= 4) hipEventSynchronize(ev[s]);
                        hipStream_t up = (mode == 8 && (k & 1)) ? st[1] : su;
                        if (mode == 5 || mode == 7) KU(k, up, s); else U(k, up, s);
                        hipEventRecord(evu[k % 8], up); hipStreamWaitEvent(st[0], evu[k % 8], 0);
                        K(k, st[0], s, d_out[s]);
                        hipEventRecord(evk[k % 8], st[0]); hipStreamWaitEvent(sd, evk[k % 8], 0);
                        if (mode == 6 || mode == 7) KD(k, sd, s); else D(k, sd, s);
                        hipEventRecord(ev[s], sd);
                    }
                } else {  // kernel reads pinned host memory and stores into it: no copies at all
                    for (int k = 0; k < chunks; ++k) { int s = k % 3; hipLaunchKernelGGL(work, dim3((per + 255) / 256), dim3(256), 0, st[s], h_in + size_t(k) * per * 12 / 5, h_out + size_t(k) * per, per); }
                }
                hipDeviceSynchronize();
                double dt = now() - t;
                if (dt < best) best = dt;
            }
            const char *names[] = {"depth-first 3 streams", "staggered U(k+1) before D(k)", "stream per direction", "kernel writes host", "kernel reads+writes host", "copy-kernel up, SDMA down", "SDMA up, copy-kernel down", "copy kernels both ways", "2 upload streams, SDMA"};
            printf("chunks %d  %-30s %7.1f us  (%.2f G genes/s)\n", chunks, names[mode], best * 1e6, N / best / 1e9);
        }
    }
    return 0;
}
