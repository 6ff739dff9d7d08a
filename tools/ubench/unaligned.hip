// Does gfx950 under ROCm take 8-byte global stores / loads at byte-granular addresses?  (vd_short would like to
// write a lane's 8 label bytes with ONE store wherever the workgroup's first gene happens to fall.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

__global__ void st8(uint8_t *base, int off) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    uint64_t v = 0;
    for (int k = 0; k < 8; ++k) v |= uint64_t((t * 8 + k) & 0xff) << (8 * k);
    uint8_t *p = base + off + t * 8;
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
__global__ void ld8(const uint8_t *base, int off, uint64_t *out) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    const uint8_t *p = base + off + t * 8;
    uint64_t v;
    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    out[t] = v;
}
int main() {
    const int n = 256 * 64;
    uint8_t *d;
    uint64_t *o;
    hipMalloc(&d, n * 8 + 64);
    hipMalloc(&o, n * 8);
    std::vector<uint8_t> h(n * 8 + 64);
    std::vector<uint64_t> ho(n);
    int bad = 0;
    for (int off = 0; off < 8; ++off) {
        hipMemset(d, 0xEE, n * 8 + 64);
        st8<<<n / 256, 256>>>(d, off);
        if (hipDeviceSynchronize() != hipSuccess) { printf("off %d: store fault\n", off); return 1; }
        hipMemcpy(h.data(), d, n * 8 + 64, hipMemcpyDeviceToHost);
        int e = 0;
        for (int i = 0; i < n * 8; ++i) e += h[off + i] != uint8_t(i & 0xff);
        for (int i = 0; i < off; ++i) e += h[i] != 0xEE;
        e += h[off + n * 8] != 0xEE;
        ld8<<<n / 256, 256>>>(d, off, o);
        if (hipDeviceSynchronize() != hipSuccess) { printf("off %d: load fault\n", off); return 1; }
        hipMemcpy(ho.data(), o, n * 8, hipMemcpyDeviceToHost);
        int el = 0;
        for (int t = 0; t < n; ++t) {
            uint64_t v;
            memcpy(&v, h.data() + off + t * 8, 8);
            el += ho[t] != v;
        }
        printf("offset %d: store errors %d, load errors %d\n", off, e, el);
        bad += e + el;
    }
    printf(bad ? "UNALIGNED 8-BYTE ACCESS BROKEN\n" : "unaligned 8-byte global access works\n");
    return bad != 0;
}
