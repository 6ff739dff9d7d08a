// Semantics of gfx950's LDS-direct buffer loads (`buffer_load_dword / dwordx4 ... offen lds`), which the streaming
// window kernel uses to park attribute ids and gathered weight pairs in LDS without a VGPR:
//   (1) lane l of a wave lands at M0 + l * size (size 4 and 16), whatever its voffset;
//   (2) an out-of-range voffset lands as zeros (bounds-checked raw buffer);
//   (3) s_waitcnt vmcnt(0) is what orders the LDS write before a ds_read;
//   (4) the load is asynchronous: an ALU loop issued behind it overlaps the memory latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void *lds_ptr;

__global__ void __launch_bounds__(256) k_check(const int *ids, const double2 *w, int n_w, int n_ids, double2 *out, int *out_ids) {
    __shared__ __attribute__((aligned(16))) double2 PARK[256];
    __shared__ int IDS[256];
    const int tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<double2 *>(w), 0, n_w * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(ids), 0, n_ids * 4, 0x00020000);
    PARK[tid] = make_double2(-1.0, -1.0);
    IDS[tid] = -7;
    __syncthreads();
    // ids: lane-contiguous 4-byte loads straight into LDS
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ri, (lds_ptr)(IDS + (tid & ~63)), 4, (blockIdx.x * 256 + tid) << 2, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int id = IDS[tid];
    out_ids[blockIdx.x * 256 + tid] = id;
    const unsigned wo = min(unsigned(id), 0x0FFFFFFFu) << 4;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(PARK + (tid & ~63)), 16, int(wo), 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[blockIdx.x * 256 + tid] = PARK[tid];
}

// latency hiding: gathers -> LDS, then `spin` dependent FMAs, then the wait
__global__ void __launch_bounds__(256) k_overlap(const int *ids, const double2 *w, int n_w, double *out, int spin, int dma) {
    __shared__ __attribute__((aligned(16))) double2 PARK[4][256];
    const int tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<double2 *>(w), 0, n_w * 16, 0x00020000);
    double acc = 0.0;
    for (int it = 0; it < 8; ++it) {
        const int id0 = ids[(blockIdx.x * 8 + it) * 256 + tid];
        if (dma) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(&PARK[a][tid & ~63]), 16, int(unsigned((id0 + a * 977) % n_w) << 4), 0, 0, 0);
        }
        double x = double(tid) * 1e-9;
        for (int s = 0; s < spin; ++s) x = fma(x, 0.999999, 1e-9);
        acc += x;
        if (dma) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll
            for (int a = 0; a < 4; ++a) acc += PARK[a][tid].x;
            __syncthreads();
        }
    }
    out[blockIdx.x * 256 + tid] = acc;
}

template <class F>
static double timeit(F f, int iters = 200) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) f();
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) f();
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1e3 / iters;
}

int main() {
    const int n_w = 35000, nb = 2048, n = nb * 256, n_ids = n - 100;  // the last 100 ids are out of range of the id buffer
    std::vector<int> ids(n);
    std::vector<double2> w(n_w);
    srand(1);
    for (int i = 0; i < n; ++i) ids[i] = (i % 17 == 3) ? 1000000 + i : rand() % n_w;  // some ids outside the weight table
    for (int i = 0; i < n_w; ++i) w[i] = make_double2(i + 0.25, -i - 0.5);
    int *d_ids, *d_oid;
    double2 *d_w, *d_out;
    double *d_acc;
    (void)hipMalloc(&d_ids, size_t(n) * 8 * 4);
    (void)hipMalloc(&d_oid, n * 4);
    (void)hipMalloc(&d_w, n_w * 16);
    (void)hipMalloc(&d_out, size_t(n) * 16);
    (void)hipMalloc(&d_acc, size_t(n) * 8);
    (void)hipMemcpy(d_ids, ids.data(), n * 4, hipMemcpyHostToDevice);
    for (int r = 1; r < 8; ++r) (void)hipMemcpy(d_ids + size_t(r) * n, ids.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_w, w.data(), n_w * 16, hipMemcpyHostToDevice);
    k_check<<<nb, 256>>>(d_ids, d_w, n_w, n_ids, d_out, d_oid);
    std::vector<double2> out(n);
    std::vector<int> oid(n);
    (void)hipMemcpy(out.data(), d_out, size_t(n) * 16, hipMemcpyDeviceToHost);
    (void)hipMemcpy(oid.data(), d_oid, n * 4, hipMemcpyDeviceToHost);
    long bad_id = 0, bad_w = 0;
    for (int i = 0; i < n; ++i) {
        const int eid = i < n_ids ? ids[i] : 0;
        if (oid[i] != eid) ++bad_id;
        const double2 e = (unsigned(eid) < unsigned(n_w)) ? w[eid] : make_double2(0.0, 0.0);
        if (out[i].x != e.x || out[i].y != e.y) ++bad_w;
    }
    printf("lds-direct loads: %ld wrong ids, %ld wrong weight pairs of %d (expect 0 0)\n", bad_id, bad_w, n);
    for (int spin : {0, 200, 800, 2000}) {
        const double t1 = timeit([&] { k_overlap<<<nb, 256>>>(d_ids, d_w, n_w, d_acc, spin, 1); });
        const double t0 = timeit([&] { k_overlap<<<nb, 256>>>(d_ids, d_w, n_w, d_acc, spin, 0); });
        printf("spin %4d: gathers->LDS + ALU %.1f us, ALU alone %.1f us\n", spin, t1, t0);
    }
    return bad_id || bad_w;
}
