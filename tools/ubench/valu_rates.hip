// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU / DPP /
// LDS-crossbar instructions the CRF kernels are built from.  Not part of the product.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int CHAINS = 8;

__device__ __forceinline__ double shr1(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rowshr1(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bperm(double v, int addr) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_ds_bpermute(addr, lo);
    hi = __builtin_amdgcn_ds_bpermute(addr, hi);
    return __hiloint2double(hi, lo);
}

template <int OP>
__global__ void __launch_bounds__(256) bench(double *out, double seed, int sel) {
    double v[CHAINS];
    for (int c = 0; c < CHAINS; ++c) v[c] = seed + c * 1e-3 + threadIdx.x * 1e-6;
    const double m = 1.0000001, a = 1e-9;
    const int addr = ((threadIdx.x + 63) & 63) * 4;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if (OP == 0) v[c] = v[c] * m;                       // v_mul_f64
            if (OP == 1) v[c] = fma(v[c], m, a);                // v_fma_f64
            if (OP == 2) v[c] = v[c] + a;                       // v_add_f64
            if (OP == 3) v[c] = shr1(v[c]);                     // 2 x v_mov_b32_dpp wave_shr:1
            if (OP == 4) v[c] = (sel & (1 << (it & 7))) ? v[c] : v[(c + 1) % CHAINS];  // scalar-cond select
            if (OP == 5) v[c] = fmax(v[c], v[(c + 1) % CHAINS] * m);  // mul + max
            if (OP == 6) v[c] = bperm(v[c], addr);              // 2 x ds_bpermute_b32
            if (OP == 7) v[c] = __builtin_amdgcn_rcp(v[c]);     // v_rcp_f64
            if (OP == 8) v[c] = rowshr1(v[c]);                  // 2 x v_mov_b32_dpp row_shr:1
            if (OP == 9) { bool t = v[c] * m > v[(c + 1) % CHAINS]; v[c] = t ? v[c] + a : v[c]; }  // mul+cmp+add+2cndmask
            if (OP == 10) v[c] = __hiloint2double(__double2hiint(v[c]) + 1, __double2loint(v[c]) ^ it);  // 2 x int ops
            if (OP == 11) v[c] = v[c] / (v[(c + 1) % CHAINS] + 3.0);  // full fp64 division
        }
    }
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += v[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
int run(const char *name, double instr_per_iter_chain, double *d_out, int waves_per_simd) {
    // one block of 256 threads = 1 wave per SIMD; blocks per CU = waves_per_simd
    int nblocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(bench<OP>, dim3(nblocks), dim3(256), 0, 0, d_out, 1.0, 0x55);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(bench<OP>, dim3(nblocks), dim3(256), 0, 0, d_out, 1.0, 0x55);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    double instrs_per_simd = double(ITERS) * CHAINS * instr_per_iter_chain * waves_per_simd;
    double cycles = ms * 1e-3 * 2.4e9;
    printf("%-34s waves/SIMD=%d  %8.3f ms  %6.2f cycles per wave-instruction (at 2.4 GHz)\n", name, waves_per_simd, ms,
           cycles / instrs_per_simd);
    return 0;
}

int main() {
    double *d_out;
    CHECK(hipMalloc(&d_out, sizeof(double) * 256 * 256 * 8));
    for (int w : {1, 4}) {
        run<0>("v_mul_f64", 1, d_out, w);
        run<1>("v_fma_f64", 1, d_out, w);
        run<2>("v_add_f64", 1, d_out, w);
        run<3>("v_mov_b32_dpp wave_shr:1 (x2)", 2, d_out, w);
        run<8>("v_mov_b32_dpp row_shr:1 (x2)", 2, d_out, w);
        run<4>("v_cndmask (scalar cond, x2)", 2, d_out, w);
        run<5>("v_mul_f64 + v_max_f64", 2, d_out, w);
        run<6>("ds_bpermute_b32 (x2)", 2, d_out, w);
        run<7>("v_rcp_f64", 1, d_out, w);
        run<9>("mul+cmp+add+2cndmask", 5, d_out, w);
        run<10>("int add + xor (x2)", 2, d_out, w);
        run<11>("fp64 divide (per division)", 1, d_out, w);
    }
    return 0;
}
