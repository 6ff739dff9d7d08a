// Micro-benchmark of the window DP alone (no global-memory stage): which part of the per-step
// instruction stream costs what on MI355X.  One lane = one window of W=20 steps; slot constants
// come from LDS filled with pseudo-random values; every workgroup runs TILES tiles back to back.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o dp_variants dp_variants.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double f64x2 __attribute__((ext_vector_type(2)));
constexpr int W = 20, NT = 512, TILES = 8;

// lane l <- src[l-1]; lane 0 <- old[0]
__device__ __forceinline__ double shr1_keep(double old, double src) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), 0x138, 0xF, 0xF, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), 0x138, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double ror1(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x13C, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x13C, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shr1(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// VARIANT 0: full (candidates + DPP shift + cross-multiplied compare + select + carries)
//         1: forward/backward chains only (sum of candidates kept alive, no shift/select)
//         2: + candidates and DPP shift, select replaced by fmax on x only
//         3: full without the lane-63 carry stores
//         4: full, running best compared through v_rcp_f64-based ratio (1 double shifted)
//         8: full DP with 16-B slot constants (e0, f) and g = f*rho derived in registers: 4+4 ops per
//            step instead of 4+3, one ds_read_b128 per step and direction instead of b128 + b64
//         7: like 1 but the slot constants are read from LDS once (registers afterwards): how much of
//            the recurrences' time is LDS latency / s_waitcnt
//         6: full, carries kept in a second DPP delay line (wave_ror + wave_shr with lane-0 insert), one
//            LDS store per tile instead of one per step
//         5: full, carries stored by EVERY lane without exec masking: lane 63 hits the real slot, the
//            others a per-wave dump area (no branch / basic-block split per step)
template <int VARIANT>
__global__ void __launch_bounds__(NT, 4) dp_kernel(double *out, double mu01, double mu11, double kap, double ikap) {
    __shared__ f64x2 fg[NT + W - 1];
    __shared__ double e0s[NT + W - 1];
    __shared__ f64x2 ef2[NT + W - 1];
    const double rho = mu11 / mu01;
    __shared__ f64x2 carry[NT / 64][W];
    __shared__ f64x2 dump[NT / 64][64 + W];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int j = tid; j < NT + W - 1; j += NT) {
        const double u = 0.3 + 0.6 * ((j * 2654435761u + blockIdx.x * 40503u) % 1000) / 1000.0;
        e0s[j] = (j & 1) ? u : 1.0;
        fg[j] = f64x2{mu01 * ((j & 1) ? 1.0 : u), mu11 * ((j & 1) ? 1.0 : u)};
        ef2[j] = f64x2{e0s[j], mu01 * ((j & 1) ? 1.0 : u)};
    }
    __syncthreads();
    double acc = 0.0;
    for (int t = 0; t < TILES; ++t) {
        const double *e0p = &e0s[tid];
        const f64x2 *fgp = &fg[tid];
        double A0[W], A1[W];
        double a0 = e0p[0], a1 = fgp[0].y * kap;
        A0[0] = a0; A1[0] = a1;
#pragma unroll
        for (int k = 1; k < W; ++k) {
            const double e0 = e0p[VARIANT == 7 ? 1 : k];
            const f64x2 g = fgp[VARIANT == 7 ? 1 : k];
            const double s = a0 + a1;
            double n1;
            if (VARIANT == 8) {
                const f64x2 ef = ef2[tid + k];
                n1 = fma(a1, rho, a0) * ef.y;
                a0 = s * ef.x; a1 = n1;
            } else {
                n1 = fma(a1, g.y, a0 * g.x);
                a0 = s * e0; a1 = n1;
            }
            A0[k] = a0; A1[k] = a1;
        }
        asm volatile("" ::: "memory");
        double b0 = 1.0, b1 = ikap, Rx = 0.0, Ry = 0.0, Rz = 0.0, Fx = 0.0, Fy = 0.0;
        f64x2 *cdst = (lane == 63 && wave < NT / 64 - 1) ? &carry[wave][0] : &dump[wave][lane];
#pragma unroll
        for (int k = W - 1; k >= 0; --k) {
            const double x = A1[k] * b1, y = A0[k] * b0;
            if (VARIANT == 1 || VARIANT == 7) {
                Rx += x; Ry += y;
            } else if (VARIANT == 2) {
                if (k < W - 1) { Rx = shr1(Rx); Ry = shr1(Ry); }
                Rx = fmax(Rx, x); Ry = fmax(Ry, y);
            } else if (VARIANT == 4) {
                if (k < W - 1) Rz = shr1(Rz);
                double r = __builtin_amdgcn_rcp(y);
                r = fma(fma(-y, r, 1.0), r, r);
                Rz = fmax(Rz, x * r);
            } else if (VARIANT == 8) {
                if (k < W - 1) {
                    if (lane == 63 && wave < NT / 64 - 1) carry[wave][k] = f64x2{Rx, Ry};
                    Rx = shr1(Rx); Ry = shr1(Ry);
                }
                const bool take = x * Ry >= Rx * y;
                Rx = take ? x : Rx; Ry = take ? y : Ry;
            } else if (VARIANT == 6) {
                if (k < W - 1) {
                    const double Tx = ror1(Rx), Ty = ror1(Ry);   // lane 0 sees lane 63's running best
                    Fx = shr1_keep(Tx, Fx); Fy = shr1_keep(Ty, Fy);
                    Rx = shr1(Rx); Ry = shr1(Ry);
                }
                const bool take = x * Ry >= Rx * y;
                Rx = take ? x : Rx; Ry = take ? y : Ry;
            } else {
                if (k < W - 1) {
                    if (VARIANT == 0 && lane == 63 && wave < NT / 64 - 1) carry[wave][k] = f64x2{Rx, Ry};
                    if (VARIANT == 5) cdst[k] = f64x2{Rx, Ry};
                    Rx = shr1(Rx); Ry = shr1(Ry);
                }
                const bool take = x * Ry >= Rx * y;
                Rx = take ? x : Rx; Ry = take ? y : Ry;
            }
            if (k > 0) {
                const double e0 = e0p[VARIANT == 7 ? 1 : k];
                const f64x2 g = fgp[VARIANT == 7 ? 1 : k];
                if (VARIANT == 8) {
                    const f64x2 ef = ef2[tid + k];
                    const double c = ef.x * b0, u = ef.y * b1;
                    b0 = c + u;
                    b1 = fma(u, rho, c);
                } else {
                const double c = e0 * b0;
                b0 = fma(g.x, b1, c);
                b1 = fma(g.y, b1, c);
                }
            }
        }
        if (VARIANT == 6) { carry[wave][lane < W ? lane : 0] = f64x2{Fx, Fy}; }
        acc += (VARIANT == 4) ? Rz : Rx / (Rx + Ry);
        __syncthreads();
    }
    out[blockIdx.x * NT + tid] = acc;
}

template <int V>
int run(const char *name, double *d_out) {
    const int nblocks = 4096;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(dp_kernel<V>, dim3(nblocks), dim3(NT), 0, 0, d_out, 2.7e-5, 0.9, 5.1e-3, 196.0);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(dp_kernel<V>, dim3(nblocks), dim3(NT), 0, 0, d_out, 2.7e-5, 0.9, 5.1e-3, 196.0);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double windows = double(nblocks) * NT * TILES;
    printf("%-52s %8.3f ms  %7.2f ns per 1000 windows  (C3-equivalent: %.1f us)\n", name, ms, ms * 1e6 / windows * 1000, ms * 1e3 * (2.0e6 / 0.963) / windows);
    return 0;
}

int main() {
    double *d_out; CHECK(hipMalloc(&d_out, sizeof(double) * 4096 * NT));
    run<0>("0 full DP (shift + cross-mult select + carries)", d_out);
    run<8>("8 full DP, 16-B slot constants (g = f*rho in registers)", d_out);
    run<3>("3 full without carry stores", d_out);
    run<5>("5 full, carries stored by every lane (no exec mask)", d_out);
    run<6>("6 full, carries in a DPP delay line (1 store/tile)", d_out);
    run<2>("2 candidates + shift, fmax instead of select", d_out);
    run<4>("4 ratio via rcp+NR, one double shifted, fmax", d_out);
    run<1>("1 forward/backward chains + candidates only", d_out);
    run<7>("7 same, slot constants in registers (no LDS in loop)", d_out);
    return 0;
}
