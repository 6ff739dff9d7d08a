// HIP kernels for GECCO's CRF hot path, written for gfx950 (MI355X / CDNA4) only.
//
// Row W of SURVEY.md §8a -- gecco/crf/__init__.py:244-258: for every window of W genes run
// an independent forward-backward ([EXT] CRFsuite crf1dc_alpha_score / crf1dc_beta_score /
// crf1dc_marginal_point) and keep, per gene, the maximum P(y = label) over the windows that
// cover it.  The reference does this one window at a time through python-crfsuite; here one
// LANE owns one window start and the whole batch is one launch.
//
// Design (two-label models, window <= 32: `crf_windowed_l2`):
//   * slot space: genes of all scored contigs end to end, short contigs centre-padded to W
//     slots (:216-227).  A 256-lane workgroup owns 256 consecutive slots as window starts;
//     its first W-1 lanes are a recomputed halo, so workgroups never exchange data.
//   * stage 1 (HBM -> LDS, coalesced CSR reads + L2-resident weight gather): per slot the
//     state scores s[y] = sum_a w[a][y] ([EXT] crf1dt_state_score), reduced to the pair
//     e = exp(s - max(s)) and parked in LDS (16 B/slot).  A per-position scale factor
//     cancels in every marginal, so e (and exp(trans - max)) replace CRFsuite's raw exps;
//     this bounds the DP vectors in (0, 2^k] and removes CRFsuite's per-step 1/sum
//     division.  Renormalisation by an exact power of two happens only at the steps the
//     host flags in `rescale_mask` (never for GECCO's model: see crf_plan.cpp).
//   * stage 2 (registers): lane s runs the W-step forward recursion for window s keeping all
//     W alpha pairs in VGPRs (fully unrolled), then the backward recursion; at step k it
//     holds both alpha_k and beta_k of slot s+k, i.e. the un-normalised pair
//     (x, y) = (alpha[label]*beta[label], alpha[other]*beta[other]),  P = x / (x + y).
//   * stage 3 (DPP, no LDS traffic, no atomics): the maximum over windows is a diagonal
//     reduction -- candidate k of lane s belongs to slot s+k.  A running best (x, y) is
//     shifted one lane up per step with DPP wave_shr:1 and compared by cross-multiplication
//     (x1*y2 > x2*y1), so the only division is the final x/(x+y) per gene.  Values leaving
//     lane 63 are handed to the next wave of the workgroup through a 16-B LDS slot per step.
//   No MFMA: L = 2 recurrences are 2x2 matrix-vector products on fp64 VALU.
#include "crf_device.hpp"

namespace gecco {
namespace {

// lane l receives lane l-1's value; lane 0 receives +0.0 (DPP wave_shr:1, bound_ctrl:1).
__device__ __forceinline__ double wave_shr1_zero(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// Workgroup b runs on XCD b % 8 (observed placement; speed only).  Give each XCD one
// contiguous range of tiles so that the W-1 slot halo shared by neighbouring tiles is
// served by the same L2.  Bijective for any nwg.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (orig >> 3);
}

__device__ __forceinline__ void rescale_pair(double &u, double &v) {
    int ex;
    (void)frexp(fmax(u, v), &ex);
    u = ldexp(u, -ex);
    v = ldexp(v, -ex);
}

template <int WMAX, int NT>
struct WinSmem {
    double2 em[NT + WMAX - 1];      // per-slot emission pair (other, label), max-normalised
    double2 carry[NT / 64][WMAX];   // running best leaving lane 63 of each wave, per step
    int32_t cslot[NT + WMAX + 1];   // slot offsets of the contigs this tile overlaps
    int32_t cgene[NT + WMAX];
    int32_t cn[NT + WMAX];
};

struct SlotInfo {
    int gene;       // global gene index, -1 for padding / out of range
    bool start_ok;  // a window may start here
};

template <class Smem>
__device__ __forceinline__ SlotInfo slot_lookup(const Smem &sm, int cnt, int q, int S, int W, int step) {
    SlotInfo r{-1, false};
    if (q < 0 || q >= S) return r;
    int lo = 0, hi = cnt - 1;  // largest k with cslot[k] <= q
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (sm.cslot[mid] <= q) lo = mid; else hi = mid - 1;
    }
    const int pos = q - sm.cslot[lo];
    const int np = sm.cslot[lo + 1] - sm.cslot[lo];  // padded length max(n, W)
    const int n = sm.cn[lo];
    const int gl = pos - ((np - n) >> 1);            // delta // 2 empty items in front (:227)
    if (gl >= 0 && gl < n) r.gene = sm.cgene[lo] + gl;
    r.start_ok = (pos + W <= np) && (step == 1 || pos % step == 0);  // _meta.py:131
    return r;
}

template <int WMAX, bool EXACT, bool RESCALE, int NT>
__global__ void __launch_bounds__(NT) crf_windowed_l2(const WinArgs P) {
    using Smem = WinSmem<WMAX, NT>;
    __shared__ Smem sm;
    const int W = EXACT ? WMAX : P.W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = xcd_remap(blockIdx.x, P.ntiles);
    const int q0 = tile * (NT - (W - 1)) - (W - 1);  // slot owned by lane 0

    // ---- contig table of this tile -> LDS
    const int2 tc = P.tile_c[tile];
    const int cnt = tc.y - tc.x + 1;
    for (int j = tid; j <= cnt; j += NT) {
        sm.cslot[j] = P.c_slot[tc.x + j];
        if (j < cnt) {
            sm.cgene[j] = P.c_gene[tc.x + j];
            sm.cn[j] = P.c_n[tc.x + j];
        }
    }
    __syncthreads();

    // ---- stage 1: state scores -> normalised emission pairs in LDS
    int my_gene = -1;
    bool my_start = false;
    for (int j = tid; j < NT + W - 1; j += NT) {
        const SlotInfo si = slot_lookup(sm, cnt, q0 + j, P.S, W, P.step);
        if (j == tid) {
            my_gene = si.gene;
            my_start = si.start_ok;
        }
        double s0 = 0.0, s1 = 0.0;
        if (si.gene >= 0) {
            const int lo = P.gene_ptr[si.gene], hi = P.gene_ptr[si.gene + 1];
            for (int k = lo; k < hi; ++k) {
                const double2 w = P.wtab2[P.attr_id[k]];
                s0 += w.x;
                s1 += w.y;
            }
        }
        const double d = s1 - s0;
        const double e = exp(-fabs(d));
        sm.em[j] = d > 0.0 ? make_double2(e, 1.0) : make_double2(1.0, e);
    }
    __syncthreads();

    const double m00 = P.m00, m01 = P.m01, m10 = P.m10, m11 = P.m11;
    const uint32_t rmask = P.rescale_mask;

    // ---- stage 2a: forward recursion, all W alpha pairs stay in registers
    double A0[WMAX], A1[WMAX];
    double a0, a1;
    {
        const double2 e = sm.em[tid];
        a0 = e.x;
        a1 = e.y;
    }
    A0[0] = a0;
    A1[0] = a1;
#pragma unroll
    for (int k = 1; k < WMAX; ++k) {
        if (EXACT || k < W) {
            const double2 e = sm.em[tid + k];
            const double t0 = fma(a1, m10, a0 * m00);
            const double t1 = fma(a1, m11, a0 * m01);
            a0 = t0 * e.x;
            a1 = t1 * e.y;
            if (RESCALE && ((rmask >> k) & 1u)) rescale_pair(a0, a1);
            A0[k] = a0;
            A1[k] = a1;
        }
    }

    // ---- stage 2b + 3: backward recursion, candidates, diagonal max via DPP shifts
    double b0 = 1.0, b1 = 1.0;
    // running best candidate; (0, 0) = "no window yet" so that DPP zero-fill is the identity
    double Rx = 0.0, Ry = 0.0;
#pragma unroll
    for (int k = WMAX - 1; k >= 0; --k) {
        if (EXACT || k < W) {
            const double x = A1[k] * b1;
            const double y = A0[k] * b0;
            if (k < W - 1) {
                if (lane == 63) sm.carry[wave][k] = make_double2(Rx, Ry);
                Rx = wave_shr1_zero(Rx);
                Ry = wave_shr1_zero(Ry);
            }
            // x/y >= Rx/Ry by cross-multiplication; always true against the (0,0) identity
            const bool take = my_start && (x * Ry >= Rx * y);
            Rx = take ? x : Rx;
            Ry = take ? y : Ry;
            if (k > 0) {
                const double2 e = sm.em[tid + k];
                const double c0 = e.x * b0, c1 = e.y * b1;
                b0 = fma(m01, c1, m00 * c0);
                b1 = fma(m11, c1, m10 * c0);
                if (RESCALE && ((rmask >> k) & 1u)) rescale_pair(b0, b1);
            }
        }
    }
    __syncthreads();
    if (wave > 0 && lane < W - 1) {
        const double2 c = sm.carry[wave - 1][lane];
        if (c.x * Ry > Rx * c.y || (Rx == 0.0 && Ry == 0.0)) {
            Rx = c.x;
            Ry = c.y;
        }
    }
    // genes no window covers (step > 1) keep 0.0 like numpy.zeros (crf/__init__.py:251)
    if (tid >= W - 1 && my_gene >= 0) P.p_out[my_gene] = (Rx + Ry > 0.0) ? Rx / (Rx + Ry) : 0.0;
}

__global__ void fill_nan_kernel(double *p, const int2 *ranges, int n_ranges) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_ranges) return;
    const int2 rg = ranges[r];
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    for (int g = rg.x; g < rg.y; ++g) p[g] = nan;
}

}  // namespace

const char *windowed_kernel_name(int W, int L) {
    if (L == 2 && W == 20) return "crf_windowed_l2<20,exact>";
    if (L == 2 && W <= kWinMaxW) return "crf_windowed_l2<32,dynamic>";
    return "unsupported";
}

int windowed_tile_out(int W, int L) {
    if (L == 2 && W <= kWinMaxW) return kWinThreads - (W - 1);
    return 0;
}

hipError_t launch_windowed(const WinArgs &a, hipStream_t stream) {
    if (a.ntiles <= 0) return hipSuccess;
    const dim3 grid(a.ntiles), block(kWinThreads);
    if (a.L == 2 && a.W == 20 && a.rescale_mask == 0) {
        hipLaunchKernelGGL((crf_windowed_l2<20, true, false, kWinThreads>), grid, block, 0, stream, a);
    } else if (a.L == 2 && a.W == 20) {
        hipLaunchKernelGGL((crf_windowed_l2<20, true, true, kWinThreads>), grid, block, 0, stream, a);
    } else if (a.L == 2 && a.W <= kWinMaxW && a.rescale_mask == 0) {
        hipLaunchKernelGGL((crf_windowed_l2<kWinMaxW, false, false, kWinThreads>), grid, block, 0, stream, a);
    } else if (a.L == 2 && a.W <= kWinMaxW) {
        hipLaunchKernelGGL((crf_windowed_l2<kWinMaxW, false, true, kWinThreads>), grid, block, 0, stream, a);
    } else {
        return hipErrorNotSupported;
    }
    return hipGetLastError();
}

hipError_t launch_fill_nan(double *p, const int2 *ranges, int n_ranges, hipStream_t stream) {
    if (n_ranges <= 0) return hipSuccess;
    hipLaunchKernelGGL(fill_nan_kernel, dim3((n_ranges + 255) / 256), dim3(256), 0, stream, p, ranges, n_ranges);
    return hipGetLastError();
}

}  // namespace gecco
