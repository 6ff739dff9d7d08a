// Any number of labels (1 <= L <= 32): rows S, A/B, P, W, F and V of SURVEY.md §8a for CRFsuite
// models other than GECCO's 2-label one.  GECCO's own model never comes
// here; `GECCO_CRF_FORCE_GENERAL=1` routes 2-label models through these kernels so the tests can
// cross-check the specialised kernels on the device.
//
// Layout: LP = L rounded up to a power of two; LP consecutive lanes ("a group", never straddling
// a wave) own one window or one contig, lane j holds component j of the alpha / beta / delta
// vector.  A step is a vector x (L x L) product: the previous vector goes through one LDS slot
// per lane and comes back as L broadcast reads; the lane's column (forward) and row (backward)
// of the transition matrix stay in registers.  LDS traffic of one wave is in order and groups
// live inside a wave, so no barrier is needed and groups may run different trip counts.
//
// Arithmetic follows [EXT] CRFsuite crf1d_context.c in its own order (alpha: sum over the
// source label in index order, then * exp(state), then 1/sum scaling; beta: row . (beta o exp(state))
// then * scale; marginal = alpha * beta / scale; Viterbi: strict `<` update, first arg max), with
// one deliberate difference: exp(state - max_y state) instead of exp(state), which cancels in
// every marginal and is added back to the log-partition.
#include "crf_device.hpp"

#include <cfloat>

namespace gecco {
namespace {

constexpr int kGT = 256;  // lanes per workgroup

template <int LP>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
    for (int o = LP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, LP);
    return v;
}
template <int LP>
__device__ __forceinline__ double group_max(double v) {
#pragma unroll
    for (int o = LP / 2; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, LP));
    return v;
}

// last gene + 1 of chunk ci: where the next chunk starts, or where its contig ends (the chunk table may list the long contigs
// of a batch only: the next chunk then belongs to a contig further on)
__device__ __forceinline__ int gl_chunk_end(const GenArgs &a, long long ci) {
    return min(a.ch_g0[ci + 1], a.contig_ptr[a.ch_contig[ci] + 1]);
}

// ---- row S: state scores of every gene (CSR order, bit-identical to sequential addition) ----
// state[g][y] = sum_a w[a][y]; E[g][y] = exp(state - max_y); smax[g] = max_y state.
template <int LP>
__global__ void __launch_bounds__(kGT) gl_state(const int32_t *__restrict__ gene_ptr, const int32_t *__restrict__ attr_id,
                                                const double *__restrict__ wtab, int L, int A, int n_genes,
                                                double *__restrict__ state, double *__restrict__ E,
                                                double *__restrict__ smax) {
    const int j = threadIdx.x & (LP - 1);
    const long long g = (static_cast<long long>(blockIdx.x) * kGT + threadIdx.x) / LP;
    if (g >= n_genes) return;
    const int lo = gene_ptr[g], hi = gene_ptr[g + 1];
    const int jj = j < L ? j : 0;
    double acc = 0.0;
    for (int a = lo; a < hi; ++a) {
        const int id = attr_id[a];  // ids outside the model's dictionary are unknown attributes: no weight
        if (unsigned(id) < unsigned(A)) acc += wtab[static_cast<size_t>(id) * L + jj];
    }
    const double m = group_max<LP>(j < L ? acc : -DBL_MAX);
    if (j < L) {
        if (state) state[static_cast<size_t>(g) * L + j] = acc;
        if (E) E[static_cast<size_t>(g) * L + j] = exp(acc - m);
    }
    if (j == 0 && smax) smax[g] = m;
}

// ---- rows A/B, P, W: one group per window start ------------------------------------------------
template <int LP>
__global__ void __launch_bounds__(kGT) gl_windowed(GenArgs a) {
    extern __shared__ double lds[];
    constexpr int G = kGT / LP;
    const int j = threadIdx.x & (LP - 1), grp = threadIdx.x / LP;
    const int W = a.W, L = a.L;
    double *al = lds + static_cast<size_t>(grp) * W * LP;        // alpha-hat of every step
    double *sc = lds + static_cast<size_t>(G) * W * LP + grp * W; // scale factors
    double *vec = lds + static_cast<size_t>(G) * W * (LP + 1) + grp * LP;
    const long long q = static_cast<long long>(blockIdx.x) * G + grp;
    bool active = q < a.S && ((a.start_bits[q >> 6] >> (q & 63)) & 1);
    int g0 = 0, n = 0, off = 0;
    if (active) {
        int lo = 0, hi = a.K - 1;  // scored contig owning slot q
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (a.c_slot[mid] <= q) lo = mid; else hi = mid - 1;
        }
        const int s0 = a.c_slot[lo], np = a.c_slot[lo + 1] - s0;
        n = a.c_n[lo];
        g0 = a.c_gene[lo];
        off = int(q - s0) - ((np - n) >> 1);  // gene index (within the contig) of window position 0
    }
    const bool lane_on = active && j < L;
    double mcol[LP], mrow[LP];
#pragma unroll
    for (int i = 0; i < LP; ++i) {
        const bool ok = i < L && j < L;
        mcol[i] = ok ? a.exp_trans[i * L + j] : 0.0;
        mrow[i] = ok ? a.exp_trans[j * L + i] : 0.0;
    }
    // emission of window position t: padding items have no attributes -> state 0 -> exp(0-0) = 1
    auto emis = [&](int t) -> double {
        if (!lane_on) return 0.0;
        const int gi = off + t;
        return (gi >= 0 && gi < n) ? a.E[static_cast<size_t>(g0 + gi) * L + j] : 1.0;
    };
    // forward
    double e = emis(0), v = 0.0, c = 1.0;
    for (int t = 0; t < W; ++t) {
        const double e_next = t + 1 < W ? emis(t + 1) : 0.0;
        if (t == 0) {
            v = e;
        } else {
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < LP; ++i) acc = fma(al[(t - 1) * LP + i], mcol[i], acc);
            v = acc * e;
        }
        const double s = group_sum<LP>(v);
        c = s != 0.0 ? 1.0 / s : 1.0;
        v *= c;
        al[t * LP + j] = v;
        if (j == 0) sc[t] = c;
        e = e_next;
        __builtin_amdgcn_wave_barrier();
    }
    // backward + marginal of `label` + per-gene maximum over windows
    unsigned long long *out = reinterpret_cast<unsigned long long *>(a.p_out);
    double b = c;  // beta_{W-1} = scale_{W-1}
    for (int t = W - 1; t >= 0; --t) {
        const double ct = sc[t];
        if (t < W - 1) {
            vec[j] = b * emis(t + 1);
            __builtin_amdgcn_wave_barrier();
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < LP; ++i) acc = fma(mrow[i], vec[i], acc);
            b = acc * ct;
            __builtin_amdgcn_wave_barrier();
        }
        const int gi = off + t;
        if (lane_on && j == a.label && gi >= 0 && gi < n) {
            const double pr = al[t * LP + j] * b / ct;
            atomicMax(out + g0 + gi, static_cast<unsigned long long>(__double_as_longlong(pr)));
        }
    }
}

// ---- rows A/B, P, W for a handful of labels (up to 8): the two-label kernel's design ---------------------------------
// One LANE per window start, everything of the window in its registers (crf_kernels.hip): with L <= 8 the vectors are
// L doubles, a step is L*L FMAs + L multiplications, and a group of lanes exchanging vector components through LDS --
// the kernel above, written for up to 32 labels -- spends most of its time on that exchange (1.6 G genes/s at L = 3).
//   * un-normalised recurrences on max-normalised factors (exp(state - max state): gl_state; exp(trans - max trans)):
//     alpha_k . beta_k = Z at every position of the window, so the marginal of the queried label is
//     alpha_k[label] beta_k[label] / Z with 1/Z folded into the initial beta; only alpha_k[label] is kept (W doubles).
//     The host checks that W - 1 steps cannot leave the range (spread of the transition weights * (W - 1) < 600);
//     models beyond that take the kernel above.  Labels are permuted so that the queried one is component 0.
//   * the maximum over the windows that cover a gene is the DPP diagonal of the two-label kernel (running best shifted
//     one lane up per step, hand-over between waves through LDS): no atomics, p_out written once.
//   * a workgroup of 256 window starts owns 256 - (W - 1) output slots; the emissions of its 256 + (W - 1) slots are
//     staged in LDS once (regular tiles: slots map to genes by a constant shift; others look every slot up).
constexpr int kSmallNT = 256;
__device__ __forceinline__ double gl_wave_shr1_zero(double v) {  // lane l <- lane l-1, lane 0 <- +0.0
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
struct SmallTrans {
    double m[64];  // exp(trans - max), labels permuted (queried label first), row-major L x L (L <= 8)
};
template <int L, int WMAX>
__global__ void __launch_bounds__(kSmallNT) gl_windowed_small(GenArgs a, SmallTrans T, const int4 *__restrict__ tile_desc) {
    constexpr int NT = kSmallNT, CAP = NT + WMAX - 1;
    __shared__ double Es[L * CAP];       // emissions of the tile's slots, one row per label (conflict-free lane stride)
    __shared__ uint32_t ginfo[CAP];      // bit 31: a window may start here; low bits: gene + 1 (0: none)
    __shared__ double carry[(NT / 64) * WMAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int W = a.W, OUT = NT - (W - 1), ns = NT + W - 1;
    const int q0 = blockIdx.x * OUT - (W - 1);
    const int4 td = tile_desc[blockIdx.x];
    // queried label first, the others in their own order
    int perm[L];
    perm[0] = a.label;
#pragma unroll
    for (int j = 1; j < L; ++j) perm[j] = j <= a.label ? j - 1 : j;
    for (int sl = tid; sl < ns; sl += NT) {
        const int q = q0 + sl;
        int gene = -1;
        bool start = false;
        if (q >= 0 && q < a.S) {
            start = (a.start_bits[q >> 6] >> (q & 63)) & 1ull;
            if (td.w & 1) {
                gene = q + td.x;
            } else {
                int lo = td.y, hi = td.z;  // largest k with c_slot[k] <= q among the contigs in reach
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (a.c_slot[mid] <= q) lo = mid; else hi = mid - 1;
                }
                const int pos = q - a.c_slot[lo], np = a.c_slot[lo + 1] - a.c_slot[lo], n = a.c_n[lo];
                const int gl = pos - ((np - n) >> 1);  // delta // 2 empty items in front (crf/__init__.py:227)
                if (gl >= 0 && gl < n) gene = a.c_gene[lo] + gl;
            }
        }
#pragma unroll
        for (int j = 0; j < L; ++j)  // padding items have no attributes: state 0, exp(0 - 0) = 1
            Es[j * CAP + sl] = gene >= 0 ? a.E[size_t(gene) * L + perm[j]] : 1.0;
        ginfo[sl] = (start ? 0x80000000u : 0u) | uint32_t(gene + 1);
    }
    __syncthreads();
    const uint32_t gi = ginfo[tid];
    const bool my_start = gi >> 31;
    const int my_gene = int(gi & 0x7fffffffu) - 1;
    const double *es = Es + tid;
    // forward: alpha_0 = E_0; alpha_k[j] = (sum_i alpha_{k-1}[i] M[i][j]) E_k[j]
    double al[L], A0[WMAX];
#pragma unroll
    for (int j = 0; j < L; ++j) al[j] = es[j * CAP];
    A0[0] = al[0];
#pragma unroll
    for (int k = 1; k < WMAX; ++k) {
        if (k < W) {
            double nx[L];
#pragma unroll
            for (int j = 0; j < L; ++j) {
                double acc = al[0] * T.m[j];
#pragma unroll
                for (int i = 1; i < L; ++i) acc = fma(al[i], T.m[i * L + j], acc);
                nx[j] = acc * es[j * CAP + k];
            }
#pragma unroll
            for (int j = 0; j < L; ++j) al[j] = nx[j];
            A0[k] = al[0];
        }
    }
    asm volatile("" ::: "memory");  // re-read the emissions in the backward pass (VGPRs)
    double z = al[0];
#pragma unroll
    for (int j = 1; j < L; ++j) z += al[j];
    double rz = __builtin_amdgcn_rcp(z);
    rz = fma(fma(-z, rz, 1.0), rz, rz);
    double be[L];
#pragma unroll
    for (int j = 0; j < L; ++j) be[j] = my_start ? rz : 0.0;  // beta_{W-1} = 1, times 1/Z; lanes that start no window: 0
    double R = 0.0;
#pragma unroll
    for (int k = WMAX - 1; k >= 0; --k) {
        if (k < W) {
            const double cand = A0[k] * be[0];
            if (k < W - 1) {
                if (lane == 63 && wave < NT / 64 - 1) carry[wave * WMAX + k] = R;
                R = gl_wave_shr1_zero(R);
            }
            R = fmax(R, cand);
            if (k > 0) {  // beta_{k-1}[i] = sum_j M[i][j] E_k[j] beta_k[j]
                double u[L];
#pragma unroll
                for (int j = 0; j < L; ++j) u[j] = es[j * CAP + k] * be[j];
#pragma unroll
                for (int i = 0; i < L; ++i) {
                    double acc = T.m[i * L] * u[0];
#pragma unroll
                    for (int j = 1; j < L; ++j) acc = fma(T.m[i * L + j], u[j], acc);
                    be[i] = acc;
                }
            }
        }
    }
    __syncthreads();
    if (wave > 0 && lane < W - 1) R = fmax(R, carry[(wave - 1) * WMAX + lane]);
    R = fmin(R, 1.0);
    // genes no window covers (step > 1) keep 0.0 like numpy.zeros (crf/__init__.py:251)
    if (tid >= W - 1 && my_gene >= 0) a.p_out[my_gene] = R;
}

// ---- rows A/B, P, W for 9 to 32 labels: sixteen windows per wave on the fp64 matrix cores -------------------------------
// Above eight labels the L-vectors of a window no longer fit one lane's registers, and the lane-group kernel at the top of
// this file pays for every step with L broadcast reads from LDS per lane (0.34 G genes/s at L = 16, 0.10 G at L = 32).
// A step of the recurrences over SIXTEEN windows at once is a matrix product, alpha'^T = M^T alpha^T (labels x windows):
//   v_mfma_f64_16x16x4_f64:  D (16 x 16) += A (16 x 4) B (4 x 16);  lane l holds A[l & 15][l >> 4], B[l >> 4][l & 15] and
//   D[(l >> 4) + 4 r][l & 15] in its result register r   (cdna_hip_programming.md, fragment layout of the f64 form).
// With windows as columns, result register r of a lane IS its B operand of K-slice r in the next step (row (l >> 4) + 4 r of
// D = row l >> 4 of slice r of B): alpha and beta never leave their registers, nothing crosses lanes between steps, the
// slices of M^T (forward) and M (backward) are per-lane constants.  Per step and 16 windows: ceil(L / 16) * ceil(L / 4)
// MFMAs + one multiplication by the emission per register.
//   * formulation of gl_windowed_small: un-normalised recurrences on max-normalised factors, alpha_k . beta_k = Z at every
//     position, 1 / Z folded into the initial beta, queried label permuted to component 0 (lanes 0-15 of a wave own its
//     alpha_k: W doubles per lane), same range guard on the transition weights (gen_small_ok).  The summation order of a
//     step differs from CRFsuite's (four terms per MFMA, in hardware order): results agree with the oracle to 1e-12, as
//     the other kernels of this file do, not bit for bit.
//   * the maximum over the windows that cover a gene: returnless LDS atomics (ds_max_u64 on the bit pattern of the
//     non-negative candidate) on best[slot] -- sixteen lanes per step, next to 4-16 MFMAs: the LDS port is idle here (in
//     the two-label kernel, where a step is seven VALU instructions, the same atomics lose: profiles/r04_window_kernel_ab.txt).
//   * a workgroup = 256 window starts (4 waves x 4 batches of 16) owning 256 - (W - 1) output slots, emissions staged once
//     in LDS, one row per label with a stride that keeps the 16 x 4 lanes of a read on distinct banks.
typedef double gl_v4d __attribute__((ext_vector_type(4)));
constexpr int kMfmaNT = 256;
__host__ __device__ constexpr int gl_mfma_stride(int wmax) { return ((kMfmaNT + wmax - 1 + 15) / 32) * 32 + 16; }  // = 16 mod 32, >= slots
template <int TILES, int NS, int WMAX>
__global__ void __launch_bounds__(kMfmaNT) gl_windowed_mfma(GenArgs a, double tmax, const int4 *__restrict__ tile_desc) {
    constexpr int NT = kMfmaNT, S = gl_mfma_stride(WMAX);
    extern __shared__ double gl_dyn[];
    const int L = a.L;
    double *Es = gl_dyn;                                                       // [L][S] emissions, label-major
    unsigned long long *best = reinterpret_cast<unsigned long long *>(gl_dyn + size_t(L) * S);  // [NT + WMAX]
    uint32_t *ginfo = reinterpret_cast<uint32_t *>(best + NT + WMAX);          // [NT + WMAX]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, w = lane & 15, g = lane >> 4;
    const int W = a.W, OUT = NT - (W - 1), ns = NT + W - 1;
    const int q0 = blockIdx.x * OUT - (W - 1);
    const int4 td = tile_desc[blockIdx.x];
    auto perm = [&](int j) { return j == 0 ? a.label : (j <= a.label ? j - 1 : j); };  // queried label first
    // per-lane constants: slices of M^T (forward: row = output label, column = input label) and of M (backward)
    double Af[TILES][NS], Ab[TILES][NS];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx) {
            const int row = 16 * t + w, col = 4 * sidx + g;  // (row: label on the matrix side of the product, col: the summed label)
            const bool ok = row < L && col < L;
            Af[t][sidx] = ok ? exp(a.trans[perm(col) * L + perm(row)] - tmax) : 0.0;  // M^T[row][col] = M[col][row]
            Ab[t][sidx] = ok ? exp(a.trans[perm(row) * L + perm(col)] - tmax) : 0.0;
        }
    // ---- stage 1: slots -> genes, emissions -> LDS (consecutive lanes read consecutive doubles of the [gene][label] array)
    for (int sl = tid; sl < NT + WMAX; sl += NT) {
        const int q = q0 + sl;
        int gene = -1;
        bool start = false;
        if (sl < ns && q >= 0 && q < a.S) {
            start = (a.start_bits[q >> 6] >> (q & 63)) & 1ull;
            if (td.w & 1) {
                gene = q + td.x;
            } else {
                int lo = td.y, hi = td.z;  // largest k with c_slot[k] <= q among the contigs in reach
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (a.c_slot[mid] <= q) lo = mid; else hi = mid - 1;
                }
                const int pos = q - a.c_slot[lo], np = a.c_slot[lo + 1] - a.c_slot[lo], n = a.c_n[lo];
                const int gl = pos - ((np - n) >> 1);  // delta // 2 empty items in front (crf/__init__.py:227)
                if (gl >= 0 && gl < n) gene = a.c_gene[lo] + gl;
            }
        }
        ginfo[sl] = (start ? 0x80000000u : 0u) | uint32_t(gene + 1);
        best[sl] = 0ull;
    }
    __syncthreads();
    for (int idx = tid; idx < ns * L; idx += NT) {
        const int sl = idx / L, j = idx - sl * L;
        const int gene = int(ginfo[sl] & 0x7fffffffu) - 1;
        // (j-th label of the gene's row in memory = permuted component pj: component 0 is the queried label)
        const int pj = j == a.label ? 0 : (j < a.label ? j + 1 : j);
        Es[pj * S + sl] = gene >= 0 ? a.E[size_t(gene) * L + j] : 1.0;  // padding items have no attributes: exp(0 - 0) = 1
    }
    __syncthreads();
    // the emission of (component of result register r of tile t, this lane's window, step k); components >= L: any finite
    // number (their alpha / beta are exact zeros: the matrix rows are)
    auto em = [&](int t, int r, int slot) { return Es[min(16 * t + 4 * r + g, L - 1) * S + slot]; };
#pragma unroll 1
    for (int b = 0; b < 4; ++b) {
        const int sl0 = wave * 64 + b * 16 + w;  // slot of this lane's window start
        gl_v4d D[TILES];
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) D[t][r] = (16 * t + 4 * r + g) < L ? em(t, r, sl0) : 0.0;
        double A0[WMAX];
        A0[0] = D[0][0];
#pragma unroll
        for (int k = 1; k < WMAX; ++k) {
            if (k < W) {
                gl_v4d N[TILES];
#pragma unroll
                for (int t = 0; t < TILES; ++t) {
                    N[t] = gl_v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int sidx = 0; sidx < NS; ++sidx)
                        N[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Af[t][sidx], D[sidx / 4][sidx % 4], N[t], 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < TILES; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) D[t][r] = N[t][r] * em(t, r, sl0 + k);
                A0[k] = D[0][0];
            }
        }
        // Z of the lane's window: all components, i.e. all registers of the four lanes w, w + 16, w + 32, w + 48
        double z = 0.0;
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) z += D[t][r];
        z += __shfl_xor(z, 16);
        z += __shfl_xor(z, 32);
        double rz = __builtin_amdgcn_rcp(z);
        rz = fma(fma(-z, rz, 1.0), rz, rz);
        const bool my_start = ginfo[sl0] >> 31;
        gl_v4d B[TILES];
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) B[t][r] = (my_start && (16 * t + 4 * r + g) < L) ? rz : 0.0;  // beta_{W-1} = 1, times 1/Z
#pragma unroll
        for (int k = WMAX - 1; k >= 0; --k) {
            if (k < W) {
                const double cand = A0[k] * B[0][0];  // lanes 0-15: component 0 = the queried label
                if (g == 0)
                    (void)__hip_atomic_fetch_max(best + sl0 + k, static_cast<unsigned long long>(__double_as_longlong(cand)),
                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (k > 0) {  // beta_{k-1} = M (E_k o beta_k)
                    gl_v4d U[TILES], N[TILES];
#pragma unroll
                    for (int t = 0; t < TILES; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) U[t][r] = B[t][r] * em(t, r, sl0 + k);
#pragma unroll
                    for (int t = 0; t < TILES; ++t) {
                        N[t] = gl_v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int sidx = 0; sidx < NS; ++sidx)
                            N[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ab[t][sidx], U[sidx / 4][sidx % 4], N[t], 0, 0, 0);
                    }
#pragma unroll
                    for (int t = 0; t < TILES; ++t) B[t] = N[t];
                }
            }
        }
    }
    __syncthreads();
    const int my_gene = int(ginfo[tid] & 0x7fffffffu) - 1;
    // genes no window covers (step > 1) keep 0.0 like numpy.zeros (crf/__init__.py:251)
    if (tid >= W - 1 && my_gene >= 0) a.p_out[my_gene] = fmin(__longlong_as_double(static_cast<long long>(best[tid])), 1.0);
}

// ---- row F: whole-contig marginals, one group per contig, CRFsuite's own sequential recursion ----
template <int LP>
__global__ void __launch_bounds__(kGT) gl_marginals_seq(GenArgs a) {
    __shared__ double vecs[kGT];
    constexpr int G = kGT / LP;
    const int j = threadIdx.x & (LP - 1), grp = threadIdx.x / LP;
    double *vec = vecs + grp * LP;
    const long long ci = static_cast<long long>(blockIdx.x) * G + grp;
    if (ci >= a.n_contigs) return;
    const int L = a.L;
    const int g0 = a.contig_ptr[ci], T = a.contig_ptr[ci + 1] - g0;
    if (T <= 0) {
        if (j == 0 && a.lognorm) a.lognorm[ci] = 0.0;
        return;
    }
    const bool on = j < L;
    const int jj = on ? j : 0;
    double mcol[LP], mrow[LP];
#pragma unroll
    for (int i = 0; i < LP; ++i) {
        const bool ok = i < L && on;
        mcol[i] = ok ? a.exp_trans[i * L + j] : 0.0;
        mrow[i] = ok ? a.exp_trans[j * L + i] : 0.0;
    }
    const double *E = a.E + static_cast<size_t>(g0) * L;
    double *alpha = a.alpha + static_cast<size_t>(g0) * L;
    double *scale = a.scale + g0;
    double lognorm = 0.0, v = 0.0, c = 1.0;
    double e = on ? E[jj] : 0.0;
    for (int t = 0; t < T; ++t) {
        const double e_next = (t + 1 < T && on) ? E[static_cast<size_t>(t + 1) * L + jj] : 0.0;
        const double sm = a.smax[g0 + t];
        if (t == 0) {
            v = e;
        } else {
            vec[j] = v;
            __builtin_amdgcn_wave_barrier();
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < LP; ++i) acc = fma(vec[i], mcol[i], acc);
            __builtin_amdgcn_wave_barrier();
            v = acc * e;
        }
        const double s = group_sum<LP>(v);
        c = s != 0.0 ? 1.0 / s : 1.0;
        v *= c;
        if (on) alpha[static_cast<size_t>(t) * L + j] = v;
        if (j == 0) scale[t] = c;
        lognorm += sm - log(c);
        e = e_next;
    }
    if (j == 0 && a.lognorm) a.lognorm[ci] = lognorm;
    // backward; alpha/scale of step t were written by this very lane / this group's lane 0
    double *marg = a.marg + static_cast<size_t>(g0) * L;
    double b = c;
    if (on) marg[static_cast<size_t>(T - 1) * L + j] = v * b / c;
    double al_p = (T >= 2 && on) ? alpha[static_cast<size_t>(T - 2) * L + j] : 0.0;
    double c_p = T >= 2 ? __shfl(j == 0 ? scale[T - 2] : 0.0, 0, LP) : 1.0;
    double e1 = (T >= 2 && on) ? E[static_cast<size_t>(T - 1) * L + jj] : 0.0;
    for (int t = T - 2; t >= 0; --t) {
        const double al_t = al_p, ct = c_p, et1 = e1;
        if (t > 0) {  // prefetch step t-1 while step t computes
            al_p = on ? alpha[static_cast<size_t>(t - 1) * L + j] : 0.0;
            c_p = __shfl(j == 0 ? scale[t - 1] : 0.0, 0, LP);
            e1 = on ? E[static_cast<size_t>(t) * L + jj] : 0.0;
        }
        vec[j] = b * et1;
        __builtin_amdgcn_wave_barrier();
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < LP; ++i) acc = fma(mrow[i], vec[i], acc);
        __builtin_amdgcn_wave_barrier();
        b = acc * ct;
        if (on) marg[static_cast<size_t>(t) * L + j] = al_t * b / ct;
    }
}

// ---- row V: whole-contig Viterbi, one group per contig ------------------------------------------
constexpr int kBackRows = 32;  // back-pointer rows walked from LDS per global round trip

template <int LP>
__global__ void __launch_bounds__(kGT) gl_viterbi_seq(GenArgs a) {
    __shared__ double vecs[kGT];
    __shared__ uint8_t rows[kGT * kBackRows];
    constexpr int G = kGT / LP;
    const int j = threadIdx.x & (LP - 1), grp = threadIdx.x / LP;
    double *vec = vecs + grp * LP;
    uint8_t *row = rows + grp * LP * kBackRows;
    const long long ci = static_cast<long long>(blockIdx.x) * G + grp;
    if (ci >= a.n_contigs) return;
    const int L = a.L;
    const int g0 = a.contig_ptr[ci], T = a.contig_ptr[ci + 1] - g0;
    if (T <= 0) {
        if (j == 0 && a.score) a.score[ci] = 0.0;
        return;
    }
    const bool on = j < L;
    const int jj = on ? j : 0;
    double tcol[LP];
#pragma unroll
    for (int i = 0; i < LP; ++i) tcol[i] = (i < L && on) ? a.trans[i * L + j] : 0.0;
    const double *st = a.state + static_cast<size_t>(g0) * L;
    uint8_t *back = a.back + static_cast<size_t>(g0) * L;
    double d = on ? st[jj] : -DBL_MAX;
    double s_next = (T > 1 && on) ? st[static_cast<size_t>(L) + jj] : 0.0;
    for (int t = 1; t < T; ++t) {
        const double s_t = s_next;
        if (t + 1 < T) s_next = on ? st[static_cast<size_t>(t + 1) * L + jj] : 0.0;
        vec[j] = d;
        __builtin_amdgcn_wave_barrier();
        double best = -DBL_MAX;
        int arg = -1;
#pragma unroll
        for (int i = 0; i < LP; ++i) {
            if (i < L) {
                const double s = vec[i] + tcol[i];
                if (best < s) {
                    best = s;
                    arg = i;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (on) back[static_cast<size_t>(t) * L + j] = static_cast<uint8_t>(arg < 0 ? 0 : arg);
        d = best + s_t;
    }
    // end label = first arg max; back-tracking by lane 0, rows staged through LDS in chunks
    vec[j] = d;
    __builtin_amdgcn_wave_barrier();
    int y = 0;
    {
        double best = -DBL_MAX;
        for (int i = 0; i < L; ++i)
            if (best < vec[i]) {
                best = vec[i];
                y = i;
            }
        if (j == 0 && a.score) a.score[ci] = best;
    }
    int8_t *yout = a.y + g0;
    if (j == 0) yout[T - 1] = static_cast<int8_t>(y);
    __threadfence();
    for (int hi = T - 1; hi >= 1; hi -= kBackRows) {
        const int lo = hi - kBackRows + 1 > 1 ? hi - kBackRows + 1 : 1;  // rows lo..hi
        const int nb = (hi - lo + 1) * L;
        const uint8_t *src = back + static_cast<size_t>(lo) * L;
        for (int k = j; k < nb; k += LP) row[k] = src[k];
        __builtin_amdgcn_wave_barrier();
        if (j == 0) {
            for (int t = hi; t >= lo; --t) {
                y = row[(t - lo) * L + y];
                yout[t - 1] = static_cast<int8_t>(y);
            }
        }
        y = __shfl(y, 0, LP);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- row V for 17 to 32 labels, batches of many contigs: one WAVE per contig ------------------------------------------
// The chunked recursions below pay for their parallelism inside a contig with L x the arithmetic (a chunk's transfer matrix is
// L forward recursions): 32 x at L = 32, where a batch of a thousand contigs has all the parallelism the chip can use in its
// contigs alone (C2's shape, L = 32: 1.42 ms chunked).  Here a contig is a wave walking it gene by gene, CRFsuite's own
// recursion ([EXT] crf1dc_viterbi: strict `<` update, first arg max).  The batch then takes as long as its LONGEST contig,
// and a lone wave issues one instruction every four cycles whatever the instruction is -- so the step is built to be short in
// instructions (~95 at L <= 32: 0.36 us), not in flops:
//   * lane (j, h), h = lane % H, holds target label j and the source labels [h PER, (h + 1) PER): H = 64 / LP partial maxima
//     per target in ONE quad of lanes, merged over h in ascending order by DPP quad permutes (ties keep the lower source
//     label, as the sequential loop does); a lane's maximum is a v_max_f64 chain, its first arg max a compare-select chain;
//   * no exec-mask juggling: lanes of labels that do not exist repeat label 0's work, every store is unconditional;
//   * the state scores of the next sixteen genes are loaded while the current sixteen are walked (one round trip to memory
//     per sixteen steps, hidden), the previous delta goes through one LDS slot per label;
//   * back-pointers leave as one dword per label and FOUR genes (a contig's region of `back`, dword-aligned inside it: the
//     regions of neighbouring contigs cannot meet); the last, incomplete quad never leaves its register;
//   * back-tracking reads sixteen quads per round (prefetched) and walks them on the scalar unit: the label is wave-uniform,
//     a step is v_readlane_b32 + s_bfe -- no dependent memory access.
// Measured (1 000 contigs, 0.22 M genes, longest 1 519; chunked -> wave): L = 32 1.42 -> 0.55 ms, L = 24 0.96 -> 0.55 ms;
// L = 16 0.30 -> 0.37 ms, L = 9 0.20 -> 0.36 ms: the plan takes this kernel above 16 labels when the longest contig is short
// against the batch (plan_run_viterbi), and tests force it from 9 labels up.
constexpr int kWaveK = 16;      // genes per staged block of state scores
constexpr int kWaveQuads = 16;  // back-pointer quads per back-tracking round

// lane l <- lane l ^ 1 / l ^ 2 inside its quad (DPP quad_perm: no LDS round trip)
template <int X>
__device__ __forceinline__ int gl_quad_xor(int v) {
    return __builtin_amdgcn_update_dpp(0, v, X == 1 ? 0xB1 : 0x4E, 0xF, 0xF, true);
}
template <int X>
__device__ __forceinline__ double gl_quad_xor(double v) {
    return __hiloint2double(gl_quad_xor<X>(__double2hiint(v)), gl_quad_xor<X>(__double2loint(v)));
}

template <int LP>
__global__ void __launch_bounds__(kGT) gl_viterbi_wave(GenArgs a) {
    constexpr int H = 64 / LP, PER = LP / H, K = kWaveK, Q = kWaveQuads;
    __shared__ double vecs[kGT / 64][LP];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    const long long ci = static_cast<long long>(blockIdx.x) * (kGT / 64) + wave;
    if (ci >= a.n_contigs) return;
    const int L = a.L, h = lane & (H - 1), j = lane / H;  // (the H lanes of a target label share a quad)
    const int g0 = __builtin_amdgcn_readfirstlane(a.contig_ptr[ci]);
    const int T = __builtin_amdgcn_readfirstlane(a.contig_ptr[ci + 1]) - g0;
    if (T <= 0) {
        if (lane == 0 && a.score) a.score[ci] = 0.0;
        return;
    }
    if (a.wave_tmax > 0 && T > a.wave_tmax) return;  // (the batch's long tail: decoded by the chunked kernels before this launch)
    // lanes of labels that do not exist (L < LP) repeat label 0's work: their stores then carry label 0's values, and no store
    // needs a test (a single wave issues an instruction every four cycles whatever it is: exec-mask juggling costs as much as
    // arithmetic here)
    const bool on = j < L;
    const int jj = on ? j : 0;
    double tcol[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = h * PER + k;
        tcol[k] = i < L ? a.trans[i * L + jj] : -DBL_MAX;  // (a label that does not exist never wins a strict `<`)
    }
    double *vec = vecs[wave];
    double *vout = vec + j;  // (slots of missing labels: written with label 0's value -- finite --, read against -DBL_MAX)
    const double *vin = vec + h * PER;
    const double *st = a.state + static_cast<size_t>(g0) * L + jj;
    // the contig's back-pointer quads: dwords [q * L + j], from the first dword boundary inside its T * L bytes.  Row t (1 ..
    // T - 1; gene 0 has no back-pointer) is byte (t - 1) & 3 of quad (t - 1) >> 2: the whole quads of the T - 1 rows take at
    // most (T - 1) * L bytes, the alignment at most 3 < L (this kernel serves L >= 9): every store stays inside the contig's
    // own T * L bytes, whatever kernel decodes its neighbours (split mode: the chunked kernels on the side stream).
    const size_t byte0 = (static_cast<size_t>(g0) * L + 3) & ~size_t(3);
    uint32_t *backq = reinterpret_cast<uint32_t *>(a.back + byte0) + jj;
    const bool writer = on && h == 0;
    double d = st[0];
    double nxt[K];
#pragma unroll
    for (int k = 0; k < K; ++k) nxt[k] = 1 + k < T ? st[static_cast<size_t>(1 + k) * L] : 0.0;
    uint32_t acc = 0;  // back-pointers of the quad in progress (byte t & 3 = row t)
    // one step of [EXT] crf1dc_viterbi for target label j: delta_t[j] = max_i (delta_{t-1}[i] + trans[i][j]) + state_t[j],
    // back-pointer = the FIRST source label that attains the maximum
    auto step = [&](const int t, const double s_t, const int sh) {
        *vout = d;  // (the H lanes of a label hold the same value)
        __builtin_amdgcn_wave_barrier();
        double sc[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) sc[u] = vin[u] + tcol[u];
        __builtin_amdgcn_wave_barrier();
        double best = sc[0];
#pragma unroll
        for (int u = 1; u < PER; ++u) best = fmax(best, sc[u]);
        int arg = h * PER + PER - 1;  // the first source label of the lane's range that attains its maximum
#pragma unroll
        for (int u = PER - 2; u >= 0; --u) arg = sc[u] == best ? h * PER + u : arg;
        // merge the partial maxima of the quad's lanes, the lower source range first (ties keep the lower label)
        if (H >= 2) {
            const double ob = gl_quad_xor<1>(best);
            const int oa = gl_quad_xor<1>(arg);
            const bool take = (lane & 1) ? !(ob < best) : (best < ob);
            best = take ? ob : best;
            arg = take ? oa : arg;
        }
        if (H >= 4) {
            const double ob = gl_quad_xor<2>(best);
            const int oa = gl_quad_xor<2>(arg);
            const bool take = (lane & 2) ? !(ob < best) : (best < ob);
            best = take ? ob : best;
            arg = take ? oa : arg;
        }
        d = best + s_t;
        const uint32_t bp = uint32_t(arg);
        acc = sh == 0 ? bp : (acc | (bp << sh));
        if (sh == 24) backq[static_cast<size_t>((t - 1) >> 2) * L] = acc;  // (every lane of the label, the same dword)
    };
    int tb = 1;
    for (; tb + K <= T; tb += K) {  // whole blocks of sixteen genes: nothing to test inside
        double cur[K];
#pragma unroll
        for (int k = 0; k < K; ++k) cur[k] = nxt[k];
        if (tb + K < T) {
            const double *sp = st + static_cast<size_t>(tb + K) * L;
#pragma unroll
            for (int k = 0; k < K; ++k) nxt[k] = tb + K + k < T ? sp[static_cast<size_t>(k) * L] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) step(tb + k, cur[k], (k & 3) * 8);  // (tb = 1 mod 16: row tb + k is byte k & 3 of its quad)
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (tb + k < T) step(tb + k, nxt[k], (k & 3) * 8);  // the last, partial block (wave-uniform tests)
    // end label = first arg max (every lane computes it: wave-uniform)
    *vout = d;
    __builtin_amdgcn_wave_barrier();
    int y = 0;
    {
        double best = -DBL_MAX;
        for (int i = 0; i < L; ++i) {
            const double v = vec[i];
            if (best < v) {
                best = v;
                y = i;
            }
        }
        if (lane == 0 && a.score) a.score[ci] = best;
    }
    y = __builtin_amdgcn_readfirstlane(y);
    int8_t *yout = a.y + g0;
    if (lane == 0) yout[T - 1] = static_cast<int8_t>(y);
    if (T == 1) return;
    __threadfence();  // (the quads this wave stored are read back through other lanes)
    // rows T - 1 .. 1; quad q = rows 4 q + 1 .. 4 q + 4; the top quad is `acc` when it is incomplete
    const int q_top = (T - 2) >> 2;
    const bool top_in_reg = ((T - 2) & 3) != 3;
    auto load_quads = [&](int q_hi, uint32_t (&w)[Q]) {  // quads q_hi - Q + 1 .. q_hi (those below 0: unused)
#pragma unroll
        for (int r = 0; r < Q; ++r) {
            const int q = q_hi - (Q - 1) + r;
            const bool stored = q >= 0 && (q < q_top || !top_in_reg);
            w[r] = (stored && writer) ? backq[static_cast<size_t>(q) * L] : 0u;
        }
    };
    uint32_t wn[Q];
    load_quads(q_top, wn);
    for (int q_hi = q_top; q_hi >= 0; q_hi -= Q) {
        uint32_t w[Q];
#pragma unroll
        for (int r = 0; r < Q; ++r) w[r] = wn[r];
        if (q_hi == q_top && top_in_reg) w[Q - 1] = acc;
        if (q_hi - Q >= 0) load_quads(q_hi - Q, wn);
#pragma unroll
        for (int r = Q - 1; r >= 0; --r) {
            const int q = q_hi - (Q - 1) + r;
            if (q < 0) continue;  // (wave-uniform)
#pragma unroll
            for (int b = 3; b >= 0; --b) {
                const int t = 4 * q + b + 1;
                if (t <= T - 1) {  // (wave-uniform)
                    const uint32_t word = uint32_t(__builtin_amdgcn_readlane(int(w[r]), y * H));  // row t as label y_t's lane holds it
                    y = int((word >> (8 * b)) & 0xffu);                                       // = label of gene t - 1
                    if (lane == 0) yout[t - 1] = static_cast<int8_t>(y);
                }
            }
        }
    }
}

// ---- row F for 17 to 32 labels, batches of many contigs: one WAVE per contig ------------------------------------------
// The same arrangement as gl_viterbi_wave for CRFsuite's scaled forward-backward recursion ([EXT] crf1dc_alpha_score /
// beta_score / marginal_point): lane (j, h) holds label j and half of the other index, the halves' partial sums meet by one DPP
// quad permute, the per-step sum over the labels (the scale factor's 1 / sum) is a DPP reduction inside the rows of sixteen
// lanes and four v_readlane across them, emissions / alpha / scale factors are staged eight or sixteen genes ahead.  A
// partial sum adds its terms in another order than the sequential loop does: results agree with the oracle to 1e-12, like
// the matrix-core kernels', not bit for bit.  log Z comes from the product of the scale factors, renormalised by frexp every
// sixteen genes (one log per contig instead of one per gene).
template <int X>
__device__ __forceinline__ double gl_row_dpp(double v) {  // X = 0: row_half_mirror, 1: row_mirror
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), X == 0 ? 0x141 : 0x140, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), X == 0 ? 0x141 : 0x140, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// sum over the labels of a value that every label's H lanes hold alike (lanes of missing labels hold 0): the lanes with one
// value of h are one lane per label, so the permutes that would add the H copies are left out; every lane gets the sum
template <int H>
__device__ __forceinline__ double gl_wave_label_sum(double v) {
    if (H < 2) v += gl_quad_xor<1>(v);
    if (H < 4) v += gl_quad_xor<2>(v);
    v += gl_row_dpp<0>(v);  // lanes 0-7 <-> 7-0 of every half row: with the quad sums in place, the half row's
    v += gl_row_dpp<1>(v);  // ... and the row's
    auto row = [&](int l) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
    };
    const double t = (row(0) + row(16)) + (row(32) + row(48));
    return t;
}

constexpr int kWaveFK = 8;  // genes per staged block of the forward-backward walk

template <int LP>
__global__ void __launch_bounds__(kGT) gl_marginals_wave(GenArgs a) {
    constexpr int H = 64 / LP, PER = LP / H, K = kWaveFK;
    __shared__ double vecs[kGT / 64][LP];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    const long long ci = static_cast<long long>(blockIdx.x) * (kGT / 64) + wave;
    if (ci >= a.n_contigs) return;
    const int L = a.L, h = lane & (H - 1), j = lane / H;
    const int g0 = __builtin_amdgcn_readfirstlane(a.contig_ptr[ci]);
    const int T = __builtin_amdgcn_readfirstlane(a.contig_ptr[ci + 1]) - g0;
    if (T <= 0) {
        if (lane == 0 && a.lognorm) a.lognorm[ci] = 0.0;
        return;
    }
    if (a.wave_tmax > 0 && T > a.wave_tmax) return;  // (the batch's long tail: the chunked kernels' next to this launch)
    // lanes of labels that do not exist repeat label 0's work (their stores carry label 0's values); only the sum over the
    // labels has to leave them out
    const bool on = j < L;
    const int jj = on ? j : 0;
    double mcol[PER], mrow[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int i = h * PER + u;
        mcol[u] = i < L ? a.exp_trans[i * L + jj] : 0.0;
        mrow[u] = i < L ? a.exp_trans[jj * L + i] : 0.0;
    }
    double *vec = vecs[wave];
    double *vout = vec + j;
    const double *vin = vec + h * PER;
    const double *E = a.E + static_cast<size_t>(g0) * L + jj;
    const double *smax = a.smax + g0;
    double *alpha = a.alpha + static_cast<size_t>(g0) * L + jj;
    double *scale = a.scale + g0;
    double *marg = a.marg + static_cast<size_t>(g0) * L + jj;
    // ---- forward ([EXT] crf1dc_alpha_score): alpha_t[j] = (sum_i alpha_{t-1}[i] M[i][j]) E_t[j], scaled to sum 1 -- at every
    // `period`-th gene and at the last one (the recursions hold for ANY positive scale factors as long as beta uses the same
    // ones and the last alpha sums to 1: a step whose factor is 1 skips the sum over the labels and two divisions; the host
    // allows period 4 when four unscaled steps cannot leave the range, as for gl_chunk_rows_mfma)
    const int period = a.rows_rescale_period;
    double v = 0.0, c = 1.0, prod = 1.0, sm_sum = 0.0;
    int ex_sum = 0;
    double en[K], sn[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        en[k] = k < T ? E[static_cast<size_t>(k) * L] : 0.0;
        sn[k] = k < T ? smax[k] : 0.0;
    }
    const size_t stride = size_t(L);
    const double *ep = E + static_cast<size_t>(K) * L;  // next emission block
    double *ap = alpha;                                  // alpha of the gene at hand
    auto fstep = [&](const int t, const double e_t, const double sm_t) {
        if (t > 0) {
            *vout = v;
            __builtin_amdgcn_wave_barrier();
            double acc = vin[0] * mcol[0];
#pragma unroll
            for (int u = 1; u < PER; ++u) acc = fma(vin[u], mcol[u], acc);
            __builtin_amdgcn_wave_barrier();
            if (H >= 2) acc += gl_quad_xor<1>(acc);
            if (H >= 4) acc += gl_quad_xor<2>(acc);
            v = acc * e_t;
        } else {
            v = e_t;
        }
        if (period == 1 || (t & (period - 1)) == period - 1 || t == T - 1) {  // (wave-uniform)
            const double s = gl_wave_label_sum<H>(on ? v : 0.0);
            c = s != 0.0 ? 1.0 / s : 1.0;
            v *= c;
            prod *= c;
        } else {
            c = 1.0;
        }
        *ap = v;
        ap += stride;
        scale[t] = c;
        sm_sum += sm_t;
    };
    for (int tb = 0; tb < T; tb += K) {
        double ec[K], sc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            ec[k] = en[k];
            sc[k] = sn[k];
        }
        if (tb + K < T) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                en[k] = tb + K + k < T ? ep[static_cast<size_t>(k) * L] : 0.0;
                sn[k] = tb + K + k < T ? smax[tb + K + k] : 0.0;
            }
            ep += static_cast<size_t>(K) * L;
        }
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (tb + k < T) fstep(tb + k, ec[k], sc[k]);
        int ex;  // (the product of the scale factors since the last renormalisation stays far inside the double range)
        prod = frexp(prod, &ex);
        ex_sum += ex;
    }
    if (lane == 0 && a.lognorm) a.lognorm[ci] = sm_sum - (log(prod) + double(ex_sum) * 0.6931471805599453);
    // ---- backward ([EXT] crf1dc_beta_score, marginal_point): beta_t[i] = (sum_k M[i][k] E_{t+1}[k] beta_{t+1}[k]) c_t,
    // p_t[j] = alpha_t[j] beta_t[j] / c_t.  alpha / scale of step t were stored by this very lane's label (or by label 0's
    // lanes for a missing label): the fence orders them before the loads below.
    __threadfence();
    double b = c;
    marg[static_cast<size_t>(T - 1) * L] = v * b / c;
    if (T == 1) return;
    // step t needs alpha_t, c_t, E_{t+1}; blocks of K steps t = hi, hi - 1, ... fetched one block ahead
    double an[K], cn[K], e1n[K];
    const double *al_p = alpha + static_cast<size_t>(T - 2) * L;  // alpha_t of the next block's first step
    const double *e1_p = E + static_cast<size_t>(T - 1) * L;      // E_{t+1} likewise
    double *mp = marg + static_cast<size_t>(T - 2) * L;
    auto fetch = [&](const int hi) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int t = hi - k;
            an[k] = t >= 0 ? *(al_p - static_cast<size_t>(k) * L) : 0.0;
            cn[k] = t >= 0 ? scale[t] : 1.0;
            e1n[k] = t >= 0 ? *(e1_p - static_cast<size_t>(k) * L) : 0.0;
        }
        al_p -= static_cast<size_t>(K) * L;  // (never dereferenced below the contig: the tests above)
        e1_p -= static_cast<size_t>(K) * L;
    };
    fetch(T - 2);
    for (int hi = T - 2; hi >= 0; hi -= K) {
        double ac[K], cc[K], e1c[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            ac[k] = an[k];
            cc[k] = cn[k];
            e1c[k] = e1n[k];
        }
        if (hi - K >= 0) fetch(hi - K);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int t = hi - k;
            if (t >= 0) {
                *vout = b * e1c[k];
                __builtin_amdgcn_wave_barrier();
                double acc = mrow[0] * vin[0];
#pragma unroll
                for (int u = 1; u < PER; ++u) acc = fma(mrow[u], vin[u], acc);
                __builtin_amdgcn_wave_barrier();
                if (H >= 2) acc += gl_quad_xor<1>(acc);
                if (H >= 4) acc += gl_quad_xor<2>(acc);
                const double ct = cc[k];
                if (__builtin_amdgcn_readfirstlane(__double2hiint(ct)) == 0x3ff00000 && __builtin_amdgcn_readfirstlane(__double2loint(ct)) == 0) {
                    b = acc;  // (a step without rescaling: c_t = 1)
                    *mp = ac[k] * b;
                } else {
                    b = acc * ct;
                    *mp = ac[k] * b / ct;
                }
                mp -= stride;
            }
        }
    }
}

// ==================================================================================================
// Long contigs, any number of labels ("matrix-product scan for C5").  The kernels
// above give a whole contig to ONE group of lanes: a 50 000-gene contig is a 50 000-step dependent chain
// (~50 ms).  Here a contig is cut into chunks of kChunk genes and the recursions become three short
// stages, each parallel over chunks or short in steps:
//   rows    one group per (chunk, row i): row i of the chunk's transfer matrix M_c = S_first .. S_last,
//           i.e. the forward recursion started from the unit vector e_i (sum-product with exact
//           power-of-two scaling and a per-row exponent; max-plus for Viterbi).  L x the work of the
//           sequential kernel, spread over n / kChunk groups;
//   vecs    one group per contig walks its CHUNKS (n / kChunk steps instead of n): v <- v M_c forwards
//           (alpha / delta entering every chunk), b <- M_c b backwards (beta leaving every chunk: the
//           backward step matrices are the same S_t);
//   replay  one group per chunk re-runs CRFsuite's own recursion inside the chunk from its entering
//           vector: alpha / scale / marginals, or back-pointers and labels.
// Values entering a chunk come from composed matrices, so they equal the strictly sequential ones
// whenever the arithmetic is exact (integer weights: ties included) and to rounding otherwise; marginals
// are normalised by their sum inside the chunk (CRFsuite divides by the scale factor, which is the same
// number up to rounding).
constexpr int kChunk = 64;

template <int LP, bool MAXPLUS>
__global__ void __launch_bounds__(kGT) gl_chunk_rows(GenArgs a) {
    __shared__ double vecs[kGT];
    constexpr int G = kGT / LP;
    const int j = threadIdx.x & (LP - 1), grp = threadIdx.x / LP;
    double *vec = vecs + grp * LP;
    const int L = a.L;
    const long long q = static_cast<long long>(blockIdx.x) * G + grp;
    if (q >= static_cast<long long>(a.n_chunks) * L) return;
    const int ci = int(q / L), i = int(q % L);
    const int g0 = a.ch_g0[ci], g1 = gl_chunk_end(a, ci);
    const int cfirst = a.contig_ptr[a.ch_contig[ci]];  // the contig's first gene has no transition into it
    const bool on = j < L;
    const int jj = on ? j : 0;
    const double ninf = -__builtin_huge_val();
    double mcol[LP];
#pragma unroll
    for (int k = 0; k < LP; ++k) mcol[k] = (k < L && on) ? (MAXPLUS ? a.trans[k * L + j] : a.exp_trans[k * L + j]) : (MAXPLUS ? ninf : 0.0);
    const double *src = MAXPLUS ? a.state : a.E;
    double v = MAXPLUS ? (j == i ? 0.0 : ninf) : (j == i ? 1.0 : 0.0);
    int ex = 0;
    double e = on ? src[static_cast<size_t>(g0) * L + jj] : (MAXPLUS ? ninf : 0.0);
    for (int t = g0; t < g1; ++t) {
        const double e_next = (t + 1 < g1 && on) ? src[static_cast<size_t>(t + 1) * L + jj] : (MAXPLUS ? ninf : 0.0);
        if (t == cfirst) {
            v = e;  // every row: the first gene forgets what entered
        } else {
            vec[j] = v;
            __builtin_amdgcn_wave_barrier();
            double acc = MAXPLUS ? ninf : 0.0;
#pragma unroll
            for (int k = 0; k < LP; ++k) {
                if (MAXPLUS) {
                    if (k < L) acc = fmax(acc, vec[k] + mcol[k]);
                } else {
                    acc = fma(vec[k], mcol[k], acc);
                }
            }
            __builtin_amdgcn_wave_barrier();
            v = MAXPLUS ? acc + e : acc * e;
        }
        if (!MAXPLUS) {  // exact power-of-two scaling of the row, exponent kept
            const double mx = group_max<LP>(on ? v : 0.0);
            if (mx > 0.0) {
                int e2;
                (void)frexp(mx, &e2);
                v = ldexp(v, -e2);
                ex += e2;
            }
        }
        e = e_next;
    }
    if (on) a.chM[(static_cast<size_t>(ci) * L + i) * L + j] = v;
    if (!MAXPLUS && j == 0) a.chEx[static_cast<size_t>(ci) * L + i] = ex;
}

// The same transfer matrices for 9 to 32 labels on the fp64 matrix cores.  The L rows of a chunk's matrix are L forward
// recursions from the unit vectors: sixteen of them at once are the columns of X^T in  X'^T = diag(E_t) M^T X^T  -- the step
// of gl_windowed_mfma with "sixteen windows" replaced by "sixteen start labels of one chunk": result register r of a lane IS
// its B operand of K-slice r in the next step, the slices of M^T are per-lane constants, nothing crosses lanes between steps.
// One wave per (chunk, group of sixteen rows); ceil(L / 16) * ceil(L / 4) MFMAs per gene and wave against L broadcast reads
// from LDS per lane and step in the lane-group kernel above.  A column's power-of-two rescaling (maximum over its labels:
// the lane's registers, then the four lanes that share the column) is taken every `period` steps (1, or 4 when four
// un-normalised steps cannot leave the range: 4 max|trans| < 600).  Summation order differs from gl_chunk_rows: 1e-12, not bits.
template <int TILES, int NS>
__global__ void __launch_bounds__(kGT) gl_chunk_rows_mfma(GenArgs a, const int period) {
    const int L = a.L;
    const int lane = threadIdx.x & 63, w = lane & 15, g = lane >> 4;
    const int groups = (L + 15) / 16;
    const long long wv = static_cast<long long>(blockIdx.x) * (kGT / 64) + (threadIdx.x >> 6);
    if (wv >= static_cast<long long>(a.n_chunks) * groups) return;
    const int ci = int(wv / groups), cg = int(wv % groups);
    const int g0 = a.ch_g0[ci], g1 = gl_chunk_end(a, ci);
    const int cfirst = a.contig_ptr[a.ch_contig[ci]];
    const int col = 16 * cg + w;  // the start label of this lane's column
    double Af[TILES][NS];         // slices of M^T: row = output label, column = summed label
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx) {
            const int row = 16 * t + w, k = 4 * sidx + g;
            Af[t][sidx] = (row < L && k < L) ? a.exp_trans[k * L + row] : 0.0;
        }
    auto label_of = [&](int t, int r) { return 16 * t + 4 * r + g; };
    gl_v4d D[TILES], E[TILES], En[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            D[t][r] = (label_of(t, r) == col && col < L) ? 1.0 : 0.0;
            E[t][r] = label_of(t, r) < L ? a.E[static_cast<size_t>(g0) * L + label_of(t, r)] : 0.0;
        }
    int ex = 0;
    for (int gi = g0; gi < g1; ++gi) {
        // (the next gene's emissions are requested before this gene's products)
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) En[t][r] = (gi + 1 < g1 && label_of(t, r) < L) ? a.E[static_cast<size_t>(gi + 1) * L + label_of(t, r)] : 0.0;
        if (gi == cfirst) {  // every row: the contig's first gene forgets what entered
#pragma unroll
            for (int t = 0; t < TILES; ++t) D[t] = col < L ? E[t] : gl_v4d{0.0, 0.0, 0.0, 0.0};
        } else {
            gl_v4d N[TILES];
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                N[t] = gl_v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int sidx = 0; sidx < NS; ++sidx) N[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Af[t][sidx], D[sidx / 4][sidx % 4], N[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < TILES; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) D[t][r] = N[t][r] * E[t][r];
        }
        if (period <= 1 || ((gi - g0) % period) == period - 1 || gi + 1 == g1) {  // exact power-of-two scaling of the column, exponent kept
            double mx = 0.0;
#pragma unroll
            for (int t = 0; t < TILES; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmax(mx, D[t][r]);
            mx = fmax(mx, __shfl_xor(mx, 16));
            mx = fmax(mx, __shfl_xor(mx, 32));
            if (mx > 0.0) {
                int e2;
                (void)frexp(mx, &e2);
#pragma unroll
                for (int t = 0; t < TILES; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) D[t][r] = ldexp(D[t][r], -e2);
                ex += e2;
            }
        }
#pragma unroll
        for (int t = 0; t < TILES; ++t) E[t] = En[t];
    }
    if (col < L) {
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (label_of(t, r) < L) a.chM[(static_cast<size_t>(ci) * L + col) * L + label_of(t, r)] = D[t][r];
        if (g == 0) a.chEx[static_cast<size_t>(ci) * L + col] = ex;
    }
}

// The walks over the chunks of a contig: forwards, the vector entering every chunk (normalised alpha / delta of the gene
// before it); backwards (marginals), beta of the last gene of every chunk up to a factor.  One group of lanes per contig
// and direction (blockIdx.y), n / chunk dependent steps.  A step is a few dozen instructions; what it waited for in
// round 2 was the load of the chunk's matrix (~0.7 us from L2/HBM per step: 41 us for the 64 chunks of the longest contig
// of a 1 000-contig batch, 0.5 ms for a 50 000-gene contig).  The matrices do not depend on the walk, so a lane fetches
// its column (row) of the next kWalkAhead chunks in one go and then takes those steps from registers.
template <int LP>
struct WalkAhead {
    static constexpr int B = LP <= 4 ? 8 : LP == 8 ? 4 : LP == 16 ? 2 : 1;
};

template <int LP, bool MAXPLUS>
__device__ __forceinline__ void chunk_walk_fwd(const GenArgs &a, long long ct, int j, double *vec) {
    constexpr int B = WalkAhead<LP>::B;
    const int L = a.L;
    const bool on = j < L;
    const double ninf = -__builtin_huge_val();
    const int c0 = a.cc_ptr[ct], c1 = a.cc_ptr[ct + 1];
    double v = MAXPLUS ? (j == 0 ? 0.0 : ninf) : (j == 0 ? 1.0 : 0.0);
    // (round 5: the columns of the NEXT B chunks are on their way while the current B steps are taken -- the round trip to
    // memory per batch of chunks was all a step waited for)
    double mn[B][LP];
    int en[B];
    auto fetch = [&](const int cb) {
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int c = cb + b;
            const bool live = on && c + 1 < c1;  // (nothing leaves the contig's last chunk)
            const double *M = a.chM + static_cast<size_t>(live ? c : c0) * L * L;
#pragma unroll
            for (int k = 0; k < LP; ++k) mn[b][k] = (live && k < L) ? M[static_cast<size_t>(k) * L + j] : (MAXPLUS ? ninf : 0.0);
            en[b] = (!MAXPLUS && live) ? a.chEx[static_cast<size_t>(c) * L + j] : 0;
        }
    };
    if (c0 < c1) fetch(c0);
    for (int cb = c0; cb < c1; cb += B) {
        double mc[B][LP];
        int ex[B];
#pragma unroll
        for (int b = 0; b < B; ++b) {
            ex[b] = en[b];
#pragma unroll
            for (int k = 0; k < LP; ++k) mc[b][k] = mn[b][k];
        }
        if (cb + B < c1) fetch(cb + B);
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int c = cb + b;
            if (c < c1) {
                if (on) a.chV[static_cast<size_t>(c) * L + j] = v;
                if (c + 1 < c1) {
                    double w = v;
                    if (!MAXPLUS) {  // rows carry their own power-of-two exponents
                        const double exm = group_max<LP>((on && v > 0.0) ? double(ex[b]) : -1e300);
                        w = (on && v > 0.0) ? ldexp(v, ex[b] - int(exm)) : 0.0;
                    }
                    vec[j] = w;
                    __builtin_amdgcn_wave_barrier();
                    double acc = MAXPLUS ? ninf : 0.0;
#pragma unroll
                    for (int k = 0; k < LP; ++k)
                        if (k < L) acc = MAXPLUS ? fmax(acc, vec[k] + mc[b][k]) : fma(vec[k], mc[b][k], acc);
                    __builtin_amdgcn_wave_barrier();
                    if (MAXPLUS) {
                        v = acc;
                    } else {
                        const double sm = group_sum<LP>(on ? acc : 0.0);
                        v = sm != 0.0 ? acc / sm : acc;
                    }
                }
            }
        }
    }
}

template <int LP>
__device__ __forceinline__ void chunk_walk_bwd(const GenArgs &a, long long ct, int j, double *vec) {
    constexpr int B = WalkAhead<LP>::B;
    const int L = a.L;
    const bool on = j < L;
    const int c0 = a.cc_ptr[ct], c1 = a.cc_ptr[ct + 1];
    double bt = 1.0;
    double mn[B][LP];
    int en[B];
    auto fetch = [&](const int cb) {  // (the rows of the next B chunks towards the front: on their way during the current steps)
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int c = cb - b;
            const bool live = on && c > c0;  // (nothing is in front of the contig's first chunk)
            const double *row = a.chM + (static_cast<size_t>(live ? c : c0) * L + (on ? j : 0)) * L;
#pragma unroll
            for (int k = 0; k < LP; ++k) mn[b][k] = (live && k < L) ? row[k] : 0.0;
            en[b] = live ? a.chEx[static_cast<size_t>(c) * L + j] : 0;
        }
    };
    if (c1 > c0) fetch(c1 - 1);
    for (int cb = c1 - 1; cb >= c0; cb -= B) {
        double mr[B][LP];
        int ex[B];
#pragma unroll
        for (int b = 0; b < B; ++b) {
            ex[b] = en[b];
#pragma unroll
            for (int k = 0; k < LP; ++k) mr[b][k] = mn[b][k];
        }
        if (cb - B >= c0) fetch(cb - B);
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int c = cb - b;
            if (c >= c0) {
                if (on) a.chB[static_cast<size_t>(c) * L + j] = bt;
                if (c > c0) {  // beta of the gene before the chunk = M_c beta: lane j takes row j
                    vec[j] = on ? bt : 0.0;
                    __builtin_amdgcn_wave_barrier();
                    double acc = 0.0;
#pragma unroll
                    for (int k = 0; k < LP; ++k)
                        if (k < L) acc = fma(mr[b][k], vec[k], acc);
                    __builtin_amdgcn_wave_barrier();
                    const double exm = group_max<LP>((on && acc > 0.0) ? double(ex[b]) : -1e300);
                    acc = (on && acc > 0.0) ? ldexp(acc, ex[b] - int(exm)) : 0.0;
                    const double mx = group_max<LP>(acc);
                    bt = mx > 0.0 ? acc / mx : 1.0;
                }
            }
        }
    }
}

template <int LP, bool MAXPLUS>
__global__ void __launch_bounds__(kGT) gl_chunk_vecs(GenArgs a) {
    __shared__ double vecs[kGT];
    constexpr int G = kGT / LP;
    const int j = threadIdx.x & (LP - 1), grp = threadIdx.x / LP;
    const long long ct = static_cast<long long>(blockIdx.x) * G + grp;
    if (ct >= a.n_contigs) return;
    // a contig without genes has no chunk, so none of the chunk kernels writes its log-partition / path score: 0, as the
    // contig-sequential kernels give it
    if (j == 0 && blockIdx.y == 0 && a.cc_ptr[ct + 1] == a.cc_ptr[ct] && a.wave_tmax == 0) {  // (split batches: the waves' contigs)
        if (!MAXPLUS && a.lognorm) a.lognorm[ct] = 0.0;
        if (MAXPLUS && a.score) a.score[ct] = 0.0;
    }
    if (MAXPLUS || blockIdx.y == 0)
        chunk_walk_fwd<LP, MAXPLUS>(a, ct, j, vecs + grp * LP);
    else
        chunk_walk_bwd<LP>(a, ct, j, vecs + grp * LP);
}

// CRFsuite's forward recursion inside a chunk, from the vector that enters it
template <int LP>
__global__ void __launch_bounds__(kGT) gl_chunk_fwd(GenArgs a) {
    __shared__ double vecs[kGT];
    constexpr int G = kGT / LP;
    const int j = threadIdx.x & (LP - 1), grp = threadIdx.x / LP;
    double *vec = vecs + grp * LP;
    const int L = a.L;
    const long long ci = static_cast<long long>(blockIdx.x) * G + grp;
    if (ci >= a.n_chunks) return;
    const int g0 = a.ch_g0[ci], g1 = gl_chunk_end(a, ci);
    const int cfirst = a.contig_ptr[a.ch_contig[ci]];
    const bool on = j < L;
    const int jj = on ? j : 0;
    double mcol[LP];
#pragma unroll
    for (int k = 0; k < LP; ++k) mcol[k] = (k < L && on) ? a.exp_trans[k * L + j] : 0.0;
    double v = on ? a.chV[static_cast<size_t>(ci) * L + j] : 0.0, z = 0.0;
    double e = on ? a.E[static_cast<size_t>(g0) * L + jj] : 0.0;
    for (int t = g0; t < g1; ++t) {
        const double e_next = (t + 1 < g1 && on) ? a.E[static_cast<size_t>(t + 1) * L + jj] : 0.0;
        if (t == cfirst) {
            v = e;
        } else {
            vec[j] = v;
            __builtin_amdgcn_wave_barrier();
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < LP; ++k) acc = fma(vec[k], mcol[k], acc);
            __builtin_amdgcn_wave_barrier();
            v = acc * e;
        }
        const double sm = group_sum<LP>(v);
        const double c = sm != 0.0 ? 1.0 / sm : 1.0;
        v *= c;
        if (on) a.alpha[static_cast<size_t>(t) * L + j] = v;
        if (j == 0) a.scale[t] = c;
        z += a.smax[t] - log(c);
        e = e_next;
    }
    if (j == 0) a.chZ[ci] = z;
}

// backward recursion inside a chunk + marginals
template <int LP>
__global__ void __launch_bounds__(kGT) gl_chunk_bwd(GenArgs a) {
    __shared__ double vecs[kGT];
    constexpr int G = kGT / LP;
    const int j = threadIdx.x & (LP - 1), grp = threadIdx.x / LP;
    double *vec = vecs + grp * LP;
    const int L = a.L;
    const long long ci = static_cast<long long>(blockIdx.x) * G + grp;
    if (ci >= a.n_chunks) return;
    const int g0 = a.ch_g0[ci], g1 = gl_chunk_end(a, ci);
    const bool on = j < L;
    const int jj = on ? j : 0;
    double mrow[LP];
#pragma unroll
    for (int k = 0; k < LP; ++k) mrow[k] = (k < L && on) ? a.exp_trans[j * L + k] : 0.0;
    if (a.lognorm) {  // the contig's log-partition from the chunks' partial sums: the group of its first chunk adds them up
        const int ct = a.ch_contig[ci];
        if (g0 == a.contig_ptr[ct]) {
            double z = 0.0;
            for (int c = a.cc_ptr[ct] + j; c < a.cc_ptr[ct + 1]; c += LP) z += a.chZ[c];
            z = group_sum<LP>(z);
            if (j == 0) a.lognorm[ct] = z;
        }
    }
    double b = on ? a.chB[static_cast<size_t>(ci) * L + j] : 0.0;
    for (int t = g1 - 1; t >= g0; --t) {
        const double al = on ? a.alpha[static_cast<size_t>(t) * L + j] : 0.0;
        const double x = al * b;
        const double sm = group_sum<LP>(x);
        if (on) a.marg[static_cast<size_t>(t) * L + j] = sm != 0.0 ? x / sm : 0.0;
        if (t > g0) {
            vec[j] = on ? b * a.E[static_cast<size_t>(t) * L + jj] : 0.0;
            __builtin_amdgcn_wave_barrier();
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < LP; ++k) acc = fma(mrow[k], vec[k], acc);
            __builtin_amdgcn_wave_barrier();
            b = acc * a.scale[t - 1];  // CRFsuite's own scaling keeps beta near 1
        }
    }
}

// Viterbi inside a chunk: back-pointers; the contig's end label and score in its last chunk
template <int LP>
__global__ void __launch_bounds__(kGT) gl_chunk_vit(GenArgs a) {
    __shared__ double vecs[kGT];
    constexpr int G = kGT / LP;
    const int j = threadIdx.x & (LP - 1), grp = threadIdx.x / LP;
    double *vec = vecs + grp * LP;
    const int L = a.L;
    const long long ci = static_cast<long long>(blockIdx.x) * G + grp;
    if (ci >= a.n_chunks) return;
    const int g0 = a.ch_g0[ci], g1 = gl_chunk_end(a, ci);
    const int ct = a.ch_contig[ci];
    const int cfirst = a.contig_ptr[ct], cend = a.contig_ptr[ct + 1];
    const bool on = j < L;
    const int jj = on ? j : 0;
    double tcol[LP];
#pragma unroll
    for (int k = 0; k < LP; ++k) tcol[k] = (k < L && on) ? a.trans[k * L + j] : 0.0;
    double d = on ? a.chV[static_cast<size_t>(ci) * L + j] : -DBL_MAX;
    for (int t = g0; t < g1; ++t) {
        const double s_t = on ? a.state[static_cast<size_t>(t) * L + jj] : 0.0;
        if (t == cfirst) {
            d = on ? s_t : -DBL_MAX;
            continue;
        }
        vec[j] = d;
        __builtin_amdgcn_wave_barrier();
        double best = -DBL_MAX;
        int arg = -1;
#pragma unroll
        for (int k = 0; k < LP; ++k) {
            if (k < L) {
                const double sc = vec[k] + tcol[k];
                if (best < sc) {
                    best = sc;
                    arg = k;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (on) a.back[static_cast<size_t>(t) * L + j] = static_cast<uint8_t>(arg < 0 ? 0 : arg);
        d = best + s_t;
    }
    if (g1 == cend) {  // first arg max of the final scores
        vec[j] = d;
        __builtin_amdgcn_wave_barrier();
        if (j == 0) {
            double best = -DBL_MAX;
            int y = 0;
            for (int k = 0; k < L; ++k)
                if (best < vec[k]) {
                    best = vec[k];
                    y = k;
                }
            a.chY[ci] = static_cast<int8_t>(y);
            if (a.score) a.score[ct] = best;
        }
    }
}

// per chunk: label of its last gene -> label of the gene before the chunk (lane y follows the back-pointers)
template <int LP>
__global__ void __launch_bounds__(kGT) gl_chunk_maps(GenArgs a) {
    constexpr int G = kGT / LP;
    const int j = threadIdx.x & (LP - 1), grp = threadIdx.x / LP;
    const int L = a.L;
    const long long ci = static_cast<long long>(blockIdx.x) * G + grp;
    if (ci >= a.n_chunks || j >= L) return;
    const int g0 = a.ch_g0[ci], g1 = gl_chunk_end(a, ci);
    if (g0 == a.contig_ptr[a.ch_contig[ci]]) return;  // nothing before a contig's first chunk
    int y = j;
    for (int t = g1 - 1; t >= g0; --t) y = a.back[static_cast<size_t>(t) * L + y];
    a.chMap[static_cast<size_t>(ci) * L + j] = static_cast<uint8_t>(y);
}

// per contig: the label of every chunk's last gene, back to front (n / chunk dependent steps).  One wave per contig:
// the maps of up to kEndsTile / L chunks are fetched into LDS in one go (they do not depend on the walk), then lane 0
// follows them there -- a step is an LDS read (~50 ns) instead of a dependent global load (~0.7 us).
constexpr int kEndsTile = 2048;  // bytes of chunk maps per wave and round
__global__ void __launch_bounds__(kGT) gl_chunk_ends(GenArgs a) {
    __shared__ uint8_t maps[(kGT / 64) * kEndsTile];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long ct = static_cast<long long>(blockIdx.x) * (kGT / 64) + wave;
    if (ct >= a.n_contigs) return;
    const int L = a.L;
    const int c0 = a.cc_ptr[ct], c1 = a.cc_ptr[ct + 1];
    if (c1 <= c0) return;
    uint8_t *mp = maps + wave * kEndsTile;
    const int T = kEndsTile / L;  // chunks per round
    int y = a.chY[c1 - 1];
    for (int hi = c1 - 1; hi > c0; hi -= T) {  // chunks (lo, hi] map their last label to the label before them
        const int lo = hi - T > c0 ? hi - T : c0;
        const uint8_t *src = a.chMap + static_cast<size_t>(lo + 1) * L;
        const int bytes = (hi - lo) * L;
        for (int o = lane; o < bytes; o += 64) mp[o] = src[o];
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            for (int c = hi; c > lo; --c) {
                y = mp[(c - lo - 1) * L + y];
                a.chY[c - 1] = static_cast<int8_t>(y);
            }
        }
        y = __shfl(y, 0);
        __builtin_amdgcn_wave_barrier();
    }
}

// per chunk: labels of its genes from the label of its last one
__global__ void __launch_bounds__(kGT) gl_chunk_backtrack(GenArgs a) {
    const long long ci = static_cast<long long>(blockIdx.x) * kGT + threadIdx.x;
    if (ci >= a.n_chunks) return;
    const int g0 = a.ch_g0[ci], g1 = gl_chunk_end(a, ci);
    if (g1 <= g0) return;
    int y = a.chY[ci];
    a.y[g1 - 1] = static_cast<int8_t>(y);
    for (int t = g1 - 1; t > g0; --t) {
        y = a.back[static_cast<size_t>(t) * a.L + y];
        a.y[t - 1] = static_cast<int8_t>(y);
    }
}

template <int LP>
hipError_t launch_chunked(int what, const GenArgs &a, hipStream_t stream) {
    constexpr int G = kGT / LP;
    // GECCO_CRF_GENERAL_ROWS=groups: the lane-group kernel for every label count (A/B runs, tests)
    static const bool rows_groups = [] {
        const char *env = std::getenv("GECCO_CRF_GENERAL_ROWS");
        return env && env[0] == 'g';
    }();
    const int rows_period = a.rows_rescale_period;
    auto blocks = [](long long items, int per) { return dim3(unsigned((items + per - 1) / per)); };
    const long long rows = static_cast<long long>(a.n_chunks) * a.L;
    if (a.n_chunks <= 0) return hipSuccess;
    if (what == 2) {
        if (LP >= 16 && a.L > 8 && !rows_groups) {  // 9 to 32 labels: sixteen rows of a chunk's transfer matrix per wave on the matrix cores
            const int groups = (a.L + 15) / 16, ns = (a.L + 3) / 4;
            const dim3 grid = blocks(static_cast<long long>(a.n_chunks) * groups, kGT / 64);
            switch (ns) {
            case 3: hipLaunchKernelGGL((gl_chunk_rows_mfma<1, 3>), grid, dim3(kGT), 0, stream, a, rows_period); break;
            case 4: hipLaunchKernelGGL((gl_chunk_rows_mfma<1, 4>), grid, dim3(kGT), 0, stream, a, rows_period); break;
            case 5: hipLaunchKernelGGL((gl_chunk_rows_mfma<2, 5>), grid, dim3(kGT), 0, stream, a, rows_period); break;
            case 6: hipLaunchKernelGGL((gl_chunk_rows_mfma<2, 6>), grid, dim3(kGT), 0, stream, a, rows_period); break;
            case 7: hipLaunchKernelGGL((gl_chunk_rows_mfma<2, 7>), grid, dim3(kGT), 0, stream, a, rows_period); break;
            default: hipLaunchKernelGGL((gl_chunk_rows_mfma<2, 8>), grid, dim3(kGT), 0, stream, a, rows_period); break;
            }
        } else {
            hipLaunchKernelGGL((gl_chunk_rows<LP, false>), blocks(rows, G), dim3(kGT), 0, stream, a);
        }
        hipLaunchKernelGGL((gl_chunk_vecs<LP, false>), dim3(blocks(a.n_contigs, G).x, 2), dim3(kGT), 0, stream, a);
        hipLaunchKernelGGL(gl_chunk_fwd<LP>, blocks(a.n_chunks, G), dim3(kGT), 0, stream, a);
        hipLaunchKernelGGL(gl_chunk_bwd<LP>, blocks(a.n_chunks, G), dim3(kGT), 0, stream, a);
    } else {
        hipLaunchKernelGGL((gl_chunk_rows<LP, true>), blocks(rows, G), dim3(kGT), 0, stream, a);
        hipLaunchKernelGGL((gl_chunk_vecs<LP, true>), blocks(a.n_contigs, G), dim3(kGT), 0, stream, a);
        hipLaunchKernelGGL(gl_chunk_vit<LP>, blocks(a.n_chunks, G), dim3(kGT), 0, stream, a);
        hipLaunchKernelGGL(gl_chunk_maps<LP>, blocks(a.n_chunks, G), dim3(kGT), 0, stream, a);
        hipLaunchKernelGGL(gl_chunk_ends, blocks(a.n_contigs, kGT / 64), dim3(kGT), 0, stream, a);
        hipLaunchKernelGGL(gl_chunk_backtrack, blocks(a.n_chunks, kGT), dim3(kGT), 0, stream, a);
    }
    return hipGetLastError();
}

template <int LP>
hipError_t launch_lp(int what, const GenArgs &a, hipStream_t stream) {
    constexpr int G = kGT / LP;
    auto blocks = [](long long items, int per) { return dim3(unsigned((items + per - 1) / per)); };
    switch (what) {
    case 0:  // state scores
        if (a.n_genes > 0)
            hipLaunchKernelGGL(gl_state<LP>, blocks(a.n_genes, G), dim3(kGT), 0, stream, a.gene_ptr, a.attr_id, a.wtab, a.L,
                               a.A, a.n_genes, a.state, a.E, a.smax);
        break;
    case 1: {
        const size_t lds = (size_t(G) * a.W * (LP + 1) + kGT) * sizeof(double);
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gl_windowed<LP>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
            if (e != hipSuccess) return e;
        }
        if (a.S > 0) hipLaunchKernelGGL(gl_windowed<LP>, blocks(a.S, G), dim3(kGT), lds, stream, a);
        break;
    }
    case 2:
        if (a.n_contigs > 0) hipLaunchKernelGGL(gl_marginals_seq<LP>, blocks(a.n_contigs, G), dim3(kGT), 0, stream, a);
        break;
    case 3:
        if (a.n_contigs > 0) hipLaunchKernelGGL(gl_viterbi_seq<LP>, blocks(a.n_contigs, G), dim3(kGT), 0, stream, a);
        break;
    }
    return hipGetLastError();
}

hipError_t launch_any(int what, const GenArgs &a, hipStream_t stream) {
    if (a.L <= 0 || a.L > kGenMaxL) return hipErrorNotSupported;
    if ((what == 2 || what == 3) && a.n_chunks > 0) {  // long contigs in the batch: chunked recursions
        if (a.L <= 2) return launch_chunked<2>(what, a, stream);
        if (a.L <= 4) return launch_chunked<4>(what, a, stream);
        if (a.L <= 8) return launch_chunked<8>(what, a, stream);
        if (a.L <= 16) return launch_chunked<16>(what, a, stream);
        return launch_chunked<32>(what, a, stream);
    }
    if (a.L <= 2) return launch_lp<2>(what, a, stream);
    if (a.L <= 4) return launch_lp<4>(what, a, stream);
    if (a.L <= 8) return launch_lp<8>(what, a, stream);
    if (a.L <= 16) return launch_lp<16>(what, a, stream);
    return launch_lp<32>(what, a, stream);
}

}  // namespace

hipError_t launch_gen_state(const GenArgs &a, hipStream_t stream) { return launch_any(0, a, stream); }
int gen_small_tile_out(int W) { return kSmallNT - (W - 1); }

// the lane-per-window kernel takes 3 to 8 labels (2 too: tests), windows of up to 32 genes (20 beyond 4 labels: W doubles of
// alpha per lane next to four L-vectors) and transition weights whose spread cannot take W - 1 un-normalised steps out
// of the range
bool gen_small_ok(int L, int W, const double *trans_host) {
    // (9 to 32 labels: the matrix-core kernel, same geometry and range guard, windows of up to 32 genes)
    if (L < 2 || L > kGenMaxL || W < 1 || W > (L <= 4 || L > 8 ? 32 : 20) || !trans_host) return false;
    double lo = trans_host[0], hi = trans_host[0];
    for (int i = 0; i < L * L; ++i) {
        lo = trans_host[i] < lo ? trans_host[i] : lo;
        hi = trans_host[i] > hi ? trans_host[i] : hi;
    }
    return hi - lo == hi - lo && (hi - lo) * double(W - 1) < 600.0;  // (finite)
}

hipError_t launch_gen_windowed_small(const GenArgs &a, const double *trans_host, const int4 *d_tile_desc, int ntiles,
                                     hipStream_t stream) {
    if (ntiles <= 0) return hipSuccess;
    const int L = a.L;
    SmallTrans T{};
    double mx = trans_host[0];
    for (int i = 0; i < L * L; ++i) mx = trans_host[i] > mx ? trans_host[i] : mx;
    if (L > 8) {  // sixteen windows per wave on the fp64 matrix cores
        const int wmax = a.W <= 20 ? 20 : 32, ns = (L + 3) / 4;
        const size_t lds = size_t(L) * gl_mfma_stride(wmax) * 8 + size_t(kMfmaNT + wmax) * 12;
        hipError_t attr_rc = hipSuccess;
#define GL_MFMA(TT, NN)                                                                                                              \
    do {                                                                                                                             \
        if (wmax == 20) {                                                                                                            \
            attr_rc = hipFuncSetAttribute(reinterpret_cast<const void *>(&gl_windowed_mfma<TT, NN, 20>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)); \
            if (attr_rc != hipSuccess) return attr_rc; /* (a device whose LDS cannot hold L label rows of a tile: fail loudly) */ \
            hipLaunchKernelGGL((gl_windowed_mfma<TT, NN, 20>), dim3(ntiles), dim3(kMfmaNT), lds, stream, a, mx, d_tile_desc);        \
        } else {                                                                                                                     \
            attr_rc = hipFuncSetAttribute(reinterpret_cast<const void *>(&gl_windowed_mfma<TT, NN, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)); \
            if (attr_rc != hipSuccess) return attr_rc;                                                                               \
            hipLaunchKernelGGL((gl_windowed_mfma<TT, NN, 32>), dim3(ntiles), dim3(kMfmaNT), lds, stream, a, mx, d_tile_desc);        \
        }                                                                                                                            \
    } while (0)
        switch (ns) {
        case 3: GL_MFMA(1, 3); break;
        case 4: GL_MFMA(1, 4); break;
        case 5: GL_MFMA(2, 5); break;
        case 6: GL_MFMA(2, 6); break;
        case 7: GL_MFMA(2, 7); break;
        case 8: GL_MFMA(2, 8); break;
        default: return hipErrorNotSupported;
        }
#undef GL_MFMA
        return hipGetLastError();
    }
    int perm[8];
    perm[0] = a.label;
    for (int j = 1; j < L; ++j) perm[j] = j <= a.label ? j - 1 : j;
    for (int i = 0; i < L; ++i)
        for (int j = 0; j < L; ++j) T.m[i * L + j] = exp(trans_host[perm[i] * L + perm[j]] - mx);
    const dim3 grid(ntiles), block(kSmallNT);
#define GL_SMALL(LL, WW) hipLaunchKernelGGL((gl_windowed_small<LL, WW>), grid, block, 0, stream, a, T, d_tile_desc)
    const bool w20 = a.W <= 20;
    switch (L) {
    case 2: if (w20) GL_SMALL(2, 20); else GL_SMALL(2, 32); break;
    case 3: if (w20) GL_SMALL(3, 20); else GL_SMALL(3, 32); break;
    case 4: if (w20) GL_SMALL(4, 20); else GL_SMALL(4, 32); break;
    case 5: GL_SMALL(5, 20); break;
    case 6: GL_SMALL(6, 20); break;
    case 7: GL_SMALL(7, 20); break;
    case 8: GL_SMALL(8, 20); break;
    default: return hipErrorNotSupported;
    }
#undef GL_SMALL
    return hipGetLastError();
}

hipError_t launch_gen_windowed(const GenArgs &a, hipStream_t stream) {
    if (a.W > kGenMaxW) return hipErrorNotSupported;
    return launch_any(1, a, stream);
}
hipError_t launch_gen_marginals(const GenArgs &a, hipStream_t stream) { return launch_any(2, a, stream); }
hipError_t launch_gen_viterbi(const GenArgs &a, hipStream_t stream) { return launch_any(3, a, stream); }
hipError_t launch_gen_marginals_wave(const GenArgs &a, hipStream_t stream) {
    if (a.L <= 8 || a.L > kGenMaxL) return hipErrorNotSupported;
    if (a.n_contigs <= 0) return hipSuccess;
    const dim3 grid(unsigned((a.n_contigs + kGT / 64 - 1) / (kGT / 64)));
    if (a.L <= 16)
        hipLaunchKernelGGL(gl_marginals_wave<16>, grid, dim3(kGT), 0, stream, a);
    else
        hipLaunchKernelGGL(gl_marginals_wave<32>, grid, dim3(kGT), 0, stream, a);
    return hipGetLastError();
}
hipError_t launch_gen_viterbi_wave(const GenArgs &a, hipStream_t stream) {
    if (a.L <= 8 || a.L > kGenMaxL) return hipErrorNotSupported;
    if (a.n_contigs <= 0) return hipSuccess;
    const dim3 grid(unsigned((a.n_contigs + kGT / 64 - 1) / (kGT / 64)));
    if (a.L <= 16)
        hipLaunchKernelGGL(gl_viterbi_wave<16>, grid, dim3(kGT), 0, stream, a);
    else
        hipLaunchKernelGGL(gl_viterbi_wave<32>, grid, dim3(kGT), 0, stream, a);
    return hipGetLastError();
}
int gen_chunk_genes() { return kChunk; }

}  // namespace gecco
