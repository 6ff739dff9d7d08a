// Reference-bits mode: windowed marginals of a 2-label model in CRFsuite's OWN operation order, so that the probabilities --
// and with them genes.tsv / features.tsv / clusters.tsv -- come out with the reference's bits, not merely within 1e-12 of them.
//
// The fast kernels (crf_kernels.hip) reorganise the arithmetic (ratio form, one reciprocal per window, a table-driven exp): their
// results sit within a few ulps of CRFsuite's (<= 13 ulps on the BGC0001866 fixture, where every one of the 48 printed
// probabilities differs in its last digits from the reference's files).  The reference's own acceptance test compares whole
// files (/root/reference/galaxy/gecco.xml:83-111), so this mode restates [EXT] crf1dc_exp_state / crf1dc_alpha_score /
// crf1dc_beta_score / crf1dc_marginal_point literally, as oracle/crf_oracle.c does (separate multiply and add roundings, the
// per-step 1 / sum scaling as an IEEE division, alpha * beta / scale) around the window loop of
// /root/reference/gecco/crf/__init__.py:244-258 -- with ONE substitution: libm's exp becomes the correctly rounded exp
// (crf_exact_exp.hpp), which is what libm returns for all but ~0.07 % of arguments (glibc: 0.51 ulp).  exp(trans) comes from
// the host's libm, as in the reference.  One lane per window start, alpha and the scale factors of a window in registers
// (W <= 32), per-gene maximum by maximum on the bit pattern (probabilities are non-negative: order-independent, exact) in LDS.
// Selected per session / plan (gecco_crf_session_set_reference_bits,
// GECCO_CRF_REFERENCE_BITS=1).
#include "crf_device.hpp"
#include "crf_exact_exp.hpp"

namespace gecco {
namespace {

constexpr int kRefT = 256;
constexpr int kRefMaxW = 32;
constexpr int kRefOwnExpTiles = 32;  // up to here (~7 500 slots) the tiles exponentiate their own scores: one launch

#pragma clang fp contract(off)

__device__ inline double2 exp_state_of(const int32_t *__restrict__ gene_ptr, const int32_t *__restrict__ attr_id,
                                       const double2 *__restrict__ wtab01, int n_attrs, int g) {
    double s0 = 0.0, s1 = 0.0;
    for (int k = gene_ptr[g]; k < gene_ptr[g + 1]; ++k) {
        const int a = attr_id[k];
        if (unsigned(a) < unsigned(n_attrs)) {  // (names CRFsuite does not know carry no weight)
            const double2 w = wtab01[a];
            s0 += w.x;
            s1 += w.y;
        }
    }
    return make_double2(ddx::exp_correctly_rounded(s0), ddx::exp_correctly_rounded(s1));
}

// exp of the state scores of every gene: E[g] = (exp(s[g][0]), exp(s[g][1])), s summed attribute by attribute in CSR order
// ([EXT] crf1dt_state_score: state[t][y] += w[a][y] * 1.0)
__global__ void __launch_bounds__(kRefT) ref_exp_states(const int32_t *__restrict__ gene_ptr, const int32_t *__restrict__ attr_id,
                                                        const double2 *__restrict__ wtab01, int n_attrs, int n_genes, double2 *__restrict__ E) {
    const int g = blockIdx.x * kRefT + threadIdx.x;
    if (g >= n_genes) return;
    E[g] = exp_state_of(gene_ptr, attr_id, wtab01, n_attrs, g);
}

struct RefArgs {
    const double2 *E;          // [n_genes] exp of the state scores, label order (OWN_EXP: unused)
    const int32_t *gene_ptr, *attr_id;  // OWN_EXP: the tiles exponentiate their own slots' scores
    const double2 *wtab01;
    int32_t n_attrs;
    const int32_t *c_slot, *c_gene, *c_n;
    const uint64_t *start_bits;
    double *p_out;             // device or host memory; skipped contigs NaN (the caller's)
    int32_t K, S, W, label;
    double t00, t01, t10, t11; // exp(trans), host libm
};

// A tile = 256 window starts; it OWNS the 256 - (W - 1) slots every one of whose windows starts inside it (the W - 1 starts in
// front are recomputed by the tile before: 8 % more arithmetic at W = 20, and no tile ever needs another's result).  The per-gene
// maximum goes through LDS (one returnless ds_max_u64 per window position, conflict-free: lane i hits slot i + t; probabilities
// are non-negative, so the order of their bit patterns is theirs) and every gene is stored once, by its owner: no atomics on
// memory, no zeroed array (a gene no window covers gets its 0.0 here: numpy.zeros, crf/__init__.py:251), and p may live in host
// memory.  The exponentials of the tile's slots (and the slot -> gene map) are staged in LDS once, padding items as (1, 1).
// OWN_EXP (a handful of tiles: one launch instead of two, the W - 1 front slots exponentiated twice).
// WT: compile-time bound of the window size (20 = GECCO's own: the registers of twelve steps fewer buy another wave per SIMD)
template <bool OWN_EXP, int WT>
__global__ void __launch_bounds__(kRefT, WT <= 20 ? 5 : 2) crf_windowed_reference_l2(const RefArgs A) {
    __shared__ double2 Es[kRefT + WT];
    __shared__ unsigned long long best[kRefT + WT];
    __shared__ int gslot[kRefT + WT];
    const int W = A.W;
    const int out = kRefT - (W - 1);                  // slots this tile owns
    const int own0 = blockIdx.x * out;                // the first of them
    const int q0 = own0 - (W - 1);                    // slot of lane 0's window start, and of LDS index 0
    const int q = q0 + int(threadIdx.x);
    for (int i = threadIdx.x; i < kRefT + W - 1; i += kRefT) {
        const int s = q0 + i;
        int g = -1;
        if (s >= 0 && s < A.S) {
            int lo = 0, hi = A.K - 1;  // contig of this slot: largest k with c_slot[k] <= s
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (A.c_slot[mid] <= s) lo = mid; else hi = mid - 1;
            }
            const int s0 = A.c_slot[lo], np = A.c_slot[lo + 1] - s0, n = A.c_n[lo];
            const int gl = s - s0 - ((np - n) >> 1);  // delta // 2 empty items in front (crf/__init__.py:227)
            if (gl >= 0 && gl < n) g = A.c_gene[lo] + gl;
        }
        gslot[i] = g;
        // a padding item has no attribute: state 0, exp 1
        Es[i] = g < 0 ? make_double2(1.0, 1.0) : OWN_EXP ? exp_state_of(A.gene_ptr, A.attr_id, A.wtab01, A.n_attrs, g) : A.E[g];
        best[i] = 0ull;
    }
    __syncthreads();
    const bool mine = q >= 0 && q < A.S && ((A.start_bits[q >> 6] >> (q & 63)) & 1ull);
    if (mine) {
        const int base = threadIdx.x;
        double al[WT], sc[WT];  // alpha of the asked label, scale factors
        double p0, p1;
        // ---- [EXT] crf1dc_alpha_score
        {
            const double2 e = Es[base];
            double x0 = e.x, x1 = e.y;
            const double sum = x0 + x1;
            const double c = (sum != 0.) ? 1. / sum : 1.;
            x0 *= c;
            x1 *= c;
            p0 = x0;
            p1 = x1;
            al[0] = A.label ? x1 : x0;
            sc[0] = c;
        }
#pragma unroll
        for (int t = 1; t < WT; ++t) {
            if (t < W) {
                const double2 e = Es[base + t];
                double x0 = p0 * A.t00, x1 = p0 * A.t01;  // cur[j] = 0 + prev[0] * trans[0][j]
                x0 = x0 + p1 * A.t10;                     //        + prev[1] * trans[1][j]
                x1 = x1 + p1 * A.t11;
                x0 *= e.x;
                x1 *= e.y;
                const double sum = x0 + x1;
                const double c = (sum != 0.) ? 1. / sum : 1.;
                x0 *= c;
                x1 *= c;
                p0 = x0;
                p1 = x1;
                al[t] = A.label ? x1 : x0;
                sc[t] = c;
            }
        }
        // ---- [EXT] crf1dc_beta_score + crf1dc_marginal_point, back to front
        double b0 = 0.0, b1 = 0.0;
#pragma unroll
        for (int t = WT - 1; t >= 0; --t) {
            if (t < W) {
                if (t == W - 1) {
                    b0 = b1 = sc[t];
                } else {
                    const double2 e = Es[base + t + 1];
                    const double r0 = b0 * e.x, r1 = b1 * e.y;  // row[j] = next[j] * exp_state[t + 1][j]
                    double y0 = A.t00 * r0, y1 = A.t10 * r0;    // s = 0 + trans[i][0] * row[0]
                    y0 = y0 + A.t01 * r1;                       //       + trans[i][1] * row[1]
                    y1 = y1 + A.t11 * r1;
                    b0 = y0 * sc[t];
                    b1 = y1 * sc[t];
                }
                const double m = (al[t] * (A.label ? b1 : b0)) / sc[t];  // alpha * beta / scale
                atomicMax(&best[base + t], static_cast<unsigned long long>(__double_as_longlong(m)));
            }
        }
    }
    __syncthreads();
    for (int i = (W - 1) + int(threadIdx.x); i < (W - 1) + out; i += kRefT) {  // the owned slots
        const int g = gslot[i];
        if (g >= 0) A.p_out[g] = __longlong_as_double(static_cast<long long>(best[i]));
    }
}

}  // namespace

bool reference_bits_ok(int L, int W) { return L == 2 && W >= 1 && W <= kRefMaxW; }

size_t reference_scratch_bytes(int n_genes) { return (size_t(n_genes) + 1) * sizeof(double2); }

// p_out: every gene of slot space is stored once (skipped contigs: NaN, filled by the caller); wtab01: the weight pairs in LABEL order (w[a][0], w[a][1]);
// exp_trans_host: exp(trans[i][j]) row-major, from the host's libm; `scratch`: reference_scratch_bytes(n_genes)
hipError_t launch_windowed_reference(const WinArgs &w, const double2 *wtab01, const double *exp_trans_host, void *scratch, hipStream_t stream) {
    if (!reference_bits_ok(w.L, w.W)) return hipErrorNotSupported;
    if (w.n_genes <= 0 || w.S <= 0) return hipSuccess;
    double2 *E = static_cast<double2 *>(scratch);
    const int out = kRefT - (w.W - 1), tiles = (w.S + out - 1) / out;
    const bool own_exp = tiles <= kRefOwnExpTiles;
    if (!own_exp)
        hipLaunchKernelGGL(ref_exp_states, dim3((w.n_genes + kRefT - 1) / kRefT), dim3(kRefT), 0, stream, w.gene_ptr, w.attr_id, wtab01, w.A,
                           w.n_genes, E);
    RefArgs a{};
    a.E = E;
    a.gene_ptr = w.gene_ptr;
    a.attr_id = w.attr_id;
    a.wtab01 = wtab01;
    a.n_attrs = w.A;
    a.c_slot = w.c_slot;
    a.c_gene = w.c_gene;
    a.c_n = w.c_n;
    a.start_bits = w.start_bits;
    a.p_out = w.p_out;
    a.K = w.K;
    a.S = w.S;
    a.W = w.W;
    a.label = w.label;
    a.t00 = exp_trans_host[0];
    a.t01 = exp_trans_host[1];
    a.t10 = exp_trans_host[2];
    a.t11 = exp_trans_host[3];
    if (own_exp)
        if (w.W <= 20)
            hipLaunchKernelGGL((crf_windowed_reference_l2<true, 20>), dim3(tiles), dim3(kRefT), 0, stream, a);
        else
            hipLaunchKernelGGL((crf_windowed_reference_l2<true, kRefMaxW>), dim3(tiles), dim3(kRefT), 0, stream, a);
    else
        if (w.W <= 20)
            hipLaunchKernelGGL((crf_windowed_reference_l2<false, 20>), dim3(tiles), dim3(kRefT), 0, stream, a);
        else
            hipLaunchKernelGGL((crf_windowed_reference_l2<false, kRefMaxW>), dim3(tiles), dim3(kRefT), 0, stream, a);
    return hipGetLastError();
}

}  // namespace gecco
