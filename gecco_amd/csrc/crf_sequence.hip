// Whole-contig inference kernels (north-star extensions; rows F and V of SURVEY.md §8a):
//   F  [EXT] CRF.predict_marginals_single(all feats)  = crf1dc_alpha/beta/marginal_point over a
//      whole contig -- NOT what GECCO's predict path computes (that is the windowed kernel);
//   V  [EXT] CRF.predict_single = crf1dc_viterbi, first-argmax tie-breaking.
// Two-label models.  Contigs range from a handful of genes to 50 000 (BASELINE.json configs[4]),
// so sequences are cut into chunks of kSeqChunk genes and scanned on two levels:
//   1. one lane per chunk folds its genes into a 2x2 transfer matrix (sum-product for F with
//      power-of-two rescaling, max-plus for V);
//   2. one lane per contig walks its chunk matrices (n/64 steps) to get the DP vectors at every
//      chunk boundary;
//   3. one lane per chunk replays its genes from the boundary vector and emits results.
// Contigs up to kSeqChunk genes are a single chunk, i.e. evaluated strictly sequentially.
// For longer contigs boundary scores are composed through the chunk matrices: identical
// whenever the additions are exact (e.g. integer-valued weights, ties included), equal to
// rounding otherwise.
#include "crf_device.hpp"

namespace gecco {
namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ double max4(double a, double b, double c, double d) { return fmax(fmax(a, b), fmax(c, d)); }

// exact power-of-two renormalisation; returns the exponent removed
__device__ __forceinline__ int renorm(Mat2 &p) {
    int ex;
    (void)frexp(max4(p.a00, p.a01, p.a10, p.a11), &ex);
    p.a00 = ldexp(p.a00, -ex);
    p.a01 = ldexp(p.a01, -ex);
    p.a10 = ldexp(p.a10, -ex);
    p.a11 = ldexp(p.a11, -ex);
    return ex;
}
__device__ __forceinline__ int renorm2(double &u, double &v) {
    int ex;
    (void)frexp(fmax(u, v), &ex);
    u = ldexp(u, -ex);
    v = ldexp(v, -ex);
    return ex;
}

// ---- row S for whole contigs: state scores of every gene, (s[label 0], s[label 1])
__global__ void __launch_bounds__(kThreads) seq_state_scores(const int32_t *__restrict__ gene_ptr,
                                                             const int32_t *__restrict__ attr_id,
                                                             const double2 *__restrict__ wtab01, int n_genes,
                                                             double2 *__restrict__ state) {
    const int g = blockIdx.x * kThreads + threadIdx.x;
    if (g >= n_genes) return;
    const int lo = gene_ptr[g], hi = gene_ptr[g + 1];
    double s0 = 0.0, s1 = 0.0;
    for (int base = lo; base < hi; base += 4) {
        int a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = base + u < hi ? attr_id[base + u] : -1;
        double2 w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = a[u] >= 0 ? wtab01[a[u]] : make_double2(0.0, 0.0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s0 += w[u].x;
            s1 += w[u].y;
        }
    }
    state[g] = make_double2(s0, s1);
}

// =====================================================================================
// F: sum-product.  alpha_t = (1,1) D_0 (M' D_1) ... (M' D_t),  beta_t = (M' D_{t+1}) ... (M' D_{n-1}) 1
// with D_t = diag(exp(s_t - max s_t)), M' = exp(trans - max trans); all scale factors cancel in
// P_t(y) = alpha_t[y] beta_t[y] / (alpha_t . beta_t) and are tracked only for log Z.
// =====================================================================================
__device__ __forceinline__ double2 emit_norm(double2 s, double &m) {
    m = fmax(s.x, s.y);
    return make_double2(exp(s.x - m), exp(s.y - m));
}

__global__ void __launch_bounds__(kThreads) f_chunk_product(const SeqArgs A) {
    const int c = blockIdx.x * kThreads + threadIdx.x;
    if (c >= A.n_chunks) return;
    const int g0 = A.ch_start[c], len = A.ch_len[c];
    Mat2 P{1.0, 0.0, 0.0, 1.0};
    double ex = 0.0, sm = 0.0;
    for (int k = 0; k < len; ++k) {
        double m;
        const double2 e = emit_norm(A.state[g0 + k], m);
        sm += m;
        Mat2 Q;
        if (k == 0 && A.ch_first[c]) {  // first gene of the contig: no transition in front
            Q = P;
        } else {
            Q.a00 = fma(P.a01, A.m10, P.a00 * A.m00);
            Q.a01 = fma(P.a01, A.m11, P.a00 * A.m01);
            Q.a10 = fma(P.a11, A.m10, P.a10 * A.m00);
            Q.a11 = fma(P.a11, A.m11, P.a10 * A.m01);
        }
        P.a00 = Q.a00 * e.x;
        P.a01 = Q.a01 * e.y;
        P.a10 = Q.a10 * e.x;
        P.a11 = Q.a11 * e.y;
        ex += double(renorm(P));
    }
    A.chP[c] = P;
    A.chAux[c] = make_double2(ex, sm);
}

__global__ void __launch_bounds__(kThreads) f_contig_scan(const SeqArgs A) {
    const int ci = blockIdx.x * kThreads + threadIdx.x;
    if (ci >= A.n_contigs) return;
    const int c0 = A.ct_chunk0[ci], c1 = A.ct_chunk0[ci + 1];
    if (c0 == c1) {
        if (A.lognorm) A.lognorm[ci] = 0.0;
        return;
    }
    double a0 = 1.0, a1 = 1.0, ex = 0.0, sm = 0.0;
    int n = 0;
    for (int c = c0; c < c1; ++c) {
        A.chIn[c] = make_double2(a0, a1);
        const Mat2 P = A.chP[c];
        const double2 aux = A.chAux[c];
        const double n0 = fma(a1, P.a10, a0 * P.a00);
        const double n1 = fma(a1, P.a11, a0 * P.a01);
        a0 = n0;
        a1 = n1;
        ex += aux.x + double(renorm2(a0, a1));
        sm += aux.y;
        n += A.ch_len[c];
    }
    if (A.lognorm) A.lognorm[ci] = sm + double(n - 1) * A.mx + log(a0 + a1) + ex * 0.6931471805599453;
    double b0 = 1.0, b1 = 1.0;
    for (int c = c1 - 1; c >= c0; --c) {
        A.chOut[c] = make_double2(b0, b1);
        const Mat2 P = A.chP[c];  // includes the transition INTO the chunk's first gene
        const double n0 = fma(P.a01, b1, P.a00 * b0);
        const double n1 = fma(P.a11, b1, P.a10 * b0);
        b0 = n0;
        b1 = n1;
        (void)renorm2(b0, b1);
    }
}

__global__ void __launch_bounds__(kThreads) f_chunk_marginals(const SeqArgs A) {
    const int c = blockIdx.x * kThreads + threadIdx.x;
    if (c >= A.n_chunks) return;
    const int g0 = A.ch_start[c], len = A.ch_len[c];
    const double2 in = A.chIn[c];
    double a0 = in.x, a1 = in.y;
    for (int k = 0; k < len; ++k) {
        double m;
        const double2 e = emit_norm(A.state[g0 + k], m);
        double n0 = a0, n1 = a1;
        if (!(k == 0 && A.ch_first[c])) {
            n0 = fma(a1, A.m10, a0 * A.m00);
            n1 = fma(a1, A.m11, a0 * A.m01);
        }
        a0 = n0 * e.x;
        a1 = n1 * e.y;
        (void)renorm2(a0, a1);
        A.alpha[g0 + k] = make_double2(a0, a1);
    }
    const double2 out = A.chOut[c];
    double b0 = out.x, b1 = out.y;
    for (int k = len - 1; k >= 0; --k) {
        const double2 al = A.alpha[g0 + k];
        const double x0 = al.x * b0, x1 = al.y * b1;
        const double z = x0 + x1;
        A.marg[2 * size_t(g0 + k)] = x0 / z;
        A.marg[2 * size_t(g0 + k) + 1] = x1 / z;
        if (k > 0) {  // beta_{t-1} = M' (D_t o beta_t); never crosses the chunk's first gene
            double m;
            const double2 e = emit_norm(A.state[g0 + k], m);
            const double c0 = e.x * b0, c1 = e.y * b1;
            b0 = fma(A.m01, c1, A.m00 * c0);
            b1 = fma(A.m11, c1, A.m10 * c0);
            (void)renorm2(b0, b1);
        }
    }
}

// =====================================================================================
// V: max-plus.  delta_0 = s_0; delta_t[j] = max_i(delta_{t-1}[i] + trans[i][j]) + s_t[j], ties -> the
// smaller i (CRFsuite updates on strict '<'); last label = first argmax; backtrack.
// =====================================================================================
__global__ void __launch_bounds__(kThreads) v_chunk_product(const SeqArgs A) {
    const int c = blockIdx.x * kThreads + threadIdx.x;
    if (c >= A.n_chunks) return;
    const int g0 = A.ch_start[c], len = A.ch_len[c];
    // V[i][j]: best score from label i just before the chunk to label j at its current gene
    double v00 = 0.0, v01 = 0.0, v10 = 0.0, v11 = 0.0;
    for (int k = 0; k < len; ++k) {
        const double2 s = A.state[g0 + k];
        if (k == 0) {
            if (A.ch_first[c]) {  // delta_0 = s_0 whatever the (virtual) previous label
                v00 = s.x; v01 = s.y; v10 = s.x; v11 = s.y;
            } else {
                v00 = A.t00 + s.x; v01 = A.t01 + s.y; v10 = A.t10 + s.x; v11 = A.t11 + s.y;
            }
        } else {
            const double n00 = fmax(v00 + A.t00, v01 + A.t10) + s.x;
            const double n01 = fmax(v00 + A.t01, v01 + A.t11) + s.y;
            const double n10 = fmax(v10 + A.t00, v11 + A.t10) + s.x;
            const double n11 = fmax(v10 + A.t01, v11 + A.t11) + s.y;
            v00 = n00; v01 = n01; v10 = n10; v11 = n11;
        }
    }
    A.chP[c] = Mat2{v00, v01, v10, v11};
}

__global__ void __launch_bounds__(kThreads) v_contig_scan(const SeqArgs A) {
    const int ci = blockIdx.x * kThreads + threadIdx.x;
    if (ci >= A.n_contigs) return;
    const int c0 = A.ct_chunk0[ci], c1 = A.ct_chunk0[ci + 1];
    double d0 = 0.0, d1 = 0.0;  // virtual delta before the contig (ignored by a first chunk)
    for (int c = c0; c < c1; ++c) {
        A.chIn[c] = make_double2(d0, d1);
        const Mat2 V = A.chP[c];
        const double n0 = fmax(d0 + V.a00, d1 + V.a10);
        const double n1 = fmax(d0 + V.a01, d1 + V.a11);
        d0 = n0;
        d1 = n1;
    }
    if (A.score) A.score[ci] = c0 == c1 ? 0.0 : fmax(d0, d1);
}

// replays a chunk from the delta entering it, leaves the two back-pointer bits of every gene in
// y[] (bit0: best predecessor of label 0, bit1: of label 1) and the chunk's end->entry label map
__global__ void __launch_bounds__(kThreads) v_chunk_backpointers(const SeqArgs A) {
    const int c = blockIdx.x * kThreads + threadIdx.x;
    if (c >= A.n_chunks) return;
    const int g0 = A.ch_start[c], len = A.ch_len[c];
    const double2 in = A.chIn[c];
    double d0 = in.x, d1 = in.y;
    for (int k = 0; k < len; ++k) {
        const double2 s = A.state[g0 + k];
        int bp = 0;
        if (k == 0 && A.ch_first[c]) {
            d0 = s.x;
            d1 = s.y;
        } else {
            const double a0 = d0 + A.t00, b0 = d1 + A.t10;  // into label 0
            const double a1 = d0 + A.t01, b1 = d1 + A.t11;  // into label 1
            const bool p0 = a0 < b0, p1 = a1 < b1;           // strict: ties keep predecessor 0
            bp = (p0 ? 1 : 0) | (p1 ? 2 : 0);
            d0 = (p0 ? b0 : a0) + s.x;
            d1 = (p1 ? b1 : a1) + s.y;
        }
        A.y[g0 + k] = int8_t(bp);
    }
    // end label j -> label just before the chunk
    for (int j = 0; j < 2; ++j) {
        int lab = j;
        for (int k = len - 1; k >= 0; --k) lab = (A.y[g0 + k] >> lab) & 1;
        A.chMap[2 * c + j] = int8_t(lab);
    }
    // last chunk of a contig also fixes the end label: first argmax (strict '<' update from label 0)
    A.chEnd[c] = int8_t(d0 < d1 ? 1 : 0);
}

__global__ void __launch_bounds__(kThreads) v_contig_backtrack(const SeqArgs A) {
    const int ci = blockIdx.x * kThreads + threadIdx.x;
    if (ci >= A.n_contigs) return;
    const int c0 = A.ct_chunk0[ci], c1 = A.ct_chunk0[ci + 1];
    if (c0 == c1) return;
    int lab = A.chEnd[c1 - 1];
    for (int c = c1 - 1; c >= c0; --c) {
        const int entry = A.chMap[2 * c + lab];
        A.chEnd[c] = int8_t(lab);
        lab = entry;
    }
}

__global__ void __launch_bounds__(kThreads) v_chunk_labels(const SeqArgs A) {
    const int c = blockIdx.x * kThreads + threadIdx.x;
    if (c >= A.n_chunks) return;
    const int g0 = A.ch_start[c], len = A.ch_len[c];
    int lab = A.chEnd[c];
    for (int k = len - 1; k >= 0; --k) {
        const int bp = A.y[g0 + k];
        A.y[g0 + k] = int8_t(lab);
        lab = (bp >> lab) & 1;
    }
}

}  // namespace

// ---- launchers ---------------------------------------------------------------------------
static inline dim3 grid_for(int n) { return dim3((n + kThreads - 1) / kThreads); }

hipError_t launch_seq_state(const int32_t *gene_ptr, const int32_t *attr_id, const double2 *wtab01, int n_genes,
                            double2 *state, hipStream_t stream) {
    if (n_genes <= 0) return hipSuccess;
    hipLaunchKernelGGL(seq_state_scores, grid_for(n_genes), dim3(kThreads), 0, stream, gene_ptr, attr_id, wtab01, n_genes,
                       state);
    return hipGetLastError();
}

hipError_t launch_seq_marginals(const SeqArgs &a, hipStream_t stream) {
    if (a.n_contigs <= 0) return hipSuccess;
    if (a.n_chunks > 0) hipLaunchKernelGGL(f_chunk_product, grid_for(a.n_chunks), dim3(kThreads), 0, stream, a);
    hipLaunchKernelGGL(f_contig_scan, grid_for(a.n_contigs), dim3(kThreads), 0, stream, a);
    if (a.n_chunks > 0) hipLaunchKernelGGL(f_chunk_marginals, grid_for(a.n_chunks), dim3(kThreads), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_seq_viterbi(const SeqArgs &a, hipStream_t stream) {
    if (a.n_contigs <= 0) return hipSuccess;
    if (a.n_chunks > 0) hipLaunchKernelGGL(v_chunk_product, grid_for(a.n_chunks), dim3(kThreads), 0, stream, a);
    hipLaunchKernelGGL(v_contig_scan, grid_for(a.n_contigs), dim3(kThreads), 0, stream, a);
    if (a.n_chunks > 0) {
        hipLaunchKernelGGL(v_chunk_backpointers, grid_for(a.n_chunks), dim3(kThreads), 0, stream, a);
        hipLaunchKernelGGL(v_contig_backtrack, grid_for(a.n_contigs), dim3(kThreads), 0, stream, a);
        hipLaunchKernelGGL(v_chunk_labels, grid_for(a.n_chunks), dim3(kThreads), 0, stream, a);
    }
    return hipGetLastError();
}

}  // namespace gecco
