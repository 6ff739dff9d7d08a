// Row R on packed arrays: threshold run-length segmentation of per-gene probabilities with the
// reference's stateful grouper, edge trimming and the "gecco" validation criterion
// (/root/reference/gecco/refine.py:51-64 GeneGrouper, :118-200 ClusterRefiner).
//   * gene "in" <=> p > threshold (strict); NaN (no probability) inherits the previous gene's
//     state -- and ONE grouper spans all contigs of a call (:186), so a contig that starts with
//     NaN genes inherits the state the previous contigs ended in;
//   * every maximal "in" run of a contig is numbered from 1 before filtering; trimming drops
//     un-annotated genes at both ends; kept iff #annotated >= n_cds and
//     #(genes that are not edge genes) >= n_cds, edge genes being the first/last
//     `edge_distance` annotated genes of the contig.
// One lane per contig walks its genes (clusters are a few genes out of hundreds: the walk is
// a coalescing-unfriendly but tiny stream of 9 B/gene); the cross-contig grouper state and the
// output offsets are resolved by single-workgroup scans.
#include "crf_device.hpp"

namespace gecco {
namespace {

constexpr int kT = 256;

struct Walk {
    const double *p;
    const uint8_t *ann;
    double thr;
    int n_cds, edge, trim;
};

// walks contig [g0,g1) starting in grouper state `st`; calls emit(number, a, b) for kept clusters
template <class Emit>
__device__ __forceinline__ int walk_contig(const Walk &w, int g0, int g1, bool st, Emit emit) {
    int n_ann = 0;
    if (w.edge > 0)
        for (int k = g0; k < g1; ++k) n_ann += w.ann[k] ? 1 : 0;
    int kept = 0, number = 0, run_start = -1;
    for (int g = g0; g <= g1; ++g) {
        bool in = false;
        if (g < g1) {
            const double pv = w.p[g];
            if (pv == pv) st = pv > w.thr;
            in = st;
        }
        if (in && run_start < 0) run_start = g;
        if (!in && run_start >= 0) {
            ++number;
            int a = run_start, b = g;
            run_start = -1;
            if (w.trim) {
                while (a < b && !w.ann[a]) ++a;
                while (b > a && !w.ann[b - 1]) --b;
            }
            int ann = 0, inner = 0, rank = 0;
            if (w.edge > 0)
                for (int k = g0; k < a; ++k) rank += w.ann[k] ? 1 : 0;
            for (int k = a; k < b; ++k) {
                bool is_edge = false;
                if (w.ann[k]) {
                    ++ann;
                    if (w.edge > 0 && (rank < w.edge || rank >= n_ann - w.edge)) is_edge = true;
                    ++rank;
                }
                if (!is_edge) ++inner;
            }
            if (ann >= w.n_cds && inner >= w.n_cds) {
                emit(number, a, b, kept);
                ++kept;
            }
        }
    }
    return kept;
}

// per contig: index of itself if it holds any gene with a probability (else -1) and the
// grouper state it leaves behind in that case
__global__ void __launch_bounds__(kT) seg_contig_state(const double *__restrict__ p, const int32_t *__restrict__ cptr,
                                                       int n_contigs, double thr, int32_t *__restrict__ last_valid,
                                                       uint8_t *__restrict__ out_state) {
    const int c = blockIdx.x * kT + threadIdx.x;
    if (c >= n_contigs) return;
    int lv = -1;
    uint8_t st = 0;
    for (int g = cptr[c + 1] - 1; g >= cptr[c]; --g) {
        const double pv = p[g];
        if (pv == pv) {
            lv = c;
            st = pv > thr;
            break;
        }
    }
    last_valid[c] = lv;
    out_state[c] = st;
}

// single workgroup: x[i] <- scan over i of op; MODE 0: inclusive running max (in place),
// MODE 1: exclusive prefix sum (in place), total written to *total
template <int MODE>
__global__ void __launch_bounds__(1024) seg_scan(int32_t *x, int n, int32_t *total) {
    __shared__ int32_t buf[1024];
    __shared__ int32_t carry;
    const int tid = threadIdx.x;
    if (tid == 0) carry = MODE == 0 ? -1 : 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const int32_t v = i < n ? x[i] : (MODE == 0 ? -1 : 0);
        buf[tid] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            int32_t t = buf[tid];
            if (tid >= off) t = MODE == 0 ? max(t, buf[tid - off]) : t + buf[tid - off];
            __syncthreads();
            buf[tid] = t;
            __syncthreads();
        }
        const int32_t c = carry;
        const int32_t incl = MODE == 0 ? max(buf[tid], c) : buf[tid] + c;
        if (i < n) x[i] = MODE == 0 ? incl : incl - v;
        __syncthreads();
        if (tid == 1023) carry = incl;
        __syncthreads();
    }
    if (MODE == 1 && tid == 0 && total) *total = carry;
}

__global__ void __launch_bounds__(kT) seg_count(const Walk w, const int32_t *__restrict__ cptr, int n_contigs,
                                                const int32_t *__restrict__ last_valid, const uint8_t *__restrict__ out_state,
                                                int32_t *__restrict__ cnt) {
    const int c = blockIdx.x * kT + threadIdx.x;
    if (c >= n_contigs) return;
    const int src = c > 0 ? last_valid[c - 1] : -1;  // running max: nearest earlier contig with a value
    const bool st = src >= 0 ? out_state[src] != 0 : false;
    cnt[c] = walk_contig(w, cptr[c], cptr[c + 1], st, [](int, int, int, int) {});
}

__global__ void __launch_bounds__(kT) seg_write(const Walk w, const int32_t *__restrict__ cptr, int n_contigs,
                                                const int32_t *__restrict__ last_valid, const uint8_t *__restrict__ out_state,
                                                const int32_t *__restrict__ off, int32_t *__restrict__ seg, int max_seg) {
    const int c = blockIdx.x * kT + threadIdx.x;
    if (c >= n_contigs) return;
    const int src = c > 0 ? last_valid[c - 1] : -1;
    const bool st = src >= 0 ? out_state[src] != 0 : false;
    const int o = off[c];
    walk_contig(w, cptr[c], cptr[c + 1], st, [&](int number, int a, int b, int k) {
        if (o + k < max_seg) {
            int32_t *r = seg + 4 * size_t(o + k);
            r[0] = c;
            r[1] = number;
            r[2] = a;
            r[3] = b;
        }
    });
}

}  // namespace

// d_work: 3*n_contigs int32 + n_contigs bytes (+ 1 int32 total); all device pointers
hipError_t launch_segment(const double *d_p, const uint8_t *d_ann, const int32_t *d_cptr, int n_contigs, double threshold,
                          int n_cds, int edge_distance, int trim, int32_t *d_seg, int max_seg, int32_t *d_work,
                          int32_t *d_total, hipStream_t stream) {
    if (n_contigs <= 0) return hipMemsetAsync(d_total, 0, 4, stream);
    int32_t *last_valid = d_work, *cnt = d_work + n_contigs;
    uint8_t *out_state = reinterpret_cast<uint8_t *>(d_work + 2 * size_t(n_contigs));
    const Walk w{d_p, d_ann, threshold, n_cds, edge_distance, trim};
    const dim3 grid((n_contigs + kT - 1) / kT), block(kT);
    hipLaunchKernelGGL(seg_contig_state, grid, block, 0, stream, d_p, d_cptr, n_contigs, threshold, last_valid, out_state);
    hipLaunchKernelGGL(seg_scan<0>, dim3(1), dim3(1024), 0, stream, last_valid, n_contigs, (int32_t *)nullptr);
    hipLaunchKernelGGL(seg_count, grid, block, 0, stream, w, d_cptr, n_contigs, last_valid, out_state, cnt);
    hipLaunchKernelGGL(seg_scan<1>, dim3(1), dim3(1024), 0, stream, cnt, n_contigs, d_total);
    hipLaunchKernelGGL(seg_write, grid, block, 0, stream, w, d_cptr, n_contigs, last_valid, out_state, cnt, d_seg, max_seg);
    return hipGetLastError();
}

}  // namespace gecco
