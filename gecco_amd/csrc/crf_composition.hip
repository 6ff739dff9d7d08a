// Weighted domain composition of called clusters on packed arrays:
// /root/reference/gecco/model.py:458-503 `Cluster.domain_composition(all_possible)` as assembled
// for the type classifier at /root/reference/gecco/types/__init__.py:118 -- one dense row per
// cluster over the classifier's domain list, entry i = numpy.sum of the weights (1 - pvalue) of
// the cluster's domains named all_possible[i], the row divided by `row.sum() or 1`.
//
// The only arithmetic is numpy.sum, so bit-identical output means reproducing numpy's pairwise
// summation ([EXT] numpy/_core/src/umath/loops_utils.h.src `pairwise_sum`): fewer than 8 terms are
// added left to right; up to 128 terms go through 8 interleaved accumulators combined as
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus a left-to-right tail; longer ranges split at
// n/2 rounded down to a multiple of 8, left + right; the reduction starts from 0.0 and takes the
// array in chunks of 8192 elements (numpy's reduction buffer), adding each chunk's pairwise sum.
//
// One workgroup per cluster.  Domain rows of a cluster are contiguous (genes of a cluster are
// consecutive, domain rows are stored per gene), so the cluster is a row range [r0, r1):
//   1. zero the output row;
//   2. every row finds its slot in a stable sort by column (rank by counting: clusters hold tens
//      of rows) and parks its weight there, so each column's terms are contiguous and in cluster
//      order; the first row of every column sums its terms;
//   3. the row total: leaves of the pairwise tree in parallel (one lane each), combined by lane 0.
#include "crf_device.hpp"

namespace gecco {
namespace {

constexpr int kCT = 256;
constexpr int kMaxLeaves = 2048;  // leaves of the row-total tree held in LDS (n_cols <= ~130 k)

// numpy's pairwise_sum for n <= 128 contiguous terms
__device__ __forceinline__ double np_leaf(const double *a, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += a[i];
        return res;
    }
    double r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = a[k];
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] += a[i + k];
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

// Walks numpy's recursion tree over [0, n) without recursion; leaf(off, len, index) supplies the
// value of the index-th leaf in left-to-right order.  Returns the root value.
template <class Leaf>
__device__ double np_pairwise_tree(int n, Leaf leaf) {
    int off_s[64], len_s[64];  // len < 0: "add the two topmost values"
    double val[40];
    int top = 0, vp = 0, idx = 0;
    off_s[0] = 0;
    len_s[0] = n;
    top = 1;
    while (top > 0) {
        --top;
        const int off = off_s[top], len = len_s[top];
        if (len < 0) {
            const double b = val[--vp], a = val[--vp];
            val[vp++] = a + b;
        } else if (len <= 128) {
            val[vp++] = leaf(off, len, idx++);
        } else {
            int n2 = len / 2;
            n2 -= n2 % 8;
            off_s[top] = 0;
            len_s[top++] = -1;
            off_s[top] = off + n2;
            len_s[top++] = len - n2;
            off_s[top] = off;
            len_s[top++] = n2;
        }
    }
    return val[0];
}

constexpr int kNpChunk = 8192;  // numpy.getbufsize()

// numpy.sum of n contiguous doubles; leaf(off, len, index) as above, indices running over all chunks
template <class Leaf>
__device__ double np_sum_with(int n, Leaf leaf) {
    double res = 0.0;
    int base = 0;
    for (int lo = 0; lo < n; lo += kNpChunk) {
        const int len = n - lo < kNpChunk ? n - lo : kNpChunk;
        int used = 0;
        res += np_pairwise_tree(len, [&](int off, int l, int idx) {
            used = idx + 1;
            return leaf(lo + off, l, base + idx);
        });
        base += used;
    }
    return res;
}

__device__ double np_sum(const double *a, int n) {
    if (n <= 128) return 0.0 + np_leaf(a, n);
    return np_sum_with(n, [&](int off, int len, int) { return np_leaf(a + off, len); });
}

__global__ void __launch_bounds__(kCT) composition_kernel(const int32_t *__restrict__ seg, const int32_t *__restrict__ dom_ptr,
                                                          const int32_t *__restrict__ dom_col,
                                                          const double *__restrict__ dom_w, double *__restrict__ tmp,
                                                          int n_cols, int normalize, double *__restrict__ out) {
    __shared__ double leafv[kMaxLeaves];
    __shared__ double total_s;
    const int b = blockIdx.x, tid = threadIdx.x;
    double *row = out + static_cast<size_t>(b) * n_cols;
    const int r0 = dom_ptr[seg[4 * b + 2]], r1 = dom_ptr[seg[4 * b + 3]];
    for (int i = tid; i < n_cols; i += kCT) row[i] = 0.0;
    // stable counting rank of every row among the rows of this cluster (absent columns last)
    for (int r = r0 + tid; r < r1; r += kCT) {
        const int c = dom_col[r];
        const unsigned key = c < 0 ? 0xffffffffu : unsigned(c);
        int before = 0;
        for (int q = r0; q < r1; ++q) {
            const int cq = dom_col[q];
            const unsigned kq = cq < 0 ? 0xffffffffu : unsigned(cq);
            before += (kq < key || (kq == key && q < r)) ? 1 : 0;
        }
        tmp[r0 + before] = dom_w[r];
    }
    __syncthreads();
    for (int r = r0 + tid; r < r1; r += kCT) {
        const int c = dom_col[r];
        if (c < 0 || c >= n_cols) continue;
        int smaller = 0, same = 0;
        bool first = true;
        for (int q = r0; q < r1; ++q) {
            const int cq = dom_col[q];
            smaller += (cq >= 0 && cq < c) ? 1 : 0;
            same += cq == c ? 1 : 0;
            first = first && !(cq == c && q < r);
        }
        if (first) row[c] = np_sum(tmp + r0 + smaller, same);
    }
    if (!normalize) return;
    __syncthreads();
    // composition.sum(): leaves in parallel, then the tree by one lane
    if (n_cols <= 128 * 8 || n_cols > kMaxLeaves * 64) {
        if (tid == 0) total_s = np_sum(row, n_cols);
    } else {
        np_sum_with(n_cols, [&](int off, int len, int idx) {
            if (idx % kCT == tid) leafv[idx] = np_leaf(row + off, len);
            return 0.0;
        });
        __syncthreads();
        if (tid == 0) total_s = np_sum_with(n_cols, [&](int, int, int idx) { return leafv[idx]; });
    }
    __syncthreads();
    const double total = total_s;
    const double den = total == 0.0 ? 1.0 : total;  // `composition.sum() or 1`
    for (int i = tid; i < n_cols; i += kCT) row[i] = row[i] / den;
}

}  // namespace

hipError_t launch_composition(const int32_t *d_seg, int n_seg, const int32_t *d_dom_ptr, const int32_t *d_dom_col,
                              const double *d_dom_w, double *d_tmp, int n_cols, int normalize, double *d_out,
                              hipStream_t stream) {
    if (n_seg <= 0 || n_cols <= 0) return hipSuccess;
    hipLaunchKernelGGL(composition_kernel, dim3(n_seg), dim3(kCT), 0, stream, d_seg, d_dom_ptr, d_dom_col, d_dom_w, d_tmp,
                       n_cols, normalize, d_out);
    return hipGetLastError();
}

}  // namespace gecco
