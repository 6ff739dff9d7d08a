// Row R on packed arrays: threshold run-length segmentation of per-gene probabilities with the
// reference's stateful grouper, edge trimming and the "gecco" validation criterion
// (/root/reference/gecco/refine.py:51-64 GeneGrouper, :118-200 ClusterRefiner).
//   * gene "in" <=> p > threshold (strict); NaN (no probability) inherits the previous gene's
//     state.  One grouper lives for one iter_clusters call (:186): the CLI makes one call per contig
//     (cli/commands/_common.py:621-623; `carry` = 0: every contig starts "out"), a single call over
//     many contigs lets a contig that starts with NaN genes inherit the state the previous contigs
//     ended in (`carry` = 1);
//   * every maximal "in" run of a contig is numbered from 1 before filtering; trimming drops
//     un-annotated genes at both ends; kept iff #annotated >= n_cds and
//     #(genes that are not edge genes) >= n_cds, edge genes being the first/last
//     `edge_distance` annotated genes of the contig.
//
// Nothing here walks a contig: all genes of all contigs form ONE flat sequence (a 50 000-gene
// contig is 25 workgroups, not one lane).  The grouper is a two-state transducer, so a span of genes
// is summarised by what it does to either entering state -- the state it leaves, how many runs it
// starts -- plus its number of annotated genes (SegE, crf_scan.hpp); spans compose associatively:
//   seg_fold     a lane folds 8 genes for both entering states; wave scan by DPP, wave totals in LDS
//   seg_replay   every workgroup first reduces the totals of the workgroups before it (n/2048 elements, 64 per step of a
//                wave: no separate scan launch, nobody waits for anybody); every lane then re-walks its 8 genes from
//                its now known entering state: writes the prefix count of annotated genes pre[] and, at run starts /
//                ends, the raw run table (runs are dense and ordered: row = number of runs started before)
//   seg_validate one lane per raw run: contig, cluster number, trimming and the annotated / edge
//                counts by binary searches in pre[] (no walk over the run either)
//   seg_compact  ordered compaction of the kept rows (the counts of the tiles before a tile reduced the same way),
//                their gene offsets and -- on request -- the probabilities of their genes (what a cluster table needs of p)
// Bound: HBM, 2 x 10 B/gene read + 4 B/gene written; four short launches (round 3: six + the gather).
#include <algorithm>
#include <cstdlib>

#include "crf_device.hpp"
#include "crf_scan.hpp"

namespace gecco {
namespace {

constexpr int kT = kScanThreads;
constexpr int kGPL = 8;  // genes per lane
constexpr int kBlockGenes = kT * kGPL;

struct SegOp {
    static __device__ __forceinline__ SegE identity() { return SegE{2u, 0u, 0u, 0u}; }
    static __device__ __forceinline__ SegE combine(const SegE &a, const SegE &b) {  // a earlier, b later
        return SegE{MapOp::combine(b.map, a.map), a.ng0 + ((a.map & 1u) ? b.ng1 : b.ng0),
                    a.ng1 + ((a.map & 2u) ? b.ng1 : b.ng0), a.ann + b.ann};
    }
};
// The workgroup's 2048 probabilities arrive with coalesced 8-B loads and reach the lanes that own 8
// consecutive ones through padded LDS rows (conflict-free both ways); the 8 annotation / contig
// flags of a lane are one aligned 8-byte word each.
struct Stage {
    double st[kT * (kGPL + 1)];
};
struct LaneIn {
    double p[kGPL];
    uint64_t ann, fl;
    int g0, cnt;
};
__device__ __forceinline__ uint64_t load_bytes8(const uint8_t *__restrict__ a, int g0, int n) {
    if (g0 + kGPL <= n) return *reinterpret_cast<const uint64_t *>(a + g0);
    uint64_t w = 0;
    for (int k = 0; k < kGPL; ++k)
        if (g0 + k < n) w |= uint64_t(a[g0 + k]) << (8 * k);
    return w;
}
__device__ __forceinline__ LaneIn load_lane(const SegArgs &A, Stage &stg) {
    const int slot = threadIdx.x, base = blockIdx.x * kBlockGenes;
#pragma unroll
    for (int j = 0; j < kGPL; ++j) {
        const int idx = j * kT + slot, g = base + idx;
        stg.st[(idx / kGPL) * (kGPL + 1) + idx % kGPL] = g < A.n_genes ? A.p[g] : 0.0;
    }
    LaneIn L;
    L.g0 = base + slot * kGPL;
    L.cnt = min(kGPL, A.n_genes - L.g0);
    L.ann = load_bytes8(A.ann, L.g0, A.n_genes);
    L.fl = load_bytes8(A.flags, L.g0, A.n_genes);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kGPL; ++k) L.p[k] = stg.st[slot * (kGPL + 1) + k];
    return L;
}

// flags[g]: bit0 = first gene of a contig, bit1 = last gene (the layout of the plan's whole-contig tables).  One lane per
// eight genes: it finds the contig of its first gene in the contig table (a binary search through L2-resident words) and
// walks from there -- every byte of the array is written, by one launch (no memset before it).
__device__ __forceinline__ void seg_flags_body(const int32_t *__restrict__ cptr, int n_contigs, int n_genes, uint8_t *__restrict__ flags,
                                               const int word_index) {
    const int g0 = word_index * 8;
    if (g0 >= n_genes + 8) return;
    uint64_t word = 0;
    if (g0 < n_genes) {
        int lo = 0, hi = n_contigs;  // largest c with cptr[c] <= g0 (cptr[0] = 0, cptr[n_contigs] = n_genes > g0)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (cptr[mid] <= g0) lo = mid; else hi = mid;
        }
        int c = lo, end = cptr[c + 1];
        while (end <= g0) end = cptr[++c + 1];  // (empty contigs that start at g0 sort before the one that holds it)
        int start = cptr[c];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int g = g0 + k;
            if (g >= n_genes) break;
            while (end <= g) {
                start = end;
                end = cptr[++c + 1];
            }
            word |= uint64_t((g == start ? 1u : 0u) | (g + 1 == end ? 2u : 0u)) << (8 * k);
        }
    }
    *reinterpret_cast<uint64_t *>(flags + g0) = word;
}
__global__ void __launch_bounds__(kT) seg_flags(const int32_t *__restrict__ cptr, int n_contigs, int n_genes, uint8_t *__restrict__ flags) {
    seg_flags_body(cptr, n_contigs, n_genes, flags, blockIdx.x * kT + threadIdx.x);
}

__device__ __forceinline__ void seg_fold_body(const SegArgs &A) {
    __shared__ Stage stg;
    __shared__ SegE lds[kT / 64];
    const LaneIn L = load_lane(A, stg);
    uint32_t st0 = 0, st1 = 1, ng0 = 0, ng1 = 0, ann = 0;
#pragma unroll
    for (int k = 0; k < kGPL; ++k) {
        if (k < L.cnt) {
            const bool first = (L.fl >> (8 * k)) & 1u;
            const double pv = L.p[k];
            const uint32_t p0 = first ? 0u : st0, p1 = first ? 0u : st1;  // a contig always opens a new run
            if (first && !A.carry) st0 = st1 = 0u;
            if (pv == pv) st0 = st1 = pv > A.thr ? 1u : 0u;
            ng0 += st0 & ~p0;
            ng1 += st1 & ~p1;
            ann += ((L.ann >> (8 * k)) & 0xffu) ? 1u : 0u;
        }
    }
    SegE total;
    const SegE excl = block_scan_exclusive<SegOp, false>(SegE{st0 | (st1 << 1), ng0, ng1, ann}, lds, &total);
    A.lane[blockIdx.x * kT + threadIdx.x] = excl;
    if (threadIdx.x == 0) A.block[blockIdx.x] = total;
}
__global__ void __launch_bounds__(kT) seg_fold(const SegArgs A) { seg_fold_body(A); }

// Product of the elements e[0 .. n) in order, by ONE wave (every wave of a workgroup for itself: no barrier): 64 at a time.
__device__ __forceinline__ SegE wave_prefix_total(const SegE *__restrict__ e, int n) {
    const int lane = threadIdx.x & 63;
    SegE acc = SegOp::identity();
    for (int base = 0; base < n; base += 64) {
        const SegE mine = base + lane < n ? e[base + lane] : SegOp::identity();
        const SegE inc = wave_scan_inclusive<SegOp, false>(mine);
        auto bc = [](uint32_t v) { return uint32_t(__builtin_amdgcn_readlane(int(v), 63)); };
        acc = SegOp::combine(acc, SegE{bc(inc.map), bc(inc.ng0), bc(inc.ng1), bc(inc.ann)});
    }
    return acc;
}

__device__ __forceinline__ void seg_replay_body(const SegArgs &A) {
    __shared__ Stage stg;
    const LaneIn L = load_lane(A, stg);
    // what the workgroups before this one do to the grouper (the only serial step of round 3, a launch of its own then)
    const SegE B = wave_prefix_total(A.block, blockIdx.x);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        const SegE all = SegOp::combine(B, A.block[blockIdx.x]);
        *A.n_raw = int32_t(all.ng0);  // the batch is entered "out"
        A.pre[A.n_genes] = int32_t(all.ann);
    }
    if (L.cnt <= 0) return;  // (no barrier below this line)
    const SegE X = A.lane[blockIdx.x * kT + threadIdx.x];
    const uint32_t sb = B.map & 1u;  // the batch is entered "out": evaluate every map at 0
    uint32_t st = (X.map >> sb) & 1u;
    uint32_t ng = B.ng0 + (sb ? X.ng1 : X.ng0);
    uint32_t ann = B.ann + X.ann;
    const bool need_next = L.cnt == kGPL && L.g0 + kGPL < A.n_genes;
    const double p_next = need_next ? A.p[L.g0 + kGPL] : 0.0;
    int32_t pre[kGPL];
#pragma unroll
    for (int k = 0; k < kGPL; ++k) {
        pre[k] = int32_t(ann);
        if (k < L.cnt) {
            const uint32_t f = uint32_t(L.fl >> (8 * k)) & 0xffu;
            const bool first = f & 1u, last = f & 2u;
            const double pv = L.p[k];
            const uint32_t prev = first ? 0u : st;
            if (first && !A.carry) st = 0u;
            if (pv == pv) st = pv > A.thr ? 1u : 0u;
            if (st & ~prev) {
                A.raw[ng].x = L.g0 + k;
                ++ng;
            }
            if (st) {
                bool ends = last;
                if (!last) {  // the state the next gene of the contig will be in
                    const double pn = k + 1 < kGPL ? L.p[k + 1 < kGPL ? k + 1 : k] : p_next;
                    ends = pn == pn ? !(pn > A.thr) : false;
                }
                if (ends) A.raw[ng - 1].y = L.g0 + k + 1;
            }
            ann += ((L.ann >> (8 * k)) & 0xffu) ? 1u : 0u;
        }
    }
    if (L.cnt == kGPL) {
        int4 *dst = reinterpret_cast<int4 *>(A.pre + L.g0);
        dst[0] = make_int4(pre[0], pre[1], pre[2], pre[3]);
        dst[1] = make_int4(pre[4], pre[5], pre[6], pre[7]);
    } else {
        for (int k = 0; k < L.cnt; ++k) A.pre[L.g0 + k] = pre[k];
    }
}
__global__ void __launch_bounds__(kT) seg_replay(const SegArgs A) { seg_replay_body(A); }

// smallest i in [lo, hi) with a[i] > v (hi if none)
__device__ __forceinline__ int upper_bound_i32(const int32_t *__restrict__ a, int lo, int hi, int v) {
    while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if (a[mid] > v) hi = mid; else lo = mid + 1;
    }
    return lo;
}

__device__ __forceinline__ void seg_validate_body(const SegArgs &A) {
    const int n_raw = *A.n_raw;
    const int ntile = (n_raw + kT - 1) / kT;
    for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
        const int i = t * kT + threadIdx.x;
        bool kept = false;
        int len = 0;
        if (i < n_raw) {
            const int2 r = A.raw[i];
            const int s = r.x, e = r.y;
            // contig of the run: largest c with cptr[c] <= s (empty contigs never win: cptr[c+1] > s is required)
            const int c = upper_bound_i32(A.cptr, 0, A.n_contigs + 1, s) - 1;
            const int g0 = A.cptr[c], g1 = A.cptr[c + 1];
            // cluster number = rank of the run among the runs of its contig (numbered before filtering)
            int lo = 0, hi = i;  // smallest j with raw[j].x >= g0
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (A.raw[mid].x >= g0) hi = mid; else lo = mid + 1;
            }
            const int number = i - lo + 1;
            int a = s, b = e;
            const int pre_s = A.pre[s], pre_e = A.pre[e];
            if (A.trim) {
                if (pre_e == pre_s) {
                    a = b = e;  // nothing annotated: the run trims to nothing
                } else {
                    a = upper_bound_i32(A.pre, s + 1, e + 1, pre_s) - 1;      // first annotated gene
                    b = upper_bound_i32(A.pre, a + 1, e + 1, pre_e - 1);      // one past the last annotated gene
                }
            }
            len = b - a;
            if (A.criterion == 0) {
                const int pa = A.pre[a], ann = A.pre[b] - pa;
                int edge = 0;
                if (A.edge > 0) {
                    const int n_ann = A.pre[g1] - A.pre[g0];
                    const int ra = pa - A.pre[g0], rb = ra + ann;  // ranks of the run's annotated genes
                    const int lo2 = max(A.edge, n_ann - A.edge);   // [0, edge) u [lo2, n_ann)
                    edge = max(0, min(rb, A.edge) - ra) + max(0, rb - max(ra, lo2));
                }
                kept = ann >= A.n_cds && len - edge >= A.n_cds;
            } else {
                // "antismash" (refine.py:157-163): mean probability, distinct marker domains, number of genes.  This
                // one does walk the run (one lane per run; runs are short where this criterion is used).  The mean
                // is the plain left-to-right sum over the count: numpy.mean's own last bit depends on the SIMD
                // width numpy was dispatched to, so the reference does not pin it.
                double sum = 0.0;
                uint64_t seen[kSegMaxMarkers / 64] = {};
                for (int g = a; g < b; ++g) {
                    sum += A.p[g];
                    for (int k = A.bio_ptr[g]; k < A.bio_ptr[g + 1]; ++k) {
                        const uint32_t id = uint32_t(A.bio_id[k]);
                        if (id < uint32_t(kSegMaxMarkers)) seen[id >> 6] |= 1ull << (id & 63u);
                    }
                }
                int markers = 0;
#pragma unroll
                for (int w = 0; w < kSegMaxMarkers / 64; ++w) markers += __popcll(seen[w]);
                // an empty (all trimmed away) run has mean NaN there: rejected like every failed comparison
                kept = len > 0 && sum / double(len) >= A.avg_thr && markers >= A.n_bio && len >= A.n_cds;
            }
            A.val[i] = make_int4(kept ? c : -1 - c, number, a, b);
        }
        const int cnt = __syncthreads_count(kept ? 1 : 0);
        // genes of the kept rows of this tile: small numbers, an LDS atomic is plenty
        __shared__ int genes;
        if (threadIdx.x == 0) genes = 0;
        __syncthreads();
        if (kept && len) atomicAdd(&genes, len);
        __syncthreads();
        if (threadIdx.x == 0) A.tile[t] = make_int2(cnt, genes);
        __syncthreads();
    }
}
__global__ void __launch_bounds__(kT) seg_validate(const SegArgs A) { seg_validate_body(A); }

// Ordered compaction of the kept rows.  A workgroup takes tiles t, t + G, ...; (rows, genes) kept in the tiles before
// a tile are summed by every wave for itself (64 tiles per step), carried from the workgroup's previous tile.  Kept rows go
// to A.seg, their gene offsets to A.seg_off; with A.gout the probabilities of a row's genes follow row after row (a wave
// copies the rows of its 64 lanes one after the other, coalesced).  The batch driver points all three at pinned HOST memory:
// everything is read from device memory and only written across PCIe (posted writes).
__device__ __forceinline__ void seg_compact_body(const SegArgs &A) {
    __shared__ U2 lds[kT / 64];
    const int n_raw = *A.n_raw;
    const int ntile = (n_raw + kT - 1) / kT;
    const int lane = threadIdx.x & 63;
    auto range_sum = [&](int t0, int t1) {  // sum of tile[t0 .. t1)
        uint32_t x = 0, y = 0;
        for (int base = t0; base < t1; base += 64) {
            const int2 v = base + lane < t1 ? A.tile[base + lane] : make_int2(0, 0);
            x += uint32_t(v.x);
            y += uint32_t(v.y);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            x += uint32_t(__shfl_xor(int(x), o));
            y += uint32_t(__shfl_xor(int(y), o));
        }
        return U2{x, y};
    };
    if (blockIdx.x == 0 && ntile == 0 && threadIdx.x == 0) {
        *A.total = 0;
        if (A.seg_off) A.seg_off[0] = 0;
    }
    U2 before{0u, 0u};
    int done = 0;  // tiles [0, done) are in `before`
    for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
        before = AddOp::combine(before, range_sum(done, t));
        const int i = t * kT + threadIdx.x;
        int4 v = make_int4(-1, 0, 0, 0);
        if (i < n_raw) v = A.val[i];
        const bool kept = v.x >= 0;
        U2 total;
        const U2 excl = block_scan_exclusive<AddOp, false>(U2{kept ? 1u : 0u, kept ? uint32_t(v.w - v.z) : 0u}, lds, &total);
        const int o = int(before.x + excl.x), off = int(before.y + excl.y);
        const bool write = kept && o < A.max_seg;
        if (write) {
            reinterpret_cast<int4 *>(A.seg)[o] = make_int4(v.x + A.row_c0, v.y, v.z + A.row_g0, v.w + A.row_g0);
            if (A.seg_off) A.seg_off[o] = off;
        }
        if (A.gout) {
            unsigned long long m = __builtin_amdgcn_ballot_w64(write);
            while (m) {
                const int src = __builtin_ctzll(m);
                m &= m - 1;
                const int a = __builtin_amdgcn_readlane(v.z, src), b = __builtin_amdgcn_readlane(v.w, src),
                          ro = __builtin_amdgcn_readlane(off, src);
                for (int k = lane; k < b - a; k += 64)
                    if (ro + k < A.gcap) A.gout[ro + k] = A.p[a + k];
            }
        }
        before = AddOp::combine(before, total);  // (total = this tile's own sums: tile[t])
        done = t + 1;
        if (t == ntile - 1 && threadIdx.x == 0) {
            *A.total = int32_t(before.x);
            if (A.seg_off && int(before.x) <= A.max_seg) A.seg_off[before.x] = int32_t(before.y);
        }
    }
}
__global__ void __launch_bounds__(kT) seg_compact(const SegArgs A) { seg_compact_body(A); }

// A batch of ONE workgroup's worth of genes (kBlockGenes = 2 048: one genome's contig, C1): the five launches above as one --
// "the workgroups before this one" are none, every reduction over workgroups is empty, and what a stage reads of the stage
// before it was written by lanes of the same workgroup (a barrier and a workgroup-scope fence in between).  The one place
// where a lane leaves a stage early (seg_replay_body) has no barrier behind it inside the stage.
__global__ void __launch_bounds__(kT) seg_small(const SegArgs A, const int build_flags) {
    if (build_flags) {
        for (int w = threadIdx.x; w * 8 < A.n_genes + 8; w += kT)
            seg_flags_body(A.cptr, A.n_contigs, A.n_genes, const_cast<uint8_t *>(A.flags), w);
        __threadfence_block();
        __syncthreads();
    }
    seg_fold_body(A);
    __threadfence_block();
    __syncthreads();
    seg_replay_body(A);
    __threadfence_block();
    __syncthreads();
    seg_validate_body(A);
    __threadfence_block();
    __syncthreads();
    seg_compact_body(A);
}

inline size_t align256s(size_t x) { return (x + 255) & ~size_t(255); }

}  // namespace

size_t segment_raw_capacity(int n_genes, int n_contigs) {
    // a run needs a gene, and two runs of one contig a gene between them
    const size_t n = size_t(n_genes), k = size_t(n_contigs);
    return std::min(n, n / 2 + k) + 1;
}

size_t segment_workspace_bytes(int n_genes, int n_contigs) {
    const size_t n = size_t(n_genes), nb = (n + kBlockGenes - 1) / kBlockGenes, cap = segment_raw_capacity(n_genes, n_contigs);
    return align256s(nb * kT * sizeof(SegE)) + align256s((nb + 1) * sizeof(SegE)) + align256s((n + 8) * 4) + align256s(cap * 8) +
           align256s(cap * 16) + align256s((cap / kT + 2) * 8) + align256s(n + 8) + 256;
}

// All pointers are device pointers; `flags` may be null (built here from d_cptr into the workspace);
// d_seg_off may be null.  d_total receives the number of kept rows (it may exceed max_seg: the rows
// beyond are not written).
hipError_t launch_segment(const double *d_p, const uint8_t *d_ann, const uint8_t *d_flags, const int32_t *d_cptr,
                          int n_genes, int n_contigs, const SegParams &params, int32_t *d_seg, int max_seg,
                          int32_t *d_seg_off, int32_t *d_total, void *d_work, hipStream_t stream, double *d_gather, int gather_cap) {
    if (n_contigs <= 0 || n_genes <= 0) {
        if (d_seg_off) (void)hipMemsetAsync(d_seg_off, 0, 4, stream);
        return hipMemsetAsync(d_total, 0, 4, stream);
    }
    const size_t n = size_t(n_genes), nb = (n + kBlockGenes - 1) / kBlockGenes, cap = segment_raw_capacity(n_genes, n_contigs);
    char *w = static_cast<char *>(d_work);
    SegArgs a{};
    a.p = d_p;
    a.ann = d_ann;
    a.cptr = d_cptr;
    a.n_genes = n_genes;
    a.n_contigs = n_contigs;
    a.thr = params.threshold;
    a.n_cds = params.n_cds;
    a.edge = params.edge_distance;
    a.trim = params.trim;
    a.carry = params.carry;
    a.criterion = params.criterion;
    a.n_bio = params.n_biopfams;
    a.avg_thr = params.average_threshold;
    a.bio_ptr = params.bio_ptr;
    a.bio_id = params.bio_id;
    a.lane = reinterpret_cast<SegE *>(w);
    w += align256s(nb * kT * sizeof(SegE));
    a.block = reinterpret_cast<SegE *>(w);
    w += align256s((nb + 1) * sizeof(SegE));
    a.pre = reinterpret_cast<int32_t *>(w);
    w += align256s((n + 8) * 4);
    a.raw = reinterpret_cast<int2 *>(w);
    w += align256s(cap * 8);
    a.val = reinterpret_cast<int4 *>(w);
    w += align256s(cap * 16);
    a.tile = reinterpret_cast<int2 *>(w);
    w += align256s((cap / kT + 2) * 8);
    uint8_t *own_flags = reinterpret_cast<uint8_t *>(w);
    w += align256s(n + 8);
    a.n_raw = reinterpret_cast<int32_t *>(w);
    a.seg = d_seg;
    a.max_seg = max_seg;
    a.seg_off = d_seg_off;
    a.total = d_total;
    a.gout = d_seg_off ? d_gather : nullptr;  // (the rows' probabilities are laid out by the rows' gene offsets)
    a.gcap = gather_cap;
    a.row_c0 = params.row_contig0;
    a.row_g0 = params.row_gene0;
    static const bool fused_small = [] {  // GECCO_CRF_SEGMENT_FUSED=0: the five launches for every batch (tests, A/B)
        const char *e = std::getenv("GECCO_CRF_SEGMENT_FUSED");
        return !(e && e[0] == '0');
    }();
    if (nb == 1 && fused_small) {
        a.flags = d_flags ? d_flags : own_flags;
        hipLaunchKernelGGL(seg_small, dim3(1), dim3(kT), 0, stream, a, d_flags ? 0 : 1);
        return hipGetLastError();
    }
    if (!d_flags) {
        hipLaunchKernelGGL(seg_flags, dim3(unsigned((n / 8 + 1 + kT - 1) / kT)), dim3(kT), 0, stream, d_cptr, n_contigs, n_genes, own_flags);
        d_flags = own_flags;
    }
    a.flags = d_flags;
    const int tiles_cap = int(std::min<size_t>(cap / kT + 1, 2048));
    hipLaunchKernelGGL(seg_fold, dim3(nb), dim3(kT), 0, stream, a);
    hipLaunchKernelGGL(seg_replay, dim3(nb), dim3(kT), 0, stream, a);
    hipLaunchKernelGGL(seg_validate, dim3(tiles_cap), dim3(kT), 0, stream, a);
    hipLaunchKernelGGL(seg_compact, dim3(tiles_cap), dim3(kT), 0, stream, a);
    return hipGetLastError();
}

// ---- wire format of the batch driver: one DEGREE BYTE per gene instead of a 4-byte row pointer -------------------------------
// A chunk's row pointers are the prefix sums of its genes' domain counts, and a gene has a handful of domains: the host
// sends the counts as bytes (a quarter of the row pointers' bytes; SURVEY.md 8d, PCIe-inclusive level) and three small
// launches on the compute stream turn them into the int32 row pointers the kernels read (base = the caller's offset of
// the chunk's first domain, so that the attribute array is still addressed with the caller's offsets).
namespace {
constexpr int kDegT = 256, kDegPer = 16, kDegBlock = kDegT * kDegPer;  // 4 096 genes per workgroup, 16 per lane (one 16-byte load)

__device__ __forceinline__ int deg_lane_counts(const uint8_t *__restrict__ deg, int n, int i0, int (&c)[kDegPer]) {
    // (the buffer is allocated with 32 bytes of slack behind the chunk: the 16-byte load may run past n, the values are masked)
    const uint4 v = i0 < n ? *reinterpret_cast<const uint4 *>(deg + i0) : make_uint4(0u, 0u, 0u, 0u);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    int sum = 0;
#pragma unroll
    for (int k = 0; k < kDegPer; ++k) {
        c[k] = (i0 + k < n) ? int((w[k >> 2] >> (8 * (k & 3))) & 0xffu) : 0;
        sum += c[k];
    }
    return sum;
}
__device__ __forceinline__ int deg_block_exclusive(int mine, int *lds /* kDegT / 64 + 1 */, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    int pre = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kDegT / 64; ++w) {
        const int t = lds[w];
        if (w < wave) pre += t;
        all += t;
    }
    __syncthreads();
    *total = all;
    return pre + incl - mine;
}
// 16-bit attribute indices of the compact wire format -> the 32-bit ones every kernel reads: 8 per lane, one 16-byte load
// and two 16-byte stores (in and out are 16-byte aligned device buffers with 8 elements of slack)
__device__ __forceinline__ void widen_attr_ids(const uint16_t *__restrict__ in, int64_t n, int32_t *__restrict__ out, int64_t block) {
    const int64_t i = (block * kDegT + threadIdx.x) * 8;
    if (i >= n) return;
    const uint4 v = *reinterpret_cast<const uint4 *>(in + i);
    int4 *o = reinterpret_cast<int4 *>(out + i);
    o[0] = make_int4(int(v.x & 0xFFFFu), int(v.x >> 16), int(v.y & 0xFFFFu), int(v.y >> 16));
    o[1] = make_int4(int(v.z & 0xFFFFu), int(v.z >> 16), int(v.w & 0xFFFFu), int(v.w >> 16));
}
// launch 1: workgroups [0, nb) total their 4 096 degree bytes; the workgroups behind them widen attribute indices
__global__ void __launch_bounds__(kDegT) wire_block_sums(const uint8_t *__restrict__ deg, int n, int nb, int32_t *__restrict__ sums,
                                                        const uint16_t *__restrict__ at16, int64_t nnz, int32_t *__restrict__ at32) {
    if (int(blockIdx.x) >= nb) {
        widen_attr_ids(at16, nnz, at32, int64_t(blockIdx.x) - nb);
        return;
    }
    __shared__ int lds[kDegT / 64];
    int c[kDegPer], total;
    const int mine = deg_lane_counts(deg, n, blockIdx.x * kDegBlock + threadIdx.x * kDegPer, c);
    (void)deg_block_exclusive(mine, lds, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
// launch 2: every workgroup adds up the totals of the workgroups before it itself (a chunk of a million genes has 245 of
// them: one word per lane) and writes its 4 096 row pointers
__global__ void __launch_bounds__(kDegT) wire_row_ptr(const uint8_t *__restrict__ deg, int n, const int32_t *__restrict__ sums,
                                                     int32_t base, int32_t *__restrict__ row_ptr) {
    __shared__ int lds[kDegT / 64];
    int before = 0;
    for (int i = threadIdx.x; i < int(blockIdx.x); i += kDegT) before += sums[i];
    int ahead;
    (void)deg_block_exclusive(before, lds, &ahead);
    int c[kDegPer], total;
    const int i0 = blockIdx.x * kDegBlock + threadIdx.x * kDegPer;
    const int mine = deg_lane_counts(deg, n, i0, c);
    int run = base + ahead + deg_block_exclusive(mine, lds, &total);
#pragma unroll
    for (int k = 0; k < kDegPer; ++k) {
        if (i0 + k <= n) row_ptr[i0 + k] = run;  // (entry n = base + the chunk's total: written by the lane that owns position n)
        run += c[k];
    }
}
}  // namespace

hipError_t launch_contig_flags(const int32_t *d_cptr, int n_contigs, int n_genes, uint8_t *d_flags, hipStream_t stream) {
    if (n_genes < 0 || n_contigs < 0) return hipErrorInvalidValue;
    if (n_contigs == 0 || n_genes == 0) return hipMemsetAsync(d_flags, 0, size_t(n_genes) + 8, stream);
    hipLaunchKernelGGL(seg_flags, dim3(unsigned((size_t(n_genes) / 8 + 1 + kT - 1) / kT)), dim3(kT), 0, stream, d_cptr, n_contigs, n_genes, d_flags);
    return hipGetLastError();
}

namespace {
__global__ void __launch_bounds__(256) copy_block(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}
}  // namespace
hipError_t launch_copy_block(const void *src, void *dst, size_t bytes, hipStream_t stream) {
    if ((bytes & 15) || (reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(dst) & 15)) return hipErrorInvalidValue;
    if (!bytes) return hipSuccess;
    const size_t n16 = bytes / 16;
    hipLaunchKernelGGL(copy_block, dim3(unsigned((n16 + 255) / 256)), dim3(256), 0, stream, static_cast<const uint4 *>(src),
                       static_cast<uint4 *>(dst), n16);
    return hipGetLastError();
}

size_t degree_scratch_bytes(int n) { return (size_t((n + kDegBlock) / kDegBlock) + 1) * 4; }

hipError_t launch_wire_format(const uint8_t *d_deg, int n, int32_t base, int32_t *d_row_ptr, int32_t *d_scratch,
                              const uint16_t *d_attr16, int64_t nnz, int32_t *d_attr32, hipStream_t stream) {
    if (n < 0 || nnz < 0) return hipErrorInvalidValue;
    const int nb = d_deg ? (n + kDegBlock) / kDegBlock : 0;  // (n + 1 positions: position n carries the total)
    const int nw = d_attr16 ? int((nnz + kDegT * 8 - 1) / (kDegT * 8)) : 0;
    if (nb + nw == 0) return hipSuccess;
    hipLaunchKernelGGL(wire_block_sums, dim3(nb + nw), dim3(kDegT), 0, stream, d_deg, n, nb, d_scratch, d_attr16, nnz, d_attr32);
    if (nb) hipLaunchKernelGGL(wire_row_ptr, dim3(nb), dim3(kDegT), 0, stream, d_deg, n, d_scratch, base, d_row_ptr);
    return hipGetLastError();
}
hipError_t launch_degree_to_row_ptr(const uint8_t *d_deg, int n, int32_t base, int32_t *d_row_ptr, int32_t *d_scratch, hipStream_t stream) {
    return launch_wire_format(d_deg, n, base, d_row_ptr, d_scratch, nullptr, 0, nullptr, stream);
}

}  // namespace gecco
