// Whole-contig inference kernels (north-star extensions; rows F and V of SURVEY.md §8a):
//   F  [EXT] CRF.predict_marginals_single(all feats) = crf1dc_alpha_score / crf1dc_beta_score /
//      crf1dc_marginal_point over a whole contig -- NOT what GECCO's predict path computes
//      (that is the windowed kernel);
//   V  [EXT] CRF.predict_single = crf1dc_viterbi, first-argmax tie-breaking.
// Two-label models.  Contigs range from a handful of genes to 50 000 (BASELINE.json
// configs[4]: scan-length-bound), so nothing here is "one lane per contig": all genes of all
// contigs form ONE flat sequence and both recursions are associative scans over it -- max-plus
// 2x2 matrices (or, without path scores, clamp maps on score differences: see COp) for V,
// sum-product 2x2 matrices with exact power-of-two rescaling for F.  The scans are segmented: an
// element that contains a contig's first gene forgets whatever came before it (`rs` flag of the
// matrix elements; a constant map in the difference form), so every contig sees exactly its own
// sequential values; the backward direction does the same with the contig's last gene.
//
// Scan levels: a lane folds kGPL consecutive genes (registers, sequential; coalesced loads
// transposed through padded LDS), a wave scans its 64 lane products with DPP row_shr/row_bcast
// (no LDS), the four wave totals meet in LDS.  Per-workgroup totals are never scanned by a launch of
// their own: the consumer kernels look back / ahead over the neighbouring workgroups' totals (see
// lookback_prefix, lookahead_suffix).  Every gene is then replayed from the exact value entering its
// lane, using CRFsuite's operation order inside the lane.  Values entering a lane come from
// composed elements, i.e. they equal the strictly sequential values whenever the additions are
// exact (integer-valued weights: ties and first-argmax included) and to rounding otherwise.
#include <type_traits>

#include "crf_device.hpp"
#include "crf_scan.hpp"
#include "crf_vd_short.hpp"

namespace gecco {
namespace {


// ---------------------------------------------------------------- scan operators (elements: crf_device.hpp)
struct VOp {
    static __device__ __forceinline__ VE identity() {
        const double ninf = -__builtin_huge_val();
        return VE{0.0, ninf, ninf, 0.0, 0.0};
    }
    static __device__ __forceinline__ VE combine(const VE &a, const VE &b) {  // a earlier, b later
        if (b.rs != 0.0) return b;  // b restarts a contig: nothing before it matters
        return VE{fmax(a.a00 + b.a00, a.a01 + b.a10), fmax(a.a00 + b.a01, a.a01 + b.a11),
                  fmax(a.a10 + b.a00, a.a11 + b.a10), fmax(a.a10 + b.a01, a.a11 + b.a11), a.rs};
    }
};
struct FOp {
    static __device__ __forceinline__ FE identity() { return FE{1.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0}; }
    // the product without the power-of-two renormalisation: for the eight steps a lane folds by itself (entries of the
    // max-normalised factors lie in (0, 1]: eight of them cannot leave the range), renormalised once at the end
    static __device__ __forceinline__ FE combine_raw(const FE &a, const FE &b) {
        if (b.rs != 0.0) return b;
        FE c;
        c.a00 = fma(a.a01, b.a10, a.a00 * b.a00);
        c.a01 = fma(a.a01, b.a11, a.a00 * b.a01);
        c.a10 = fma(a.a11, b.a10, a.a10 * b.a00);
        c.a11 = fma(a.a11, b.a11, a.a10 * b.a01);
        c.ex = a.ex + b.ex;
        c.ms = a.ms + b.ms;
        c.rs = a.rs;
        return c;
    }
    static __device__ __forceinline__ FE renorm(FE c) {
        int e;
        (void)frexp(fmax(fmax(c.a00, c.a01), fmax(c.a10, c.a11)), &e);
        c.a00 = ldexp(c.a00, -e);
        c.a01 = ldexp(c.a01, -e);
        c.a10 = ldexp(c.a10, -e);
        c.a11 = ldexp(c.a11, -e);
        c.ex += double(e);
        return c;
    }
    static __device__ __forceinline__ FE combine(const FE &a, const FE &b) {
        if (b.rs != 0.0) return b;
        return renorm(combine_raw(a, b));
    }
};
// Backward matrices B_t (beta_t = B_t beta_{t+1}): a contig's last gene contributes 1 1^T, so whatever
// follows it only scales the product; `rs` marks "contains a last gene" and the EARLIER element absorbs.
struct FOpB {
    static __device__ __forceinline__ FE identity() { return FOp::identity(); }
    static __device__ __forceinline__ FE combine(const FE &a, const FE &b) {  // a earlier in the sequence
        if (a.rs != 0.0) return a;
        FE bb = b;
        const double flag = b.rs;
        bb.rs = 0.0;
        FE c = FOp::combine(a, bb);
        c.rs = flag;
        return c;
    }
    static __device__ __forceinline__ FE combine_raw(const FE &a, const FE &b) {
        if (a.rs != 0.0) return a;
        FE bb = b;
        const double flag = b.rs;
        bb.rs = 0.0;
        FE c = FOp::combine_raw(a, bb);
        c.rs = flag;
        return c;
    }
    static __device__ __forceinline__ FE renorm(const FE &c) { return FOp::renorm(c); }
};
// Flag-free products for workgroups that own WHOLE contigs and only want marginals (f_short): a contig's first
// gene contributes the rank-one matrix 1 e^T, so whatever stands before it only scales the product's rows, and
// its last gene contributes 1 1^T, so whatever follows only scales the columns -- directions are all a marginal
// needs.  No `rs` test (7 selects per combine), no exponent / maxima sums, 4 doubles through the DPP network.
struct F4Op {
    static __device__ __forceinline__ F4 identity() { return F4{1.0, 0.0, 0.0, 1.0}; }
    static __device__ __forceinline__ F4 combine_raw(const F4 &a, const F4 &b) {  // a earlier; no renormalisation (lane folds)
        F4 c;
        c.a00 = fma(a.a01, b.a10, a.a00 * b.a00);
        c.a01 = fma(a.a01, b.a11, a.a00 * b.a01);
        c.a10 = fma(a.a11, b.a10, a.a10 * b.a00);
        c.a11 = fma(a.a11, b.a11, a.a10 * b.a01);
        return c;
    }
    static __device__ __forceinline__ F4 renorm(F4 c) {
        int e;
        (void)frexp(fmax(fmax(c.a00, c.a01), fmax(c.a10, c.a11)), &e);
        c.a00 = ldexp(c.a00, -e);
        c.a01 = ldexp(c.a01, -e);
        c.a10 = ldexp(c.a10, -e);
        c.a11 = ldexp(c.a11, -e);
        return c;
    }
    static __device__ __forceinline__ F4 combine(const F4 &a, const F4 &b) { return renorm(combine_raw(a, b)); }
};


// Look-back over per-workgroup totals instead of a separate scan launch: the exclusive prefix of
// workgroup b is total[j] (x) ... (x) total[b-1] from the nearest workgroup j whose span contains a
// contig start (`rs`), because everything before a contig start is forgotten.  With contigs of a
// few hundred genes and 2048-gene workgroups that is one 64-wide coalesced read of neighbouring
// totals and one wave scan; a 50 000-gene contig walks back 25 totals, still one read.  Every wave
// does it for itself (no barrier).
__device__ __forceinline__ VE wave_bcast63(const VE &v) {
    auto b = [](double x) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 63), __builtin_amdgcn_readlane(__double2loint(x), 63));
    };
    return VE{b(v.a00), b(v.a01), b(v.a10), b(v.a11), b(v.rs)};
}
__device__ __forceinline__ VE lookback_prefix(const VE *__restrict__ totals, int b) {
    const int lane = threadIdx.x & 63;
    VE acc = VOp::identity();
    for (int hi = b; hi > 0; hi -= 64) {
        const int idx = hi - 1 - lane;  // lanes hold the totals back to front
        const VE e = idx >= 0 ? totals[idx] : VOp::identity();
        const VE r = wave_bcast63(wave_scan_inclusive<VOp, true>(e));
        acc = VOp::combine(r, acc);
        if (r.rs != 0.0) break;
    }
    return acc;
}
// Same for the back-to-front label maps: the suffix of workgroup b is map[b+1] o map[b+2] o ...
// up to the first constant map (a workgroup that contains a contig end, or in which the two
// Viterbi paths have merged): whatever lies to its right cannot matter.
__device__ __forceinline__ uint32_t lookahead_suffix(const uint32_t *__restrict__ maps, int b, int nb) {
    const int lane = threadIdx.x & 63;
    uint32_t acc = MapOp::identity();
    for (int lo = b + 1; lo < nb; lo += 64) {
        const int idx = lo + lane;
        const uint32_t e = idx < nb ? maps[idx] : MapOp::identity();
        const uint32_t r = uint32_t(__builtin_amdgcn_readlane(int(wave_scan_inclusive<MapOp, false>(e)), 63));
        acc = MapOp::combine(acc, r);
        if (acc == 0u || acc == 3u) break;
    }
    return acc;
}

__device__ __forceinline__ CE lookback_prefix(const CE *__restrict__ totals, int b) {
    const int lane = threadIdx.x & 63;
    auto bc = [](double x) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 63), __builtin_amdgcn_readlane(__double2loint(x), 63));
    };
    CE acc = COp::identity();
    for (int hi = b; hi > 0; hi -= 64) {
        const int idx = hi - 1 - lane;  // lanes hold the totals back to front
        const CE e = idx >= 0 ? totals[idx] : COp::identity();
        const CE s = wave_scan_inclusive<COp, true>(e);
        const CE r{bc(s.a), bc(s.L), bc(s.H)};
        acc = COp::combine(r, acc);
        if (acc.L == acc.H) break;  // a constant map: nothing further back can matter
    }
    return acc;
}

__device__ __forceinline__ FE wave_bcast63(const FE &v) {
    auto b = [](double x) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 63), __builtin_amdgcn_readlane(__double2loint(x), 63));
    };
    return FE{b(v.a00), b(v.a01), b(v.a10), b(v.a11), b(v.ex), b(v.ms), b(v.rs)};
}
__device__ __forceinline__ FE lookback_prefix(const FE *__restrict__ totals, int b) {
    const int lane = threadIdx.x & 63;
    FE acc = FOp::identity();
    for (int hi = b; hi > 0; hi -= 64) {
        const int idx = hi - 1 - lane;
        const FE e = idx >= 0 ? totals[idx] : FOp::identity();
        const FE r = wave_bcast63(wave_scan_inclusive<FOp, true>(e));
        acc = FOp::combine(r, acc);
        if (r.rs != 0.0) break;
    }
    return acc;
}
// suffix of the backward matrices of the workgroups to the right, up to the first one that contains a
// contig end
__device__ __forceinline__ FE lookahead_suffix(const FE *__restrict__ totals, int b, int nb) {
    const int lane = threadIdx.x & 63;
    FE acc = FOpB::identity();
    for (int lo = b + 1; lo < nb; lo += 64) {
        const int idx = lo + lane;
        const FE e = idx < nb ? totals[idx] : FOpB::identity();
        const FE r = wave_bcast63(wave_scan_inclusive<FOpB, false>(e));
        acc = FOpB::combine(acc, r);
        if (acc.rs != 0.0) break;
    }
    return acc;
}

// ---------------------------------------------------------------- row S: state scores
// s[y] = sum over the gene's attributes of w[a][y], added in CSR order ([EXT] crf1dt_state_score).  A workgroup
// takes 512 consecutive genes: their attribute ids are ONE contiguous stretch of the CSR, which is loaded
// attribute-per-lane (coalesced; every id gathers its 16-byte weight pair once), parked in LDS 1024 pairs at a
// time, and every gene then adds up its own run from there.  (One lane per gene walking its run through global
// memory -- the first version -- is a chain of three dependent round trips per gene: 18 us for the 2 M genes of
// C3; this arrangement is the windowed kernel's stage 1 and takes about half.)
// MODE 0: d = s[1] - s[0];  1: d and max(s[0], s[1]);  2: the pair (s[0], s[1]).
constexpr int kStateGenes = 512, kStatePark = 1024;
typedef double st_f64x2 __attribute__((ext_vector_type(2)));
typedef int st_i32x4 __attribute__((ext_vector_type(4)));
// the sums of genes [g_first, g_end) of one workgroup: lane `tid` owns genes g_first + k kT + tid (coalesced row pointers
// and outputs), PARK weight pairs are parked at a time
template <int GPLS, int PARK>
__device__ __forceinline__ void block_state_sums(const int32_t *__restrict__ gene_ptr, const int32_t *__restrict__ attr_id,
                                                 const double2 *__restrict__ wtab01, int n_attrs, int g_first, int g_end,
                                                 st_f64x2 *park, double (&s0)[GPLS], double (&s1)[GPLS]) {
    constexpr int APL = PARK / kT;
    const int tid = threadIdx.x;
    const uint32_t lo_tile = uint32_t(gene_ptr[g_first]), hi_tile = uint32_t(gene_ptr[g_end]);
    const uint32_t n_run = hi_tile - lo_tile;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(attr_id + lo_tile), 0, n_run << 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double2 *>(wtab01), 0, uint32_t(n_attrs) << 4, 0x00020000);
    uint32_t lo[GPLS], hi[GPLS];
#pragma unroll
    for (int k = 0; k < GPLS; ++k) {
        const int g = g_first + k * kT + tid;
        lo[k] = hi[k] = 0;
        if (g < g_end) {
            lo[k] = uint32_t(gene_ptr[g]);
            hi[k] = uint32_t(gene_ptr[g + 1]);
        }
        s0[k] = s1[k] = 0.0;
    }
#pragma unroll 1
    for (uint32_t base = 0; base < n_run; base += PARK) {
        int id[APL];
#pragma unroll
        for (int a = 0; a < APL; ++a) id[a] = __builtin_amdgcn_raw_buffer_load_b32(ra, int((base + a * kT + tid) << 2), 0, 0);
        st_i32x4 w[APL];
#pragma unroll
        for (int a = 0; a < APL; ++a)  // ids outside the dictionary land outside the table and read (+0.0, +0.0)
            w[a] = __builtin_amdgcn_raw_buffer_load_b128(rw, int(min(uint32_t(id[a]), 0x0FFFFFFFu) << 4), 0, 0);
#pragma unroll
        for (int a = 0; a < APL; ++a) park[a * kT + tid] = st_f64x2{__hiloint2double(w[a].y, w[a].x), __hiloint2double(w[a].w, w[a].z)};
        __syncthreads();
        const uint32_t c0 = lo_tile + base, c1 = c0 + PARK;
#pragma unroll
        for (int k = 0; k < GPLS; ++k) {
            uint32_t q = max(lo[k], c0) - c0;
            const uint32_t e = min(hi[k], c1) - c0;
            for (; q < e && hi[k] > c0; ++q) {
                const st_f64x2 v = park[q];
                s0[k] += v.x;
                s1[k] += v.y;
            }
        }
        if (base + PARK < n_run) __syncthreads();
    }
}
template <int MODE>
__global__ void __launch_bounds__(kT) seq_state_blocks(const int32_t *__restrict__ gene_ptr, const int32_t *__restrict__ attr_id,
                                                        const double2 *__restrict__ wtab01, int n_attrs, int n_genes,
                                                        double *__restrict__ out_d, double *__restrict__ out_m,
                                                        double2 *__restrict__ out_s) {
    __shared__ st_f64x2 park[kStatePark];
    constexpr int GPLS = kStateGenes / kT;
    const int tid = threadIdx.x;
    const int g_first = blockIdx.x * kStateGenes, g_end = min(g_first + kStateGenes, n_genes);
    double s0[GPLS], s1[GPLS];
    block_state_sums<GPLS, kStatePark>(gene_ptr, attr_id, wtab01, n_attrs, g_first, g_end, park, s0, s1);
#pragma unroll
    for (int k = 0; k < GPLS; ++k) {
        const int g = g_first + k * kT + tid;
        if (g < g_end) {
            if (MODE == 2) {
                out_s[g] = make_double2(s0[k], s1[k]);
            } else {
                out_d[g] = s1[k] - s0[k];
                if (MODE == 1) out_m[g] = fmax(s0[k], s1[k]);
            }
        }
    }
}

// ---------------------------------------------------------------- lane-local loads
// lane `slot` of the workgroup owns genes [g0, g0 + kGPL); flags bit0 = first gene of a contig,
// bit1 = last gene of a contig.
struct LaneGenes {
    double2 s[kGPL];
    uint32_t first, last;  // bit k
    int g0, cnt;
};
// The workgroup's kT*kGPL genes are read with perfectly coalesced 16-B loads (lane i takes
// entries i, i+kT, ...) and handed to their owners through LDS: a lane reading its own kGPL
// consecutive entries straight from global memory touches 64 different cache lines per load
// instruction (measured 2.3 TB/s effective on v_fold; LDS staging is the standard fix).  Rows
// are padded to kGPL+1 entries (144 B) so that both the b128 writes and the b128 reads of the
// transpose are bank-conflict free.
struct LaneStage {
    double2 st[kT * (kGPL + 1)];
    uint8_t fl[kT * kGPL];
};
__device__ __forceinline__ LaneGenes load_lane(const SeqArgs &A, int slot, LaneStage &stg) {
    const int base = blockIdx.x * kT * kGPL;
#pragma unroll
    for (int j = 0; j < kGPL; ++j) {
        const int idx = j * kT + slot, g = base + idx;
        const bool ok = g < A.n_genes;
        stg.st[(idx / kGPL) * (kGPL + 1) + idx % kGPL] = ok ? A.state[g] : make_double2(0.0, 0.0);
    }
    {   // flags: kGPL bytes per lane = one 8-byte word per lane, coalesced
        static_assert(kGPL == 8, "flag word assumes 8 genes per lane");
        const int g0 = base + slot * kGPL;
        uint64_t w = 0;
        if (g0 + kGPL <= A.n_genes) {
            w = *reinterpret_cast<const uint64_t *>(A.flags + g0);
        } else {
            for (int k = 0; k < kGPL; ++k)
                if (g0 + k < A.n_genes) w |= uint64_t(A.flags[g0 + k]) << (8 * k);
        }
        *reinterpret_cast<uint64_t *>(stg.fl + slot * kGPL) = w;
    }
    __syncthreads();
    LaneGenes L;
    L.g0 = base + slot * kGPL;
    L.cnt = min(kGPL, A.n_genes - L.g0);
    L.first = L.last = 0;
    const uint64_t w = *reinterpret_cast<const uint64_t *>(stg.fl + slot * kGPL);
#pragma unroll
    for (int k = 0; k < kGPL; ++k) {
        L.s[k] = stg.st[slot * (kGPL + 1) + k];
        const uint32_t f = uint32_t(w >> (8 * k)) & 0xffu;
        L.first |= (f & 1u) << k;
        L.last |= ((f >> 1) & 1u) << k;
    }
    __syncthreads();  // the stage may be reused
    return L;
}

// Per-lane rows of kGPL consecutive 16-byte entries <-> global memory with coalesced accesses: a lane
// touching its own kGPL entries directly hits 64 different cache lines per instruction.  Same
// padded LDS transpose as load_lane; `stg` must not be in use (both end with a barrier).
__device__ __forceinline__ void store_lane_rows(double2 *__restrict__ gmem, int n_genes, int slot, const double2 (&v)[kGPL],
                                                LaneStage &stg) {
    const int base = blockIdx.x * kT * kGPL;
#pragma unroll
    for (int k = 0; k < kGPL; ++k) stg.st[slot * (kGPL + 1) + k] = v[k];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kGPL; ++j) {
        const int idx = j * kT + slot, g = base + idx;
        if (g < n_genes) gmem[g] = stg.st[(idx / kGPL) * (kGPL + 1) + idx % kGPL];
    }
    __syncthreads();
}
__device__ __forceinline__ void load_lane_rows(const double2 *__restrict__ gmem, int n_genes, int slot, double2 (&v)[kGPL],
                                               LaneStage &stg) {
    const int base = blockIdx.x * kT * kGPL;
#pragma unroll
    for (int j = 0; j < kGPL; ++j) {
        const int idx = j * kT + slot, g = base + idx;
        stg.st[(idx / kGPL) * (kGPL + 1) + idx % kGPL] = g < n_genes ? gmem[g] : make_double2(0.0, 0.0);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kGPL; ++k) v[k] = stg.st[slot * (kGPL + 1) + k];
    __syncthreads();
}

// =====================================================================================
// V: max-plus.  delta_0 = s_0; delta_t[j] = max_i(delta_{t-1}[i] + trans[i][j]) + s_t[j]; ties keep
// the smaller i (CRFsuite updates on strict '<'); the end label is the first argmax.
// =====================================================================================
__global__ void __launch_bounds__(kT) v_fold(const SeqArgs A) {
    __shared__ VE lds[kT / 64];
    __shared__ LaneStage stg;
    const LaneGenes L = load_lane(A, threadIdx.x, stg);
    VE P = VOp::identity();
#pragma unroll
    for (int k = 0; k < kGPL; ++k) {
        if (k < L.cnt) {
            const bool first = (L.first >> k) & 1u;
            const VE S{(first ? 0.0 : A.t00) + L.s[k].x, (first ? 0.0 : A.t01) + L.s[k].y,
                       (first ? 0.0 : A.t10) + L.s[k].x, (first ? 0.0 : A.t11) + L.s[k].y, first ? 1.0 : 0.0};
            P = VOp::combine(P, S);
        }
    }
    VE total;
    const VE excl = block_scan_exclusive<VOp, false>(P, lds, &total);
    A.vLane[blockIdx.x * kT + threadIdx.x] = excl;
    if (threadIdx.x == 0) A.vBlock[blockIdx.x] = total;
}

// replay with the exact entering delta; per gene the map label_{t+1} -> label_t (its own end label
// for a contig's last gene, else the back-pointers the NEXT gene will compute from delta_t); fold
// the lane's maps and scan them back to front
__global__ void __launch_bounds__(kT) v_replay(const SeqArgs A) {
    __shared__ uint32_t lds[kT / 64];
    __shared__ LaneStage stg;
    const int slot = threadIdx.x;
    const LaneGenes L = load_lane(A, slot, stg);
    const VE M = VOp::combine(lookback_prefix(A.vBlock, blockIdx.x), A.vLane[blockIdx.x * kT + slot]);
    double d0 = M.a00, d1 = M.a01;  // rows are identical once a contig has started (M.rs)
    uint32_t maps = 0;                                         // 2 bits per gene
    uint32_t lane_map = MapOp::identity();
#pragma unroll
    for (int k = 0; k < kGPL; ++k) {
        if (k < L.cnt) {
            if ((L.first >> k) & 1u) {
                d0 = L.s[k].x;
                d1 = L.s[k].y;
            } else {
                const double a0 = d0 + A.t00, b0 = d1 + A.t10, a1 = d0 + A.t01, b1 = d1 + A.t11;
                d0 = (a0 < b0 ? b0 : a0) + L.s[k].x;
                d1 = (a1 < b1 ? b1 : a1) + L.s[k].y;
            }
            uint32_t m;
            if ((L.last >> k) & 1u) {
                const uint32_t end = d0 < d1 ? 1u : 0u;  // first argmax
                m = end | (end << 1);
                A.contigTmp[L.g0 + k] = make_double2(fmax(d0, d1), 0.0);
            } else {
                m = ((d0 + A.t00 < d1 + A.t10) ? 1u : 0u) | ((d0 + A.t01 < d1 + A.t11) ? 2u : 0u);
            }
            maps |= m << (2 * k);
        }
    }
    // lane map: label entering from the right -> label of the lane's first gene
#pragma unroll
    for (int k = kGPL - 1; k >= 0; --k)
        if (k < L.cnt) lane_map = MapOp::combine((maps >> (2 * k)) & 3u, lane_map);
    A.vMaps[blockIdx.x * kT + slot] = maps;
    // back-to-front scan of the lane maps
    uint32_t total;
    A.vLaneMap[blockIdx.x * kT + slot] = block_scan_exclusive_back<MapOp>(lane_map, lds, &total);
    if (slot == 0) A.vBlockMap[blockIdx.x] = total;
}

__device__ __forceinline__ void v_labels_block(const SeqArgs &A, const int blk, const int n_blocks) {
    const int slot = threadIdx.x;
    const int g0 = (blk * kT + slot) * kGPL;
    const int cnt = min(kGPL, A.n_genes - g0);
    const uint32_t block_suf = lookahead_suffix(A.vBlockMap, blk, n_blocks);
    if (cnt <= 0) return;
    // suffix map of everything to the right of this lane, applied to a dummy label (the last gene
    // of the batch ends a contig, so the composition is constant)
    const uint32_t suf = MapOp::combine(A.vLaneMap[blk * kT + slot], block_suf);
    uint32_t lab = suf & 1u;
    const uint32_t maps = A.vMaps[blk * kT + slot];
    uint64_t packed = 0;
#pragma unroll
    for (int k = kGPL - 1; k >= 0; --k) {
        if (k < cnt) {
            lab = (maps >> (2 * k + lab)) & 1u;
            packed |= uint64_t(lab) << (8 * k);
        }
    }
    if (cnt == kGPL) {
        *reinterpret_cast<uint64_t *>(A.y + g0) = packed;
    } else {
        for (int k = 0; k < cnt; ++k) A.y[g0 + k] = int8_t((packed >> (8 * k)) & 0xff);
    }
}
__global__ void __launch_bounds__(kT) v_labels(const SeqArgs A) { v_labels_block(A, blockIdx.x, gridDim.x); }

// long contigs: vd_replay flags the contigs that hold a decision inside the margin, this kernel decodes them again.
// Two LDS buffers: lanes 64.. sum the state scores of the next chunk while lane 0 walks the current one (the walk,
// ~50 cycles per gene, is what a 50 000-gene contig costs: about a millisecond).
__global__ void __launch_bounds__(kT) vd_exact_fix(const SeqArgs A) {
    __shared__ FixStage fx[2];
    // this is the last launch of a decode: it leaves the bound, the candidate counter and the contig flags at zero for the
    // next one (until round 4 a memset per decode did that: two fill kernels, 12 us of the stream per C5 step)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        A.vBound[0] = 0ull;
        A.vBound[1] = 0ull;
    }
    for (int c = blockIdx.x; c < A.n_contigs; c += gridDim.x) {
        if (!A.fix_flag[c]) continue;  // (workgroup-uniform)
        __syncthreads();               // (every lane has seen the flag)
        if (threadIdx.x == 0) A.fix_flag[c] = 0;
        const int gs = A.contig_ptr[c], ge = A.contig_ptr[c + 1];
        if (ge <= gs) continue;
        if (threadIdx.x == 0 && A.vd_stats) {
            atomicAdd(A.vd_stats + 2, 1u);
            atomicAdd(A.vd_stats + 3, uint32_t(ge - gs));
        }
        uint32_t *bpc = exact_bp_words(A, gs);
        double d0 = 0.0, d1 = 0.0;  // delta of the last gene walked so far (every lane holds a copy)
        exact_states(A, gs, min(kFixChunk, ge - gs), fx[0], 0);
        __syncthreads();
        int k = 0;
        for (int cb = gs; cb < ge; cb += kFixChunk, k ^= 1) {
            const int m = min(kFixChunk, ge - cb);
            const double e0 = d0, e1 = d1;
            if (threadIdx.x == 0) exact_walk(A, fx[k], m, cb == gs, d0, d1);
            if (cb + kFixChunk < ge) exact_states(A, cb + kFixChunk, min(kFixChunk, ge - cb - kFixChunk), fx[k ^ 1], 64);
            __syncthreads();
            d0 = fx[k].st[m - 1].x;  // (lane 0's result, for everybody)
            d1 = fx[k].st[m - 1].y;
            exact_backpointers(A, fx[k], m, cb == gs, e0, e1);
            __syncthreads();
            for (int i = threadIdx.x; i < (m + 15) / 16; i += kT) bpc[(cb - gs) / 16 + i] = fx[k].bpw[i];
            __syncthreads();
        }
        exact_backtrack(A, gs, ge, d0, d1);
    }
}

// ---- difference form: fold / replay on 8-byte inputs ------------------------------------------
struct LaneGenesD {
    double d[kGPL];
    uint32_t first, last;
    int g0, cnt;
};
// A lane loads ITS OWN kGPL genes (64 consecutive bytes; a wave's loads cover 4 KB between them, every line used whole): no
// transposition through LDS, no barrier, all loads in flight at once (clamped indices, not a branch per load) -- as in
// crf_vd_short.hpp since round 4.  Positions past the last gene of the batch get the padding value.
__device__ __forceinline__ LaneGenesD load_lane_d(const SeqArgs &A, int slot) {
    LaneGenesD L;
    L.g0 = (int(blockIdx.x) * kT + slot) * kGPL;
    L.cnt = min(kGPL, A.n_genes - L.g0);
    // which of the lane's genes start / end a contig: two bytes per lane, packed by the host (positions past the last
    // gene of the batch count as one-gene contigs)
    const uint32_t bits = A.flat_bits[blockIdx.x * kT + slot];
    L.first = bits & 0xffu;
    L.last = bits >> 8;
#pragma unroll
    for (int k = 0; k < kGPL; ++k) L.d[k] = A.dstate[max(min(L.g0 + k, A.n_genes - 1), 0)];
#pragma unroll
    for (int k = 0; k < kGPL; ++k) L.d[k] = k < L.cnt ? L.d[k] : kVdPad;
    return L;
}

__global__ void __launch_bounds__(kT) vd_fold(const SeqArgs A) {
    __shared__ CE lds[kT / 64];
    const LaneGenesD L = load_lane_d(A, threadIdx.x);
    CE P = COp::identity();
#pragma unroll
    for (int k = 0; k < kGPL; ++k) {
        // a contig's first gene is the constant map L = H = d (its `a` is never used again: a constant map stays one)
        const bool fst = (L.first >> k) & 1u;
        const double c = A.v_k + L.d[k];
        P = COp::combine(P, CE{c, fst ? L.d[k] : A.v_lo + c, fst ? L.d[k] : A.v_hi + c});
    }
    CE total;
    const CE excl = block_scan_exclusive<COp, false>(P, lds, &total);
    reinterpret_cast<CE *>(A.vLane)[blockIdx.x * kT + threadIdx.x] = excl;
    if (threadIdx.x == 0) reinterpret_cast<CE *>(A.vBlock)[blockIdx.x] = total;
    // the largest bound on CRFsuite's accumulated scores over the batch's contigs (crf_vd_short.hpp: vd_bound), for the
    // margins of vd_replay / vd_refine: a contig per lane, one atomic per wave
    if (A.fix_flag && A.csr_gene_ptr) {
        double mx = 0.0;
        for (int c = blockIdx.x * kT + threadIdx.x; c < A.n_contigs; c += gridDim.x * kT) {
            const int gs = A.contig_ptr[c], ge = A.contig_ptr[c + 1];
            if (ge > gs) mx = fmax(mx, vd_bound(A, double(A.csr_gene_ptr[ge] - A.csr_gene_ptr[gs]), double(ge - gs)));
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
        if ((threadIdx.x & 63) == 0 && mx > 0.0)
            atomicMax(A.vBound, static_cast<unsigned long long>(__double_as_longlong(mx)));  // (non-negative: bit patterns order like values)
    }
}

__global__ void __launch_bounds__(kT) vd_replay(const SeqArgs A) {
    __shared__ uint32_t lds[kT / 64];
    const int slot = threadIdx.x;
    const LaneGenesD L = load_lane_d(A, slot);
    const CE M = COp::combine(lookback_prefix(reinterpret_cast<const CE *>(A.vBlock), blockIdx.x),
                              reinterpret_cast<const CE *>(A.vLane)[blockIdx.x * kT + slot]);
    double D = M.L;  // the map entering a lane is constant once a contig has started
    const double d_enter = D;
    uint32_t maps = 0, lane_map = MapOp::identity();
    // Coarse margin (crf_vd_short.hpp): (4 R + 4) ulp(M) with M the batch's largest per-contig bound (vd_fold) and R = 1023,
    // which holds as long as every wave (512 genes) contains a reset -- a contig's first gene or a decision that saturates
    // for certain; a wave without one is reported like a candidate.  `sat8`: which of the lane's genes saturate for certain.
    const bool exact = A.v_exact && A.fix_flag && A.csr_gene_ptr;
    const double margin = exact ? vd_margin(1023.0, __longlong_as_double(static_cast<long long>(A.vBound[0])) * kVdEps)
                                : 1e-6 * fmax(1.0, fmax(fabs(A.v_lo), fabs(A.v_hi)));
    bool sensitive = false;  // some decision of this lane lies within the coarse margin of its threshold
    uint32_t sat8 = 0;
#pragma unroll
    for (int k = 0; k < kGPL; ++k) {
        D = ((L.first >> k) & 1u) ? L.d[k] : fmin(fmax(D, A.v_lo), A.v_hi) + (A.v_k + L.d[k]);
        // back-pointers the next gene will take; a contig's last gene decides the end label (first arg max): both of
        // its "thresholds" are 0
        const bool lst = (L.last >> k) & 1u;
        const double thi = lst ? 0.0 : A.v_hi, tlo = lst ? 0.0 : A.v_lo;
        maps |= ((D > thi ? 1u : 0u) | (D > tlo ? 2u : 0u)) << (2 * k);
        if (k < L.cnt) sensitive |= fabs(D - thi) <= margin || fabs(D - tlo) <= margin;
        sat8 |= ((D >= A.v_hi + margin || D <= A.v_lo - margin) ? 1u : 0u) << k;
    }
    if (exact) {
        // candidates go to vd_refine (next launch), which finds their distance to the last reset and judges them
        if (sensitive) {
            const uint32_t at = atomicAdd(reinterpret_cast<uint32_t *>(A.vBound + 1), 1u);
            A.vCand[at] = SeqArgs::VdCand{uint32_t(blockIdx.x * kT + slot), 0u, d_enter};
        }
        const bool none = __builtin_amdgcn_ballot_w64((sat8 | L.first) != 0u) == 0ull;  // (positions past the batch count as contig starts)
        if (none && (slot & 63) == 0) {
            const uint32_t at = atomicAdd(reinterpret_cast<uint32_t *>(A.vBound + 1), 1u);
            A.vCand[at] = SeqArgs::VdCand{uint32_t(blockIdx.x * kT + slot) | 0x80000000u, 0u, 0.0};
        }
    }
#pragma unroll
    for (int k = kGPL - 1; k >= 0; --k) lane_map = MapOp::combine((maps >> (2 * k)) & 3u, lane_map);
    A.vMaps[blockIdx.x * kT + slot] = maps | (sat8 << 16);  // (v_labels reads the low 16 bits)
    uint32_t total;
    A.vLaneMap[blockIdx.x * kT + slot] = block_scan_exclusive_back<MapOp>(lane_map, lds, &total);
    if (slot == 0) A.vBlockMap[blockIdx.x] = total;
}

// flat layout: the candidates of vd_replay, one per lane of this (small) launch.  A candidate lane learns r -- the genes since
// the last reset before its first gene -- from the reset bits of the lanes to its left (vMaps, flat_bits), repeats its
// eight steps from the Delta that entered it and tests every decision against the margin of ITS r (vd_margin); only a
// decision inside that margin sends its contig to CRFsuite's own recursion (vd_exact_fix).
__device__ __forceinline__ void vd_flag_contigs(const SeqArgs &A, int g_lo, int g_hi) {  // contigs with a gene in [g_lo, g_hi)
    int lo = 0, hi = A.n_contigs - 1;  // largest c with contig_ptr[c] <= g_lo
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (A.contig_ptr[mid] <= g_lo) lo = mid; else hi = mid - 1;
    }
    for (int c = lo; c < A.n_contigs && A.contig_ptr[c] < g_hi; ++c) A.fix_flag[c] = 1;
}
__device__ __forceinline__ void vd_refine_block(const SeqArgs &A, const int blk, const int n_blocks) {
    const uint32_t count = *reinterpret_cast<const uint32_t *>(A.vBound + 1);
    const double ulpM = __longlong_as_double(static_cast<long long>(A.vBound[0])) * kVdEps;
    const double margin = vd_margin(1023.0, ulpM);
    for (uint32_t i = blk * kT + threadIdx.x; i < count; i += n_blocks * kT) {
        const SeqArgs::VdCand rec = A.vCand[i];
        const int lane = int(rec.lane & 0x7fffffffu), g0 = lane * kGPL;
        if (rec.lane & 0x80000000u) {  // 512 genes without a reset: r is not bounded by 1023 there
            vd_flag_contigs(A, g0, min(g0 + 64 * kGPL, A.n_genes));
            if (A.vd_stats) atomicAdd(A.vd_stats + 1, 1u);
            continue;
        }
        const int cnt = min(kGPL, A.n_genes - g0);
        const uint32_t bits = A.flat_bits[lane], first = bits & 0xffu, last = bits >> 8;
        // r of the gene before the lane's first: its distance to the last gene s that starts a contig or saturates for certain
        int r = 0;
        if (!(first & 1u)) {
            int l = lane - 1;
            uint32_t reset = 0;
            for (; l >= 0; --l) {  // (gene 0 of the batch starts a contig)
                reset = ((A.vMaps[l] >> 16) | A.flat_bits[l]) & 0xffu;
                if (reset) break;
            }
            const int s_gene = l * kGPL + (31 - __builtin_clz(reset));
            r = g0 - 1 - s_gene;
        }
        double D = rec.d_enter;
        bool flagged = false;
        for (int k = 0; k < cnt; ++k) {
            const double dk = A.dstate[g0 + k];
            const bool fst = (first >> k) & 1u;
            r = fst ? 0 : ((D >= A.v_hi + margin || D <= A.v_lo - margin) ? 1 : r + 1);
            D = fst ? dk : fmin(fmax(D, A.v_lo), A.v_hi) + (A.v_k + dk);
            const bool lst = (last >> k) & 1u;
            const double thi = lst ? 0.0 : A.v_hi, tlo = lst ? 0.0 : A.v_lo;
            const double mr = vd_margin(double(r), ulpM);
            flagged |= fabs(D - thi) <= mr || fabs(D - tlo) <= mr;
        }
        if (A.vd_stats) atomicAdd(A.vd_stats + 0, 1u);
        if (flagged) {
            vd_flag_contigs(A, g0, g0 + cnt);
            if (A.vd_stats) atomicAdd(A.vd_stats + 1, 1u);
        }
    }
}

// The labels of the batch and the judgement of vd_replay's candidates in ONE launch (they do not depend on each other; both
// come before vd_exact_fix): the first `nb` workgroups write labels, the last kRefineBlocks judge -- a launch less in the
// chain fold -> replay -> labels + refine -> exact_fix that a batch of long contigs waits for.  (vd_exact_fix stays a launch of
// its own: it overwrites labels and reads flags other workgroups wrote, and inside one launch that ordering needs a
// `__threadfence()` per workgroup -- a write-back of an L2 that another stream's tiles are filling: 350 instead of 83 us
// per C5 step, measured in round 5.)
constexpr int kRefineBlocks = 8;
__global__ void __launch_bounds__(kT) v_labels_refine(const SeqArgs A, const int nb) {
    if (int(blockIdx.x) < nb)
        v_labels_block(A, blockIdx.x, nb);
    else
        vd_refine_block(A, int(blockIdx.x) - nb, kRefineBlocks);
}

// short contigs: ONE launch per decoder (crf_vd_short.hpp)
__global__ void __launch_bounds__(kT) vd_short(const SeqArgs A) {
    __shared__ VdShortSmem stg;
    vd_short_block(A, blockIdx.x, stg);
}

__global__ void __launch_bounds__(kT) v_scores(const SeqArgs A, const int32_t *__restrict__ contig_ptr) {
    const int c = blockIdx.x * kT + threadIdx.x;
    if (c >= A.n_contigs) return;
    const int g1 = contig_ptr[c + 1];
    A.score[c] = g1 > contig_ptr[c] ? A.contigTmp[g1 - 1].x : 0.0;
}

// =====================================================================================
// F: sum-product.  T_t = M' D_t (first gene of a contig: 1 e_t^T, which forgets the past),
// D_t = diag(exp(s_t - max s_t)), M' = exp(trans - max trans).  alpha_t ~ (1,1) T_0..T_t,
// beta_t ~ B_t B_{t+1} ... 1 with B_t = T_{t+1} (last gene of a contig: 1 1^T).  All scale
// factors cancel in P_t(y) = alpha_t[y] beta_t[y] / (alpha_t . beta_t); exponents and emission
// maxima are carried along only for log Z.
// =====================================================================================
// exp(s - max s): one of the two is exp(0) = 1 exactly, the other one exp(-|s1 - s0|)
__device__ __forceinline__ double2 emit_norm(const SeqArgs &A, double2 s, double &m) {
    const double d = s.y - s.x;
    const double e = exp_neg(fabs(d), A.expc);
    m = d > 0.0 ? s.y : s.x;
    return d > 0.0 ? make_double2(e, 1.0) : make_double2(1.0, e);
}
__device__ __forceinline__ FE f_step(const SeqArgs &A, double2 s, bool first) {
    double m;
    const double2 e = emit_norm(A, s, m);
    return FE{(first ? 1.0 : A.m00) * e.x, (first ? 1.0 : A.m01) * e.y, (first ? 1.0 : A.m10) * e.x,
              (first ? 1.0 : A.m11) * e.y, 0.0, m, first ? 1.0 : 0.0};
}

__global__ void __launch_bounds__(kT) f_fold(const SeqArgs A) {
    __shared__ FE lds[kT / 64];
    __shared__ LaneStage stg;
    const LaneGenes L = load_lane(A, threadIdx.x, stg);
    FE P = FOp::identity();
#pragma unroll
    for (int k = 0; k < kGPL; ++k)
        if (k < L.cnt) P = FOp::combine(P, f_step(A, L.s[k], (L.first >> k) & 1u));
    FE total;
    const FE excl = block_scan_exclusive<FOp, false>(P, lds, &total);
    A.fLane[blockIdx.x * kT + threadIdx.x] = excl;
    if (threadIdx.x == 0) A.fBlock[blockIdx.x] = total;
}

// forward replay (alpha of every gene, cumulative log-mass at contig ends) + fold of the
// backward matrices B_t, scanned back to front
__global__ void __launch_bounds__(kT) f_replay(const SeqArgs A) {
    __shared__ FE lds[kT / 64];
    __shared__ FE xch[kT];
    __shared__ LaneStage stg;
    const int slot = threadIdx.x;
    const LaneGenes L = load_lane(A, slot, stg);
    const FE M = FOp::combine(lookback_prefix(A.fBlock, blockIdx.x), A.fLane[blockIdx.x * kT + slot]);
    // alpha entering the lane = a row of M (rows are identical once a contig has started), with the
    // exponent and emission-maximum sums accumulated since that contig's first gene
    double a0 = M.a00, a1 = M.a01, ex = M.ex, ms = M.ms;
    const double2 s_next = (L.cnt == kGPL && L.g0 + kGPL < A.n_genes) ? A.state[L.g0 + kGPL] : make_double2(0.0, 0.0);
    FE Bfold = FOp::identity();
    double2 al[kGPL];
#pragma unroll
    for (int k = 0; k < kGPL; ++k) {
        al[k] = make_double2(0.0, 0.0);
        if (k < L.cnt) {
            double m;
            const double2 e = emit_norm(A, L.s[k], m);
            double n0, n1;
            if ((L.first >> k) & 1u) {  // alpha_0 = exp(s_0): restart exactly
                n0 = n1 = 1.0;
                ex = 0.0;
                ms = 0.0;
            } else {
                n0 = fma(a1, A.m10, a0 * A.m00);
                n1 = fma(a1, A.m11, a0 * A.m01);
            }
            a0 = n0 * e.x;
            a1 = n1 * e.y;
            int ee;
            (void)frexp(fmax(a0, a1), &ee);
            a0 = ldexp(a0, -ee);
            a1 = ldexp(a1, -ee);
            ex += double(ee);
            ms += m;
            al[k] = make_double2(a0, a1);
            if ((L.last >> k) & 1u) {
                // log Z' of the contig (max-normalised emissions / transitions) and its emission maxima
                A.contigTmp[L.g0 + k] = make_double2(ex * 0.6931471805599453 + log(a0 + a1), ms);
            }
            // backward matrix of this gene
            const bool last = (L.last >> k) & 1u;
            const double2 sn = k + 1 < kGPL ? L.s[k + 1 < kGPL ? k + 1 : k] : s_next;
            FE B;
            if (last) {
                B = FE{1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 1.0};  // rs: everything after this gene only scales the product
            } else {
                B = f_step(A, sn, false);
            }
            Bfold = FOpB::combine(Bfold, B);  // B_{g0} B_{g0+1} ... in sequence order
        }
    }
    store_lane_rows(A.alpha, A.n_genes, slot, al, stg);
    xch[kT - 1 - slot] = Bfold;
    __syncthreads();
    const FE mine = xch[slot];
    FE total;
    const FE excl = block_scan_exclusive<FOpB, true>(mine, lds, &total);
    __syncthreads();
    xch[kT - 1 - slot] = excl;
    __syncthreads();
    A.fLaneSuf[blockIdx.x * kT + slot] = xch[slot];
    if (slot == 0) A.fBlockSuf[blockIdx.x] = total;
}

__global__ void __launch_bounds__(kT) f_marginals(const SeqArgs A) {
    __shared__ LaneStage stg;
    const int slot = threadIdx.x;
    const LaneGenes L = load_lane(A, slot, stg);
    double2 al[kGPL], out[kGPL];
    load_lane_rows(A.alpha, A.n_genes, slot, al, stg);
    // beta entering from the right of the lane: (suffix of the lanes to the right) 1
    const FE S = FOpB::combine(A.fLaneSuf[blockIdx.x * kT + slot], lookahead_suffix(A.fBlockSuf, blockIdx.x, gridDim.x));
    double b0 = S.a00 + S.a01, b1 = S.a10 + S.a11;
    const double2 s_next = (L.cnt == kGPL && L.g0 + kGPL < A.n_genes) ? A.state[L.g0 + kGPL] : make_double2(0.0, 0.0);
#pragma unroll
    for (int k = kGPL - 1; k >= 0; --k) {
        out[k] = make_double2(0.0, 0.0);
        if (k < L.cnt) {
            // beta_k = B_k beta_{k+1}
            if ((L.last >> k) & 1u) {
                b0 = b1 = 1.0;
            } else {
                double m;
                const double2 e = emit_norm(A, k + 1 < kGPL ? L.s[k + 1 < kGPL ? k + 1 : k] : s_next, m);
                const double c0 = e.x * b0, c1 = e.y * b1;
                b0 = fma(A.m01, c1, A.m00 * c0);
                b1 = fma(A.m11, c1, A.m10 * c0);
                int ee;
                (void)frexp(fmax(b0, b1), &ee);
                b0 = ldexp(b0, -ee);
                b1 = ldexp(b1, -ee);
            }
            const double x0 = al[k].x * b0, x1 = al[k].y * b1, z = x0 + x1;
            double r = __builtin_amdgcn_rcp(z);  // one reciprocal (+ Newton step) instead of two divisions
            r = fma(fma(-z, r, 1.0), r, r);
            out[k] = make_double2(x0 * r, x1 * r);
        }
    }
    store_lane_rows(reinterpret_cast<double2 *>(A.marg), A.n_genes, slot, out, stg);
}

// ---- short contigs: whole-contig marginals in ONE launch on 8-byte inputs --------------------------------
// Workgroups own whole contigs (`cblk`), so the forward products, alpha, the backward products and the marginals
// of a gene all stay in the registers of the lane that owns it: per gene 8 B are read (s[1] - s[0]: every
// marginal depends on the emissions only through that difference; + 8 B of emission maxima when log Z is
// wanted) and 16 B written.  The general path above reads 16-byte states three times and parks alpha in HBM.
__device__ __forceinline__ double2 emit_d(const SeqArgs &A, double d) {
    const double e = exp_neg(fabs(d), A.expc);
    return d > 0.0 ? make_double2(e, 1.0) : make_double2(1.0, e);
}
// one step of the forward / backward products from a gene's emission pair e = exp(s - max s)
__device__ __forceinline__ FE f_step_e(const SeqArgs &A, double2 e, double m, bool first) {
    return FE{(first ? 1.0 : A.m00) * e.x, (first ? 1.0 : A.m01) * e.y, (first ? 1.0 : A.m10) * e.x,
              (first ? 1.0 : A.m11) * e.y, 0.0, m, first ? 1.0 : 0.0};
}

// MODE 0 (f_short): workgroups own whole contigs, everything in one launch.
// MODE 1 / 2 (any contig length, the flat layout: lane l of workgroup b owns genes 2048 b + 8 l ..): the same design in two
// launches -- 1: only the workgroup's forward and backward products (`fBlock`, `fBlockSuf`); 2: the body of f_short, with
// the vector entering the workgroup from the left looked back over the forward products of the workgroups before it (up
// to the nearest one that holds a contig start) and the one entering from the right looked ahead over the backward
// products (lookback_prefix / lookahead_suffix).  Per gene 8 B (+ 8 B of maxima for log Z) are read twice and 16 B
// written; the previous general path read 16-byte states three times and parked alpha in HBM (204 us for the 5 M genes
// of BASELINE.json's configs[4]).  The flat elements carry the contig flags (FE) whether or not log Z is wanted: the
// look-back needs them.
template <bool WANT_Z, int MODE>
// (with log Z the replaying kernels hold exponents and maxima too: at 128 registers they spill, and three unspilled
// workgroups per CU beat four spilling ones)
__global__ void __launch_bounds__(kT, (WANT_Z && MODE != 1) ? 3 : 4) f_short(const SeqArgs A) {
    constexpr bool FLAT = MODE != 0;
    // with log Z: elements carry exponents, emission maxima and the contig-start flag; without: bare 2x2 products
    using E_t = typename std::conditional<WANT_Z || FLAT, FE, F4>::type;
    using OpF = typename std::conditional<WANT_Z || FLAT, FOp, F4Op>::type;
    using OpB = typename std::conditional<WANT_Z || FLAT, FOpB, F4Op>::type;
    constexpr bool ELEM_FE = WANT_Z || FLAT;
    __shared__ E_t lds[kT / 64];
    __shared__ struct {
        double2 st[kT * (kGPL + 1)];  // d (and maxima) in, marginals out
    } stg;
    // the exchange area of the mirrored scans lies OVER the stage (which is dead between the emissions read from it at the
    // start and the marginals written to it at the end): 37 KB per workgroup, FOUR of them per CU at <= 128 VGPRs (round 4;
    // 51 KB and three until then -- C5's 2 442 workgroups took four residency rounds of 768 instead of three of 1 024)
    E_t *const xch = reinterpret_cast<E_t *>(stg.st);
    static_assert(sizeof(E_t) * kT <= sizeof(stg.st), "exchange area over the stage");
    const int slot = threadIdx.x;
    const int g0 = FLAT ? int(blockIdx.x) * kBlockGenes : A.cblk[blockIdx.x];
    const int n = FLAT ? min(kBlockGenes, A.n_genes - g0) : A.cblk[blockIdx.x + 1] - g0;
    constexpr bool want_z = WANT_Z;
    if constexpr (MODE == 1) {
        // the flat layout's first launch sums the state scores of its 2 048 genes itself (attribute per lane, the weight
        // pairs parked over the stage, 2 048 at a time) and leaves s[1] - s[0] (and the maxima) for the second launch:
        // the separate state kernel, its 8 (16) B/gene round trip and a kernel boundary are gone
        static_assert(sizeof(stg.st) >= 2048 * sizeof(st_f64x2), "parking area");
        double s0[kGPL], s1[kGPL];
        block_state_sums<kGPL, 2048>(A.csr_gene_ptr, A.csr_attr_id, A.csr_wtab01, A.csr_n_attrs, g0, g0 + n,
                                     reinterpret_cast<st_f64x2 *>(stg.st), s0, s1);
        __syncthreads();  // (every lane has summed its runs: the stage takes the differences)
#pragma unroll
        for (int j = 0; j < kGPL; ++j) {
            const int idx = j * kT + slot;
            const bool ok = idx < n;
            const double d = s1[j] - s0[j], m = fmax(s0[j], s1[j]);
            if (ok) {
                const_cast<double *>(A.dstate)[g0 + idx] = d;
                if (want_z) const_cast<double *>(A.smax)[g0 + idx] = m;
            }
            stg.st[(idx / kGPL) * (kGPL + 1) + idx % kGPL] = make_double2(ok ? d : 0.0, (ok && want_z) ? m : 0.0);
        }
    } else {
#pragma unroll
        for (int j = 0; j < kGPL; ++j) {
            const int idx = j * kT + slot;
            const bool ok = idx < n;
            stg.st[(idx / kGPL) * (kGPL + 1) + idx % kGPL] =
                make_double2(ok ? A.dstate[g0 + idx] : 0.0, (ok && want_z) ? A.smax[g0 + idx] : 0.0);
        }
    }
    const int cnt = min(kGPL, n - slot * kGPL);
    // which of the lane's genes start / end a contig (host-packed; positions past the last gene: one-gene contigs)
    const uint32_t bits = FLAT ? A.flat_bits[blockIdx.x * kT + slot] : A.lane_bits[blockIdx.x * kT + slot];
    // flat layout: the gene behind the workgroup's last one (its emission enters the last lane's backward step)
    double d_next_block = 0.0;
    if (FLAT && g0 + kBlockGenes < A.n_genes) {
        if constexpr (MODE == 1) {  // (the next workgroup's first gene: not summed by anybody yet -- one lane's worth of attributes)
            const int g = g0 + kBlockGenes;
            double t0 = 0.0, t1 = 0.0;
            for (int q = A.csr_gene_ptr[g]; q < A.csr_gene_ptr[g + 1]; ++q) {
                const int a = A.csr_attr_id[q];
                const double2 w = unsigned(a) < unsigned(A.csr_n_attrs) ? A.csr_wtab01[a] : make_double2(0.0, 0.0);
                t0 += w.x;
                t1 += w.y;
            }
            d_next_block = t1 - t0;
        } else {
            d_next_block = A.dstate[g0 + kBlockGenes];
        }
    }
    __syncthreads();
    // the emission pair of every gene the lane touches (its own 8 and its right neighbour's first): ONE exp per gene,
    // used by the forward fold, the forward replay, the backward fold and the backward replay
    double2 E[kGPL + 1];
    double mx[kGPL];
#pragma unroll
    for (int k = 0; k < kGPL; ++k) {
        const double2 v = stg.st[slot * (kGPL + 1) + k];
        E[k] = emit_d(A, v.x);
        mx[k] = v.y;
    }
    E[kGPL] = emit_d(A, slot + 1 < kT ? stg.st[(slot + 1) * (kGPL + 1)].x : d_next_block);
    const uint32_t first = bits & 0xffu, last = bits >> 8;
    auto step = [&](double2 e, double m, bool fst) {
        if constexpr (ELEM_FE) {
            return f_step_e(A, e, m, fst);
        } else {
            return F4{(fst ? 1.0 : A.m00) * e.x, (fst ? 1.0 : A.m01) * e.y, (fst ? 1.0 : A.m10) * e.x, (fst ? 1.0 : A.m11) * e.y};
        }
    };
    auto bstep = [&](int k) {  // backward matrix of the lane's gene k
        const bool lst = (last >> k) & 1u;
        if constexpr (ELEM_FE) {
            return lst ? FE{1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 1.0} : f_step_e(A, E[k + 1], 0.0, false);
        } else {
            const F4 nx = step(E[k + 1], 0.0, false);
            return lst ? F4{1.0, 1.0, 1.0, 1.0} : nx;
        }
    };
    E_t P = OpF::identity();
#pragma unroll
    for (int k = 0; k < kGPL; ++k)
    {   // (positions past the last gene: one-gene contigs, d = 0.)  The eight steps a lane folds by itself are renormalised
        // once at the end when the transition weights allow it (A.raw_fold, set by the host: entries of eight
        // max-normalised factors then stay far inside the range), after every step otherwise
        P = OpF::combine_raw(P, step(E[k], mx[k], (first >> k) & 1u));
        if (!A.raw_fold) P = OpF::renorm(P);
    }
    P = OpF::renorm(P);
    E_t total;
    E_t M = block_scan_exclusive<OpF, false>(P, lds, &total);
    if constexpr (MODE == 1) {
        // products only: forward total, backward total (the lanes' backward folds, scanned back to front)
        E_t Bf = OpB::identity();
#pragma unroll
        for (int k = 0; k < kGPL; ++k) {
            Bf = OpB::combine_raw(Bf, bstep(k));
            if (!A.raw_fold) Bf = OpB::renorm(Bf);
        }
        Bf = OpB::renorm(Bf);
        __syncthreads();
        xch[kT - 1 - slot] = Bf;
        __syncthreads();
        const E_t mine1 = xch[slot];
        E_t btot;
        (void)block_scan_exclusive<OpB, true>(mine1, lds, &btot);
        if (slot == 0) {
            A.fBlock[blockIdx.x] = total;
            A.fBlockSuf[blockIdx.x] = btot;
        }
        return;
    }
    __shared__ E_t looked[2];  // MODE 2: products of the workgroups before / behind this one, up to the contig's start / end
    if constexpr (MODE == 2) {
        // ONE wave looks back over the forward products of the workgroups before this one, another ahead over the backward
        // products, at the same time; the results cross LDS.  (Every wave used to do both looks itself: a wave scan of
        // 56-byte elements is ~320 vector instructions, and the launch is issue-bound.)
        if (slot < 64) {
            const E_t r = lookback_prefix(A.fBlock, blockIdx.x);
            if (slot == 0) looked[0] = r;
        } else if (slot >= kT - 64) {
            const E_t r = lookahead_suffix(A.fBlockSuf, blockIdx.x, gridDim.x);
            if (slot == kT - 64) looked[1] = r;
        }
        __syncthreads();
        M = FOp::combine(looked[0], M);  // (looked[1] is read where it is needed: it would cost 14 registers until then)
    }
    // contig ends before this lane (workgroup-wide count): which contig a log Z belongs to
    uint32_t ends_before = 0;
    if (want_z && !FLAT) {
        __shared__ U2 ldsc[kT / 64];
        U2 ctot;
        ends_before = block_scan_exclusive<AddOp, false>(U2{uint32_t(__builtin_popcount(last & ((1u << (cnt > 0 ? cnt : 0)) - 1u))), 0u}, ldsc, &ctot).x;
    }
    // forward replay: alpha of every gene of the lane (registers), log Z at contig ends; backward matrices folded
    double a0 = M.a00, a1 = M.a01, ex = 0.0, ms = 0.0;
    if constexpr (ELEM_FE) {
        ex = M.ex;
        ms = M.ms;
    }
    E_t Bfold = OpB::identity();
    double2 al[kGPL];
#pragma unroll
    for (int k = 0; k < kGPL; ++k) {
        {
            double n0, n1;
            if ((first >> k) & 1u) {
                n0 = n1 = 1.0;
                ex = 0.0;
                ms = 0.0;
            } else {
                n0 = fma(a1, A.m10, a0 * A.m00);
                n1 = fma(a1, A.m11, a0 * A.m01);
            }
            a0 = n0 * E[k].x;
            a1 = n1 * E[k].y;
            int ee;
            (void)frexp(fmax(a0, a1), &ee);
            a0 = ldexp(a0, -ee);
            a1 = ldexp(a1, -ee);
            ex += double(ee);
            ms += mx[k];
            al[k] = make_double2(a0, a1);
            const bool lst = (last >> k) & 1u;
            if (lst && want_z && k < cnt) {
                if constexpr (FLAT) {
                    // (log Z', emission maxima) parked at the contig's last gene: f_lognorm finishes the sum per contig
                    A.contigTmp[g0 + slot * kGPL + k] = make_double2(ex * 0.6931471805599453 + log(a0 + a1), ms);
                } else {
                    // log Z = log Z' (max-normalised emissions and transitions) + the emission maxima + (n - 1) max(trans)
                    const int c = A.ne_contig[A.cblk_rank[blockIdx.x] + int(ends_before) + __builtin_popcount(last & ((1u << k) - 1u))];
                    const int len = A.contig_ptr[c + 1] - A.contig_ptr[c];
                    A.lognorm[c] = (ex * 0.6931471805599453 + log(a0 + a1)) + ms + double(len - 1) * A.mx;
                }
            }
            Bfold = OpB::combine_raw(Bfold, bstep(k));
            if (!A.raw_fold) Bfold = OpB::renorm(Bfold);
        }
    }
    Bfold = OpB::renorm(Bfold);
    __syncthreads();  // every lane has read its neighbour's d: the stage may be overwritten below
    xch[kT - 1 - slot] = Bfold;
    __syncthreads();
    const E_t mine = xch[slot];
    E_t btotal;
    const E_t bexcl = block_scan_exclusive<OpB, true>(mine, lds, &btotal);
    __syncthreads();
    xch[kT - 1 - slot] = bexcl;
    __syncthreads();
    E_t S = xch[slot];  // product of the backward matrices of the lanes to the right (up to the contig's end)
    __syncthreads();    // (the marginals go over the exchange area)
    if constexpr (MODE == 2) S = FOpB::combine(S, looked[1]);
    double b0 = S.a00 + S.a01, b1 = S.a10 + S.a11;
#pragma unroll
    for (int k = kGPL - 1; k >= 0; --k) {
        double2 o;
        {
            if ((last >> k) & 1u) {
                b0 = b1 = 1.0;
            } else {
                const double c0 = E[k + 1].x * b0, c1 = E[k + 1].y * b1;
                b0 = fma(A.m01, c1, A.m00 * c0);
                b1 = fma(A.m11, c1, A.m10 * c0);
                int ee;
                (void)frexp(fmax(b0, b1), &ee);
                b0 = ldexp(b0, -ee);
                b1 = ldexp(b1, -ee);
            }
            const double x0 = al[k].x * b0, x1 = al[k].y * b1, z = x0 + x1;
            double r = __builtin_amdgcn_rcp(z);
            r = fma(fma(-z, r, 1.0), r, r);
            o = make_double2(x0 * r, x1 * r);
        }
        stg.st[slot * (kGPL + 1) + k] = o;
    }
    __syncthreads();
    double2 *out = reinterpret_cast<double2 *>(A.marg);
#pragma unroll
    for (int j = 0; j < kGPL; ++j) {
        const int idx = j * kT + slot;
        if (idx < n) out[g0 + idx] = stg.st[(idx / kGPL) * (kGPL + 1) + idx % kGPL];
    }
}

// log Z of contig c = log Z' + its emission maxima + (n-1) max(trans)
__global__ void __launch_bounds__(kT) f_lognorm(const SeqArgs A, const int32_t *__restrict__ contig_ptr) {
    const int c = blockIdx.x * kT + threadIdx.x;
    if (c >= A.n_contigs) return;
    const int g0 = contig_ptr[c], g1 = contig_ptr[c + 1];
    if (g1 <= g0) {
        A.lognorm[c] = 0.0;
        return;
    }
    const double2 cur = A.contigTmp[g1 - 1];
    A.lognorm[c] = cur.x + cur.y + double(g1 - g0 - 1) * A.mx;
}

}  // namespace

// ---- launchers ---------------------------------------------------------------------------
static inline dim3 grid_for(int n, int per) { return dim3((n + per - 1) / per); }

hipError_t launch_seq_state(const int32_t *gene_ptr, const int32_t *attr_id, const double2 *wtab01, int n_attrs, int n_genes,
                            double2 *state, hipStream_t stream) {
    if (n_genes <= 0) return hipSuccess;
    hipLaunchKernelGGL(seq_state_blocks<2>, grid_for(n_genes, kStateGenes), dim3(kT), 0, stream, gene_ptr, attr_id, wtab01, n_attrs,
                       n_genes, (double *)nullptr, (double *)nullptr, state);
    return hipGetLastError();
}

hipError_t launch_seq_viterbi(const SeqArgs &a, const int32_t *d_contig_ptr, hipStream_t stream) {
    if (a.n_contigs <= 0) return hipSuccess;
    if (a.n_genes > 0) {
        const int nb = (a.n_genes + kBlockGenes - 1) / kBlockGenes;
        hipLaunchKernelGGL(v_fold, dim3(nb), dim3(kT), 0, stream, a);
        hipLaunchKernelGGL(v_replay, dim3(nb), dim3(kT), 0, stream, a);
        hipLaunchKernelGGL(v_labels, dim3(nb), dim3(kT), 0, stream, a);
    }
    if (a.score) hipLaunchKernelGGL(v_scores, grid_for(a.n_contigs, kT), dim3(kT), 0, stream, a, d_contig_ptr);
    return hipGetLastError();
}

hipError_t launch_seq_viterbi_delta(const SeqArgs &a, hipStream_t stream) {
    if (a.n_contigs <= 0 || a.n_genes <= 0) return hipSuccess;
    const int nb = (a.n_genes + kBlockGenes - 1) / kBlockGenes;
    if (a.short_contigs) {  // whole contigs per workgroup: one launch
        hipLaunchKernelGGL(vd_short, dim3(a.n_cblocks), dim3(kT), 0, stream, a);
        return hipGetLastError();
    }
    const bool exact = a.v_exact && a.csr_gene_ptr && a.fix_flag;
    SeqArgs b = a;
    if (!exact) b.fix_flag = nullptr;
    // (the bound, the candidate counter and the contig flags are zero here: zeroed with the workspace, and again by
    // vd_exact_fix at the end of every decode)
    hipLaunchKernelGGL(vd_fold, dim3(nb), dim3(kT), 0, stream, b);
    hipLaunchKernelGGL(vd_replay, dim3(nb), dim3(kT), 0, stream, b);
    if (exact)
        hipLaunchKernelGGL(v_labels_refine, dim3(nb + kRefineBlocks), dim3(kT), 0, stream, b, nb);
    else
        hipLaunchKernelGGL(v_labels, dim3(nb), dim3(kT), 0, stream, b);
    // contigs that vd_replay flagged (a decision inside the rounding margin): CRFsuite's own recursion
    if (exact) hipLaunchKernelGGL(vd_exact_fix, dim3(min(a.n_contigs, 1024)), dim3(kT), 0, stream, b);
    return hipGetLastError();
}

hipError_t launch_seq_state_delta(const int32_t *gene_ptr, const int32_t *attr_id, const double2 *wtab01, int n_attrs, int n_genes,
                                  double *dstate, hipStream_t stream) {
    if (n_genes <= 0) return hipSuccess;
    hipLaunchKernelGGL(seq_state_blocks<0>, grid_for(n_genes, kStateGenes), dim3(kT), 0, stream, gene_ptr, attr_id, wtab01, n_attrs,
                       n_genes, dstate, (double *)nullptr, (double2 *)nullptr);
    return hipGetLastError();
}

// short contigs: state differences (+ maxima when log Z is wanted) and the fused kernel; a.dstate / a.smax are
// workspace arrays filled here
hipError_t launch_seq_marginals_short(const SeqArgs &a, const int32_t *gene_ptr, const int32_t *attr_id, const double2 *wtab01,
                                      int n_attrs, const int32_t *d_contig_ptr, hipStream_t stream) {
    if (a.n_contigs <= 0) return hipSuccess;
    if (a.n_genes > 0) {
        if (!a.short_contigs) {
            // contigs of any length, flat layout: the first launch sums the state scores itself
        } else if (a.lognorm)
            hipLaunchKernelGGL(seq_state_blocks<1>, grid_for(a.n_genes, kStateGenes), dim3(kT), 0, stream, gene_ptr, attr_id, wtab01,
                               n_attrs, a.n_genes, const_cast<double *>(a.dstate), const_cast<double *>(a.smax), (double2 *)nullptr);
        else
            hipLaunchKernelGGL(seq_state_blocks<0>, grid_for(a.n_genes, kStateGenes), dim3(kT), 0, stream, gene_ptr, attr_id, wtab01,
                               n_attrs, a.n_genes, const_cast<double *>(a.dstate), (double *)nullptr, (double2 *)nullptr);
        if (a.short_contigs) {
            if (a.lognorm)
                hipLaunchKernelGGL((f_short<true, 0>), dim3(a.n_cblocks), dim3(kT), 0, stream, a);  // writes log Z itself
            else
                hipLaunchKernelGGL((f_short<false, 0>), dim3(a.n_cblocks), dim3(kT), 0, stream, a);
        } else {
            // contigs of any length, flat layout: state sums + the workgroups' products first, then the fused kernel looks them up
            const dim3 nb((a.n_genes + kBlockGenes - 1) / kBlockGenes);
            SeqArgs b = a;
            b.csr_gene_ptr = gene_ptr;
            b.csr_attr_id = attr_id;
            b.csr_wtab01 = wtab01;
            b.csr_n_attrs = n_attrs;
            if (a.lognorm) {
                hipLaunchKernelGGL((f_short<true, 1>), nb, dim3(kT), 0, stream, b);
                hipLaunchKernelGGL((f_short<true, 2>), nb, dim3(kT), 0, stream, b);
            } else {
                hipLaunchKernelGGL((f_short<false, 1>), nb, dim3(kT), 0, stream, b);
                hipLaunchKernelGGL((f_short<false, 2>), nb, dim3(kT), 0, stream, b);
            }
        }
    }
    if (!a.short_contigs && a.lognorm) hipLaunchKernelGGL(f_lognorm, grid_for(a.n_contigs, kT), dim3(kT), 0, stream, a, d_contig_ptr);
    return hipGetLastError();
}

hipError_t launch_seq_marginals(const SeqArgs &a, const int32_t *d_contig_ptr, hipStream_t stream) {
    if (a.n_contigs <= 0) return hipSuccess;
    if (a.n_genes > 0) {
        const int nb = (a.n_genes + kBlockGenes - 1) / kBlockGenes;
        hipLaunchKernelGGL(f_fold, dim3(nb), dim3(kT), 0, stream, a);
        hipLaunchKernelGGL(f_replay, dim3(nb), dim3(kT), 0, stream, a);
        hipLaunchKernelGGL(f_marginals, dim3(nb), dim3(kT), 0, stream, a);
    }
    if (a.lognorm) hipLaunchKernelGGL(f_lognorm, grid_for(a.n_contigs, kT), dim3(kT), 0, stream, a, d_contig_ptr);
    return hipGetLastError();
}

}  // namespace gecco
