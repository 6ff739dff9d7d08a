// Workgroup-level scan plumbing shared by the flat segmented scans (crf_sequence.hip: rows F / V;
// crf_segment.hip: row R): DPP moves of whole scan elements, a 64-lane inclusive scan without LDS
// traffic, and a workgroup-wide exclusive scan whose wave totals meet in LDS.  gfx950 only.
#pragma once
#include "crf_device.hpp"

namespace gecco {
namespace {

constexpr int kScanThreads = 256;  // lanes per workgroup of every flat-scan kernel

struct MapOp {  // maps {0,1}->{0,1} packed in 2 bits: bit x = image of x
    static __device__ __forceinline__ uint32_t identity() { return 2u; }
    // result(x) = a(b(x)): b is applied first.  In the backward label scans the element closer
    // to the END of the sequence acts first.
    static __device__ __forceinline__ uint32_t combine(uint32_t a, uint32_t b) {
        return ((a >> (b & 1u)) & 1u) | (((a >> ((b >> 1) & 1u)) & 1u) << 1);
    }
};


// ---------------------------------------------------------------- DPP plumbing
template <int CTRL, int RM>
__device__ __forceinline__ double dpp_f64(double old, double src) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, RM, 0xF, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, RM, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int RM>
__device__ __forceinline__ VE dpp_elem(const VE &old, const VE &s) {
    return VE{dpp_f64<CTRL, RM>(old.a00, s.a00), dpp_f64<CTRL, RM>(old.a01, s.a01), dpp_f64<CTRL, RM>(old.a10, s.a10),
              dpp_f64<CTRL, RM>(old.a11, s.a11), dpp_f64<CTRL, RM>(old.rs, s.rs)};
}
template <int CTRL, int RM>
__device__ __forceinline__ FE dpp_elem(const FE &old, const FE &s) {
    return FE{dpp_f64<CTRL, RM>(old.a00, s.a00), dpp_f64<CTRL, RM>(old.a01, s.a01), dpp_f64<CTRL, RM>(old.a10, s.a10),
              dpp_f64<CTRL, RM>(old.a11, s.a11), dpp_f64<CTRL, RM>(old.ex, s.ex),   dpp_f64<CTRL, RM>(old.ms, s.ms),
              dpp_f64<CTRL, RM>(old.rs, s.rs)};
}
template <int CTRL, int RM>
__device__ __forceinline__ CE dpp_elem(const CE &old, const CE &s) {
    return CE{dpp_f64<CTRL, RM>(old.a, s.a), dpp_f64<CTRL, RM>(old.L, s.L), dpp_f64<CTRL, RM>(old.H, s.H)};
}
template <int CTRL, int RM>
__device__ __forceinline__ SegE dpp_elem(const SegE &old, const SegE &s) {
    auto mv = [](uint32_t o, uint32_t v) { return uint32_t(__builtin_amdgcn_update_dpp(int(o), int(v), CTRL, RM, 0xF, false)); };
    return SegE{mv(old.map, s.map), mv(old.ng0, s.ng0), mv(old.ng1, s.ng1), mv(old.ann, s.ann)};
}
struct F4 {  // sum-product 2x2 scan element without bookkeeping (marginals only: every scale factor cancels)
    double a00, a01, a10, a11;
};
template <int CTRL, int RM>
__device__ __forceinline__ F4 dpp_elem(const F4 &old, const F4 &s) {
    return F4{dpp_f64<CTRL, RM>(old.a00, s.a00), dpp_f64<CTRL, RM>(old.a01, s.a01), dpp_f64<CTRL, RM>(old.a10, s.a10),
              dpp_f64<CTRL, RM>(old.a11, s.a11)};
}
struct U2 {  // pair of counters (plain sums)
    uint32_t x, y;
};
template <int CTRL, int RM>
__device__ __forceinline__ U2 dpp_elem(const U2 &old, const U2 &s) {
    auto mv = [](uint32_t o, uint32_t v) { return uint32_t(__builtin_amdgcn_update_dpp(int(o), int(v), CTRL, RM, 0xF, false)); };
    return U2{mv(old.x, s.x), mv(old.y, s.y)};
}
struct AddOp {
    static __device__ __forceinline__ U2 identity() { return U2{0u, 0u}; }
    static __device__ __forceinline__ U2 combine(const U2 &a, const U2 &b) { return U2{a.x + b.x, a.y + b.y}; }
};
template <int CTRL, int RM>
__device__ __forceinline__ uint32_t dpp_elem(const uint32_t &old, const uint32_t &s) {
    return uint32_t(__builtin_amdgcn_update_dpp(int(old), int(s), CTRL, RM, 0xF, false));
}

// REV = false: out[l] = e[0] (x) e[1] (x) ... (x) e[l]   (lane order = sequence order)
// REV = true : out[l] = e[l] (x) e[l-1] (x) ... (x) e[0]  (lanes hold the sequence back to front)
template <class Op, bool REV, class E>
__device__ __forceinline__ E comb(const E &earlier_lane, const E &later_lane) {
    return REV ? Op::combine(later_lane, earlier_lane) : Op::combine(earlier_lane, later_lane);
}
template <class Op, bool REV, class E>
__device__ __forceinline__ E wave_scan_inclusive(E v) {
    const E id = Op::identity();
    v = comb<Op, REV>(dpp_elem<0x111, 0xF>(id, v), v);  // row_shr:1
    v = comb<Op, REV>(dpp_elem<0x112, 0xF>(id, v), v);  // row_shr:2
    v = comb<Op, REV>(dpp_elem<0x114, 0xF>(id, v), v);  // row_shr:4
    v = comb<Op, REV>(dpp_elem<0x118, 0xF>(id, v), v);  // row_shr:8
    v = comb<Op, REV>(dpp_elem<0x142, 0xA>(id, v), v);  // row_bcast:15 -> rows 1,3
    v = comb<Op, REV>(dpp_elem<0x143, 0xC>(id, v), v);  // row_bcast:31 -> rows 2,3
    return v;
}
// The same scan without the identity: a lane that receives nothing in a step skips the step (an execution mask from one
// compare on its position) instead of combining with an identity that six v_mov_b32 have to lay down first, and the
// combine works in place (Op::combine_into(t, v): v = t (x) v, t is scratch) so that the masked step needs no copies
// either -- 14 instead of 19 vector instructions per step of a three-double element.  For kernels bound by vector issue.
template <int CTRL>
__device__ __forceinline__ double dpp_f64_z(double src) {  // every lane written; lanes without a source read zero
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(src), CTRL, 0xF, 0xF, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(src), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <class Op>
__device__ __forceinline__ CE wave_scan_inclusive_masked(CE v) {
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint32_t col = lane & 15u;
#define GECCO_MASKED_STEP(CTRL, COND)                                                  \
    {                                                                                  \
        CE t{dpp_f64_z<CTRL>(v.a), dpp_f64_z<CTRL>(v.L), dpp_f64_z<CTRL>(v.H)};         \
        if (COND) Op::combine_into(t, v);                                              \
    }
    GECCO_MASKED_STEP(0x111, col >= 1u)           // row_shr:1
    GECCO_MASKED_STEP(0x112, col >= 2u)           // row_shr:2
    GECCO_MASKED_STEP(0x114, col >= 4u)           // row_shr:4
    GECCO_MASKED_STEP(0x118, col >= 8u)           // row_shr:8
    GECCO_MASKED_STEP(0x142, (lane & 16u) != 0u)  // row_bcast:15, taken by rows 1 and 3
    GECCO_MASKED_STEP(0x143, lane >= 32u)         // row_bcast:31, taken by rows 2 and 3
#undef GECCO_MASKED_STEP
    return v;
}
template <class E>
__device__ __forceinline__ E wave_shift_up(const E &id, const E &v) {  // lane l <- lane l-1, lane 0 <- id
    return dpp_elem<0x138, 0xF>(id, v);                                 // wave_shr:1
}
// Workgroup-wide EXCLUSIVE scan of one element per lane (kT lanes); *total = product of all (total may be nullptr).
// LAST_USE: the caller does not touch lds_totals again before its next barrier (spares the closing barrier here).
template <class Op, bool REV, class E, int NTH = kScanThreads, bool MASKED = false, bool LAST_USE = false>
__device__ __forceinline__ E block_scan_exclusive(const E &mine, E *lds_totals /* NTH/64 */, E *total) {
    // (the wave index as a scalar: `w < wave` below is then a branch, not a select per component of E per wave)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    const E id = Op::identity();
    E incl;
    if constexpr (MASKED) {
        static_assert(!REV, "the masked scan runs in lane order");
        incl = wave_scan_inclusive_masked<Op>(mine);
    } else {
        incl = wave_scan_inclusive<Op, REV>(mine);
    }
    if (lane == 63) lds_totals[wave] = incl;
    E excl = wave_shift_up(id, incl);
    __syncthreads();
    E pre = id, all = id;
#pragma unroll
    for (int w = 0; w < NTH / 64; ++w) {
        const E t = lds_totals[w];
        if (w < wave) pre = comb<Op, REV>(pre, t);
        if (total) all = comb<Op, REV>(all, t);  // (a caller that has no use for the total passes nullptr)
    }
    if (!LAST_USE) __syncthreads();  // lds_totals may be reused by the caller
    if (total) *total = all;
    return comb<Op, REV>(pre, excl);
}

// Workgroup-wide exclusive scan from the BACK for 32-bit elements: lane l receives e[l+1] (x) e[l+2] (x) ... (x) e[NTH-1]
// with the element closest to the end acting first (Op::combine(a, b) applies b first); *total = e[0] (x) ... (x) e[NTH-1].
// Every wave mirrors its lanes through the LDS crossbar (ds_bpermute: no memory, no barrier), scans forward and mirrors
// back; the waves' totals meet in LDS in reverse order -- one barrier instead of the four of a mirrored exchange
// through LDS around a forward scan.  `lds_totals` must not be reused before the next barrier of the caller.
// `vote` (may be null): in / out, the workgroup-wide OR of the lanes' values rides on the scan's one barrier.
template <class Op, int NTH = kScanThreads>
__device__ __forceinline__ uint32_t block_scan_exclusive_back(uint32_t mine, uint32_t *lds_totals /* NTH/64 */, uint32_t *total,
                                                              int *vote = nullptr) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    const uint32_t id = Op::identity();
    const int mirror = (63 - lane) << 2;
    const uint32_t m = uint32_t(__builtin_amdgcn_ds_bpermute(mirror, int(mine)));
    const uint32_t incl = wave_scan_inclusive<Op, true>(m);
    if (lane == 63) lds_totals[wave] = incl;
    const uint32_t excl = wave_shift_up(id, incl);
    if (vote) *vote = __syncthreads_or(*vote);
    else __syncthreads();
    uint32_t pre = id, all = id;
#pragma unroll
    for (int w = NTH / 64 - 1; w >= 0; --w) {
        const uint32_t t = lds_totals[w];
        if (w > wave) pre = comb<Op, true>(pre, t);
        all = comb<Op, true>(all, t);
    }
    *total = all;
    return uint32_t(__builtin_amdgcn_ds_bpermute(mirror, int(comb<Op, true>(pre, excl))));
}

}  // namespace
}  // namespace gecco
