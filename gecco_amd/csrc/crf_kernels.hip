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
//     slots (:216-227).  A 256-lane workgroup owns TT tiles of 256-(W-1) output slots; every
//     lane is a window start in each of the TT DP phases, and the W-1 slots either side of the
//     workgroup's range are recomputed, so workgroups never exchange data.
//   * stage 1 (HBM -> LDS, once per workgroup): per slot the state scores s[y] = sum_a w[a][y]
//     ([EXT] crf1dt_state_score) through bounds-checked buffer descriptors (no branches, sums in
//     CSR order), reduced to e = exp(s - max(s)) and parked in LDS pre-multiplied with the
//     transition constant (16 B/slot).  A per-position scale factor cancels in every marginal, so
//     e (and transition ratios) replace CRFsuite's raw exps; this bounds the DP vectors and removes
//     CRFsuite's per-step 1/sum division.  Renormalisation by an exact power of two happens only
//     at the steps the host flags in `rescale_mask` (never for GECCO's model: see crf_plan.cpp).
//   * stage 2 (registers): lane s runs the W-step forward recursion for window s keeping the
//     label component of every alpha in VGPRs (fully unrolled), then the backward recursion; at
//     step k it holds alpha_k[label] and beta_k[label] of slot s+k.  Un-normalised vectors satisfy
//     alpha_k . beta_k = Z at every k, so P = alpha_k[label] beta_k[label] / Z with one reciprocal
//     per window, folded into the initial beta.
//   * stage 3 (DPP, no atomics): the maximum over windows is a diagonal reduction -- candidate k
//     of lane s belongs to slot s+k.  The running best is shifted one lane up per step with DPP
//     wave_shr:1 and meets the candidate in a v_max_f64.  Values leaving lane 63 are handed to the
//     next wave of the workgroup through an 8-B LDS slot per step.
//   No MFMA: L = 2 recurrences are 2x2 matrix-vector products on fp64 VALU.
#include "crf_device.hpp"
#include "crf_vd_short.hpp"

namespace gecco {
namespace {

// 16-byte LDS/global accesses must stay single b128 instructions: a struct double2 gets
// scalarised and re-paired by the compiler into bank-conflicting ds_read2_b64.
typedef double f64x2 __attribute__((ext_vector_type(2)));

// lane l receives lane l-1's value; lane 0 receives +0.0 (DPP wave_shr:1, bound_ctrl:1).
__device__ __forceinline__ double wave_shr1_zero(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// fmax() on a value that went through DPP bit moves first "canonicalises" it (v_max_f64 x, x, x: the
// compiler cannot see that it is not a signalling NaN), i.e. two VALU ops per step of the diagonal
// maximum instead of one.  The operands here are finite and non-negative.
__device__ __forceinline__ double max_nocanon(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Workgroup b runs on XCD b % 8 (observed placement; speed only).  Give each XCD one
// contiguous range of tiles so that the W-1 slot halo shared by neighbouring tiles is
// served by the same L2.  Bijective for any nwg.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (orig >> 3);
}

// The W = 20 kernel's outputs (probabilities, score differences: 16 B per gene, written once, read by a later launch)
// leave as write-through stores (agent-scope relaxed atomic store = `global_store_dwordx2 ... sc1`): the lines do not
// stay dirty in the XCD's L2, so the write-back at the end of the kernel -- which the next launch waits for -- has
// less to do (decode step 40.5 -> 39.7 us on C3; plain and `nt` stores: 40.5 / 40.1).  Pairing two genes per lane into one
// 16-byte `sc1` store (the neighbour's value by DPP) was measured too: 28.3 against 27.7 us per step -- the epilogue's
// extra VALU work costs more than the halved store count saves (profiles/r04_window_kernel_ab.txt).
__device__ __forceinline__ void store_wt(double *p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), static_cast<unsigned long long>(__double_as_longlong(v)),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A slot constant of the ratio form, read as ONE ds_read_b64: as plain loads the compiler pairs neighbouring reads into
// ds_read2_b64, which occupies the LDS array for 8 cycles per wave where two ds_read_b64 take 2 + 2 (MI355X_MICROARCH.md, LDS
// table; measured: SQ_LDS_IDX_ACTIVE -17 %, window kernel 25.05 -> 24.3 us, profiles/r04_diag_lds_ab.txt).  A relaxed
// workgroup-scope atomic load is never merged -- and never moved by the scheduler either: the DP requests its constants
// one step ahead by itself.
__device__ __forceinline__ double lds_read1(const double *p) {
    return __longlong_as_double(static_cast<long long>(
        __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)));
}

__device__ __forceinline__ void rescale_pair(double &u, double &v) {
    int ex;
    (void)frexp(fmax(u, v), &ex);
    u = ldexp(u, -ex);
    v = ldexp(v, -ex);
}


constexpr int kGatherUnroll = 8;  // attribute loads in flight per slot before the first use

// A workgroup of NT lanes owns TT consecutive "tiles": TT*(NT-(W-1)) output slots, for which it
// needs the slot constants of those slots plus W-1 on either side.  Stage 1 runs ONCE for all of
// them (every lane owns JMAX slots and has their dependent load chains in flight together: the
// chain of four memory round trips is what a workgroup spends most of its life on), then TT DP
// phases of NT window starts each.
template <int WMAX, int NT, int TT, bool EXACT>
struct WinSmem {
    static constexpr int JMAX = TT == 1 ? 2 : TT;             // slots a lane may own
    static constexpr int CAP = TT == 1 ? NT + WMAX - 1 : TT * NT;  // slot capacity of the workgroup
    // a contig occupies at least W slots (shorter ones are padded or not scored at all)
    static constexpr int CMAX = EXACT ? CAP / WMAX + 3 : CAP + 2;  // contigs a workgroup can overlap
    f64x2 ef[2 * CAP];              // first CAP entries, per slot: (e0, f = mu01*e1) with e = exp(s - max s), "other" first; or,
                                    // in the ratio form, CAP doubles r = f / e0 = mu01 exp(s[label] - s[other]) in its first quarter.
                                    // Before that, all 2 CAP entries park the weight pairs of the workgroup's attributes (stage 1).
    uint32_t ginfo[CAP];            // per slot: bit 31 = a window may start here; low bits = gene + 1 (0: none)
    f64x2 carry[NT / 64][WMAX];     // running best leaving lane 63 of each wave, per step
    int32_t cslot[CMAX];            // slot offsets of the contigs this workgroup overlaps (irregular tiles)
    int32_t cgene[CMAX];
    int32_t cn[CMAX];
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

// Row S: state scores of one gene, s[y] = sum over its attributes of w[a][y], added in CSR
// order ([EXT] crf1dt_state_score).  kGatherUnroll attribute ids are requested before the
// first weight gather and all gathers before the first add, so a gene costs two memory round
// trips instead of two per attribute; padding lanes add +0.0, which leaves the sum bit-exact.
__device__ __forceinline__ void state_scores_l2(const int32_t *__restrict__ attr_id,
                                                const double2 *__restrict__ wtab2, int n_attrs, int lo, int hi,
                                                double &s0, double &s1) {
    for (int base = lo; base < hi; base += kGatherUnroll) {
        int a[kGatherUnroll];
#pragma unroll
        for (int u = 0; u < kGatherUnroll; ++u) a[u] = base + u < hi ? attr_id[base + u] : -1;
        double2 w[kGatherUnroll];
#pragma unroll
        for (int u = 0; u < kGatherUnroll; ++u) w[u] = unsigned(a[u]) < unsigned(n_attrs) ? wtab2[a[u]] : make_double2(0.0, 0.0);
#pragma unroll
        for (int u = 0; u < kGatherUnroll; ++u) {
            s0 += w[u].x;
            s1 += w[u].y;
        }
    }
}

// Branch-free variant for the windowed kernel, on buffer descriptors: an out-of-range raw buffer
// load returns 0 instead of faulting, so (a) the kGatherUnroll attribute ids of a gene are loaded
// unconditionally (the ones past the gene's run belong to the next genes or lie past the end of
// the array and are simply not used), and (b) an unused slot gathers from offset 0xFFFFFFF0, which
// is past the weight table and yields the pair (+0.0, +0.0): the sum stays bit-exact and a gene
// costs 3 VALU ops per attribute slot (shift, compare, select) + 2 adds, no exec-mask branches.
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void state_scores_buf(__amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rw, uint32_t off,
                                                 uint32_t cnt, double &s0, double &s1) {
    for (uint32_t base = 0; base < cnt; base += kGatherUnroll) {  // one trip unless a gene has > 8 domains
        int a[kGatherUnroll];
#pragma unroll
        for (int u = 0; u < kGatherUnroll; ++u)
            a[u] = __builtin_amdgcn_raw_buffer_load_b32(ra, int((off + base + u) << 2), 0, 0);
        i32x4 w[kGatherUnroll];
#pragma unroll
        for (int u = 0; u < kGatherUnroll; ++u) {
            const uint32_t wo = base + u < cnt ? min(uint32_t(a[u]), 0x0FFFFFFFu) << 4 : 0xFFFFFFF0u;
            w[u] = __builtin_amdgcn_raw_buffer_load_b128(rw, int(wo), 0, 0);
        }
#pragma unroll
        for (int u = 0; u < kGatherUnroll; ++u) {
            s0 += __hiloint2double(w[u].y, w[u].x);
            s1 += __hiloint2double(w[u].w, w[u].z);
        }
    }
}

// Recurrences in the transformed basis (see crf_plan.cpp for the constants):
//   alpha~ = alpha * diag(1, kappa), beta~ = diag(1, 1/kappa) * beta, transitions divided by
//   m00 and conjugated so that their first column is (1, 1):  M~ = [[1, mu01], [1, mu11]].
//   forward : t = a0 + a1;  a0' = t * e0;  a1' = a0 * f + a1 * g        (f = mu01 e1, g = mu11 e1)
//   Only (e0, f) is kept per slot (16 B, one ds_read_b128): g = rho * f with rho = mu11/mu01, so
//     a1' = f * (a0 + rho a1)         and        u = f b1;  b0' = c + u;  b1' = c + rho u.
//   That is 4 + 4 VALU ops per step instead of 4 + 3, but a third less LDS traffic -- the DP is
//   LDS-read bound as much as VALU bound (tools/ubench/dp_variants.hip: 38.7 -> 33.2 us).
//   backward: c = e0 * b0;  b0' = c + f * b1;  b1' = c + g * b1
//   candidate for slot s+k: x = a1 * b1 (label), y = a0 * b0 (other); all scale factors cancel
//   in x / (x + y).
// One window tile (workgroup `tile` of the launch).
// LEAN: the tile shares its kernel with the Viterbi workgroups (crf_decode_pipelined), whose SGPR spills take one VGPR of the 64.
// SMALL: a batch of ONE regular tile whose CSR extent the host knows (the batch driver's direct path on a contig of a few dozen
// genes, BASELINE.json configs[0]): nothing is looked up in front of the first attribute load, and a phase without output
// slots is skipped.  A kernel of its own (crf_windowed_small_l2): the extra scalars would cost the tiles of the big launches
// a register spill, and occupancy means nothing to one workgroup.
template <int WMAX, bool EXACT, bool RESCALE, int NT, int TT, bool LEAN = false, bool SMALL = false>
__device__ __forceinline__ void windowed_tile(const WinArgs &P, WinSmem<WMAX, NT, TT, EXACT> &sm, const int tile) {
    using Smem = WinSmem<WMAX, NT, TT, EXACT>;
    constexpr int JMAX = Smem::JMAX;
    // the ratio form exists for fixed-length windows only (with a run-time W <= 32 the second copy of
    // the unrolled DP costs more registers than it saves instructions)
    constexpr bool RATIO = !RESCALE && EXACT;
    const int W = EXACT ? WMAX : P.W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int OUT = NT - (W - 1);             // output slots of one DP phase
    const int ns = TT * OUT + 2 * (W - 1);    // slots this workgroup needs constants for
    const int q0 = tile * (TT * OUT) - (W - 1);  // slot owned by lane 0

    // ---- slot -> gene.  A "regular" workgroup (no padded or skipped contig in reach: the normal
    // case) maps slots to genes by a constant shift and takes its window-start flags from a
    // host-built bit array, so the CSR loads can leave immediately; otherwise the contig
    // table of its reach goes through LDS and every lane searches it.
    // (gene - slot shift, first contig, last contig, flags).  A batch without any padded or skipped contig (the
    // normal case) has slot space = gene space everywhere: no descriptor to wait for, the first CSR load leaves at once
    const int4 td = P.all_regular ? make_int4(0, 0, 0, 1) : P.tile_desc[tile];
    int gene[JMAX];
    bool start[JMAX];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        gene[j] = -1;
        start[j] = false;
    }
    if (td.w & 1) {
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            const int sl = tid + j * NT, q = q0 + sl;
            if (sl < ns && q >= 0 && q < P.S) {
                gene[j] = q + td.x;
                start[j] = (reinterpret_cast<const uint32_t *>(P.start_bits)[q >> 5] >> (q & 31)) & 1u;  // (one v_bfe_u32)
            }
        }
    } else {
        const int cnt = td.z - td.y + 1;
        for (int j = tid; j <= cnt; j += NT) {
            sm.cslot[j] = P.c_slot[td.y + j];
            if (j < cnt) {
                sm.cgene[j] = P.c_gene[td.y + j];
                sm.cn[j] = P.c_n[td.y + j];
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            const int sl = tid + j * NT;
            if (sl < ns) {
                const SlotInfo si = slot_lookup(sm, cnt, q0 + sl, P.S, W, P.step);
                gene[j] = si.gene;
                start[j] = si.start_ok;
            }
        }
    }

    // buffer descriptors: attribute ids relative to the workgroup's first run (keeps byte offsets in
    // 32 bits for any batch), the weight-pair table whole.  All operands are wave-uniform (SGPRs).
    const uint32_t nnz = SMALL ? uint32_t(P.csr_end) : uint32_t(P.gene_ptr[P.n_genes]);
    const int g_first = (td.w & 1) ? (q0 > 0 ? q0 : 0) + td.x : P.c_gene[td.y];
    const uint32_t lo_tile = SMALL ? uint32_t(P.csr_begin) : uint32_t(P.gene_ptr[g_first]);
    const uint64_t abytes = uint64_t(nnz - lo_tile) << 2;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int32_t *>(P.attr_id + lo_tile), 0, abytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : uint32_t(abytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double2 *>(P.wtab2), 0, uint32_t(P.A) << 4, 0x00020000);

    // (no store in front of the wave-uniform loads above: it turns them into vector loads, and every buffer load then
    // grows a readfirstlane ("waterfall") loop for its descriptor)
    // ---- stage 1: state scores of the workgroup's slots -> slot constants in LDS.
    double sc0[JMAX], sc1[JMAX];  // s[other], s[label] of the lane's slots
#pragma unroll
    for (int j = 0; j < JMAX; ++j) sc0[j] = sc1[j] = 0.0;
    if (td.w & 1) {
        // Regular workgroup (the normal case): its slots are CONSECUTIVE GENES, so their attribute ids are one
        // contiguous stretch of the CSR.  The stretch is loaded attribute-per-lane -- consecutive lanes take
        // consecutive ids (coalesced), every id gathers its weight pair once -- parked in LDS (over the area the
        // slot constants will occupy afterwards), and each slot then adds up its own run from there, in CSR
        // order (bit-exact with sequential addition).  The first version of this stage gave every SLOT eight
        // speculative id loads and eight weight gathers whatever its number of domains (1.4 on average): 36
        // vector memory instructions per lane, of which the 16-byte gathers alone kept the texture path of a CU
        // busy for ~9 us per launch; the stage took 19.6 us on its own (DP alone: 21.2 us, both: 28.3 us).
        uint32_t lo[JMAX], hi[JMAX];
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            lo[j] = hi[j] = 0;
            if (gene[j] >= 0) {
                lo[j] = uint32_t(P.gene_ptr[gene[j]]);
                hi[j] = uint32_t(P.gene_ptr[gene[j] + 1]);
            }
        }
        const int q_end = min(q0 + ns, P.S);  // one past the workgroup's last slot
        const uint32_t hi_tile = SMALL ? uint32_t(P.csr_end) : uint32_t(P.gene_ptr[q_end + td.x]);  // (one tile: it ends where the batch does)
        const uint32_t n_attr = hi_tile - lo_tile;
        // parking area: the upper three quarters of `ef` -- the ratio form's slot constants (the lower quarter) never
        // touch it, so no barrier is needed between the sums and those writes; the max-normalised pairs (lower
        // half) are only built behind the workgroup-wide vote below
        constexpr int SBASE = (Smem::CAP + 1) / 2;
        constexpr int APL = (2 * Smem::CAP - SBASE) / NT;  // attributes per lane and round
        constexpr int SCAP = APL * NT;                     // weight pairs parked at a time
        static_assert(APL >= 1, "parking area smaller than one round of the workgroup");
        f64x2 *park = sm.ef + SBASE;
#pragma unroll 1
        for (uint32_t base = 0; base < n_attr; base += SCAP) {
            int id[APL];
#pragma unroll
            for (int a = 0; a < APL; ++a) id[a] = __builtin_amdgcn_raw_buffer_load_b32(ra, int((base + a * NT + tid) << 2), 0, 0);
            i32x4 w[APL];
#pragma unroll
            for (int a = 0; a < APL; ++a) {
                // ids past the stretch belong to later genes (or read 0 past the array): their pairs are parked
                // and never used; ids outside the dictionary land outside the table and read (+0.0, +0.0)
                const uint32_t wo = min(uint32_t(id[a]), 0x0FFFFFFFu) << 4;
                w[a] = __builtin_amdgcn_raw_buffer_load_b128(rw, int(wo), 0, 0);
            }
#pragma unroll
            for (int a = 0; a < APL; ++a)
                park[a * NT + tid] = f64x2{__hiloint2double(w[a].y, w[a].x), __hiloint2double(w[a].w, w[a].z)};
            __syncthreads();
            const uint32_t c0 = lo_tile + base, c1 = c0 + SCAP;
#pragma unroll
            for (int j = 0; j < JMAX; ++j) {
                // the run [lo, hi) cut to this round, as addresses in the parking area (one add and one compare per pair)
                const f64x2 *k = park + (max(lo[j], c0) - c0);
                const f64x2 *const e = park + (min(hi[j], c1) - c0);  // (k >= e when the run lies outside this round)
                for (; k < e && hi[j] > c0; ++k) {
                    const f64x2 v = *k;
                    sc0[j] += v.x;
                    sc1[j] += v.y;
                }
            }
            if (base + SCAP < n_attr) __syncthreads();  // the parking area is reused by the next round
        }
        // kernels without the ratio form write 16-byte slot constants over the whole lower half right away
        if (!RATIO && n_attr > 0) __syncthreads();
    } else {
        // Irregular workgroup (a padded or skipped contig in reach): slots are looked up one by one; every slot
        // requests its row bounds, then its first kGatherUnroll attribute ids, then their weight pairs, all of
        // them before anything is waited for.
        uint32_t off[JMAX], cnt[JMAX];
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            int lo = 0, hi = 0;
            if (gene[j] >= 0) {
                lo = P.gene_ptr[gene[j]];
                hi = P.gene_ptr[gene[j] + 1];
            }
            off[j] = uint32_t(lo) - lo_tile;
            cnt[j] = uint32_t(hi - lo);
        }
        int ids[JMAX][kGatherUnroll];
#pragma unroll
        for (int j = 0; j < JMAX; ++j)
            if (TT > 1 || j == 0 || wave == 0) {
#pragma unroll
                for (int u = 0; u < kGatherUnroll; ++u)
                    ids[j][u] = __builtin_amdgcn_raw_buffer_load_b32(ra, int((off[j] + u) << 2), 0, 0);
            }
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            if (TT > 1 || j == 0 || wave == 0) {
                double s0 = 0.0, s1 = 0.0;
                i32x4 w[kGatherUnroll];
#pragma unroll
                for (int u = 0; u < kGatherUnroll; ++u) {
                    const uint32_t wo = uint32_t(u) < cnt[j] ? min(uint32_t(ids[j][u]), 0x0FFFFFFFu) << 4 : 0xFFFFFFF0u;
                    w[u] = __builtin_amdgcn_raw_buffer_load_b128(rw, int(wo), 0, 0);
                }
#pragma unroll
                for (int u = 0; u < kGatherUnroll; ++u) {
                    s0 += __hiloint2double(w[u].y, w[u].x);
                    s1 += __hiloint2double(w[u].w, w[u].z);
                }
                if (cnt[j] > uint32_t(kGatherUnroll))  // rare: a gene with more than 8 domains
                    state_scores_buf(ra, rw, off[j] + kGatherUnroll, cnt[j] - kGatherUnroll, s0, s1);
                sc0[j] = s0;
                sc1[j] = s1;
            }
        }
    }
    double *rr = reinterpret_cast<double *>(sm.ef);  // ratio form: r per slot, in the first half of the (e0, f) array
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        if (TT > 1 || j == 0 || wave == 0) {
            const int sl = tid + j * NT;
            const double s0 = sc0[j], s1 = sc1[j];
            // decode = windowed marginals + Viterbi of the same batch: the raw scores of the genes this
            // workgroup owns are handed to the whole-contig kernels instead of being gathered again
            if (sl >= W - 1 && sl < TT * OUT + (W - 1) && gene[j] >= 0) {
                if (P.state_out) reinterpret_cast<f64x2 *>(P.state_out)[gene[j]] = P.label ? f64x2{s0, s1} : f64x2{s1, s0};
            }
            const double d = s1 - s0;
            if (sl < ns) {
                if (RATIO) {
                    // r = mu01 exp(d), with "a window may start here" in its sign bit (r > 0: the DP reads |r|).  A regular
                    // tile maps slots to genes by a constant shift, so only irregular ones park their genes.
                    const double r = mu_exp_tab(d, P.rtab, P.expc);
                    rr[sl] = __hiloint2double(__double2hiint(r) | (start[j] ? int(0x80000000u) : 0), __double2loint(r));
                    if (!(td.w & 1)) sm.ginfo[sl] = uint32_t(gene[j] + 1);
                } else {
                    const double e = exp_neg(fabs(d), P.expc);
                    const double e1 = d > 0.0 ? 1.0 : e;
                    sm.ef[sl] = f64x2{d > 0.0 ? e : 1.0, P.mu01 * e1};
                    sm.ginfo[sl] = (start[j] ? 0x80000000u : 0u) | uint32_t(gene[j] + 1);
                }
            }
        }
    }
    // The score differences of the genes this workgroup owns, for the Viterbi decoder of the same batch.  Stored BEHIND the
    // slot constants (loads and stores retire in order -- one vmcnt -- so nothing that is waited for follows them).
    if (P.dstate_out) {
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            if (TT > 1 || j == 0 || wave == 0) {
                const int sl = tid + j * NT;
                if (sl >= W - 1 && sl < TT * OUT + (W - 1) && gene[j] >= 0) {
                    const double d = sc1[j] - sc0[j];
                    // (s[1] - s[0] = d or -d: one XOR on the sign word)
                    const double dd = __hiloint2double(__double2hiint(d) ^ (P.label ? 0 : int(0x80000000u)), __double2loint(d));
                    store_wt(P.dstate_out + gene[j], dd);
                }
            }
        }
    }
    // Ratio form of the DP: dividing every position's emission pair by e0 turns the slot constant
    // into the single number r = f / e0 and removes one multiplication from either recursion
    //   forward : t = a0 + a1;  a1' = (a0 + rho a1) r;  a0' = t
    //   backward: u = r b1;  b1' = b0 + rho u;  b0' = b0 + u
    // (3 + 3 ops, 8 B of LDS per step).  The vectors then grow like exp(sum of the window's positive score
    // differences); a0 never decreases, so the window's Z bounds everything its forward pass has seen.  Every window is
    // therefore run in this form first, and a WAVE one of whose windows ends on Z >= 1e250 (a run of strongly
    // label-leaning genes: about one workgroup in a hundred holds one, under either weight law of the benchmark)
    // repeats its phase in the max-normalised form, whose pairs (e0, f) it derives from r on the fly.  No vote, no
    // second LDS layout: the candidates that cross lanes are probabilities whichever form produced them.
    // (Until round 3 a workgroup took the max-normalised form as soon as ONE of its slots had d > 600 / W: 3 % of the
    // workgroups under the benchmark's default weights, 96 % under SURVEY.md 8d's law.)
    __syncthreads();

    const uint32_t rmask = P.rescale_mask;
    const double rho = P.rho;
#pragma unroll 1
    for (int ph = 0; ph < TT; ++ph) {
        // (the last workgroup of a batch -- the only one of a 50-gene contig -- may have no output slot left for a later phase)
        if (SMALL && ph > 0 && q0 + (W - 1) + ph * OUT >= P.S) break;  // (no output slot left for this phase)
        const int sbase = ph * OUT + tid;  // slot of this lane's window start (and of its output)
        // the lane's output gene and whether a window may start at its slot: ratio-form kernels carry the start flag in
        // the sign bit of the slot constant and compute the gene of a regular tile from the slot; the others read both
        // from the slot table
        bool my_start = false;
        int my_gene;
        if (RATIO) {
            const int q = q0 + sbase;
            my_gene = (td.w & 1) ? ((q >= 0 && q < P.S) ? q + td.x : -1) : int(sm.ginfo[sbase]) - 1;
        } else {
            const uint32_t gi = sm.ginfo[sbase];
            my_start = gi >> 31;
            my_gene = int(gi & 0x7fffffffu) - 1;
        }
        const f64x2 *ef = sm.ef + sbase;
        if constexpr (RATIO) {
            const double *rrs = rr + sbase;
            double A1[WMAX];
            const double r0 = lds_read1(rrs);
            double a0 = 1.0, a1 = fabs(r0) * P.kappa_over_mu01;
            A1[0] = a1;
#pragma unroll
            for (int k = 1; k < WMAX; ++k) {
                if (EXACT || k < W) {
                    const double r = fabs(lds_read1(rrs + k));
                    const double t = a0 + a1;
                    a1 = fma(a1, rho, a0) * r;
                    a0 = t;
                    A1[k] = a1;
                }
            }
            asm volatile("" ::: "memory");
            double z = fma(a1, P.inv_kappa, a0);
            // (ratio_zmax = 1e250; GECCO_CRF_RATIO=0 sets it to -1: every window takes the max-normalised form -- A/B runs, tests)
            const bool renorm = __builtin_amdgcn_ballot_w64(!(z < P.ratio_zmax)) != 0;
            // max-normalised pair of a slot from its ratio constant: r > mu01 <=> d > 0, where (e0, f) = (exp(-d), mu01)
            // = (mu01 / r, mu01); else (1, mu01 exp(d)) = (1, r).
            // (The fallback is taken by the WAVE: a window that would have stayed in the ratio form gets the other form's
            // rounding when a neighbour in its wave needs it, so the last bits of such a window -- 1e-15 -- depend on how the
            // batch was tiled.  Making the choice per window was tried in round 5 -- the flag in a register, in LDS, in the
            // sign bit of A1[0] --: each cost the 64-register kernels a spill on the hot path.)
            auto pair_of = [&](double r, double &e0, double &f) {
                const bool pos = r > P.mu01;
                const double rc = fmin(r, 1.0e300);  // (r = inf for d > 709: e0 = 0 as exp(-d) would be)
                double q = __builtin_amdgcn_rcp(rc);
                q = fma(fma(-rc, q, 1.0), q, q);  // one Newton step: a division's worth of digits, a third of its registers
                e0 = pos ? P.mu01 * q : 1.0;
                f = pos ? P.mu01 : r;
            };
            if (renorm) {  // rare: the forward pass again, max-normalised
                double e0, f;
                pair_of(fabs(r0), e0, f);
                a0 = e0;
                a1 = f * P.kappa_over_mu01;
                A1[0] = a1;
#pragma unroll  // (a rolled loop would index A1 dynamically and send the whole array -- the hot path's too -- to scratch)
                for (int k = 1; k < WMAX; ++k) {
                    if (EXACT || k < W) {
                        pair_of(fabs(rrs[k]), e0, f);
                        const double t = a0 + a1;
                        a1 = fma(a1, rho, a0) * f;
                        a0 = t * e0;
                        A1[k] = a1;
                    }
                }
                z = fma(a1, P.inv_kappa, a0);
            }
            double b0, b1;
            {
                double r = __builtin_amdgcn_rcp(z);
                r = fma(fma(-z, r, 1.0), r, r);
                b0 = __double2hiint(r0) < 0 ? r : 0.0;  // (the sign bit itself: a window may start here; r may have underflowed to -0.0)
                b1 = b0 * P.inv_kappa;
            }
            double R = 0.0;
            double *carry = reinterpret_cast<double *>(&sm.carry[0][0]);
            if (!renorm) {
                double rcur = W > 1 ? lds_read1(rrs + (W - 1)) : 0.0;  // (the constant of a step is requested one step ahead)
#pragma unroll
                for (int k = WMAX - 1; k >= 0; --k) {
                    if (EXACT || k < W) {
                        double rnext = 0.0;
                        if (k > 1) rnext = lds_read1(rrs + (k - 1));
                        const double cand = A1[k] * b1;
                        if (k < W - 1) {
                            if (lane == 63 && wave < NT / 64 - 1) carry[wave * WMAX + k] = R;
                            R = wave_shr1_zero(R);
                        }
                        R = max_nocanon(R, cand);
                        if (k > 0) {
                            const double u = fabs(rcur) * b1;
                            b1 = fma(u, rho, b0);
                            b0 = b0 + u;
                        }
                        rcur = rnext;
                    }
                }
            } else {
#pragma unroll
                for (int k = WMAX - 1; k >= 0; --k) {
                    if (EXACT || k < W) {
                        const double cand = A1[k] * b1;
                        if (k < W - 1) {
                            if (lane == 63 && wave < NT / 64 - 1) carry[wave * WMAX + k] = R;
                            R = wave_shr1_zero(R);
                        }
                        R = max_nocanon(R, cand);
                        if (k > 0) {
                            double e0, f;
                            pair_of(fabs(rrs[k]), e0, f);
                            const double cc = e0 * b0, u = f * b1;
                            b0 = cc + u;
                            b1 = fma(u, rho, cc);
                        }
                    }
                }
            }
            __syncthreads();
            if (wave > 0 && lane < W - 1) {
                int t2 = tid;
                if (LEAN) asm volatile("" : "+v"(t2));  // (the index is recomputed per phase: one VGPR less across the DP)
                R = fmax(R, carry[((t2 >> 6) - 1) * WMAX + (t2 & 63)]);
            }
            R = fmin(R, 1.0);
            if (tid >= W - 1 && my_gene >= 0) store_wt(P.p_out + my_gene, R);
        } else if constexpr (!RESCALE) {
            // ---- stage 2a: forward recursion.  Without rescaling the un-normalised vectors satisfy
            //   alpha_k[0] beta_k[0] + alpha_k[1] beta_k[1] = Z   at EVERY position k of the window
            // (all per-position scale factors and the basis change cancel), so the marginal of the
            // queried label at position k is x_k / Z with x_k = alpha_k[label] beta_k[label] and one
            // reciprocal per window: only the label component of alpha has to be kept (W doubles).
            double A1[WMAX];
            double a0, a1;
            {
                const f64x2 c = ef[0];
                a0 = c.x;
                a1 = fabs(c.y) * P.kappa_over_mu01;  // kappa * e1
            }
            A1[0] = a1;
#pragma unroll
            for (int k = 1; k < WMAX; ++k) {
                if (EXACT || k < W) {
                    const f64x2 c = ef[k];
                    const double t = a0 + a1;
                    a1 = fma(a1, rho, a0) * fabs(c.y);
                    a0 = t * c.x;
                    A1[k] = a1;
                }
            }
            asm volatile("" ::: "memory");  // re-read the slot constants in the backward pass (VGPRs)

            // ---- stage 2b + 3: backward recursion; candidate k of lane s is P(slot s+k) in window s.
            // The maximum over windows is a diagonal reduction: the running best moves one lane up per
            // step (DPP wave_shr:1, zero fill = "no window yet" = numpy.zeros) and meets the candidate.
            // 1/Z goes into the initial beta (the backward recursion is linear), so a candidate is one
            // multiplication; lanes that may not start a window get beta = 0, i.e. candidates 0 = the
            // identity of the maximum.  Reciprocal: v_rcp_f64 + one Newton step (< 1 ulp from exact).
            double b0, b1;
            {
                const double z = fma(a1, P.inv_kappa, a0);
                double r = __builtin_amdgcn_rcp(z);
                r = fma(fma(-z, r, 1.0), r, r);
                b0 = my_start ? r : 0.0;
                b1 = b0 * P.inv_kappa;
            }
            double R = 0.0;
            double *carry = reinterpret_cast<double *>(&sm.carry[0][0]);
#pragma unroll
            for (int k = WMAX - 1; k >= 0; --k) {
                if (EXACT || k < W) {
                    const double cand = A1[k] * b1;
                    if (k < W - 1) {
                        if (lane == 63 && wave < NT / 64 - 1) carry[wave * WMAX + k] = R;
                        R = wave_shr1_zero(R);
                    }
                    R = max_nocanon(R, cand);
                    if (k > 0) {
                        const f64x2 c = ef[k];
                        const double cc = c.x * b0, u = fabs(c.y) * b1;
                        b0 = cc + u;
                        b1 = fma(u, rho, cc);
                    }
                }
            }
            __syncthreads();
            if (wave > 0 && lane < W - 1) R = fmax(R, carry[(wave - 1) * WMAX + lane]);
            R = fmin(R, 1.0);  // x_k and Z are rounded independently: x_k / Z may land one ulp above 1
            // genes no window covers (step > 1) keep 0.0 like numpy.zeros (crf/__init__.py:251)
            if (tid >= W - 1 && my_gene >= 0) P.p_out[my_gene] = R;
        } else {
            // ---- rescaling variant (transition weights far apart): power-of-two renormalisation at the
            // flagged steps changes Z from position to position, so candidates stay un-normalised pairs
            // (x, y) = (alpha[label] beta[label], alpha[other] beta[other]) compared by cross-multiplication
            // (x1*y2 > x2*y1); the only division is the final x/(x+y) per gene.
            double A0[WMAX], A1[WMAX];
            double a0, a1;
            {
                const f64x2 c = ef[0];
                a0 = c.x;
                a1 = c.y * P.kappa_over_mu01;  // kappa * e1
            }
            A0[0] = a0;
            A1[0] = a1;
#pragma unroll
            for (int k = 1; k < WMAX; ++k) {
                if (EXACT || k < W) {
                    const f64x2 c = ef[k];
                    const double t = a0 + a1;
                    const double n1 = fma(a1, rho, a0) * c.y;
                    a0 = t * c.x;
                    a1 = n1;
                    if ((rmask >> k) & 1u) rescale_pair(a0, a1);
                    A0[k] = a0;
                    A1[k] = a1;
                }
            }
            asm volatile("" ::: "memory");
            double b0 = 1.0, b1 = P.inv_kappa;
            // running best candidate; (0, 0) = "no window yet" so that DPP zero-fill is the identity
            double Rx = 0.0, Ry = 0.0;
#pragma unroll
            for (int k = WMAX - 1; k >= 0; --k) {
                if (EXACT || k < W) {
                    const double x = A1[k] * b1;
                    const double y = A0[k] * b0;
                    if (k < W - 1) {
                        if (lane == 63 && wave < NT / 64 - 1) sm.carry[wave][k] = f64x2{Rx, Ry};
                        Rx = wave_shr1_zero(Rx);
                        Ry = wave_shr1_zero(Ry);
                    }
                    // x/y >= Rx/Ry by cross-multiplication; always true against the (0,0) identity
                    const bool take = my_start && (x * Ry >= Rx * y);
                    Rx = take ? x : Rx;
                    Ry = take ? y : Ry;
                    if (k > 0) {
                        const f64x2 c = ef[k];
                        const double cc = c.x * b0, u = c.y * b1;
                        b0 = cc + u;
                        b1 = fma(u, rho, cc);
                        if ((rmask >> k) & 1u) rescale_pair(b0, b1);
                    }
                }
            }
            __syncthreads();
            if (wave > 0 && lane < W - 1) {
                const f64x2 c = sm.carry[wave - 1][lane];
                if (c.x * Ry > Rx * c.y || (Rx == 0.0 && Ry == 0.0)) {
                    Rx = c.x;
                    Ry = c.y;
                }
            }
            if (tid >= W - 1 && my_gene >= 0) P.p_out[my_gene] = (Rx + Ry > 0.0) ? Rx / (Rx + Ry) : 0.0;
        }
        if (TT > 1) __syncthreads();  // the carry slots are reused by the next phase
    }
}

template <int WMAX, bool EXACT, bool RESCALE, int NT, int TT>
// minimum waves per SIMD the register allocation must allow: the Z-constant DP keeps W doubles of
// alpha (8 waves at W = 20, 3-5 at W <= 32), the rescaling variant 2 W (5 / 3)
__global__ void __launch_bounds__(NT, (RESCALE ? (WMAX <= 20 ? 5 : 3) : (WMAX <= 20 ? (TT <= 2 ? 8 : 5) : 3))) crf_windowed_l2(const WinArgs P) {
    __shared__ WinSmem<WMAX, NT, TT, EXACT> sm;
    windowed_tile<WMAX, EXACT, RESCALE, NT, TT>(P, sm, xcd_remap(blockIdx.x, P.ntiles));
}

// one regular tile, CSR extent in the arguments (launch_windowed picks it when the plan carries the extent)
__global__ void __launch_bounds__(kWinThreads, 4) crf_windowed_small_l2(const WinArgs P) {
    __shared__ WinSmem<20, kWinThreads, 2, true> sm;
    windowed_tile<20, true, false, kWinThreads, 2, false, true>(P, sm, 0);
}

// ---- the decode step, software-pipelined over batches: ONE launch, NO hand-over inside it ---------------------------------
// Launch k carries the window tiles of batch k and the Viterbi workgroups of batch k - 1, whose score differences the
// tiles of launch k - 1 left in that plan's workspace (two buffers per plan, alternating, so a plan may follow itself).
// Nothing is exchanged inside the launch -- no flags, no acquire, no block roles: the first blocks are the Viterbi
// workgroups (rounded up to a multiple of 8, so the tiles keep their XCDs), the rest the tiles.  The Viterbi workgroups
// -- chains of dependent memory operations that leave the CUs idle when they run alone (vd_short: 10.5 us for 18 MB) --
// run under the VALU-bound tiles: 32 us for both against 24.3 + 10.5 us and a kernel boundary (C3: 40.7 -> 36.4 us per
// batch on the first measurement).  K batches take K + 1 launches (plan_run_decode_pipelined: the last call flushes).
// TILES: window tiles per workgroup, as in crf_windowed_l2 (1 for batches too small to fill the CUs' wave slots: a workgroup's
// tiles run one after the other, and with a wave or two per SIMD the chain of one tile is all the latency hiding there is).
// The launch's arguments, compact: the fields of WinArgs / SeqArgs this kernel reads and nothing else (376 bytes instead of the two
// full blocks' 936).  A step of a small batch is bound by what the HOST spends per launch (tools/host_issue_probe.py: 4.7 us per
// call at every batch size up to 0.2 M genes), and the runtime's argument copy grows with the block (tools/ubench/launch_floor:
// +0.8 us for 400 bytes).  The kernel rebuilds the two blocks in registers; fields it leaves zero fold away as constants.
struct PipeArgs {
    // window tiles (batch k)
    const int32_t *gene_ptr, *attr_id;
    const double2 *wtab2;
    const int32_t *c_slot, *c_gene, *c_n;
    const int4 *tile_desc;
    const uint64_t *start_bits;
    double *p_out, *dstate_out;
    const double *rtab;
    int32_t S, ntiles, step, label, n_genes, A, all_regular, nvd8;
    double mu01, rho, kappa_over_mu01, inv_kappa, ratio_zmax, expc5[5];
    // Viterbi workgroups (batch k - 1)
    const double *dstate;
    const uint8_t *flags;
    const uint16_t *lane_bits;
    const int32_t *cblk, *csr_gene_ptr, *csr_attr_id;
    const double2 *csr_wtab01;
    double2 *alpha;
    uint32_t *vd_stats;
    int8_t *y;
    int32_t s_n_genes, csr_n_attrs, v_exact, n_cblocks;
    double t00, t01, t10, t11, v_lo, v_hi, v_k, v_wmax2, v_tmax;
};

template <int TILES, int WGS>
__global__ void __launch_bounds__(kWinThreads, WGS) crf_decode_pipelined(const PipeArgs K) {
    using Smem = WinSmem<20, kWinThreads, TILES, true>;
    constexpr size_t kBytes = sizeof(Smem) > sizeof(VdShortSmem) ? sizeof(Smem) : sizeof(VdShortSmem);
    static_assert(kBytes <= 20480, "eight workgroups per CU");
    __shared__ __attribute__((aligned(16))) unsigned char raw[kBytes];
    const int b = blockIdx.x;
    if (b >= K.nvd8) {
        WinArgs P{};
        P.gene_ptr = K.gene_ptr;
        P.attr_id = K.attr_id;
        P.wtab2 = K.wtab2;
        P.c_slot = K.c_slot;
        P.c_gene = K.c_gene;
        P.c_n = K.c_n;
        P.tile_desc = K.tile_desc;
        P.start_bits = K.start_bits;
        P.p_out = K.p_out;
        P.dstate_out = K.dstate_out;
        P.rtab = K.rtab;
        P.S = K.S;
        P.ntiles = K.ntiles;
        P.W = 20;
        P.step = K.step;
        P.L = 2;
        P.label = K.label;
        P.n_genes = K.n_genes;
        P.A = K.A;
        P.tiles_per_wg = TILES;
        P.all_regular = K.all_regular;
        P.mu01 = K.mu01;
        P.rho = K.rho;
        P.kappa_over_mu01 = K.kappa_over_mu01;
        P.inv_kappa = K.inv_kappa;
        P.ratio_zmax = K.ratio_zmax;
#pragma unroll
        for (int i = 0; i < 5; ++i) P.expc[7 + i] = K.expc5[i];  // (mu_exp_tab reads 1/6! ... 1/2! only)
        P.csr_begin = P.csr_end = -1;
        windowed_tile<20, true, false, kWinThreads, TILES, true>(P, *reinterpret_cast<Smem *>(raw), xcd_remap(b - K.nvd8, K.ntiles));
        return;
    }
    if (b >= K.n_cblocks) return;
    SeqArgs A{};
    A.dstate = K.dstate;
    A.flags = K.flags;
    A.lane_bits = K.lane_bits;
    A.cblk = K.cblk;
    A.csr_gene_ptr = K.csr_gene_ptr;
    A.csr_attr_id = K.csr_attr_id;
    A.csr_wtab01 = K.csr_wtab01;
    A.csr_n_attrs = K.csr_n_attrs;
    A.alpha = K.alpha;
    A.vd_stats = K.vd_stats;
    A.y = K.y;
    A.n_genes = K.s_n_genes;
    A.n_cblocks = K.n_cblocks;
    A.short_contigs = 1;
    A.v_exact = K.v_exact;
    A.t00 = K.t00;
    A.t01 = K.t01;
    A.t10 = K.t10;
    A.t11 = K.t11;
    A.v_lo = K.v_lo;
    A.v_hi = K.v_hi;
    A.v_k = K.v_k;
    A.v_wmax2 = K.v_wmax2;
    A.v_tmax = K.v_tmax;
    vd_short_block(A, b, *reinterpret_cast<VdShortSmem *>(raw));
}

// ---- generic window kernel (2 labels, ANY window size, ANY transition spread) -------------
// Fallback for shapes the register-resident kernel does not take (W > 32, or transition
// weights so far apart that un-normalised vectors would need rescaling more than once per
// step).  One lane per window start, CRFsuite-style: both DP vectors are renormalised (by an
// exact power of two) after every step; alpha of every step is parked in a global scratch
// laid out [step][window] (coalesced); the per-gene maximum is an atomic max on the bit
// pattern of the (non-negative) probability.  Slow (a division and ~W*32 B of scratch traffic
// per window position) but shape-agnostic; also used by the tests as an on-device cross-check
// of the fast kernel.
__device__ __forceinline__ double2 slot_emission(const WinArgs &P, int gene) {
    double s0 = 0.0, s1 = 0.0;
    if (gene >= 0) state_scores_l2(P.attr_id, P.wtab2, P.A, P.gene_ptr[gene], P.gene_ptr[gene + 1], s0, s1);
    const double m = fmax(s0, s1);
    return make_double2(exp(s0 - m), exp(s1 - m));  // (other, label)
}

__global__ void __launch_bounds__(256) crf_windowed_generic_l2(const WinArgs P) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= P.S) return;
    if (!((P.start_bits[q >> 6] >> (q & 63)) & 1ull)) return;
    // contig of this window: largest k with c_slot[k] <= q
    int lo = 0, hi = P.K - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (P.c_slot[mid] <= q) lo = mid; else hi = mid - 1;
    }
    const int s0 = P.c_slot[lo], np = P.c_slot[lo + 1] - s0, n = P.c_n[lo], g0 = P.c_gene[lo];
    const int pos = q - s0, lpad = (np - n) >> 1;
    const int W = P.W;
    const size_t stride = size_t(P.S);
    double2 *scr = reinterpret_cast<double2 *>(P.scratch);
    auto gene_of = [&](int k) {
        const int gl = pos + k - lpad;
        return (gl >= 0 && gl < n) ? g0 + gl : -1;
    };
    // plain max-normalised recurrences: alpha' = (alpha M') o e, beta = M' (e o beta')
    const double m00 = P.g00, m01 = P.g01, m10 = P.g10, m11 = P.g11;
    double a0, a1;
    {
        const double2 e = slot_emission(P, gene_of(0));
        a0 = e.x;
        a1 = e.y;
    }
    scr[q] = make_double2(a0, a1);
    for (int k = 1; k < W; ++k) {
        const double2 e = slot_emission(P, gene_of(k));
        const double t0 = fma(a1, m10, a0 * m00), t1 = fma(a1, m11, a0 * m01);
        a0 = t0 * e.x;
        a1 = t1 * e.y;
        rescale_pair(a0, a1);
        scr[size_t(k) * stride + q] = make_double2(a0, a1);
    }
    double b0 = 1.0, b1 = 1.0;
    for (int k = W - 1; k >= 0; --k) {
        const double2 al = scr[size_t(k) * stride + q];
        const double x = al.y * b1, y = al.x * b0;
        const int gene = gene_of(k);
        if (gene >= 0) {
            const double pr = x / (x + y);
            atomicMax(reinterpret_cast<unsigned long long *>(P.p_out + gene), (unsigned long long)__double_as_longlong(pr));
        }
        if (k > 0) {
            const double2 e = slot_emission(P, gene);
            const double c0 = e.x * b0, c1 = e.y * b1;
            b0 = fma(m01, c1, m00 * c0);
            b1 = fma(m11, c1, m10 * c0);
            rescale_pair(b0, b1);
        }
    }
}

__global__ void fill_nan_kernel(double *p, const int2 *ranges, int n_ranges) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_ranges) return;
    const int2 rg = ranges[r];
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    for (int g = rg.x; g < rg.y; ++g) p[g] = nan;
}

}  // namespace

const char *windowed_kernel_name(int W, int L, bool fast) {
    if (L == 2 && fast && W == 20) return "crf_windowed_l2<20,exact>";
    if (L == 2 && fast && W <= kWinMaxW) return "crf_windowed_l2<32,dynamic>";
    if (L == 2) return "crf_windowed_generic_l2";
    return "unsupported";
}

int windowed_tile_out(int W, int L, int tt) {
    if (L == 2 && W <= kWinMaxW) return tt * (kWinThreads - (W - 1));
    return kWinThreads;
}

template <int TT>
static hipError_t launch_windowed_tt(const WinArgs &a, hipStream_t stream) {
    const dim3 grid(a.ntiles), block(kWinThreads);
    if (a.W == 20 && a.rescale_mask == 0) {
        hipLaunchKernelGGL((crf_windowed_l2<20, true, false, kWinThreads, TT>), grid, block, 0, stream, a);
    } else if (a.W == 20) {
        hipLaunchKernelGGL((crf_windowed_l2<20, true, true, kWinThreads, TT>), grid, block, 0, stream, a);
    } else if (a.rescale_mask == 0) {
        hipLaunchKernelGGL((crf_windowed_l2<kWinMaxW, false, false, kWinThreads, TT>), grid, block, 0, stream, a);
    } else {
        hipLaunchKernelGGL((crf_windowed_l2<kWinMaxW, false, true, kWinThreads, TT>), grid, block, 0, stream, a);
    }
    return hipGetLastError();
}

hipError_t launch_windowed(const WinArgs &a, hipStream_t stream) {
    if (a.ntiles <= 0) return hipSuccess;
    if (a.L == 2 && a.generic) {
        hipLaunchKernelGGL(crf_windowed_generic_l2, dim3((a.S + 255) / 256), dim3(256), 0, stream, a);
        return hipGetLastError();
    }
    if (a.L != 2 || a.W > kWinMaxW) return hipErrorNotSupported;
    if (a.ntiles == 1 && a.all_regular && a.csr_end >= 0 && a.csr_begin >= 0 && a.W == 20 && a.rescale_mask == 0 && a.tiles_per_wg == 2) {
        hipLaunchKernelGGL(crf_windowed_small_l2, dim3(1), dim3(kWinThreads), 0, stream, a);
        return hipGetLastError();
    }
    switch (a.tiles_per_wg) {
    case 1: return launch_windowed_tt<1>(a, stream);
    case 2: return launch_windowed_tt<2>(a, stream);
    case 3: return launch_windowed_tt<3>(a, stream);
    default: return hipErrorNotSupported;
    }
}

bool decode_pipelined_ok(const WinArgs &w, const SeqArgs &s) {
    return w.ntiles > 0 && w.W == 20 && w.rescale_mask == 0 && (w.tiles_per_wg == 1 || w.tiles_per_wg == 2) && !w.generic && w.L == 2 && !w.state_out &&
           s.short_contigs && s.n_cblocks > 0 && s.n_genes > 0;
}

hipError_t launch_decode_pipelined(const WinArgs &w, const SeqArgs &s, hipStream_t stream) {
    if (!decode_pipelined_ok(w, s)) return hipErrorNotSupported;
    PipeArgs k{};
    k.gene_ptr = w.gene_ptr;
    k.attr_id = w.attr_id;
    k.wtab2 = w.wtab2;
    k.c_slot = w.c_slot;
    k.c_gene = w.c_gene;
    k.c_n = w.c_n;
    k.tile_desc = w.tile_desc;
    k.start_bits = w.start_bits;
    k.p_out = w.p_out;
    k.dstate_out = w.dstate_out;
    k.rtab = w.rtab;
    k.S = w.S;
    k.ntiles = w.ntiles;
    k.step = w.step;
    k.label = w.label;
    k.n_genes = w.n_genes;
    k.A = w.A;
    k.all_regular = w.all_regular;
    k.nvd8 = (s.n_cblocks + 7) & ~7;
    k.mu01 = w.mu01;
    k.rho = w.rho;
    k.kappa_over_mu01 = w.kappa_over_mu01;
    k.inv_kappa = w.inv_kappa;
    k.ratio_zmax = w.ratio_zmax;
    for (int i = 0; i < 5; ++i) k.expc5[i] = w.expc[7 + i];
    k.dstate = s.dstate;
    k.flags = s.flags;
    k.lane_bits = s.lane_bits;
    k.cblk = s.cblk;
    k.csr_gene_ptr = s.csr_gene_ptr;
    k.csr_attr_id = s.csr_attr_id;
    k.csr_wtab01 = s.csr_wtab01;
    k.csr_n_attrs = s.csr_n_attrs;
    k.alpha = s.alpha;
    k.vd_stats = s.vd_stats;
    k.y = s.y;
    k.s_n_genes = s.n_genes;
    k.v_exact = s.v_exact;
    k.n_cblocks = s.n_cblocks;
    k.t00 = s.t00;
    k.t01 = s.t01;
    k.t10 = s.t10;
    k.t11 = s.t11;
    k.v_lo = s.v_lo;
    k.v_hi = s.v_hi;
    k.v_k = s.v_k;
    k.v_wmax2 = s.v_wmax2;
    k.v_tmax = s.v_tmax;
    if (w.tiles_per_wg == 1)
        hipLaunchKernelGGL((crf_decode_pipelined<1, 6>), dim3(k.nvd8 + w.ntiles), dim3(kWinThreads), 0, stream, k);
    else
        hipLaunchKernelGGL((crf_decode_pipelined<2, 8>), dim3(k.nvd8 + w.ntiles), dim3(kWinThreads), 0, stream, k);
    return hipGetLastError();
}

hipError_t launch_fill_nan(double *p, const int2 *ranges, int n_ranges, hipStream_t stream) {
    if (n_ranges <= 0) return hipSuccess;
    hipLaunchKernelGGL(fill_nan_kernel, dim3((n_ranges + 255) / 256), dim3(256), 0, stream, p, ranges, n_ranges);
    return hipGetLastError();
}

}  // namespace gecco
