// The one-launch Viterbi decoder for batches of short contigs (difference form on 8-byte inputs) as a device function,
// with CRFsuite's own recursion for the contigs that need it: shared by the kernel `vd_short` (crf_sequence.hip) and
// the pipelined decode kernel (crf_kernels.hip: crf_decode_pipelined), which runs the window tiles of a batch and the
// Viterbi workgroups of the batch before in ONE launch.
#pragma once

#include "crf_device.hpp"
#include "crf_scan.hpp"

namespace gecco {
namespace {

constexpr int kT = 256;                 // lanes per workgroup
constexpr int kGPL = kSeqGenesPerLane;  // genes folded by one lane
constexpr int kBlockGenes = kT * kGPL;
static_assert(kBlockGenes == kSeqBlockGenes, "host tables assume this block size");

// Difference form of the 2-label Viterbi recursion.  With Delta = delta[1] - delta[0] and d = s[1] - s[0],
//   Delta_t = clamp(Delta_{t-1}, lo, hi) + (t11 - t00) + d_t,   lo = t01 - t11,  hi = t00 - t10  (lo <= hi),
// the back-pointers of gene t are (Delta_{t-1} > hi, Delta_{t-1} > lo) and the end label is Delta_T > 0
// (strict, = CRFsuite's first arg max).  x -> min(max(x + a, L), H) is closed under composition, a
// contig's first gene is the constant map L = H = d_0, so the whole batch is again ONE scan -- with
// 24-byte elements, a 7-op combine, no reset flag, and 8 instead of 16 bytes of input per gene.
// Differences of accumulated scores stay O(1), so this form is if anything closer to exact
// arithmetic than the delta recursion itself; with integer-valued weights both are exact.
// fmin / fmax on a value the compiler cannot prove canonical (it came through LDS, DPP moves or a select) cost a
// canonicalising v_max_f64 x, x, x each on top; nothing here is ever a NaN (score differences, +-inf of the identity map,
// 1e30 of the padding positions), so the bare instructions are used.
__device__ __forceinline__ double vd_max(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double vd_min(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// ... against a wave-uniform bound: the second operand straight from its SGPR pair (a "v" operand would cost a v_mov_b64 per use)
__device__ __forceinline__ double vd_max_s(double a, double bound) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(bound));
    return r;
}
__device__ __forceinline__ double vd_min_s(double a, double bound) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(bound));
    return r;
}
// acc = 2 acc + (x > bound): the comparison's result is the carry of an add-with-carry -- two instructions per decision bit
// instead of compare, select, shift-or
__device__ __forceinline__ void vd_shift_in_gt(uint32_t &acc, double x, double bound) {
    asm("v_cmp_lt_f64 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(x), "s"(bound) : "vcc");
}
struct COp {
    static __device__ __forceinline__ CE identity() { return CE{0.0, -__builtin_huge_val(), __builtin_huge_val()}; }
    static __device__ __forceinline__ CE combine(const CE &a, const CE &b) {  // a applied first
        return CE{a.a + b.a, vd_min(vd_max(a.L + b.a, b.L), b.H), vd_min(vd_max(a.H + b.a, b.L), b.H)};
    }
    // v = combine(t, v) in place (t is used up): the same seven instructions in an order that needs no copy, for the
    // masked wave scan (crf_scan.hpp) whose lanes skip a step under an execution mask
    static __device__ __forceinline__ void combine_into(CE &t, CE &v) {
        asm volatile(
            "v_add_f64 %[tl], %[tl], %[a]\n\t"
            "v_add_f64 %[th], %[th], %[a]\n\t"
            "v_add_f64 %[a], %[ta], %[a]\n\t"
            "v_max_f64 %[tl], %[tl], %[L]\n\t"
            "v_max_f64 %[th], %[th], %[L]\n\t"
            "v_min_f64 %[L], %[tl], %[H]\n\t"
            "v_min_f64 %[H], %[th], %[H]"
            : [a] "+v"(v.a), [L] "+v"(v.L), [H] "+v"(v.H), [tl] "+v"(t.L), [th] "+v"(t.H)
            : [ta] "v"(t.a));
    }
};

// ---- when is a decision of the difference form PROVABLY crf1dc_viterbi's? -----------------------------------------
// CRFsuite compares fl(delta_t[0] + T[0][j]) with fl(delta_t[1] + T[1][j]) on accumulated scores of magnitude up to
//   M = nnz max|w| + n max|trans|        (a contig of n genes and nnz attribute entries; vd_bound),
// every step of which rounds twice (<= ulp(M) per step).  But the two paths it compares share their history up to the last
// gene u at which BOTH labels took the same predecessor -- a decision beyond [lo, hi], where the clamp of the difference
// form saturates -- or up to the contig's first gene, where delta = state exactly: from there on both computed scores are
// x + (exact terms) + (at most r ulp(M) of rounding each), r = genes since u, with the SAME x.  So its decision at gene t
// equals the real-arithmetic one -- which the difference form evaluates with an error far below ulp(M) per step -- whenever
//   |Delta_t - threshold| > (4 r_t + 4) ulp(M)                                                      (vd_margin)
// (2 r + 1 for CRFsuite's two scores and its two additions, as much again for the difference form's own roundings and
// the rounded thresholds).  r_t <= n, so a first test against the coarse margin (4 n + 4) ulp(M) -- which also defines
// "saturates for certain" -- finds the candidates (about one gene in 10^8 on metagenome-sized contigs); a candidate's
// lane then finds its r_t by walking back to the last certain saturation and tests again.  Only a decision inside THAT
// margin -- exact ties of integer-valued models; otherwise practically never -- sends its contig to CRFsuite's own
// recursion below.  (Until round 4 the margin was a flat 1e-6: 10^5 times too wide for 200-gene contigs, and on C5 it
// sent two 50 000-gene contigs per launch through the sequential walk: 1.27 ms per step instead of 0.09.)
__device__ __forceinline__ double vd_bound(const SeqArgs &A, double nnz, double n) { return nnz * A.v_wmax2 + (n + 2.0) * A.v_tmax; }
constexpr double kVdEps = 2.220446049250313e-16;  // 2^-52: ulp(M) <= M * 2^-52
__device__ __forceinline__ double vd_margin(double r, double ulpM) { return (4.0 * r + 4.0) * ulpM; }

// ---- CRFsuite's own recursion for the contigs that need it ------------------------------------------------
// A contig with a decision INSIDE the margin is decoded again here, the way CRFsuite does it: state scores summed attribute by
// attribute, delta_t[j] = max_i(delta_{t-1}[i] + trans[i][j]) + state_t[j] with the strict-< first-arg-max update, one
// gene after the other from the contig's first (oracle_viterbi_seq is this recursion).  The workgroup computes the state
// scores of 1024 genes at a time in parallel (LDS), one lane walks them; back-pointers go to a byte per gene in global
// scratch.  Rare (real-valued weights: about one gene in a million lies inside the margin), so nothing here is tuned.
constexpr int kFixChunk = 1024;
struct FixStage {
    double2 st[kFixChunk];            // state scores of one chunk
    uint32_t bpw[kFixChunk / 16];      // its back-pointers, 2 bits per gene (bit y: predecessor of label y)
};
// state scores of genes [g0, g0 + m) -> fx.st, by the lanes `first_lane` .. kT-1 of the workgroup; a lane takes four
// genes at a time so that their (dependent) row-pointer, id and weight loads are in flight together.  Sums in CSR order.
__device__ __forceinline__ void exact_states(const SeqArgs &A, int g0, int m, FixStage &fx, int first_lane) {
    const int nl = kT - first_lane, me = int(threadIdx.x) - first_lane;
    if (me < 0) return;
    for (int base = me; base < m; base += 4 * nl) {
        int lo[4], hi[4];
        double s0[4], s1[4];
        int longest = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + u * nl;
            lo[u] = hi[u] = 0;
            if (i < m) {
                lo[u] = A.csr_gene_ptr[g0 + i];
                hi[u] = A.csr_gene_ptr[g0 + i + 1];
            }
            s0[u] = s1[u] = 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) longest = max(longest, hi[u] - lo[u]);
        for (int k = 0; k < longest; ++k) {
            int a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = lo[u] + k < hi[u] ? A.csr_attr_id[lo[u] + k] : -1;
            double2 w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = unsigned(a[u]) < unsigned(A.csr_n_attrs) ? A.csr_wtab01[a[u]] : make_double2(0.0, 0.0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // (+0.0 for the genes that have run out: their sums do not change)
                s0[u] += w[u].x;
                s1[u] += w[u].y;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + u * nl;
            if (i < m) fx.st[i] = make_double2(s0[u], s1[u]);
        }
    }
}
// ONE lane: the recursion over the chunk's m genes (the contig's first gene, if `first`, only initialises).  The lane
// leaves delta of every gene where the gene's state scores were; the back-pointers -- which of the two candidates won,
// the same comparisons on the same numbers -- are recomputed from them by all lanes afterwards (exact_backpointers), so
// the sequential chain is 8 instructions per gene (4 adds, 2 max, 2 adds), 3 of them in a dependent row.
__device__ __forceinline__ void exact_walk(const SeqArgs &A, FixStage &fx, int m, bool first, double &d0, double &d1) {
    int i = 0;
    if (first) {
        d0 = fx.st[0].x;
        d1 = fx.st[0].y;
        i = 1;
    }
    auto step = [&](const double2 s) {
        const double a0 = d0 + A.t00, b0 = d1 + A.t10;
        const double a1 = d0 + A.t01, b1 = d1 + A.t11;
        d0 = fmax(a0, b0) + s.x;  // (on a tie both candidates are the same number)
        d1 = fmax(a1, b1) + s.y;
    };
    for (; i < m && (i & 7); ++i) {
        step(fx.st[i]);
        fx.st[i] = make_double2(d0, d1);
    }
    for (; i + 8 <= m; i += 8) {  // the eight state pairs are requested before the first step
        double2 sv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) sv[u] = fx.st[i + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            step(sv[u]);
            fx.st[i + u] = make_double2(d0, d1);
        }
    }
    for (; i < m; ++i) {
        step(fx.st[i]);
        fx.st[i] = make_double2(d0, d1);
    }
}
// all lanes: back-pointer words of the chunk from the delta values exact_walk left in fx.st; (e0, e1) = delta of the gene
// before the chunk.  [EXT] crf1dc_viterbi: candidates from label 0, then 1; `max_score < s` keeps the first on ties.
__device__ __forceinline__ void exact_backpointers(const SeqArgs &A, FixStage &fx, int m, bool first, double e0, double e1) {
    for (int w = threadIdx.x; 16 * w < m; w += kT) {
        uint32_t word = 0;
        for (int i = max(16 * w, first ? 1 : 0); i < min(16 * w + 16, m); ++i) {
            const double p0 = i ? fx.st[i - 1].x : e0, p1 = i ? fx.st[i - 1].y : e1;
            const uint32_t arg0 = (p0 + A.t00) < (p1 + A.t10) ? 1u : 0u, arg1 = (p0 + A.t01) < (p1 + A.t11) ? 1u : 0u;
            word |= (arg0 | (arg1 << 1)) << (2 * (i & 15));
        }
        fx.bpw[w] = word;
    }
}
// Back-pointer words in global scratch (alpha: marginals only), 16 genes of the CONTIG per word, the contig's words
// starting at word `gs`: a contig of n genes takes ceil(n / 16) <= n words, so contigs that different workgroups
// decode at the same time never share one.  A second array of the same shape behind it (n_genes words further) takes
// the label at the end of every word for the parallel backtrack.
__device__ __forceinline__ uint32_t *exact_bp_words(const SeqArgs &A, int gs) { return reinterpret_cast<uint32_t *>(A.alpha) + size_t(gs); }
__device__ __forceinline__ uint32_t *exact_end_labels(const SeqArgs &A, int gs) {
    return reinterpret_cast<uint32_t *>(A.alpha) + size_t(A.n_genes) + size_t(gs);
}
// Labels from the back-pointers of contig [gs, ge) and the scores of its last gene.  Every lane composes the sixteen
// steps of a word into one map (label behind the word -> label in front of it), one lane walks the words, every lane
// expands its words.
__device__ __forceinline__ void exact_backtrack(const SeqArgs &A, int gs, int ge, double d0, double d1) {
    const int n = ge - gs, nw = (n + 15) / 16;
    uint32_t *bpc = exact_bp_words(A, gs), *endl = exact_end_labels(A, gs);
    // position t of the contig (t >= 1) holds the predecessors of gene t; word w covers t in [16 w, 16 w + 15]
    auto word_map = [&](uint32_t bits, int w, uint32_t y) {  // label at the word's last position -> label at 16 w - 1
        const int hi = min(16 * w + 15, n - 1), lo = max(16 * w, 1);
        for (int t = hi; t >= lo; --t) y = (bits >> (2 * (t & 15) + y)) & 1u;
        return y;
    };
    for (int w = threadIdx.x; w < nw; w += kT) {
        const uint32_t bits = bpc[w];
        endl[w] = word_map(bits, w, 0) | (word_map(bits, w, 1) << 1);  // (bit y: where label y leads)
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t y = d0 < d1 ? 1u : 0u;  // first arg max: the label of the last gene
        for (int w = nw - 1; w >= 0; --w) {
            const uint32_t mp = endl[w];
            endl[w] = y;  // the label at the word's last position
            y = (mp >> y) & 1u;
        }
    }
    __syncthreads();
    for (int w = threadIdx.x; w < nw; w += kT) {
        const uint32_t bits = bpc[w];
        uint32_t y = endl[w];
        const int hi = min(16 * w + 15, n - 1), lo = 16 * w;
        for (int t = hi; t >= lo; --t) {
            A.y[gs + t] = int8_t(y);
            if (t >= 1) y = (bits >> (2 * (t & 15) + y)) & 1u;
        }
    }
    __syncthreads();
}

// a contig of a few chunks, inside vd_short (one LDS buffer: the state scores of a chunk, then the walk)
__device__ __forceinline__ void exact_delta_contig(const SeqArgs &A, int gs, int ge, FixStage &fx) {
    uint32_t *bpc = exact_bp_words(A, gs);
    double d0 = 0.0, d1 = 0.0;  // delta of the last gene walked so far (every lane holds a copy)
    for (int cb = gs; cb < ge; cb += kFixChunk) {
        const int m = min(kFixChunk, ge - cb);
        const double e0 = d0, e1 = d1;
        exact_states(A, cb, m, fx, 0);
        __syncthreads();
        if (threadIdx.x == 0) exact_walk(A, fx, m, cb == gs, d0, d1);
        __syncthreads();
        d0 = fx.st[m - 1].x;  // (lane 0's result, for everybody)
        d1 = fx.st[m - 1].y;
        exact_backpointers(A, fx, m, cb == gs, e0, e1);
        __syncthreads();
        for (int i = threadIdx.x; i < (m + 15) / 16; i += kT) bpc[(cb - gs) / 16 + i] = fx.bpw[i];
        __syncthreads();
    }
    exact_backtrack(A, gs, ge, d0, d1);
}

// Positions of a workgroup past its last gene behave as one-gene contigs with a score difference far from every
// threshold (the host sets their start / end bits in `lane_bits`): they decide nothing, perturb nothing that comes before
// them, and the hot loops need no "is this gene there" test.
constexpr double kVdPad = 1e30;

// ---- short contigs: ONE launch per decoder --------------------------------------------------------------
// When no contig is longer than one scan block (metagenome assemblies: the headline workload) the host packs
// WHOLE contigs into workgroups of at most 2048 genes (`cblk`, plan_ensure_seq): a workgroup then owns
// complete contigs, so nothing enters from the left, nothing from the right, and fold / scan / replay /
// back-to-front label pass all happen in this one kernel with the genes read once (8 B + 1 B per gene) and the
// labels written once (1 B per gene).  (The previous arrangement -- fixed 2048-gene spans, look-back by
// recomputation, labels in a second launch over three intermediate arrays -- took 12.5 + 5.0 us on C3.)
// LDS of one workgroup: 18.4 KB, so that eight of them fit a CU next to nothing else (and the pipelined decode kernel of
// crf_kernels.hip, whose window tiles take 19.6 KB, keeps its eight workgroups per CU).  The 9th double of every
// lane's row is padding against bank conflicts; the approximate pass keeps its marks there (one byte per gene).  The
// label bytes are staged over the first 2 KB of `st` once nothing reads the values any more.
struct VdShortSmem {
    double st[kT * (kGPL + 1)];
    uint32_t mark[kBlockGenes / 32];  // exact pass: first genes of the contigs to decode again
    int c_end;
    CE tot[kT / 64];
    uint32_t mtot[kT / 64];
};
static_assert(sizeof(VdShortSmem) <= 20480, "eight workgroups per CU");
// genes [g0, g0 + n) of the workgroup -> the lanes that own kGPL consecutive ones.  A lane loads ITS OWN genes (64
// consecutive bytes; a wave's loads cover 4 KB between them, every line used whole) and parks them in its padded LDS row
// for the later passes: no exchange between lanes, so no barrier (until round 4: coalesced loads, a transposition through
// LDS and a barrier).  All loads leave before the first value is used: clamped indices, not a branch around every load.
__device__ __forceinline__ void load_short(const double *__restrict__ v, int g0, int n, VdShortSmem &stg, double (&x)[kGPL]) {
    const int base = int(threadIdx.x) * kGPL;
    const double *vp = v + g0;
#pragma unroll
    for (int k = 0; k < kGPL; ++k) x[k] = vp[max(min(base + k, n - 1), 0)];
    double *row = stg.st + threadIdx.x * (kGPL + 1);
#pragma unroll
    for (int k = 0; k < kGPL; ++k) {
        x[k] = base + k < n ? x[k] : kVdPad;
        row[k] = x[k];
    }
}

// OR of a 16-bit value over the wave, as a wave-uniform number (DPP row shifts / broadcasts, zero fill; lane 63 ends up
// with everything)
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v) {
    v |= uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x111, 0xF, 0xF, true));  // row_shr:1
    v |= uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x112, 0xF, 0xF, true));  // row_shr:2
    v |= uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x114, 0xF, 0xF, true));  // row_shr:4
    v |= uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x118, 0xF, 0xF, true));  // row_shr:8
    v |= uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x142, 0xA, 0xF, true));  // row_bcast:15 -> rows 1, 3
    v |= uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x143, 0xC, 0xF, true));  // row_bcast:31 -> rows 2, 3
    return uint32_t(__builtin_amdgcn_readlane(int(v), 63));
}

// workgroup `blk` of the short-contig decoder (kernel vd_short in crf_sequence.hip; the pipelined decode kernel in
// crf_kernels.hip runs the same body for its Viterbi workgroups)
__device__ __forceinline__ void vd_short_block(const SeqArgs &A, const int blk, VdShortSmem &stg) {
    CE *lds = stg.tot;
    uint32_t *ldsm = stg.mtot;
    const int slot = threadIdx.x;
    const int g0 = __builtin_amdgcn_readfirstlane(A.cblk[blk]), n = __builtin_amdgcn_readfirstlane(A.cblk[blk + 1]) - g0;
    // which of the lane's genes start / end a contig: two bytes the host packed per lane (the per-gene flag bytes
    // took eight loads, an LDS round trip and 48 VALU instructions to unpack)
    const uint32_t bits = A.lane_bits[blk * kT + slot];
    const uint32_t first = bits & 0xffu, last = bits >> 8;
    // contigs are hundreds of genes long: at a given k most waves hold no contig start / end among their 64 genes, and a
    // wave-uniform test (a scalar branch) spares them the selects -- four v_cndmask_b32 per gene in either pass
    const uint32_t wave_bits = wave_or_u32(bits);
    double x[kGPL];
    load_short(A.dstate, g0, n, stg, x);
    const int cnt = min(kGPL, n - slot * kGPL);
    // (the first pass takes the lane's values as they arrive; the later ones read them from its LDS row instead of keeping
    // them in 16 VGPRs across the workgroup scans: the body has to fit the 64 registers of the pipelined decode kernel)
    const double *row = stg.st + slot * (kGPL + 1);
    // The lane's eight genes as ONE map x -> min(max(x + a, L), H): applying gene k to the map built so far is
    //   a += c_k;  L = clamp(L, lo, hi) + c_k;  H = clamp(H, lo, hi) + c_k      (c_k = (t11 - t00) + d_k)
    // -- the same bits as COp::combine with the gene's own map (c, lo + c, hi + c), because rounding is monotone
    // (fl(max(L, lo) + c) = max(fl(L + c), fl(lo + c))), in 8 instead of 13 fp64 instructions per gene.  A contig's
    // first gene is the constant map L = H = d (its `a` is never used again: a constant map stays one).
    CE P = COp::identity();
#pragma unroll
    for (int k = 0; k < kGPL; ++k) {
        const double dvk = x[k];
        const bool fst = (first >> k) & 1u;
        const double c = A.v_k + dvk;
        const double l2 = vd_min_s(vd_max_s(P.L, A.v_lo), A.v_hi) + c, h2 = vd_min_s(vd_max_s(P.H, A.v_lo), A.v_hi) + c;
        P.a += c;
        P.L = l2;
        P.H = h2;
        if ((wave_bits >> k) & 1u) {
            asm volatile("" ::: "memory");  // (keeps the scalar branch: the selects are what it is there to skip)
            P.L = fst ? dvk : l2;
            P.H = fst ? dvk : h2;
        }
    }
    const CE M = block_scan_exclusive<COp, false, CE, kScanThreads, true, true>(P, lds, static_cast<CE *>(nullptr));  // the workgroup starts at a contig start
    // ---- exact entering values.  M comes from COMPOSED maps: its additions are associated differently from the
    // sequential recursion, so M.L may differ from the sequential Delta in the last bits, and a decision that
    // lies within that noise of a threshold would depend on how the scan happens to be cut.  The clamp FORGETS:
    // wherever Delta_t lies beyond [lo, hi] by more than the noise, Delta_{t+1} = bound + c_{t+1} whatever came
    // before.  So: one approximate pass marks those positions, every lane walks back to the nearest one (or
    // to its contig's first gene) and re-runs the recursion sequentially from there -- a few genes on
    // average -- and the decisions below are those of the strictly sequential difference recursion, bit for
    // bit, independent of lane and workgroup boundaries (oracle_viterbi_delta is that recursion).
    // coarse margin of the workgroup: (4 n + 4) ulp(M) with n, nnz of all its (whole) contigs together; without the CSR
    // arrays (no exact pass possible) the flat 1e-6 of rounds 2 and 3
    double ulpM = 0.0, margin = 1e-6 * fmax(1.0, fmax(fabs(A.v_lo), fabs(A.v_hi)));
    if (A.csr_gene_ptr) {
        ulpM = vd_bound(A, double(A.csr_gene_ptr[g0 + n] - A.csr_gene_ptr[g0]), double(n)) * kVdEps;
        margin = vd_margin(double(n), ulpM);
    }
    // (the same number in every lane: kept in scalar registers, not in four of the 64 vector registers)
    ulpM = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(ulpM)), __builtin_amdgcn_readfirstlane(__double2loint(ulpM)));
    margin = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(margin)), __builtin_amdgcn_readfirstlane(__double2loint(margin)));
    uint32_t maps = 0, lane_map;
    bool sensitive = false;  // some decision of this lane lies within the noise of its threshold
    {
        double Dq = M.L;
#pragma unroll
        for (int k = 0; k < kGPL; ++k) {
            const double dvk = row[k];
            const double stepped = vd_min_s(vd_max_s(Dq, A.v_lo), A.v_hi) + (A.v_k + dvk);
            Dq = stepped;
            if ((wave_bits >> k) & 1u) {
                asm volatile("" ::: "memory");
                Dq = ((first >> k) & 1u) ? dvk : stepped;
            }
            // the gene's decisions (bit 0: Delta > hi, bit 1: Delta > lo) enter `maps` from the right: gene k ends up in bits
            // 2 (kGPL - 1 - k) + {0, 1}
            vd_shift_in_gt(maps, Dq, A.v_lo);
            vd_shift_in_gt(maps, Dq, A.v_hi);
            bool sens = fabs(Dq - A.v_hi) <= margin || fabs(Dq - A.v_lo) <= margin;
            if ((wave_bits >> (8 + k)) & 1u) {
                // a contig's last gene decides the end label: both of its "thresholds" are 0 (maps 3 / 0)
                asm volatile("" ::: "memory");
                const bool lst = (last >> k) & 1u;
                maps = lst ? ((maps & ~3u) | (Dq > 0.0 ? 3u : 0u)) : maps;
                sens = lst ? fabs(Dq) <= margin : sens;
            }
            sensitive |= sens;
        }
    }
    // the lane's eight genes as ONE label map (label after its last gene -> label before its first): both labels are walked
    // through the genes' maps, back to front -- an or and a bit-field extract per gene and label
    auto label_map = [&]() {
        uint32_t y0 = 0u, y1 = 1u;
#pragma unroll
        for (int k = kGPL - 1; k >= 0; --k) {
            y0 = __builtin_amdgcn_ubfe(maps, uint32_t(2 * (kGPL - 1 - k)) | y0, 1u);
            y1 = __builtin_amdgcn_ubfe(maps, uint32_t(2 * (kGPL - 1 - k)) | y1, 1u);
        }
        return y0 | (y1 << 1);
    };
    // back-to-front scan of the lane maps: the workgroup ends at a contig end, so the map entering from its right is
    // irrelevant (the last gene's map is constant).  Whether any lane of the workgroup has to look back rides on the
    // scan's barrier; in a workgroup where one has to (rare), the scan is repeated on the rebuilt decisions below.
    const bool lane_sensitive = sensitive && A.v_exact;
    int vote = lane_sensitive ? 1 : 0;
    uint32_t mtotal;
    lane_map = label_map();
    uint32_t lab = block_scan_exclusive_back<MapOp>(lane_map, ldsm, &mtotal, &vote) & 1u;
    const bool wg_sensitive = vote != 0;
    if (wg_sensitive) {
        // marks for the walk: per gene 1 = beyond hi, 2 = beyond lo (after this gene), 0 = inside or too close to tell
        double Dq = M.L;
        uint64_t sat = 0;
#pragma unroll
        for (int k = 0; k < kGPL; ++k) {
            if (k < cnt) {
                const double dvk = row[k];
            Dq = ((first >> k) & 1u) ? dvk : fmin(fmax(Dq, A.v_lo), A.v_hi) + (A.v_k + dvk);
                const uint64_t c = Dq >= A.v_hi + margin ? 1u : (Dq <= A.v_lo - margin ? 2u : 0u);
                sat |= c << (8 * k);
            }
        }
        *reinterpret_cast<uint64_t *>(&stg.st[slot * (kGPL + 1) + kGPL]) = sat;  // (the row's padding double)
        __syncthreads();
    } else {
        sensitive = false;
    }
    // A lane whose decisions all keep their distance from the thresholds has the sequential decisions already (its
    // values differ from the sequential ones by less than the margin at every gene: the clamp does not amplify, the
    // additions are the same).  Any other lane (ties of integer-weight models; otherwise one gene in a million)
    // rebuilds its entering value sequentially and decides again.
    bool lane_flagged = false;  // a decision of this lane lies inside the margin of ITS distance r to the last certain saturation
    if (sensitive && cnt > 0) {
        double D = 0.0;  // Delta of the gene before the lane's first, rebuilt sequentially
        int r = 0;       // genes since both labels last shared a predecessor for certain (0: a contig's first gene)
        auto certain = [&](double x) { return x >= A.v_hi + margin || x <= A.v_lo - margin; };
        if (!(first & 1u)) {
            // nearest restart at or before gene t0 - 1: a contig's first gene, or a gene whose predecessor is marked
            const int t0 = slot * kGPL;  // local index of the lane's first gene (> 0 here: gene 0 starts a contig)
            int p = t0 - 1;
            auto dval = [&](int t) { return stg.st[(t / kGPL) * (kGPL + 1) + t % kGPL]; };
            for (;; --p) {
                if (A.flags[g0 + p] & 1u) {
                    D = dval(p);
                    r = 0;
                    break;
                }
                // p >= 1: local gene 0 is a contig's first
                const uint32_t c = reinterpret_cast<const uint8_t *>(&stg.st[((p - 1) / kGPL) * (kGPL + 1) + kGPL])[(p - 1) % kGPL];
                if (c) {
                    D = (c == 1u ? A.v_hi : A.v_lo) + (A.v_k + dval(p));
                    r = 1;
                    break;
                }
            }
            for (int t = p + 1; t < t0; ++t) {
                r = certain(D) ? 1 : r + 1;
                D = fmin(fmax(D, A.v_lo), A.v_hi) + (A.v_k + dval(t));
            }
        }
        maps = 0;
#pragma unroll
        for (int k = 0; k < kGPL; ++k) {
            if (k < cnt) {
                const double dvk = row[k];
                const bool fst = (first >> k) & 1u;
                r = fst ? 0 : (certain(D) ? 1 : r + 1);
                D = fst ? dvk : fmin(fmax(D, A.v_lo), A.v_hi) + (A.v_k + dvk);
                const bool lst = (last >> k) & 1u;
                const double thi = lst ? 0.0 : A.v_hi, tlo = lst ? 0.0 : A.v_lo;
                maps |= ((D > thi ? 1u : 0u) | (D > tlo ? 2u : 0u)) << (2 * (kGPL - 1 - k));
                const double mr = vd_margin(double(r), ulpM);
                lane_flagged |= fabs(D - thi) <= mr || fabs(D - tlo) <= mr;
            }
        }
        if (A.vd_stats) {
            atomicAdd(A.vd_stats + 0, 1u);
            if (lane_flagged) atomicAdd(A.vd_stats + 1, 1u);
        }
    }
    // (a workgroup with candidates learns here whether any of them stayed inside its own margin, and scans the rebuilt
    // maps: the barrier of the vote also separates the two uses of the scan's LDS words)
    bool wg_flagged = false;
    if (wg_sensitive) {
        wg_flagged = __syncthreads_or(lane_flagged ? 1 : 0);
        lane_map = label_map();
        lab = block_scan_exclusive_back<MapOp>(lane_map, ldsm, &mtotal) & 1u;
    }
    uint64_t packed = 0;
#pragma unroll
    for (int k = kGPL - 1; k >= 0; --k) {
        lab = __builtin_amdgcn_ubfe(maps, uint32_t(2 * (kGPL - 1 - k)) | lab, 1u);
        packed |= uint64_t(lab) << (8 * k);
    }
    // a lane's eight labels are eight consecutive bytes of the output, a wave's 512: one 8-byte store per lane (at whatever
    // alignment the contig happens to start: the hardware takes unaligned global accesses); the workgroup's last lane
    // with genes may own fewer than eight
    {
        struct __attribute__((packed)) Bytes8 {
            uint64_t v;
        };
        int8_t *yp = A.y + g0 + slot * kGPL;
        if (cnt >= kGPL) {
            reinterpret_cast<Bytes8 *>(yp)->v = packed;
        } else {
            for (int k = 0; k < cnt; ++k) yp[k] = int8_t(packed >> (8 * k));
        }
    }
    // ---- contigs with a decision inside the margin: CRFsuite's own recursion decides (exact_delta_contig).  The
    // sensitive lanes mark the first genes of the contigs they touch; the workgroup then takes the marked contigs one by one.
    if (wg_flagged && A.csr_gene_ptr) {
        uint32_t *mark = stg.mark;  // kBlockGenes bits
        int *c_end = &stg.c_end;
        __syncthreads();  // (drains the label stores above: the exact pass overwrites some of them)
        if (slot < kBlockGenes / 32) mark[slot] = 0;
        __syncthreads();
        if (lane_flagged) {
            int prev = -1;
            for (int k = 0; k < cnt; ++k) {
                int t = slot * kGPL + k;
                while (!(A.flags[g0 + t] & 1u)) --t;  // the contig's first gene (local gene 0 starts one)
                if (t != prev) atomicOr(&mark[t >> 5], 1u << (t & 31));
                prev = t;
            }
        }
        __syncthreads();
        FixStage &st2 = *reinterpret_cast<FixStage *>(stg.st);  // 16.6 KB: kFixChunk pairs + their back-pointer words
        static_assert(sizeof(stg.st) >= sizeof(FixStage), "exact pass: LDS for one chunk of state scores");
        for (int wi = 0; wi < kBlockGenes / 32; ++wi) {
            uint32_t bits = mark[wi];  // (every lane reads the same word: uniform control flow)
            while (bits) {
                const int cs = wi * 32 + __builtin_ctz(bits);
                bits &= bits - 1;
                if (slot == 0) *c_end = n;
                __syncthreads();
                for (int i = cs + slot; i < n; i += kT)
                    if (A.flags[g0 + i] & 2u) {
                        atomicMin(c_end, i + 1);
                        break;
                    }
                __syncthreads();
                const int ce = *c_end;
                __syncthreads();
                if (slot == 0 && A.vd_stats) {
                    atomicAdd(A.vd_stats + 2, 1u);
                    atomicAdd(A.vd_stats + 3, uint32_t(ce - cs));
                }
                exact_delta_contig(A, g0 + cs, g0 + ce, st2);
            }
        }
    }
}

}  // namespace
}  // namespace gecco
