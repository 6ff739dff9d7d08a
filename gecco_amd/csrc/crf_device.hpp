// Device-side interface between the plan/C-ABI layer and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace gecco {

// Arguments of the windowed-marginals kernels (row W of SURVEY.md §8a).
// "slot space" = genes of all scored contigs laid end to end, each contig padded to at
// least `W` slots (gecco/crf/__init__.py:216-227); windows start at slots.
struct WinArgs {
    const int32_t *gene_ptr;   // [n_genes+1]  CSR attribute offsets            (HBM, streamed)
    const int32_t *attr_id;    // [nnz]        attribute ids                    (HBM, streamed)
    const double2 *wtab2;      // [A]          (w[a][other], w[a][label]) L == 2 (L2-resident)
    const double *wtab;        // [A*L]        dense state weights, any L        (L2-resident)
    const int32_t *c_slot;     // [K+1]        first slot of every scored contig
    const int32_t *c_gene;     // [K]          first gene of every scored contig
    const int32_t *c_n;        // [K]          number of genes of every scored contig
    const int4 *tile_desc;     // [ntiles]     (gene-slot shift, first contig, last contig, flags: 1 = regular)
    const uint64_t *start_bits;// [S/64+2]     bit q: a window may start at slot q; zero words in front (1) and behind (24)
    double *p_out;             // [n_genes]
    double2 *state_out;        // [n_genes] or null: raw state scores (s[0], s[1]) as a by-product (fast L == 2 kernel)
    double *dstate_out;        // [n_genes] or null: s[1] - s[0] as a by-product
    int32_t K, S, ntiles, W, step, L, label;
    int32_t n_genes, A;        // CSR extent and weight-table rows (buffer descriptors)
    int32_t tiles_per_wg;      // DP phases per workgroup of the fast kernel (plan geometry)
    int32_t all_regular;       // no padded or skipped contig in the whole batch: slot space = gene space, tile_desc unused
    uint32_t rescale_mask;     // bit k: renormalise the DP vectors after step k
    // transitions in the transformed basis (rows/cols ordered (other, label)):
    //   mu01 = m01*m10/m00^2, mu11 = m11/m00, kappa = m10/m00, rho = mu11/mu01  with m = exp(trans)
    double mu01, rho, kappa_over_mu01, inv_kappa;
    double expc[12];            // 1/13!, 1/12!, ..., 1/2!: Taylor coefficients of exp (SGPR-resident)
    double ratio_zmax;          // a window whose forward pass ends on Z >= this (1e250) is repeated in the max-normalised form; -1: every
                                // window is (GECCO_CRF_RATIO=0: A/B runs, tests)
    double g00, g01, g10, g11;  // exp(trans - max), (other, label) order: generic kernel
    int32_t generic;            // 1: dispatch to the generic window kernel
    const double *exp_trans;   // [L*L] exp(trans) for the generic kernel
    double *scratch;           // generic kernel workspace
    const double *rtab;        // [32] mu01 * 2^(j/32): exp table of the ratio-form slot constants (mu_exp_tab)
    // What the host may know of the CSR arrays (batch driver's direct path: they are its own copies; a caller of the resident
    // API keeps them in device memory, where the kernel has to look: -1).  csr_end = gene_ptr[n_genes], csr_begin = gene_ptr[0]:
    // a batch of ONE regular tile (crf_windowed_small_l2) takes them from here instead of loading them in front of its first
    // attribute load -- a round trip over PCIe when the arrays live in pinned host memory.
    int32_t csr_end, csr_begin;
};

// Geometry of the fast L==2 kernel.
constexpr int kWinThreads = 256;  // lanes (= window starts) per workgroup: 4 waves, 8 workgroups per CU at <= 64 VGPRs
constexpr int kWinMaxW = 32;      // largest window the register-resident kernel handles
constexpr int kWinTilesPerWg = 2; // default DP phases per workgroup (GECCO_CRF_TILES_PER_WG=1..3 overrides; A/B runs)
constexpr int kWinTiles1MaxSlots = 350000;  // batches of up to this many slots: one tile per workgroup (tools/tiles_sweep.py: the window kernel wins up to 0.4 M genes, the pipelined launch -- seven workgroups per CU -- up to 0.3 M)

// ---- shared device helpers ----------------------------------------------------------------------
// exp(-t) for t >= 0: n = rint(t log2 e), r = n ln2 - t in two pieces (|r| <= ln2/2), degree-13
// Taylor polynomial (truncation 2e-18), v_ldexp_f64 for 2^-n (flushes to 0 by itself for huge t).
// The coefficients 1/13! .. 1/2! (fill_exp_coefficients) travel as kernel arguments so that they sit in SGPRs: as literals
// every one of them costs two v_mov per use, as many VALU slots as the polynomial itself.
__device__ __forceinline__ double exp_neg(double t, const double (&kExpC)[12]) {
    const double n = rint(t * 1.4426950408889634);
    double r = fma(n, 0.6931471805599453094, -t);   // ln2 hi
    r = fma(n, 2.3190468138462996e-17, r);           // ln2 lo
    // exp(-t) = 2^-n exp(r),  r = n ln2 - t
    double p = kExpC[0];
#pragma unroll
    for (int i = 1; i < 12; ++i) p = fma(p, r, kExpC[i]);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, -int(n));
}

// exp(x) for any sign with the same polynomial: overflows to +inf / flushes to 0 through v_ldexp_f64
__device__ __forceinline__ double exp_signed(double x, const double (&kExpC)[12]) {
    const double n = rint(x * 1.4426950408889634);
    double r = fma(-n, 0.6931471805599453094, x);
    r = fma(-n, 2.3190468138462996e-17, r);
    double p = kExpC[0];
#pragma unroll
    for (int i = 1; i < 12; ++i) p = fma(p, r, kExpC[i]);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, int(n));
}
// mu exp(x) = 2^e (mu 2^(j/32)) exp(r),  n = rint(x 32 / ln 2) = 32 e + j,  r = x - n ln2 / 32 in two pieces
// (|r| <= ln2 / 64), degree-6 Taylor polynomial (truncation 3e-18 relative), the factor mu 2^(j/32) from a 32-entry
// table (`rtab`, built by the host in extended precision; cache-resident, requested before the polynomial),
// v_ldexp_f64 for 2^e (overflows to +inf / flushes to 0 by itself).  17 VALU instructions and five coefficients,
// against 28 and twelve for exp_signed followed by the multiplication.
__device__ __forceinline__ double mu_exp_tab(double x, const double *__restrict__ rtab, const double (&kExpC)[12]) {
    const double n = rint(x * 46.166241308446828384);  // 32 / ln 2
    const int ni = int(n);
    const double t = rtab[ni & 31];
    double r = fma(-n, 0.02166084939249829, x);          // ln2 / 32, high part
    r = fma(-n, 7.247021293269686e-19, r);                // low part
    double p = fma(kExpC[7], r, kExpC[8]);                 // 1/6!, 1/5!
    p = fma(p, r, kExpC[9]);
    p = fma(p, r, kExpC[10]);
    p = fma(p, r, kExpC[11]);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p * t, ni >> 5);
}
inline void fill_exp_coefficients(double (&c)[12]) {
    double f = 1.0;  // k!
    for (int k = 2; k <= 13; ++k) {
        f *= double(k);
        c[13 - k] = 1.0 / f;
    }
}

// ---- whole-contig kernels (crf_sequence.hip) ------------------------------------------------
constexpr int kSeqGenesPerLane = 8;  // genes folded sequentially by one lane of the flat scans
constexpr int kSeqBlockGenes = 256 * kSeqGenesPerLane;  // genes per workgroup of the flat scans

// `rs` != 0: the span contains the first gene of a contig; everything before that gene is
// forgotten (segmented scan), so rows are identical and hold the exact values since the reset.
struct VE {  // max-plus 2x2 scan element
    double a00, a01, a10, a11, rs;
};
struct FE {  // sum-product 2x2 scan element + power-of-two exponent + sum of emission maxima
    double a00, a01, a10, a11, ex, ms, rs;
};

struct CE {  // Viterbi of a 2-label model on score differences: Delta -> min(max(Delta + a, L), H)
    double a, L, H;
};

struct SeqArgs {
    const double2 *state;    // [n_genes]  (s[label 0], s[label 1])
    const double *dstate;    // [n_genes]  s[1] - s[0]: all the difference-form Viterbi needs (8 B/gene)
    const uint8_t *flags;    // [n_genes]  bit0: first gene of a contig, bit1: last gene
    const uint16_t *flat_bits;  // [ceil(n_genes / 2048) * 256] the same for the flat layout (lane l owns genes 8 l .. 8 l + 7)
    const uint16_t *lane_bits;  // [n_cblocks * 256] short contigs: low byte = which of the lane's 8 genes start a contig, high byte = end one
    const int32_t *cblk;     // [n_cblocks+1] short contigs only: first gene of every workgroup of WHOLE contigs (<= kSeqBlockGenes genes)
    const int32_t *cblk_rank;   // [n_cblocks] non-empty contigs before the workgroup's first
    const int32_t *ne_contig;   // [non-empty contigs] their indices in contig_ptr
    const int32_t *contig_ptr;  // [n_contigs+1] (device)
    int32_t n_cblocks;
    int32_t short_contigs;   // 1: no contig is longer than one scan block (kSeqBlockGenes)
    const double *smax;      // [n_genes] or null: max(s[0], s[1]) next to dstate (whole-contig marginals with log Z)
    int32_t n_contigs, n_genes;
    double m00, m01, m10, m11;  // exp(trans - mx)
    double t00, t01, t10, t11;  // raw transition weights (Viterbi)
    double mx;                  // max(trans)
    double v_lo, v_hi, v_k;     // difference-form Viterbi: t01-t11, t00-t10, t11-t00
    int32_t v_exact;            // 1: decisions within rounding noise of a threshold are re-derived sequentially (GECCO_CRF_VD_EXACT=0: A/B runs)
    // What bounds CRFsuite's accumulated scores, for the margin inside which a decision of the difference form is not
    // PROVABLY crf1dc_viterbi's (crf_vd_short.hpp): |delta_t| <= nnz * max|w| + n * max|trans| over a contig of n genes and
    // nnz attribute entries.  v_wmax2 = 2 max_a max_y |w[a][y]| (the factor 2 also covers |s[1] - s[0]|), v_tmax = max |trans|.
    double v_wmax2, v_tmax;
    int32_t v_nmax;             // genes of the batch's longest contig (flat layout: one bound for the whole batch)
    // flat layout: candidates for that margin (lanes with a decision within the coarse margin; waves without any reset),
    // written by vd_replay, judged by vd_refine: {lane | bit 31: "wave without reset", 0, Delta entering the lane}
    struct VdCand {
        uint32_t lane, pad;
        double d_enter;
    };
    VdCand *vCand;              // [lanes + 4 * workgroups]
    unsigned long long *vBound; // [0]: bit pattern of the largest per-contig bound (vd_fold); [1] low word: number of candidates
    uint32_t *vd_stats;         // [4] or null, accumulated over launches: coarse candidates, decisions inside the margin,
                                // contigs decoded again with CRFsuite's recursion, their genes
    // CSR of the batch + weight pairs (w[a][0], w[a][1]): contigs with such a decision are decoded again with
    // CRFsuite's own delta recursion on freshly summed state scores (null: no such pass)
    const int32_t *csr_gene_ptr, *csr_attr_id;
    const double2 *csr_wtab01;
    int32_t csr_n_attrs;
    uint8_t *fix_flag;          // [n_contigs] long contigs: 1 = decode this contig again (vd_replay -> vd_exact_fix)
    int32_t raw_fold;           // 1: eight max-normalised factors multiply without renormalisation (transition spread * 8 < 600)
    double expc[12];            // Taylor coefficients of exp (SGPR-resident), see exp_neg
    // workspaces: one element per lane (n_genes / kSeqGenesPerLane) or per workgroup
    VE *vLane, *vBlock;
    uint32_t *vMaps, *vLaneMap, *vBlockMap;
    FE *fLane, *fBlock, *fLaneSuf, *fBlockSuf;
    double2 *alpha;      // [n_genes] F: alpha_t
    double2 *contigTmp;  // [n_genes] sparse: values parked at contig ends
    // outputs
    double *marg;     // [n_genes*2]
    double *lognorm;  // [n_contigs] or null
    int8_t *y;        // [n_genes]
    double *score;    // [n_contigs] or null
};

hipError_t launch_seq_state(const int32_t *gene_ptr, const int32_t *attr_id, const double2 *wtab01, int n_attrs, int n_genes,
                            double2 *state, hipStream_t stream);
hipError_t launch_seq_marginals(const SeqArgs &a, const int32_t *d_contig_ptr, hipStream_t stream);
hipError_t launch_seq_marginals_short(const SeqArgs &a, const int32_t *gene_ptr, const int32_t *attr_id, const double2 *wtab01,
                                      int n_attrs, const int32_t *d_contig_ptr, hipStream_t stream);
hipError_t launch_seq_viterbi(const SeqArgs &a, const int32_t *d_contig_ptr, hipStream_t stream);
// The decode step pipelined over batches (crf_decode_pipelined): the window tiles of `w`'s batch and the Viterbi workgroups of
// `s`'s batch (short contigs, score differences already in s.dstate) in one launch, nothing exchanged inside it.
bool decode_pipelined_ok(const WinArgs &w, const SeqArgs &s);
hipError_t launch_decode_pipelined(const WinArgs &w, const SeqArgs &s, hipStream_t stream);

// labels only, from a.dstate; needs trans[0][1] - trans[1][1] <= trans[0][0] - trans[1][0]
hipError_t launch_seq_viterbi_delta(const SeqArgs &a, hipStream_t stream);
hipError_t launch_seq_state_delta(const int32_t *gene_ptr, const int32_t *attr_id, const double2 *wtab01, int n_attrs, int n_genes,
                                  double *dstate, hipStream_t stream);

// ---- row R on packed arrays (crf_segment.hip) ----------------------------------------------
struct SegE {  // what a span of genes does to the grouper, as a function of the state it is entered in
    uint32_t map;       // bit s = state left behind when entered in state s
    uint32_t ng0, ng1;  // cluster runs started inside the span when entered in state 0 / 1
    uint32_t ann;       // annotated genes in the span
};
// What the refiner is asked for (refine.py:75-116).  criterion 0 = "gecco" (annotated genes and genes away from the
// contig edges >= n_cds, :142-156), 1 = "antismash" (mean p >= average_threshold, distinct marker domains >=
// n_biopfams, genes >= n_cds, :157-163).  `bio_ptr` / `bio_id`: per gene the marker domains it carries (CSR over genes,
// ids in [0, kSegMaxMarkers)), device pointers, antismash only.
constexpr int kSegMaxMarkers = 256;
struct SegParams {
    double threshold = 0.8;
    int32_t n_cds = 5, edge_distance = 0, trim = 1, carry = 0;
    int32_t criterion = 0, n_biopfams = 5;
    double average_threshold = 0.6;
    const int32_t *bio_ptr = nullptr, *bio_id = nullptr;
    int32_t row_contig0 = 0, row_gene0 = 0;  // added to the contig / gene indices of the rows written (a chunk of a larger batch)
};
struct SegArgs {
    const double *p;        // [n_genes] probabilities (NaN: none)
    const uint8_t *ann;     // [n_genes] the gene has at least one domain
    const uint8_t *flags;   // [n_genes] bit0: first gene of a contig, bit1: last gene
    const int32_t *cptr;    // [n_contigs+1]
    int32_t n_genes, n_contigs;
    double thr;
    int32_t n_cds, edge, trim, carry;
    int32_t criterion, n_bio;
    double avg_thr;
    const int32_t *bio_ptr, *bio_id;
    // workspace
    SegE *lane, *block;     // exclusive prefix per lane inside its workgroup; per workgroup
    int32_t *pre;           // [n_genes+1] annotated genes before gene g
    int2 *raw;              // (first gene, one past the last gene) of every run, in order
    int4 *val;              // (contig or -1-contig if rejected, number, first, last+1) per raw run
    int2 *tile;             // per 256 raw runs: kept rows, their genes -> exclusive prefixes
    int32_t *n_raw;
    // outputs
    int32_t *seg;           // [max_seg][4]
    int32_t max_seg;
    int32_t *seg_off;       // [max_seg+1] or null: gene offsets of the kept rows
    int32_t *total;
    double *gout;           // [gcap] or null: probabilities of the genes of the kept rows, row after row
    int32_t gcap;
    int32_t row_c0, row_g0; // offsets of the rows' contig / gene indices as written
};
size_t segment_workspace_bytes(int n_genes, int n_contigs);
// d_gather (may be null; needs d_seg_off): the probabilities of the rows' genes, row after row.  d_seg / d_seg_off / d_total /
// d_gather may live in pinned host memory (they are only written).
hipError_t launch_segment(const double *d_p, const uint8_t *d_ann, const uint8_t *d_flags, const int32_t *d_cptr,
                          int n_genes, int n_contigs, const SegParams &params, int32_t *d_seg, int max_seg,
                          int32_t *d_seg_off, int32_t *d_total, void *d_work, hipStream_t stream, double *d_gather = nullptr,
                          int gather_cap = 0);
// batch driver's wire format: degree bytes of a chunk's n genes -> its n + 1 row pointers (base + prefix sums); scratch of
// degree_scratch_bytes(n); d_deg readable for 32 bytes past n
size_t degree_scratch_bytes(int n);
hipError_t launch_degree_to_row_ptr(const uint8_t *d_deg, int n, int32_t base, int32_t *d_row_ptr, int32_t *d_scratch, hipStream_t stream);

// flags[g] for g in [0, n_genes + 8): bit 0 = first gene of its contig, bit 1 = last (d_flags 8-byte aligned, room for
// n_genes + 16 bytes; d_cptr in device memory: every lane searches it)
hipError_t launch_contig_flags(const int32_t *d_cptr, int n_contigs, int n_genes, uint8_t *d_flags, hipStream_t stream);
// `bytes` (a multiple of 16, 16-byte aligned both sides) from device-visible memory -- a pinned host block -- to device memory
hipError_t launch_copy_block(const void *src, void *dst, size_t bytes, hipStream_t stream);
// both halves of the wire format in two launches: the row pointers (d_deg may be null: none) and, with d_attr16, nnz 16-bit
// attribute indices widened to 32-bit ones (both 16-byte aligned device buffers with 8 elements of slack)
hipError_t launch_wire_format(const uint8_t *d_deg, int n, int32_t base, int32_t *d_row_ptr, int32_t *d_scratch,
                              const uint16_t *d_attr16, int64_t nnz, int32_t *d_attr32, hipStream_t stream);

// ---- any number of labels (crf_general.hip) -------------------------------------------------
constexpr int kGenMaxL = 32;  // labels: a group of next-pow2(L) lanes must fit in half a wave
constexpr int kGenMaxW = 48;  // window length: alpha-hat of a whole window lives in LDS (<= 3 KB per step)
struct GenArgs {
    const int32_t *gene_ptr, *attr_id;
    const double *wtab;       // [A*L] state weights
    const double *exp_trans;  // [L*L] exp(trans)
    const double *trans;      // [L*L] raw transition weights (Viterbi)
    const int32_t *contig_ptr;
    int32_t L, A, n_genes, n_contigs;
    // per-gene workspace
    double *state;  // [n*L] raw state scores            (Viterbi)
    double *E;      // [n*L] exp(state - max_y state)    (marginals)
    double *smax;   // [n]   max_y state
    double *alpha;  // [n*L] scaled forward vectors
    double *scale;  // [n]
    uint8_t *back;  // [n*L] Viterbi back-pointers
    // outputs
    double *marg, *lognorm, *score;
    int8_t *y;
    // long contigs (n_chunks > 0): contigs cut into chunks of gen_chunk_genes() genes
    const int32_t *ch_g0;      // [n_chunks+1] first gene of every chunk (chunks never straddle contigs)
    const int32_t *ch_contig;  // [n_chunks]
    const int32_t *cc_ptr;     // [n_contigs+1] chunks of every contig
    int32_t n_chunks;
    double *chM;               // [n_chunks*L*L] rows of the chunks' transfer matrices
    int32_t *chEx;             // [n_chunks*L]   power-of-two exponents of the rows (sum-product)
    double *chV, *chB;         // [n_chunks*L]   vector entering every chunk / beta of its last gene
    double *chZ;               // [n_chunks]     partial log-partition
    uint8_t *chMap;            // [n_chunks*L]   label of the chunk's last gene -> label of the gene before the chunk
    int8_t *chY;               // [n_chunks]     label of the chunk's last gene
    int32_t wave_tmax;            // gl_viterbi_wave: contigs longer than this are left to the chunked kernels (0: none is)
    int32_t rows_rescale_period;  // gl_chunk_rows_mfma: steps between two power-of-two rescalings of a column (1 or 4; host: 4 max|trans| < 600)
    // windowed path: slot space of the plan
    const int32_t *c_slot, *c_gene, *c_n;
    const uint64_t *start_bits;
    double *p_out;
    int32_t K, S, W, label;
};
hipError_t launch_gen_state(const GenArgs &a, hipStream_t stream);
hipError_t launch_gen_windowed(const GenArgs &a, hipStream_t stream);   // p_out must be zeroed first
// 3 or 4 labels: one lane per window start (the two-label kernel's design); tile geometry gen_small_tile_out(W)
bool gen_small_ok(int L, int W, const double *trans_host);
int gen_small_tile_out(int W);
hipError_t launch_gen_windowed_small(const GenArgs &a, const double *trans_host, const int4 *d_tile_desc, int ntiles,
                                     hipStream_t stream);
hipError_t launch_gen_marginals(const GenArgs &a, hipStream_t stream);
hipError_t launch_gen_viterbi(const GenArgs &a, hipStream_t stream);
// 9 to 32 labels, batches of many contigs: one wave per contig, CRFsuite's sequential recursion (no chunk tables needed)
hipError_t launch_gen_viterbi_wave(const GenArgs &a, hipStream_t stream);
hipError_t launch_gen_marginals_wave(const GenArgs &a, hipStream_t stream);  // row F likewise (a.E, a.smax, a.alpha, a.scale)
int gen_chunk_genes();

// weighted domain composition of called clusters (crf_composition.hip); d_tmp: one double per domain row
hipError_t launch_composition(const int32_t *d_seg, int n_seg, const int32_t *d_dom_ptr, const int32_t *d_dom_col,
                              const double *d_dom_w, double *d_tmp, int n_cols, int normalize, double *d_out,
                              hipStream_t stream);

// ---- reference-bits mode (crf_exact.hip): CRFsuite's operation order, correctly rounded exp ----------------------
bool reference_bits_ok(int L, int W);
size_t reference_scratch_bytes(int n_genes);
hipError_t launch_windowed_reference(const WinArgs &w, const double2 *wtab01, const double *exp_trans_host, void *scratch, hipStream_t stream);

const char *windowed_kernel_name(int W, int L, bool fast);
// tile_out = output slots per workgroup for the kernel that (W, L) dispatches to.
int windowed_tile_out(int W, int L, int tiles_per_wg);
hipError_t launch_windowed(const WinArgs &a, hipStream_t stream);
hipError_t launch_fill_nan(double *p, const int2 *ranges, int n_ranges, hipStream_t stream);

}  // namespace gecco
