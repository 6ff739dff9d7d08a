// Batch plan: host-side layout of one batch of contigs in "slot space" + the device copies
// the kernels need.  Mirrors the bookkeeping half of gecco/crf/__init__.py:209-240
// (which contigs are scored, how they are padded, how many windows there are).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

#include "crf_device.hpp"
#include "crf_model.hpp"

namespace gecco {

// Per-device copies of a model's tables (owned by the Model, created on first use).
struct DeviceTables {
    int device = -1;
    double *wtab = nullptr;       // [A*L]
    double2 *wtab2[2] = {nullptr, nullptr};  // L == 2: [A] (other, label) for label = 0 / 1
    double *exp_trans = nullptr;  // [L*L]
    double *trans = nullptr;      // [L*L] raw weights (general-L Viterbi)
    double wmax_abs = 0.0, tmax_abs = 0.0;  // largest |state weight| / |transition weight| (bounds of the Viterbi exactness margin)
    double *rtab[2] = {nullptr, nullptr};  // L == 2: [32] mu01(label) * 2^(j/32), the exp table of the window kernel's slot constants
    // Host-side constants of the kernels' argument blocks that depend on the model alone (L == 2), computed ONCE here: a launch
    // used to recompute them (17 exp calls, 36 divisions, three getenv scans per pipelined decode call: ~1 us of a 5 us
    // launch-bound step, tools/host_issue_probe.py).
    struct WinConsts {
        double mu01, rho, kappa_over_mu01, inv_kappa, g00, g01, g10, g11, expc[12], ratio_zmax;
    } win[2];  // by queried label
    struct SeqConsts {
        double mx, m00, m01, m10, m11, v_lo, v_hi, v_k, expc[12];
        int32_t raw_fold, v_exact;
    } seq;
    bool consts_ok = false;
};

// A pinned host block mirrored by a device block: plan tables are written on the host side and reach
// the device with ONE asynchronous copy.  Grow-only, so a plan that is rebuilt for the next chunk of a
// batch (crf_session.cpp) allocates nothing once it has seen its largest chunk.
struct Arena {
    char *h = nullptr, *d = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes, const char *what);  // the plan's device must be current
    void release();
};

struct Plan {
    const Model *model = nullptr;
    int device = -1;  // -1: host-only plan (layout queries work, launches return ENODEV)
    int32_t W = 0, step = 1, pad = 1;
    int32_t n_contigs = 0, n_genes = 0, n_max = 0;  // (n_max: genes of the longest contig)
    int64_t n_windows = 0;
    // slot-space layout (host)
    int32_t K = 0, S = 0, ntiles = 0, tile_out = 0, tiles_per_wg = 1;
    std::vector<int32_t> c_slot, c_gene, c_n;
    std::vector<int4> tile_desc;
    std::vector<uint64_t> start_bits;
    std::vector<int2> skipped;  // gene ranges of contigs skipped by pad == 0
    std::vector<int32_t> contig_ptr;
    std::vector<int32_t> irr_prefix;  // scratch of plan_build (kept for its capacity)
    uint32_t rescale_mask = 0;
    bool all_regular = false;   // no padded or skipped contig: every tile maps slots to genes by the identity
    bool fast_ok = false;       // the register-resident kernel takes this shape
    bool force_generic = false; // GECCO_CRF_FORCE_GENERIC=1 (tests): always use the generic kernel
    bool general = false;       // any-L kernels (crf_general.hip): L != 2, or GECCO_CRF_FORCE_GENERAL=1 (tests)
    bool gen_small = false;     // ... 3 or 4 labels: the lane-per-window kernel (gl_windowed_small) and its tile geometry
    std::string kernel_name;
    // device copies (all inside `tables`)
    Arena tables;
    int32_t *d_c_slot = nullptr, *d_c_gene = nullptr, *d_c_n = nullptr, *d_contig_ptr = nullptr;
    int4 *d_tile_desc = nullptr;
    uint64_t *d_start_bits = nullptr;
    int2 *d_skipped = nullptr;
    const DeviceTables *tables_model = nullptr;
    // whole-contig scans (rows F, V) and the segmenter: contig flags + scan block table, built on first use
    Arena seq;
    bool seq_ready = false;
    uint8_t *d_seq_flags = nullptr;
    const uint16_t *d_seq_flat_bits = nullptr;  // [ceil(n / 2048) * 256] the same bits for the flat layout (lane l owns genes 8 l ..)
    const uint16_t *d_seq_lane_bits = nullptr;  // [n_cblocks * 256] contig start / end bits of the 8 genes of every lane (short contigs)
    int32_t *d_seq_cblk = nullptr;    // short contigs: first gene of every workgroup of whole contigs (<= 2048 genes), [n_cblocks+1]
    int32_t n_cblocks = 0;
    int32_t *d_seq_cblk_rank = nullptr, *d_seq_ne_contig = nullptr;  // non-empty contigs before every workgroup; their indices
    int32_t n_empty_contigs = 0;
    bool seq_short = false;           // no contig longer than one scan block: whole contigs per workgroup, one launch per decoder
    // workspaces, allocated on first use, grow-only
    char *d_seq_ws = nullptr, *d_gen_ws = nullptr, *d_seg_ws = nullptr;
    size_t vbound_sig = 0;  // layout for which the exactness test's bound / counter / contig flags were last zeroed
    double *d_win_scratch = nullptr;
    size_t seq_ws_cap = 0, gen_ws_cap = 0, seg_ws_cap = 0, win_scratch_cap = 0;
    // chunk tables of the any-L whole-contig kernels (first gene / contig of every chunk, chunks of every contig): they
    // depend on the plan's contigs and the chunk length only, so they are built and uploaded on first use
    // Two sets: every contig (marginals, Viterbi of small models), and the contigs longer than `min_len` only (Viterbi of
    // 17-32 labels: the long tail of a batch whose other contigs take the wave-per-contig kernel).
    struct GenTab {
        char *d = nullptr;
        int32_t chunk = 0, min_len = -1;  // (chunk == 0: not built)
        size_t nch = 0, off1 = 0, off2 = 0;
    } gen_tab[2];
    // Viterbi of 17-32 labels: the chunked kernels of a batch's long contigs run on a stream of the plan's own next to the
    // wave-per-contig kernel of the others (forked from and joined to the caller's stream by events)
    int32_t gen_wave_tmax = INT32_MIN, gen_wave_tmax_f = INT32_MIN;  // the splits chosen for this layout (Viterbi, marginals; INT32_MIN: not yet)
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // Pipelined decode (plan_run_decode_pipelined): what the last call left for the next one -- the batch's score differences
    // in buffer `parity` of the workspace (pending) and the batch's CSR arrays, which the caller keeps alive until then.
    struct Pipe {
        bool pending = false;
        int parity = 0;
        const int32_t *gene_ptr = nullptr, *attr_id = nullptr;
    } pipe;
    bool async_tables = false;  // the owner launches everything on ONE stream (batch driver): table uploads are not waited for
    bool tables_by_kernel = false;  // copied tables are fetched from the pinned block by a small launch on the upload stream, not by
                                    // the copy engine (batch driver: the engine is busy with the next chunk's arrays by then)
    bool tables_in_host_memory = false;  // the window kernel reads the plan tables from the pinned block itself (batch driver:
                                         // one copy and one inter-copy gap less per chunk; they are ~0.1 MB, read once)
    int64_t csr_begin = -1, csr_end = -1;  // gene_ptr[0] / gene_ptr[n_genes] when the owner knows them (batch driver's direct path), else -1
    // Reference-bits mode (crf_exact.hip): windowed marginals in CRFsuite's own operation order with a correctly rounded exp --
    // the reference's output files bit for bit, at about six times the fast kernels' time.  Set by the owner before
    // plan_build (batch driver: gecco_crf_session_set_reference_bits); GECCO_CRF_REFERENCE_BITS=1 sets it for every 2-label plan that runs windowed marginals.
    bool reference_bits = false;
    bool windowed_use = true;  // the owner runs windowed marginals on this layout (batch driver: false for Viterbi-only and
                               // whole-contig-marginal requests): the environment switch applies to such layouts only
    bool reference_now = false;  // (this layout runs in reference-bits mode: reference_bits, or the environment)
    bool seq_in_host_memory = false;     // small batches (batch driver's direct path): the whole-contig tables AND the contig flags
                                         // stay in the pinned block, flags built by the host -- no copy, no launch in front of the decoder
    std::mutex ws_mutex;  // guards the lazy workspace / table creation: launches of one plan may come from several threads
    ~Plan();
};

// `upload_stream` / `sync`: the device copy of the tables is ONE asynchronous copy on that stream; with
// sync it is waited for before returning (the caller may then launch on any stream).  `p` may be a plan
// that has been built before: its allocations are reused.
int plan_build(const Model &m, int device, const int32_t *contig_ptr, int32_t n_contigs, int32_t W, int32_t step,
               int32_t pad, Plan &p, hipStream_t upload_stream = nullptr, bool sync = true);
// contig flags / scan block table on the device (what rows F, V and R need), uploaded on `stream`
int plan_ensure_seq(Plan &p, hipStream_t stream, bool sync = true);
// row R chained behind the marginals on the same stream: d_p and d_annotated are device arrays over the
// plan's genes; rows go to d_seg (device-accessible memory), their number to d_total
int plan_run_segment(Plan &p, const double *d_p, const uint8_t *d_annotated, const SegParams &params, int32_t *d_seg,
                     int32_t max_seg, int32_t *d_seg_off, int32_t *d_total, hipStream_t stream, double *d_gather = nullptr,
                     int32_t gather_cap = 0);
int plan_run_windowed(Plan &p, const int32_t *d_gene_ptr, const int32_t *d_attr_id, int32_t label, double *d_p_out,
                      hipStream_t stream);
// windowed marginals + whole-contig Viterbi of the same batch in one pass over the CSR
// Decode pipelined over batches: enqueue the windowed marginals of `cur`'s batch (cur may be null: flush) and the Viterbi
// labels of the batch the previous call scored on `prev` (null on the first call; may be the same plan) -- in ONE launch
// when both sides qualify (2-label model, W = 20, short contigs: crf_decode_pipelined), in separate launches otherwise.
int plan_run_decode_pipelined(Plan *cur, const int32_t *d_gene_ptr, const int32_t *d_attr_id, int32_t label, double *d_p_out,
                              Plan *prev, int8_t *d_prev_y, hipStream_t stream);
int plan_run_decode(Plan &p, const int32_t *d_gene_ptr, const int32_t *d_attr_id, int32_t label, double *d_p_out,
                    int8_t *d_y, double *d_score, hipStream_t stream);
int plan_run_marginals_full(Plan &p, const int32_t *d_gene_ptr, const int32_t *d_attr_id, double *d_marg,
                            double *d_lognorm, hipStream_t stream);
int plan_run_viterbi(Plan &p, const int32_t *d_gene_ptr, const int32_t *d_attr_id, int8_t *d_y, double *d_score,
                     hipStream_t stream);
// counters of the 2-label Viterbi decoder (crf_device.hpp: SeqArgs::vd_stats); waits for the device
int plan_viterbi_stats(Plan &p, int64_t out[4], bool reset);
// returns GECCO_CRF_* ; on HIP failure sets the error text
int check_hip(hipError_t e, const char *what);
int get_device_tables(const Model &m, int device, const DeviceTables **out);

}  // namespace gecco
