/*
 * gecco_crf.h -- C ABI of the MI355X-native linear-chain CRF inference engine for GECCO's
 * `gecco.crf` hot path.
 *
 * This is the drop-in boundary: the native interface GECCO reaches today for this path is
 * python-crfsuite's `Tagger` ([EXT] CRFsuite 0.12 `crfsuite_tagger_t`: open / labels / set /
 * marginal_point / viterbi), crossed once per sliding window at
 *     /root/reference/gecco/crf/__init__.py:253   self.model.predict_marginals_single(feats[win])
 * inside the per-contig window loop at :244-258.  The entry points below replace that
 * per-window boundary with one call per *batch of contigs*; the reference-side binding a
 * maintainer would add (a ctypes stub inside a `ClusterCRF` subclass handed to
 * `gecco.cli.main(crf_type=...)`, gecco/cli/commands/__init__.py:127-137) is shown in
 * INTEGRATION.md and implemented in gecco_amd/crf.py.
 *
 * Conventions: plain pointers and sizes; every function returns an int status (0 = OK,
 * <0 = error, text via gecco_crf_last_error()); caller owns all buffers.  Model handles are
 * immutable after creation and may be shared between threads; a plan may be launched from several
 * threads (its lazily created work space is guarded), but concurrent launches of ONE plan share that
 * work space and must therefore go to one stream; sessions serialise their batches.  Batches are CSR:
 *   contig_ptr[n_contigs+1]  gene offsets of each contig          (host memory, always)
 *   gene_ptr  [n_genes+1]    attribute offsets of each gene
 *   attr_id   [nnz]          attribute ids; ids outside [0, num_attrs) count as unknown attributes
 *                            and carry no weight, as names CRFsuite does not know do
 * Genes are in the reference's order: sorted by (contig id, start) -- crf/__init__.py:199.
 * There is NO CPU fallback in this library: without a HIP device the compute entry points
 * return GECCO_CRF_ENODEV.
 */
#ifndef GECCO_CRF_H
#define GECCO_CRF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GECCO_CRF_OK 0
#define GECCO_CRF_EINVAL (-1)       /* bad argument; window errors mirror _meta.py:127-130 */
#define GECCO_CRF_EFORMAT (-2)      /* malformed CRFsuite model blob */
#define GECCO_CRF_ENOMEM (-3)
#define GECCO_CRF_EHIP (-4)         /* HIP runtime error */
#define GECCO_CRF_ENODEV (-5)       /* no usable HIP device */
#define GECCO_CRF_EUNSUPPORTED (-6) /* model/window shape outside every kernel's range */

typedef struct gecco_crf_model gecco_crf_model;
typedef struct gecco_crf_plan gecco_crf_plan;

/* Thread-local description of the last error returned on this thread. */
const char *gecco_crf_last_error(void);
/* ABI version: major*100 + minor*10 + patch (2.2.1 = 221). */
int gecco_crf_version(void);

/* ---- model (replaces [EXT] pycrfsuite.Tagger.open / labels() / info(); the blob is the
 * `__FILE_RESOURCE_DATA__` bytes held by the pickle that ClusterCRF.trained() loads,
 * gecco/crf/__init__.py:61-99) ------------------------------------------------------- */
int gecco_crf_model_load(const uint8_t *lcrf, size_t n_bytes, gecco_crf_model **out);
/* Model from dense tables (row-major state[A][L], trans[L][L]); names are "0".."L-1" and
 * "a<id>".  Used for synthetic benchmark models. */
int gecco_crf_model_from_tables(const double *state, const double *trans, int32_t num_attrs,
                                int32_t num_labels, gecco_crf_model **out);
void gecco_crf_model_free(gecco_crf_model *m);
int32_t gecco_crf_model_num_labels(const gecco_crf_model *m);
int32_t gecco_crf_model_num_attrs(const gecco_crf_model *m);
int32_t gecco_crf_model_num_features(const gecco_crf_model *m); /* state + transition */
const char *gecco_crf_model_label_name(const gecco_crf_model *m, int32_t id); /* NULL if out of range */
const char *gecco_crf_model_attr_name(const gecco_crf_model *m, int32_t id);
int32_t gecco_crf_model_label_id(const gecco_crf_model *m, const char *name); /* -1 if unknown */
int32_t gecco_crf_model_attr_id(const gecco_crf_model *m, const char *name);  /* -1 if unknown */
/* Bulk name->id (ids[i] = -1 for names the model does not know). */
int gecco_crf_model_map_attrs(const gecco_crf_model *m, const char *const *names, int32_t n, int32_t *ids);
/* Dense weights; `present` (may be NULL) flags which entries are actual model features
 * ([EXT] CRF.state_features_ / transition_features_ list only those). */
int gecco_crf_model_state_weights(const gecco_crf_model *m, double *w /* A*L */, uint8_t *present /* A*L */);
int gecco_crf_model_trans_weights(const gecco_crf_model *m, double *w /* L*L */, uint8_t *present /* L*L */);

/* ---- devices ---------------------------------------------------------------------- */
int gecco_crf_device_count(int32_t *n);

/* ---- one-shot, host buffers in / host buffers out, synchronous ----------------------
 * Row W (+D,S,A/B,P): p_out[g] = max over sliding windows covering gene g of
 * P(y_g = label) from an independent forward-backward on each window of `window` genes
 * (gecco/crf/__init__.py:209-258).  Contigs shorter than the window are centre-padded
 * with empty genes when pad != 0, else skipped and their genes get NaN ("no prediction",
 * :228-234,246-248).  Genes covered by no window (step > 1) get 0.0 (:251). */
int gecco_crf_windowed_marginals(const gecco_crf_model *m, int32_t device,
                                 const int32_t *contig_ptr, int32_t n_contigs,
                                 const int32_t *gene_ptr, const int32_t *attr_id,
                                 int32_t window, int32_t step, int32_t label, int32_t pad,
                                 double *p_out /* n_genes */);
/* Row F (extension; [EXT] CRF.predict_marginals_single on a whole contig): marginals of
 * every label, marg[n_genes][L]; lognorm[n_contigs] may be NULL. */
int gecco_crf_marginals_full(const gecco_crf_model *m, int32_t device,
                             const int32_t *contig_ptr, int32_t n_contigs,
                             const int32_t *gene_ptr, const int32_t *attr_id,
                             double *marg, double *lognorm);
/* Row V (extension; [EXT] CRF.predict_single / crf1dc_viterbi): best label path per
 * contig, first-argmax tie-breaking; score[n_contigs] may be NULL. */
int gecco_crf_viterbi(const gecco_crf_model *m, int32_t device,
                      const int32_t *contig_ptr, int32_t n_contigs,
                      const int32_t *gene_ptr, const int32_t *attr_id,
                      int8_t *y_out /* n_genes */, double *score);
/* Row R (gecco/refine.py:51-64,118-200, criterion "gecco"): threshold run-length
 * segmentation with the stateful grouper, optional trimming of un-annotated edge genes and
 * validation.  seg_out rows = (contig, cluster_number, first_gene, last_gene_exclusive);
 * *n_seg receives the number of rows; GECCO_CRF_EINVAL if max_seg is too small.
 * carry_state: the grouper lives for one `iter_clusters` call (refine.py:186).  0 = one call per
 * contig, what the CLI does (cli/commands/_common.py:621-623): every contig starts "out".
 * 1 = one call over all contigs: a contig that starts with genes without probability inherits the
 * state the previous contig ended in. */
int gecco_crf_segment(int32_t device, const double *p, const uint8_t *annotated,
                      const int32_t *contig_ptr, int32_t n_contigs,
                      double threshold, int32_t n_cds, int32_t edge_distance, int32_t trim,
                      int32_t carry_state, int32_t *seg_out, int32_t max_seg, int32_t *n_seg);

/* ClusterRefiner's parameters (gecco/refine.py:75-116) for the *_ex entry points.  criterion 0 = "gecco"
 * (:142-156: annotated genes, and genes away from the contig edges, >= n_cds); 1 = "antismash" (:157-163: mean
 * probability of the member genes >= average_threshold, distinct marker domains among ALL their domains >=
 * n_biopfams, member genes >= n_cds).  The marker domains (the reference's BIO_PFAMS list, refine.py:19-27) come
 * as a CSR over the batch's genes: marker_ptr[n_genes+1], marker_id[...] in [0, 256) = index of the domain in the
 * caller's marker list; only read when criterion == 1 (gecco_crf_segment_ex wants marker_ptr[0] == 0; the batch driver and
 * the plan take any non-decreasing offsets, like gene_ptr).  (The mean is the left-to-right sum over the count;
 * numpy.mean's own last bit depends on the SIMD width numpy dispatches to, so the reference does not pin it.) */
typedef struct {
    double threshold;         /* 0.8 */
    double average_threshold; /* 0.6 */
    int32_t criterion;        /* 0 */
    int32_t n_cds;            /* 5 */
    int32_t n_biopfams;       /* 5 */
    int32_t edge_distance;    /* 0 */
    int32_t trim;             /* 1 */
    int32_t carry_state;      /* see gecco_crf_segment; the batch driver always works per contig */
    const int32_t *marker_ptr;
    const int32_t *marker_id;
} gecco_crf_refine_params;
int gecco_crf_segment_ex(int32_t device, const double *p, const uint8_t *annotated,
                         const int32_t *contig_ptr, int32_t n_contigs, const gecco_crf_refine_params *params,
                         int32_t *seg_out, int32_t max_seg, int32_t *n_seg);

/* Weighted domain composition of called clusters (gecco/model.py:458-503
 * `Cluster.domain_composition(all_possible, normalize)`, assembled per cluster for the type
 * classifier at gecco/types/__init__.py:118).  seg rows as written by gecco_crf_segment
 * (only first_gene / last_gene_exclusive are read); dom_ptr[n_genes+1] = CSR of the domain
 * rows of every gene in the order of `gene.protein.domains`; dom_col[row] = index of the
 * domain's name in all_possible or -1; dom_weight[row] = 1 - pvalue (or -log10, the caller's
 * choice as in the reference).  comp_out[n_seg][n_cols], every entry numpy.sum of the matching
 * weights and every row divided by `row.sum() or 1` when normalize != 0: bit-identical to
 * numpy's pairwise summation. */
int gecco_crf_domain_composition(int32_t device, const int32_t *seg, int32_t n_seg,
                                 const int32_t *dom_ptr, int32_t n_genes,
                                 const int32_t *dom_col, const double *dom_weight,
                                 int32_t n_cols, int32_t normalize, double *comp_out);

/* ---- resident / asynchronous API ----------------------------------------------------
 * A plan owns the device copies of the model tables and of the contig layout of one
 * batch; bulk arrays stay in caller-owned DEVICE memory and launches go to the caller's
 * stream (`stream` is a hipStream_t passed as void*; NULL = the default stream).  This is
 * what bench.py and multi-GPU drivers use: one plan per rank/shard, no collectives. */
int gecco_crf_plan_create(const gecco_crf_model *m, int32_t device,
                          const int32_t *contig_ptr /* host */, int32_t n_contigs,
                          int32_t window, int32_t step, int32_t pad, gecco_crf_plan **out);
void gecco_crf_plan_free(gecco_crf_plan *p);
int32_t gecco_crf_plan_num_genes(const gecco_crf_plan *p);
int64_t gecco_crf_plan_num_windows(const gecco_crf_plan *p); /* = the reference's progress `total` */
int32_t gecco_crf_plan_num_tiles(const gecco_crf_plan *p);   /* workgroups of the windowed kernel */
int32_t gecco_crf_plan_tile_out(const gecco_crf_plan *p);    /* output slots per workgroup of the windowed kernel */
/* Name of the kernel variant the plan dispatches to (for profiles / bench). */
const char *gecco_crf_plan_kernel_name(const gecco_crf_plan *p);
int gecco_crf_plan_run_windowed(gecco_crf_plan *p, const int32_t *d_gene_ptr, const int32_t *d_attr_id,
                                int32_t label, double *d_p_out, void *stream);
int gecco_crf_plan_run_marginals_full(gecco_crf_plan *p, const int32_t *d_gene_ptr, const int32_t *d_attr_id,
                                      double *d_marg, double *d_lognorm, void *stream);
int gecco_crf_plan_run_viterbi(gecco_crf_plan *p, const int32_t *d_gene_ptr, const int32_t *d_attr_id,
                               int8_t *d_y, double *d_score, void *stream);
/* Windowed marginals (gecco/crf/__init__.py:244-258) and whole-contig Viterbi ([EXT]
 * CRF.predict_single) of the same batch in one pass over the CSR: the state scores
 * ([EXT] crf1dt_state_score) are accumulated once and shared.  Same outputs, bit for bit, as
 * gecco_crf_plan_run_windowed followed by gecco_crf_plan_run_viterbi; d_score may be NULL. */
int gecco_crf_plan_run_decode(gecco_crf_plan *p, const int32_t *d_gene_ptr, const int32_t *d_attr_id,
                              int32_t label, double *d_p_out, int8_t *d_y, double *d_score, void *stream);
/* The same decode, software-pipelined over a sequence of batches (throughput form): call k enqueues the windowed marginals
 * of batch k (plan p, its CSR arrays, d_p_out) and the Viterbi labels of batch k - 1 (plan `prev`, the plan of the
 * previous call -- it may be the same plan --, labels to d_prev_y) -- in ONE launch when both qualify (2-label model,
 * window 20, no contig longer than 2048 genes): the Viterbi workgroups, which leave the CUs idle when they run alone,
 * run under the window tiles of the next batch.  First call: prev = NULL.  Last call: p = NULL (labels of the last batch
 * only), so K batches take K + 1 calls.  Outputs are the bits of gecco_crf_plan_run_decode.  The caller keeps the CSR
 * arrays of a batch alive until the call that delivers its labels has been enqueued, and does not use `prev` for other
 * whole-contig calls in between (they would recompute the state scores; results stay correct). */
int gecco_crf_plan_run_decode_pipelined(gecco_crf_plan *p, const int32_t *d_gene_ptr, const int32_t *d_attr_id,
                                        int32_t label, double *d_p_out, gecco_crf_plan *prev, int8_t *d_prev_y,
                                        void *stream);
/* Row R chained behind the marginals, on the same stream, without moving them: d_p (e.g. the output
 * of gecco_crf_plan_run_windowed) and d_annotated are device arrays over the plan's genes; rows go
 * to d_seg[max_seg][4] and their number to *d_n_seg, both device-accessible (device memory, or
 * memory from gecco_crf_host_alloc).  The reference runs this right behind predict_probabilities
 * (cli/commands/_common.py:595-625 -> refine.py:118-200). */
int gecco_crf_plan_run_segment(gecco_crf_plan *p, const double *d_p, const uint8_t *d_annotated,
                               double threshold, int32_t n_cds, int32_t edge_distance, int32_t trim,
                               int32_t carry_state, int32_t *d_seg, int32_t max_seg, int32_t *d_n_seg,
                               void *stream);
/* Same with the full parameter set; marker_ptr / marker_id are DEVICE arrays over the plan's genes. */
int gecco_crf_plan_run_segment_ex(gecco_crf_plan *p, const double *d_p, const uint8_t *d_annotated,
                                  const gecco_crf_refine_params *params, int32_t *d_seg, int32_t max_seg,
                                  int32_t *d_n_seg, void *stream);
/* Average milliseconds per launch of `iters` back-to-back windowed launches, measured with
 * HIP events on `stream` (after `warmup` untimed launches). */
int gecco_crf_plan_time_windowed(gecco_crf_plan *p, const int32_t *d_gene_ptr, const int32_t *d_attr_id,
                                 int32_t label, double *d_p_out, void *stream,
                                 int32_t warmup, int32_t iters, float *ms_per_launch);
/* The same for the pipelined decode launch (the plan following itself: window tiles of the batch + Viterbi workgroups of
 * the batch before, one launch per iteration); the interval includes the boundaries between the launches. */
int gecco_crf_plan_time_decode_pipelined(gecco_crf_plan *p, const int32_t *d_gene_ptr, const int32_t *d_attr_id,
                                         int32_t label, double *d_p_out, int8_t *d_y, void *stream, int32_t warmup,
                                         int32_t iters, float *ms_per_launch);
/* What the 2-label Viterbi decoder of this plan has met since the last call with reset != 0 (waits for the device):
 * out[0] decisions inside the coarse margin of their threshold (candidates), out[1] decisions inside the margin in which the
 * difference form is not provably [EXT] crf1dc_viterbi's, out[2] contigs decoded again with CRFsuite's own recursion
 * because of those, out[3] the genes of these contigs.  All zero for models the any-label kernels serve. */
int gecco_crf_plan_viterbi_stats(gecco_crf_plan *p, int64_t out[4], int32_t reset);

/* ---- batch driver: host buffers in, host buffers out, one or several devices ----------------
 * What `gecco run` reaches through ClusterCRF.predict_probabilities (gecco/crf/__init__.py:244-258:
 * one loop iteration per contig, nothing shared).  A session owns, per device, a ring of lanes
 * (stream + reusable plan + device buffers); a batch is cut into chunks at contig boundaries, the
 * chunks are dealt to the devices longest-first by gene count (per-GPU queues, no collective), and
 * chunk k+1 is uploaded while chunk k computes and chunk k-1 is downloaded.  Host buffers from
 * gecco_crf_host_alloc (pinned) make every copy asynchronous; any other host memory works too.
 * Nothing is allocated once a session has seen its largest chunk.  One batch at a time per session
 * (calls are serialised); free a session before its model.  The one-shot entry points above run on
 * a per-device session owned by the model. */
typedef struct gecco_crf_session gecco_crf_session;
int gecco_crf_host_alloc(size_t n_bytes, void **out);
void gecco_crf_host_free(void *p);
int gecco_crf_session_create(const gecco_crf_model *m, const int32_t *devices, int32_t n_devices,
                             gecco_crf_session **out);
void gecco_crf_session_free(gecco_crf_session *s);
int gecco_crf_session_set_chunk_genes(gecco_crf_session *s, int32_t genes); /* default 2^19 */
/* Small batches -- what `gecco run` on ONE genome hands over: a contig of a few dozen genes, BASELINE.json configs[0],
 * /root/reference/tests/test_cli/test_run.py:35-70 -- take the DIRECT path: the batch is one chunk, its arrays are read by the
 * kernels from pinned host memory (the caller's own buffers when they come from gecco_crf_host_alloc and are large enough for
 * that to matter, a pinned staging copy otherwise) and its outputs are written there; no copy command, no second stream, the
 * call is its launches and one wait (input arrays of 256 KB and more are copied to device memory by a copy command on that same
 * stream).  `genes` = largest batch that takes it (half of that for cluster calls; 0 = never; never more than the chunk size;
 * -1 = the defaults: every batch that is one chunk anyway -- 2^19 genes -- on a one-device session, 131072 on a session over
 * several devices, 65536 for cluster calls).  Same output bits as the chunked path up to the last ulp of the 0.3 % of windows
 * that take the max-normalised form (reference-bits mode: the same bits).  Whole-contig marginals and path
 * scores always take the chunked path. */
int gecco_crf_session_set_direct_genes(gecco_crf_session *s, int32_t genes);
/* Figures of the last batch (any pointer may be NULL). */
int gecco_crf_session_stats(const gecco_crf_session *s, int32_t *n_chunks, int64_t *h2d_bytes,
                            int64_t *d2h_bytes, double *host_plan_seconds, double *wall_seconds);
/* REFERENCE-BITS MODE.  The fast kernels reorganise CRFsuite's arithmetic; their probabilities lie within a few ulps of the
 * reference's (<= 13 on the BGC0001866 fixture) -- enough for bit-identical cluster calls, not for the reference's own acceptance
 * test, which compares whole output files (/root/reference/galaxy/gecco.xml:83-111; probabilities are printed with 16-17
 * digits).  With this switch on, the session's windowed marginals -- and so its cluster calls and the p of its decode calls --
 * are computed in CRFsuite's OWN operation order ([EXT] crf1dc_exp_state / alpha_score / beta_score / marginal_point as
 * restated in oracle/crf_oracle.c) with a CORRECTLY ROUNDED exp in place of libm's.  What that is and is not: bit-identical to
 * the oracle run with a correctly rounded exp (libquadmath) on every gene, string-identical to the reference's fixture files
 * (85 of 85 float cells); against the oracle run with the host's glibc exp -- what CRFsuite calls -- 2 881 of 1 999 989 genes of
 * the C3 benchmark batch differ, by at most 14 ulps (glibc's exp is not correctly rounded on every argument, and which ones
 * depends on its version and the CPU's FMA path); the fast kernels differ on 89 % of the genes, by at most 64 ulps.  About five
 * times the fast window kernel's time.  2-label models, windows of at most 32 genes (GECCO_CRF_EUNSUPPORTED otherwise).
 * GECCO_CRF_REFERENCE_BITS=1 (read once per process) switches it on for the windowed marginals of every 2-label session and plan
 * whose window it covers; other layouts keep their kernels. */
int gecco_crf_session_set_reference_bits(gecco_crf_session *s, int32_t on);
/* The exp that mode uses, on the host (a double-double evaluation; same code as the device's): out[i] = the double nearest to
 * exp(x[i]). */
int gecco_crf_exp_correctly_rounded(const double *x, int64_t n, double *out);
/* The same and more as one struct: whether the batch took the direct path, and the time the submitting host threads spent
 * issuing work (HIP API calls + chunk layouts), which is what bounds a session over many devices. */
typedef struct {
    int32_t n_chunks, n_devices, direct, host_threads;
    int64_t h2d_bytes, d2h_bytes;
    double host_plan_seconds, host_issue_seconds, wall_seconds;
} gecco_crf_session_stats_t;
int gecco_crf_session_stats_ex(const gecco_crf_session *s, gecco_crf_session_stats_t *out);
/* = gecco_crf_windowed_marginals over the session's devices. */
int gecco_crf_session_windowed(gecco_crf_session *s, const int32_t *contig_ptr, int32_t n_contigs,
                               const int32_t *gene_ptr, const int32_t *attr_id, int32_t window,
                               int32_t step, int32_t label, int32_t pad, double *p_out);
/* gecco_crf_session_windowed with a lighter wire format: `degree[i]` = gene_ptr[i + 1] - gene_ptr[i] as ONE BYTE per gene
 * (the caller guarantees the equality and that no gene has more than 255 domains; the packers know the counts anyway).
 * The degrees cross PCIe instead of the row pointers (2 instead of 8 MB per 2 M genes: a third of the upload); the row
 * pointers are rebuilt on the device by a prefix sum.  gene_ptr is still passed -- host memory, read at chunk boundaries
 * only.  Same output bits. */
int gecco_crf_session_windowed_degrees(gecco_crf_session *s, const int32_t *contig_ptr, int32_t n_contigs,
                                       const int32_t *gene_ptr, const uint8_t *degree, const int32_t *attr_id,
                                       int32_t window, int32_t step, int32_t label, int32_t pad, double *p_out);
/* Windowed marginals + Viterbi labels of the same batch (state scores gathered once). */
int gecco_crf_session_decode(gecco_crf_session *s, const int32_t *contig_ptr, int32_t n_contigs,
                             const int32_t *gene_ptr, const int32_t *attr_id, int32_t window,
                             int32_t step, int32_t label, int32_t pad, double *p_out, int8_t *y_out);
/* gecco_crf_session_windowed (y_out NULL) / gecco_crf_session_decode with the compact wire format: `degree` (or NULL) =
 * the genes' domain counts as bytes, sent instead of the row pointers (gecco_crf_session_windowed_degrees); `attr_id16`
 * (or NULL) = the attribute indices as 16-bit words, read instead of attr_id, for a model with at most 65536 attributes
 * (EINVAL otherwise).  Same bits out. */
int gecco_crf_session_decode_wire(gecco_crf_session *s, const int32_t *contig_ptr, int32_t n_contigs,
                                  const int32_t *gene_ptr, const uint8_t *degree, const int32_t *attr_id,
                                  const uint16_t *attr_id16, int32_t window, int32_t step, int32_t label, int32_t pad,
                                  double *p_out, int8_t *y_out /* or NULL */);
/* predict_probabilities + ClusterRefiner in one pass, one grouper per contig like the CLI
 * (cli/commands/_common.py:595-625): the probabilities never leave the device unless p_out is given;
 * what comes back is the rows (batch-wide contig / gene indices) and, if seg_p_out is given, the
 * probabilities of the genes of every row (row k at seg_off_out[k] .. seg_off_out[k+1]), which is
 * all a cluster table needs of them (average_p, max_p: gecco/model.py:442-454). */
int gecco_crf_session_clusters(gecco_crf_session *s, const int32_t *contig_ptr, int32_t n_contigs,
                               const int32_t *gene_ptr, const int32_t *attr_id, const uint8_t *annotated,
                               int32_t window, int32_t step, int32_t label, int32_t pad,
                               double threshold, int32_t n_cds, int32_t edge_distance, int32_t trim,
                               double *p_out /* n_genes or NULL */, int32_t *seg_out, int32_t max_seg,
                               int32_t *n_seg, double *seg_p_out /* or NULL */, int64_t max_seg_genes,
                               int64_t *seg_off_out /* max_seg + 1, with seg_p_out */);

/* Same with the full parameter set (host marker arrays over the batch's genes; carry_state is ignored). */
int gecco_crf_session_clusters_ex(gecco_crf_session *s, const int32_t *contig_ptr, int32_t n_contigs,
                                  const int32_t *gene_ptr, const int32_t *attr_id, const uint8_t *annotated,
                                  int32_t window, int32_t step, int32_t label, int32_t pad,
                                  const gecco_crf_refine_params *params, double *p_out, int32_t *seg_out,
                                  int32_t max_seg, int32_t *n_seg, double *seg_p_out, int64_t max_seg_genes,
                                  int64_t *seg_off_out);
/* gecco_crf_session_clusters_ex with the degree-byte wire format of gecco_crf_session_windowed_degrees: `degree` crosses
 * PCIe instead of the row pointers.  Same rows, same probabilities.  With `degree`, `annotated` may be NULL: a gene then
 * counts as annotated iff it carries a domain the model knows (degree > 0), and nothing else is uploaded for it. */
int gecco_crf_session_clusters_degrees(gecco_crf_session *s, const int32_t *contig_ptr, int32_t n_contigs,
                                       const int32_t *gene_ptr, const uint8_t *degree, const int32_t *attr_id,
                                       const uint8_t *annotated, int32_t window, int32_t step, int32_t label, int32_t pad,
                                       const gecco_crf_refine_params *params, double *p_out, int32_t *seg_out,
                                       int32_t max_seg, int32_t *n_seg, double *seg_p_out, int64_t max_seg_genes,
                                       int64_t *seg_off_out);

/* gecco_crf_session_clusters_degrees with one more option of the wire format: `attr_id16` (or NULL), the attribute indices
 * as 16-bit words, read instead of attr_id -- for a model with at most 65536 attributes (GECCO's has 2766; EINVAL
 * otherwise).  2 instead of 4 bytes per domain cross PCIe and are widened on the device.  Same rows, same probabilities. */
int gecco_crf_session_clusters_wire(gecco_crf_session *s, const int32_t *contig_ptr, int32_t n_contigs,
                                    const int32_t *gene_ptr, const uint8_t *degree, const int32_t *attr_id,
                                    const uint16_t *attr_id16, const uint8_t *annotated, int32_t window, int32_t step,
                                    int32_t label, int32_t pad, const gecco_crf_refine_params *params, double *p_out,
                                    int32_t *seg_out, int32_t max_seg, int32_t *n_seg, double *seg_p_out,
                                    int64_t max_seg_genes, int64_t *seg_off_out);

/* ---- columnar host side: table columns -> CSR batch, called clusters -> clusters.tsv rows --------
 * Strings travel as Arrow-style columns: one byte buffer + int64 offsets[n+1] per column (what
 * pandas / polars / pyarrow hold them in).  gecco_crf_pack_columns does, on columns, what the reference
 * does on Gene objects before the tagger sees them: genes by (sequence_id, start) with ties in
 * first-appearance order (sorted() is stable, gecco/crf/__init__.py:199-206), a gene's domains by
 * domain_start (:200-201), repeated names collapsed (crf/features.py:31-35), names the model does not know
 * dropped ([EXT] CRFsuite attribute lookup).  Feature rows: one per domain hit (gecco/model.py:629-642);
 * gene rows (optional, so that genes without any domain are kept): one per gene (:781-789).  The CSR lands
 * in pinned memory when a device is present.  Pointers returned by the accessors live as long as the handle. */
typedef struct {
    const uint8_t *data;
    const int64_t *offsets;
} gecco_crf_strings;
typedef struct {
    int64_t n_rows; /* feature table */
    gecco_crf_strings sequence_id, protein_id, domain;
    const int64_t *start, *domain_start;
    int64_t n_genes; /* gene table, may be 0 */
    gecco_crf_strings gene_sequence_id, gene_protein_id;
    const int64_t *gene_start;
    int64_t n_markers; /* marker domain names for the antismash criterion (<= 256), may be 0 */
    gecco_crf_strings markers;
} gecco_crf_table_columns;
typedef struct gecco_crf_packed gecco_crf_packed;
typedef struct gecco_crf_cluster_rows gecco_crf_cluster_rows;
int gecco_crf_pack_columns(const gecco_crf_model *m, const gecco_crf_table_columns *t, gecco_crf_packed **out);
void gecco_crf_packed_free(gecco_crf_packed *p);
/* Sizes and diagnostics (any pointer may be NULL): duplicated ids in the gene table (the last row of an id
 * stands for it) and proteins of the feature table the gene table does not list -- the reference raises on
 * both (`annotate_genes`, gecco/cli/commands/_common.py), callers that mirror it check these. */
int gecco_crf_packed_info(const gecco_crf_packed *p, int32_t *n_genes, int32_t *n_contigs, int64_t *nnz,
                          int32_t *n_duplicate_gene_ids, int32_t *n_unlisted_proteins, int32_t *pinned);
const int32_t *gecco_crf_packed_contig_ptr(const gecco_crf_packed *p); /* [n_contigs+1] */
const int32_t *gecco_crf_packed_gene_ptr(const gecco_crf_packed *p);   /* [n_genes+1] */
const int32_t *gecco_crf_packed_attr_id(const gecco_crf_packed *p);    /* [nnz] */
const uint8_t *gecco_crf_packed_annotated(const gecco_crf_packed *p);  /* [n_genes] has >= 1 feature row */
const int64_t *gecco_crf_packed_gene_row(const gecco_crf_packed *p);   /* [n_genes] gene-table row, or -1 - first feature row */
const int32_t *gecco_crf_packed_row_gene(const gecco_crf_packed *p);   /* [n_rows] gene position of every feature row */
const int64_t *gecco_crf_packed_row_order(const gecco_crf_packed *p);  /* [n_rows] rows by (gene position, domain_start) */
const int64_t *gecco_crf_packed_row_ptr(const gecco_crf_packed *p);    /* [n_genes+1] */
/* per gene the distinct marker domains among its rows (index into `markers`); NULL without markers */
const int32_t *gecco_crf_packed_marker_ptr(const gecco_crf_packed *p); /* [n_genes+1] */
const int32_t *gecco_crf_packed_marker_id(const gecco_crf_packed *p);
/* Rows of clusters.tsv (gecco/model.py:731-760) for segments as returned by gecco_crf_session_clusters:
 * start / end over the member genes (gene_end: the gene table's `end`, feature_end: the feature table's),
 * average_p = statistics.mean of the members' probabilities, exactly rounded (:442-447), max_p, the sorted
 * protein ids and the sorted domain names joined by ';', "<sequence_id>_cluster_<number>". */
int gecco_crf_cluster_rows_build(const gecco_crf_packed *p, const gecco_crf_table_columns *t, const int64_t *gene_end,
                                 const int64_t *feature_end, const int32_t *seg, int32_t n_seg, const double *seg_p,
                                 const int64_t *seg_off, gecco_crf_cluster_rows **out);
void gecco_crf_cluster_rows_free(gecco_crf_cluster_rows *r);
const int64_t *gecco_crf_cluster_rows_start(const gecco_crf_cluster_rows *r);
const int64_t *gecco_crf_cluster_rows_end(const gecco_crf_cluster_rows *r);
const double *gecco_crf_cluster_rows_average_p(const gecco_crf_cluster_rows *r);
const double *gecco_crf_cluster_rows_max_p(const gecco_crf_cluster_rows *r);
/* which: 0 sequence_id, 1 cluster_id, 2 proteins, 3 domains */
int gecco_crf_cluster_rows_strings(const gecco_crf_cluster_rows *r, int32_t which, const uint8_t **data,
                                   const int64_t **offsets);
/* statistics.mean of the non-NaN values: the exact sum divided by the count, rounded once.  Any finite doubles of either
 * sign; infinite values follow float arithmetic (inf, -inf, NaN for both signs); NaN when there is no value. */
double gecco_crf_exact_mean(const double *v, int64_t n);
/* Output columns of the columnar path, on several host threads (ABI 2.2.1).  gather: out[i] = src[idx[i]] -- the feature
 * table's `cluster_probability` is its genes' probabilities (gecco/crf/features.py:92-96).  order_info, given the gene table's
 * `start` / `end` columns: rows_in_order = the genes in scoring order are the gene table's rows 0 .. n - 1 (the table can be
 * handed back as it is); refiner_order_differs = two genes of a contig share a start with their ends in decreasing order, so the
 * refiner's (start, end) order (gecco/refine.py:190) is not the CRF's (contig, start) order (gecco/crf/__init__.py:199). */
int gecco_crf_gather_f64(const double *src, int64_t n_src, const int32_t *idx, int64_t n, double *out);
int gecco_crf_packed_order_info(const gecco_crf_packed *p, const int64_t *gene_start, const int64_t *gene_end, int64_t n_gene_rows,
                                int32_t *rows_in_order, int32_t *refiner_order_differs);
/* TSV text of a table, the wire format either side of the path (gecco/_base.py:133-152): `header` first, then
 * n_rows lines of tab-separated cells.  kinds[c]: 0 text (data[c] bytes + offsets[c]), 1 int64, 2 float64; floats
 * are written with the shortest digits that round-trip, laid out as Python's repr() does, NaN as an empty field.
 * *out is malloc'ed: release it with gecco_crf_buffer_free. */
int gecco_crf_tsv_format(int64_t n_rows, int32_t n_cols, const int32_t *kinds, const void *const *data,
                         const int64_t *const *offsets, const char *header, uint8_t **out, int64_t *out_len);
void gecco_crf_buffer_free(uint8_t *p);

#ifdef __cplusplus
}
#endif
#endif /* GECCO_CRF_H */
