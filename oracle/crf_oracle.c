/*
 * ORACLE / TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the arithmetic on GECCO's CRF hot path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the
 * product (gecco_amd/) never does.
 *
 * The arithmetic itself lives in a third-party dependency that is NOT vendored in
 * /root/reference: sklearn-crfsuite ~=0.5.0 (pyproject.toml:43) -> python-crfsuite
 * -> CRFsuite 0.12 (C).  It is restated here from CRFsuite's published algorithm
 * ([EXT] crf1d_tag.c / crf1d_context.c), keeping its operation order (separate
 * mul/add roundings, libm exp, per-step 1/sum scaling) and anchored on GECCO's own
 * call sites:
 *   gecco/crf/__init__.py:253   predict_marginals_single(feats[win])   (rows S,A/B,P)
 *   gecco/crf/__init__.py:209-258  pad / sliding window / per-gene max    (rows D,W)
 *   gecco/_meta.py:124-132      sliding_window                          (row W)
 *   gecco/refine.py:51-64,118-200  GeneGrouper / ClusterRefiner         (row R)
 * Parity pin: tests/golden/BGC0001866.{features,genes,clusters}.tsv (the reference's
 * tests/test_cli/data fixture) -- tests/test_oracle_golden.py.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off; no FMA contraction so that
 * roundings match a baseline x86-64 build of CRFsuite).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define ORACLE_API __attribute__((visibility("default")))

/* exp of the state scores ([EXT] crf1dc_exp_state).  Mode 0 (default): libm's exp, what CRFsuite calls.  Mode 1: the correctly
 * rounded exp -- libquadmath's expq (113-bit) rounded to double --, the checker of the product's reference-bits mode
 * (gecco_amd/csrc/crf_exact.hip), whose own correctly rounded exp is a double-double evaluation: the two are independent
 * implementations of the same mathematical function, and glibc's exp agrees with both on all but ~0.07 % of arguments. */
#include <quadmath.h>
static int g_exp_mode = 0;
ORACLE_API void oracle_set_exp_mode(int mode) { g_exp_mode = mode; }
static inline double oracle_exp(double x) { return g_exp_mode ? (double)expq((__float128)x) : exp(x); }

/* ---- row S: [EXT] crf1dt_state_score -------------------------------------------
 * state[t][y] += w[a][y] * value, value == 1.0 for GECCO's {name: True} items
 * (gecco/crf/features.py:31-35).  Attributes are visited in item order. */
static void state_scores(const double *w, int L, const int32_t *gene_ptr, const int32_t *attr_id,
                         int g0, int T, int lpad, int n_real, double *state)
{
    for (int t = 0; t < T; ++t) {
        double *s = state + (size_t)t * L;
        for (int y = 0; y < L; ++y) s[y] = 0.0;
        int g = t - lpad;
        if (g < 0 || g >= n_real) continue; /* padding item: {} (crf/__init__.py:227) */
        for (int k = gene_ptr[g0 + g]; k < gene_ptr[g0 + g + 1]; ++k) {
            const double *wa = w + (size_t)attr_id[k] * L;
            for (int y = 0; y < L; ++y) s[y] += wa[y] * 1.0;
        }
    }
}

/* ---- rows A/B/P: [EXT] crf1dc_exp_state, crf1dc_exp_transition, crf1dc_alpha_score,
 * crf1dc_beta_score, crf1dc_marginal_point --------------------------------------- */
typedef struct {
    int T, L;
    double *exp_state, *alpha, *beta, *scale, *row;
} fb_ctx;

static int fb_alloc(fb_ctx *c, int T, int L)
{
    c->T = T; c->L = L;
    c->exp_state = (double *)malloc(sizeof(double) * (size_t)T * L);
    c->alpha = (double *)malloc(sizeof(double) * (size_t)T * L);
    c->beta = (double *)malloc(sizeof(double) * (size_t)T * L);
    c->scale = (double *)malloc(sizeof(double) * (size_t)T);
    c->row = (double *)malloc(sizeof(double) * (size_t)L);
    return (c->exp_state && c->alpha && c->beta && c->scale && c->row) ? 0 : -1;
}
static void fb_free(fb_ctx *c)
{
    free(c->exp_state); free(c->alpha); free(c->beta); free(c->scale); free(c->row);
}

static double forward_backward(fb_ctx *c, const double *state, const double *exp_trans, int T)
{
    const int L = c->L;
    double *a = c->alpha, *b = c->beta, *sc = c->scale, *es = c->exp_state;
    for (size_t i = 0; i < (size_t)T * L; ++i) es[i] = oracle_exp(state[i]);

    /* alpha */
    double sum = 0.0;
    for (int j = 0; j < L; ++j) { a[j] = es[j]; }
    for (int j = 0; j < L; ++j) sum += a[j];
    sc[0] = (sum != 0.) ? 1. / sum : 1.;
    for (int j = 0; j < L; ++j) a[j] *= sc[0];
    for (int t = 1; t < T; ++t) {
        const double *prev = a + (size_t)(t - 1) * L;
        double *cur = a + (size_t)t * L;
        const double *st = es + (size_t)t * L;
        for (int j = 0; j < L; ++j) cur[j] = 0.0;
        for (int i = 0; i < L; ++i) {
            const double *tr = exp_trans + (size_t)i * L;
            for (int j = 0; j < L; ++j) cur[j] += prev[i] * tr[j];
        }
        for (int j = 0; j < L; ++j) cur[j] *= st[j];
        sum = 0.0;
        for (int j = 0; j < L; ++j) sum += cur[j];
        sc[t] = (sum != 0.) ? 1. / sum : 1.;
        for (int j = 0; j < L; ++j) cur[j] *= sc[t];
    }
    /* beta */
    {
        double *cur = b + (size_t)(T - 1) * L;
        for (int i = 0; i < L; ++i) cur[i] = sc[T - 1];
    }
    for (int t = T - 2; t >= 0; --t) {
        double *cur = b + (size_t)t * L;
        const double *next = b + (size_t)(t + 1) * L;
        const double *st = es + (size_t)(t + 1) * L;
        for (int j = 0; j < L; ++j) c->row[j] = next[j];
        for (int j = 0; j < L; ++j) c->row[j] *= st[j];
        for (int i = 0; i < L; ++i) {
            const double *tr = exp_trans + (size_t)i * L;
            double s = 0.0;
            for (int j = 0; j < L; ++j) s += tr[j] * c->row[j];
            cur[i] = s;
        }
        for (int i = 0; i < L; ++i) cur[i] *= sc[t];
    }
    double lognorm = 0.0;
    for (int t = 0; t < T; ++t) lognorm += log(sc[t]);
    return -lognorm;
}

static inline double marginal_point(const fb_ctx *c, int l, int t)
{
    return c->alpha[(size_t)t * c->L + l] * c->beta[(size_t)t * c->L + l] / c->scale[t];
}

ORACLE_API int oracle_marginals_seq(const double *state, const double *trans, int T, int L, double *marg, double *lognorm)
{
    if (T <= 0) { if (lognorm) *lognorm = 0; return 0; }
    fb_ctx c;
    if (fb_alloc(&c, T, L)) return -1;
    double *et = (double *)malloc(sizeof(double) * L * L);
    for (int i = 0; i < L * L; ++i) et[i] = exp(trans[i]);
    double ln = forward_backward(&c, state, et, T);
    for (int t = 0; t < T; ++t)
        for (int l = 0; l < L; ++l) marg[(size_t)t * L + l] = marginal_point(&c, l, t);
    if (lognorm) *lognorm = ln;
    free(et);
    fb_free(&c);
    return 0;
}

/* ---- row V: [EXT] crf1dc_viterbi (log domain, first-argmax tie-breaking) -------- */
ORACLE_API int oracle_viterbi_seq(const double *state, const double *trans, int T, int L, int32_t *labels, double *score)
{
    if (T <= 0) { if (score) *score = 0; return 0; }
    double *delta = (double *)malloc(sizeof(double) * (size_t)T * L);
    int32_t *back = (int32_t *)malloc(sizeof(int32_t) * (size_t)T * L);
    if (!delta || !back) return -1;
    for (int j = 0; j < L; ++j) delta[j] = state[j];
    for (int t = 1; t < T; ++t) {
        const double *prev = delta + (size_t)(t - 1) * L;
        double *cur = delta + (size_t)t * L;
        for (int j = 0; j < L; ++j) {
            double max_score = -DBL_MAX;
            int arg = -1;
            for (int i = 0; i < L; ++i) {
                double s = prev[i] + trans[(size_t)i * L + j];
                if (max_score < s) { max_score = s; arg = i; }
            }
            if (arg >= 0) back[(size_t)t * L + j] = arg;
            cur[j] = max_score + state[(size_t)t * L + j];
        }
    }
    double max_score = -DBL_MAX;
    const double *last = delta + (size_t)(T - 1) * L;
    labels[T - 1] = 0;
    for (int i = 0; i < L; ++i)
        if (max_score < last[i]) { max_score = last[i]; labels[T - 1] = i; }
    for (int t = T - 2; t >= 0; --t) labels[t] = back[(size_t)(t + 1) * L + labels[t + 1]];
    if (score) *score = max_score;
    free(delta); free(back);
    return 0;
}

ORACLE_API void oracle_state_scores(const double *w, int L, const int32_t *gene_ptr, const int32_t *attr_id, int n, double *state)
{
    state_scores(w, L, gene_ptr, attr_id, 0, n, 0, n, state);
}

/* ---- rows D + W: GECCO's pad / sliding-window / per-gene max wrapper --------------
 * gecco/crf/__init__.py:209-258.  Driven exactly like the reference: every window
 * re-accumulates its state scores, exps them, runs alpha and beta and reads W
 * marginals (no cross-window reuse).  Genes of contigs skipped by pad=0 get NaN
 * ("no prediction": the reference returns those Gene objects untouched, :246-248).
 * Genes never covered by a window (step>1) keep 0.0 (numpy.zeros, :251). */
ORACLE_API int oracle_windowed_marginals(const double *w, const double *trans, int A, int L,
                                         const int32_t *contig_ptr, int n_contigs,
                                         const int32_t *gene_ptr, const int32_t *attr_id,
                                         int W, int step, int label, int pad, double *p_out)
{
    (void)A;
    if (W <= 0 || step <= 0 || step > W) return -2; /* _meta.py:127-130 */
    fb_ctx c;
    if (fb_alloc(&c, W, L)) return -1;
    double *et = (double *)malloc(sizeof(double) * L * L);
    double *state = (double *)malloc(sizeof(double) * (size_t)W * L);
    for (int i = 0; i < L * L; ++i) et[i] = exp(trans[i]);
    for (int ci = 0; ci < n_contigs; ++ci) {
        int g0 = contig_ptr[ci], n = contig_ptr[ci + 1] - g0;
        int delta = 0, np = n;
        if (n < W) {
            if (!pad) {
                for (int g = 0; g < n; ++g) p_out[g0 + g] = NAN;
                continue;
            }
            delta = W - n; np = W;
        }
        int lpad = delta / 2;
        double *prob = (double *)calloc((size_t)np, sizeof(double));
        for (int s = 0; s < np + 1 - W; s += step) {
            /* feats[win] -> tagger.set -> state scores of this window only */
            state_scores(w, L, gene_ptr, attr_id, g0, W, lpad - s, n, state);
            forward_backward(&c, state, et, W);
            for (int t = 0; t < W; ++t) {
                double m = marginal_point(&c, label, t);
                /* numpy.maximum propagates NaN; m is never NaN for finite weights */
                if (m > prob[s + t] || m != m) prob[s + t] = m;
            }
        }
        for (int g = 0; g < n; ++g) p_out[g0 + g] = prob[lpad + g];
        free(prob);
    }
    free(state); free(et);
    fb_free(&c);
    return 0;
}

/* ---- row F (extension): whole-contig marginals = predict_marginals_single(all feats) */
ORACLE_API int oracle_full_marginals(const double *w, const double *trans, int A, int L,
                                     const int32_t *contig_ptr, int n_contigs,
                                     const int32_t *gene_ptr, const int32_t *attr_id,
                                     double *marg /* n_genes x L */, double *lognorm /* n_contigs or NULL */)
{
    (void)A;
    for (int ci = 0; ci < n_contigs; ++ci) {
        int g0 = contig_ptr[ci], n = contig_ptr[ci + 1] - g0;
        if (n == 0) { if (lognorm) lognorm[ci] = 0; continue; }
        double *state = (double *)malloc(sizeof(double) * (size_t)n * L);
        state_scores(w, L, gene_ptr, attr_id, g0, n, 0, n, state);
        int rc = oracle_marginals_seq(state, trans, n, L, marg + (size_t)g0 * L, lognorm ? lognorm + ci : NULL);
        free(state);
        if (rc) return rc;
    }
    return 0;
}

/* ---- row V (extension): whole-contig Viterbi = CRF.predict_single(all feats) */
ORACLE_API int oracle_viterbi(const double *w, const double *trans, int A, int L,
                              const int32_t *contig_ptr, int n_contigs,
                              const int32_t *gene_ptr, const int32_t *attr_id,
                              int32_t *labels /* n_genes */, double *score /* n_contigs or NULL */)
{
    (void)A;
    for (int ci = 0; ci < n_contigs; ++ci) {
        int g0 = contig_ptr[ci], n = contig_ptr[ci + 1] - g0;
        if (n == 0) { if (score) score[ci] = 0; continue; }
        double *state = (double *)malloc(sizeof(double) * (size_t)n * L);
        state_scores(w, L, gene_ptr, attr_id, g0, n, 0, n, state);
        int rc = oracle_viterbi_seq(state, trans, n, L, labels + g0, score ? score + ci : NULL);
        free(state);
        if (rc) return rc;
    }
    return 0;
}

/* ---- all host cores ("1 thread and all host cores (OpenMP over contigs)"): contigs are independent
 * (gecco/crf/__init__.py:244), so the loops above run contig ranges in parallel, every thread with its own scratch.
 * Same arithmetic, same outputs.  Ranges of `grain` contigs are handed out dynamically (contig lengths differ). */
#ifdef _OPENMP
#include <omp.h>
#endif
ORACLE_API int oracle_windowed_marginals_omp(const double *w, const double *trans, int A, int L,
                                             const int32_t *contig_ptr, int n_contigs,
                                             const int32_t *gene_ptr, const int32_t *attr_id,
                                             int W, int step, int label, int pad, double *p_out, int threads, int grain)
{
    if (grain < 1) grain = 1;
    const int n_ranges = (n_contigs + grain - 1) / grain;
    int rc_all = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(| : rc_all)
    for (int r = 0; r < n_ranges; ++r) {
        const int c0 = r * grain, c1 = c0 + grain < n_contigs ? c0 + grain : n_contigs;
        rc_all |= oracle_windowed_marginals(w, trans, A, L, contig_ptr + c0, c1 - c0, gene_ptr, attr_id, W, step, label, pad, p_out);
    }
    return rc_all;
}

ORACLE_API int oracle_viterbi_omp(const double *w, const double *trans, int A, int L,
                                  const int32_t *contig_ptr, int n_contigs,
                                  const int32_t *gene_ptr, const int32_t *attr_id,
                                  int32_t *labels, double *score, int threads, int grain)
{
    if (grain < 1) grain = 1;
    const int n_ranges = (n_contigs + grain - 1) / grain;
    int rc_all = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(| : rc_all)
    for (int r = 0; r < n_ranges; ++r) {
        const int c0 = r * grain, c1 = c0 + grain < n_contigs ? c0 + grain : n_contigs;
        rc_all |= oracle_viterbi(w, trans, A, L, contig_ptr + c0, c1 - c0, gene_ptr, attr_id, labels, score ? score + c0 : NULL);
    }
    return rc_all;
}

ORACLE_API int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- row V, difference form: the SPECIFICATION of the device's 2-label label-only Viterbi kernels
 * (gecco_amd/csrc/crf_sequence.hip).  With Delta = delta[1] - delta[0] and d = s[1] - s[0], crf1dc_viterbi's
 * recursion reads  Delta_t = clamp(Delta_{t-1}, lo, hi) + (t11 - t00) + d_t,  lo = t01 - t11, hi = t00 - t10
 * (requires lo <= hi), back-pointers (Delta_{t-1} > hi, Delta_{t-1} > lo), end label Delta_T > 0 (strict:
 * the first arg max).  Algebraically identical to oracle_viterbi; numerically it rounds differently (its
 * quantities stay O(1)), so labels can only differ where a decision lies within rounding noise.  The device
 * reproduces THIS recursion bit for bit for contigs of up to 2048 genes. */
ORACLE_API int oracle_viterbi_delta(const double *w, const double *trans, int A, int L,
                                    const int32_t *contig_ptr, int n_contigs,
                                    const int32_t *gene_ptr, const int32_t *attr_id, int32_t *labels)
{
    (void)A;
    if (L != 2) return -1;
    const double t00 = trans[0], t01 = trans[1], t10 = trans[2], t11 = trans[3];
    const double lo = t01 - t11, hi = t00 - t10, k = t11 - t00;
    if (!(lo <= hi)) return -2;
    for (int ci = 0; ci < n_contigs; ++ci) {
        int g0 = contig_ptr[ci], n = contig_ptr[ci + 1] - g0;
        if (n == 0) continue;
        double *state = (double *)malloc(sizeof(double) * (size_t)n * 2);
        unsigned char *bp = (unsigned char *)malloc((size_t)n);
        state_scores(w, L, gene_ptr, attr_id, g0, n, 0, n, state);
        double D = state[1] - state[0];
        for (int t = 1; t < n; ++t) {
            bp[t] = (unsigned char)((D > hi ? 1 : 0) | (D > lo ? 2 : 0)); /* bit y: predecessor of label y */
            double c = D < lo ? lo : D;
            c = c > hi ? hi : c;
            D = c + (k + (state[2 * t + 1] - state[2 * t]));
        }
        int y = D > 0.0 ? 1 : 0;
        labels[g0 + n - 1] = y;
        for (int t = n - 1; t >= 1; --t) {
            y = (bp[t] >> y) & 1;
            labels[g0 + t - 1] = y;
        }
        free(state); free(bp);
    }
    return 0;
}

/* ---- row R: GeneGrouper + ClusterRefiner, criterion "gecco" ------------------------
 * gecco/refine.py:51-64 (stateful grouper: a gene without probability inherits the
 * previous gene's state; ONE grouper instance spans all contigs of an iter_clusters call, :186),
 * :182-200 (runs of in_cluster genes per contig, numbered from 1 BEFORE filtering),
 * :167-180 (trim unannotated edge genes), :139-157 (validate: #annotated >= n_cds and
 * #(cluster genes - edge genes) >= n_cds, edge genes = first/last `edge_distance`
 * ANNOTATED genes of the contig).
 * Genes are expected in the reference's order (contig, then (start,end)).
 * Output rows: (contig, cluster_number, first_gene, last_gene_exclusive) in global
 * gene indices, after trimming; returns number of clusters kept, or <0 on overflow. */
ORACLE_API int oracle_segment(const double *p, const uint8_t *annotated,
                              const int32_t *contig_ptr, int n_contigs,
                              double threshold, int n_cds, int edge_distance, int trim,
                              int carry_state, int32_t *seg_out, int max_seg)
{
    int kept = 0;
    int in_cluster = 0; /* one grouper per iter_clusters call (refine.py:186) */
    for (int ci = 0; ci < n_contigs; ++ci) {
        int g0 = contig_ptr[ci], g1 = contig_ptr[ci + 1];
        /* carry_state = 0: one iter_clusters call per contig, as the CLI does
         * (cli/commands/_common.py:621-623) -> a fresh grouper; 1: one call over all contigs */
        if (!carry_state) in_cluster = 0;
        int number = 0;
        int g = g0;
        /* edge genes: indices of annotated genes near both ends */
        int n_ann = 0;
        for (int k = g0; k < g1; ++k) n_ann += annotated[k] ? 1 : 0;
        while (g < g1) {
            /* key of this gene */
            if (p[g] == p[g]) in_cluster = p[g] > threshold;
            int key = in_cluster;
            int h = g + 1;
            while (h < g1) {
                int k2;
                if (p[h] == p[h]) k2 = p[h] > threshold; else k2 = in_cluster;
                if (k2 != key) break;
                in_cluster = k2;
                ++h;
            }
            if (key) {
                ++number;
                int a = g, b = h;
                if (trim) {
                    while (a < b && !annotated[a]) ++a;
                    while (b > a && !annotated[b - 1]) --b;
                }
                int ann = 0, non_edge = 0;
                /* rank of annotated genes inside the contig to decide edge membership */
                int rank = 0;
                for (int k = g0; k < a; ++k) rank += annotated[k] ? 1 : 0;
                for (int k = a; k < b; ++k) {
                    int is_edge = 0;
                    if (annotated[k]) {
                        ++ann;
                        if (edge_distance > 0 && (rank < edge_distance || rank >= n_ann - edge_distance)) is_edge = 1;
                        ++rank;
                    }
                    if (!is_edge) ++non_edge;
                }
                if (ann >= n_cds && non_edge >= n_cds) {
                    if (kept >= max_seg) return -1;
                    seg_out[4 * kept + 0] = ci;
                    seg_out[4 * kept + 1] = number;
                    seg_out[4 * kept + 2] = a;
                    seg_out[4 * kept + 3] = b;
                    ++kept;
                }
            }
            g = h;
        }
    }
    return kept;
}

/* The same walk with the "antismash" validation of refine.py:157-163: mean probability of the (trimmed) run's genes
 * >= average_threshold, distinct marker domains (the reference's BIO_PFAMS) among ALL domains of those genes >=
 * n_biopfams, number of genes >= n_cds.  marker_ptr / marker_id: CSR over genes of marker indices (< 256).  The mean
 * is the left-to-right sum over the count (numpy.mean's last bit depends on numpy's SIMD dispatch: not pinned). */
ORACLE_API int oracle_segment_antismash(const double *p, const uint8_t *annotated, const int32_t *contig_ptr, int n_contigs,
                                        const int32_t *marker_ptr, const int32_t *marker_id, double threshold, int n_cds,
                                        int n_biopfams, double average_threshold, int trim, int carry_state,
                                        int32_t *seg_out, int max_seg)
{
    int kept = 0;
    int in_cluster = 0;
    for (int ci = 0; ci < n_contigs; ++ci) {
        int g0 = contig_ptr[ci], g1 = contig_ptr[ci + 1];
        if (!carry_state) in_cluster = 0;
        int number = 0;
        int g = g0;
        while (g < g1) {
            if (p[g] == p[g]) in_cluster = p[g] > threshold;
            int key = in_cluster;
            int h = g + 1;
            while (h < g1) {
                int k2;
                if (p[h] == p[h]) k2 = p[h] > threshold; else k2 = in_cluster;
                if (k2 != key) break;
                in_cluster = k2;
                ++h;
            }
            if (key) {
                ++number;
                int a = g, b = h;
                if (trim) {
                    while (a < b && !annotated[a]) ++a;
                    while (b > a && !annotated[b - 1]) --b;
                }
                double sum = 0.0;
                uint8_t seen[256];
                memset(seen, 0, sizeof seen);
                int markers = 0;
                for (int k = a; k < b; ++k) {
                    sum += p[k];
                    for (int q = marker_ptr[k]; q < marker_ptr[k + 1]; ++q) {
                        int id = marker_id[q];
                        if (id >= 0 && id < 256 && !seen[id]) {
                            seen[id] = 1;
                            ++markers;
                        }
                    }
                }
                /* an empty run has mean NaN in the reference: every comparison fails */
                if (b > a && sum / (double)(b - a) >= average_threshold && markers >= n_biopfams && b - a >= n_cds) {
                    if (kept >= max_seg) return -1;
                    seg_out[4 * kept + 0] = ci;
                    seg_out[4 * kept + 1] = number;
                    seg_out[4 * kept + 2] = a;
                    seg_out[4 * kept + 3] = b;
                    ++kept;
                }
            }
            g = h;
        }
    }
    return kept;
}
