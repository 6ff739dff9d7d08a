// extern "C" surface declared in include/gecco_crf.h.
#include <cstring>
#include <new>

#include "../../include/gecco_crf.h"
#include "crf_model.hpp"
#include "crf_plan.hpp"

using namespace gecco;

struct gecco_crf_model {
    Model m;
};
struct gecco_crf_plan {
    Plan p;
};

#define GECCO_API extern "C" __attribute__((visibility("default")))

namespace {
struct DeviceGuard {  // restores the caller's current device
    int prev = -1;
    DeviceGuard() {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

template <class T>
struct DevBuf {
    T *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t n, const char *what) {
        return check_hip(hipMalloc(reinterpret_cast<void **>(&p), (n ? n : 1) * sizeof(T)), what);
    }
};

int check_device(int32_t device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (this library has no CPU fallback)");
        return GECCO_CRF_ENODEV;
    }
    if (device < 0 || device >= n) {
        set_error("device index out of range");
        return GECCO_CRF_ENODEV;
    }
    return GECCO_CRF_OK;
}
}  // namespace

GECCO_API const char *gecco_crf_last_error(void) { return last_error(); }
GECCO_API int gecco_crf_version(void) { return 100; }

GECCO_API int gecco_crf_model_load(const uint8_t *lcrf, size_t n_bytes, gecco_crf_model **out) {
    if (!out) return GECCO_CRF_EINVAL;
    *out = nullptr;
    auto *h = new (std::nothrow) gecco_crf_model();
    if (!h) return GECCO_CRF_ENOMEM;
    int rc = parse_lcrf(lcrf, n_bytes, h->m);
    if (rc) {
        delete h;
        return rc;
    }
    *out = h;
    return GECCO_CRF_OK;
}

GECCO_API int gecco_crf_model_from_tables(const double *state, const double *trans, int32_t num_attrs,
                                          int32_t num_labels, gecco_crf_model **out) {
    if (!out) return GECCO_CRF_EINVAL;
    *out = nullptr;
    auto *h = new (std::nothrow) gecco_crf_model();
    if (!h) return GECCO_CRF_ENOMEM;
    int rc = model_from_tables(state, trans, num_attrs, num_labels, h->m);
    if (rc) {
        delete h;
        return rc;
    }
    *out = h;
    return GECCO_CRF_OK;
}

GECCO_API void gecco_crf_model_free(gecco_crf_model *m) { delete m; }
GECCO_API int32_t gecco_crf_model_num_labels(const gecco_crf_model *m) { return m ? m->m.L : 0; }
GECCO_API int32_t gecco_crf_model_num_attrs(const gecco_crf_model *m) { return m ? m->m.A : 0; }
GECCO_API int32_t gecco_crf_model_num_features(const gecco_crf_model *m) { return m ? m->m.n_features : 0; }
GECCO_API const char *gecco_crf_model_label_name(const gecco_crf_model *m, int32_t id) {
    return (m && id >= 0 && id < m->m.L) ? m->m.labels[id].c_str() : nullptr;
}
GECCO_API const char *gecco_crf_model_attr_name(const gecco_crf_model *m, int32_t id) {
    return (m && id >= 0 && id < m->m.A) ? m->m.attrs[id].c_str() : nullptr;
}
GECCO_API int32_t gecco_crf_model_label_id(const gecco_crf_model *m, const char *name) {
    if (!m || !name) return -1;
    auto it = m->m.label_index.find(name);
    return it == m->m.label_index.end() ? -1 : it->second;
}
GECCO_API int32_t gecco_crf_model_attr_id(const gecco_crf_model *m, const char *name) {
    if (!m || !name) return -1;
    auto it = m->m.attr_index.find(name);
    return it == m->m.attr_index.end() ? -1 : it->second;
}
GECCO_API int gecco_crf_model_map_attrs(const gecco_crf_model *m, const char *const *names, int32_t n, int32_t *ids) {
    if (!m || (n > 0 && (!names || !ids))) return GECCO_CRF_EINVAL;
    std::string key;
    for (int32_t i = 0; i < n; ++i) {
        if (!names[i]) {
            ids[i] = -1;
            continue;
        }
        key.assign(names[i]);
        auto it = m->m.attr_index.find(key);
        ids[i] = it == m->m.attr_index.end() ? -1 : it->second;
    }
    return GECCO_CRF_OK;
}
GECCO_API int gecco_crf_model_state_weights(const gecco_crf_model *m, double *w, uint8_t *present) {
    if (!m) return GECCO_CRF_EINVAL;
    if (w) std::memcpy(w, m->m.state.data(), m->m.state.size() * sizeof(double));
    if (present) std::memcpy(present, m->m.state_mask.data(), m->m.state_mask.size());
    return GECCO_CRF_OK;
}
GECCO_API int gecco_crf_model_trans_weights(const gecco_crf_model *m, double *w, uint8_t *present) {
    if (!m) return GECCO_CRF_EINVAL;
    if (w) std::memcpy(w, m->m.trans.data(), m->m.trans.size() * sizeof(double));
    if (present) std::memcpy(present, m->m.trans_mask.data(), m->m.trans_mask.size());
    return GECCO_CRF_OK;
}

GECCO_API int gecco_crf_device_count(int32_t *n) {
    if (!n) return GECCO_CRF_EINVAL;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    *n = (e == hipSuccess) ? c : 0;
    return GECCO_CRF_OK;
}

// ---- plans -----------------------------------------------------------------------------
GECCO_API int gecco_crf_plan_create(const gecco_crf_model *m, int32_t device, const int32_t *contig_ptr,
                                    int32_t n_contigs, int32_t window, int32_t step, int32_t pad,
                                    gecco_crf_plan **out) {
    if (!m || !out) return GECCO_CRF_EINVAL;
    *out = nullptr;
    if (device >= 0) {
        int rc = check_device(device);
        if (rc) return rc;
    }
    DeviceGuard guard;
    auto *h = new (std::nothrow) gecco_crf_plan();
    if (!h) return GECCO_CRF_ENOMEM;
    int rc = plan_build(m->m, device, contig_ptr, n_contigs, window, step, pad, h->p);
    if (rc) {
        delete h;
        return rc;
    }
    *out = h;
    return GECCO_CRF_OK;
}
GECCO_API void gecco_crf_plan_free(gecco_crf_plan *p) {
    DeviceGuard guard;
    delete p;
}
GECCO_API int32_t gecco_crf_plan_num_genes(const gecco_crf_plan *p) { return p ? p->p.n_genes : 0; }
GECCO_API int64_t gecco_crf_plan_num_windows(const gecco_crf_plan *p) { return p ? p->p.n_windows : 0; }
GECCO_API int32_t gecco_crf_plan_num_tiles(const gecco_crf_plan *p) { return p ? p->p.ntiles : 0; }
GECCO_API const char *gecco_crf_plan_kernel_name(const gecco_crf_plan *p) { return p ? p->p.kernel_name.c_str() : ""; }

GECCO_API int gecco_crf_plan_run_windowed(gecco_crf_plan *p, const int32_t *d_gene_ptr, const int32_t *d_attr_id,
                                          int32_t label, double *d_p_out, void *stream) {
    if (!p) return GECCO_CRF_EINVAL;
    DeviceGuard guard;
    return plan_run_windowed(p->p, d_gene_ptr, d_attr_id, label, d_p_out, static_cast<hipStream_t>(stream));
}

GECCO_API int gecco_crf_plan_time_windowed(gecco_crf_plan *p, const int32_t *d_gene_ptr, const int32_t *d_attr_id,
                                           int32_t label, double *d_p_out, void *stream, int32_t warmup,
                                           int32_t iters, float *ms_per_launch) {
    if (!p || !ms_per_launch || iters <= 0) return GECCO_CRF_EINVAL;
    DeviceGuard guard;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc;
    for (int i = 0; i < warmup; ++i)
        if ((rc = plan_run_windowed(p->p, d_gene_ptr, d_attr_id, label, d_p_out, s))) return rc;
    hipEvent_t e0, e1;
    if ((rc = check_hip(hipEventCreate(&e0), "hipEventCreate"))) return rc;
    if ((rc = check_hip(hipEventCreate(&e1), "hipEventCreate"))) return rc;
    rc = check_hip(hipEventRecord(e0, s), "hipEventRecord");
    for (int i = 0; i < iters && !rc; ++i) rc = plan_run_windowed(p->p, d_gene_ptr, d_attr_id, label, d_p_out, s);
    if (!rc) rc = check_hip(hipEventRecord(e1, s), "hipEventRecord");
    if (!rc) rc = check_hip(hipEventSynchronize(e1), "hipEventSynchronize");
    float ms = 0.f;
    if (!rc) rc = check_hip(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime");
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_per_launch = ms / float(iters);
    return rc;
}

GECCO_API int gecco_crf_plan_run_decode(gecco_crf_plan *p, const int32_t *d_gene_ptr, const int32_t *d_attr_id,
                                        int32_t label, double *d_p_out, int8_t *d_y, double *d_score, void *stream) {
    if (!p) return GECCO_CRF_EINVAL;
    DeviceGuard guard;
    return plan_run_decode(p->p, d_gene_ptr, d_attr_id, label, d_p_out, d_y, d_score, static_cast<hipStream_t>(stream));
}
GECCO_API int gecco_crf_plan_run_marginals_full(gecco_crf_plan *p, const int32_t *d_gene_ptr, const int32_t *d_attr_id,
                                                double *d_marg, double *d_lognorm, void *stream) {
    if (!p) return GECCO_CRF_EINVAL;
    DeviceGuard guard;
    return plan_run_marginals_full(p->p, d_gene_ptr, d_attr_id, d_marg, d_lognorm, static_cast<hipStream_t>(stream));
}
GECCO_API int gecco_crf_plan_run_viterbi(gecco_crf_plan *p, const int32_t *d_gene_ptr, const int32_t *d_attr_id,
                                         int8_t *d_y, double *d_score, void *stream) {
    if (!p) return GECCO_CRF_EINVAL;
    DeviceGuard guard;
    return plan_run_viterbi(p->p, d_gene_ptr, d_attr_id, d_y, d_score, static_cast<hipStream_t>(stream));
}

// ---- one-shot host entry points ------------------------------------------------------------
GECCO_API int gecco_crf_windowed_marginals(const gecco_crf_model *m, int32_t device, const int32_t *contig_ptr,
                                           int32_t n_contigs, const int32_t *gene_ptr, const int32_t *attr_id,
                                           int32_t window, int32_t step, int32_t label, int32_t pad, double *p_out) {
    if (!m) return GECCO_CRF_EINVAL;
    // argument errors first, so that they surface even on a box without a GPU
    if (window <= 0) {
        set_error("Window size must be strictly positive");
        return GECCO_CRF_EINVAL;
    }
    if (step <= 0 || step > window) {
        set_error("Window step must be strictly positive and under `window_size`");
        return GECCO_CRF_EINVAL;
    }
    if (label < 0 || label >= m->m.L) {
        set_error("label out of range");
        return GECCO_CRF_EINVAL;
    }
    int rc = check_device(device);
    if (rc) return rc;
    DeviceGuard guard;
    gecco_crf_plan h;
    if ((rc = plan_build(m->m, device, contig_ptr, n_contigs, window, step, pad, h.p))) return rc;
    const int32_t n = h.p.n_genes;
    if (n == 0) return GECCO_CRF_OK;
    if (!gene_ptr || !p_out) {
        set_error("null buffer");
        return GECCO_CRF_EINVAL;
    }
    const size_t nnz = size_t(gene_ptr[n]);
    for (size_t k = 0; k < nnz; ++k)
        if (attr_id[k] < 0 || attr_id[k] >= m->m.A) {
            set_error("attr_id out of range (unknown attributes must be dropped by the packer)");
            return GECCO_CRF_EINVAL;
        }
    DevBuf<int32_t> d_gp, d_at;
    DevBuf<double> d_p;
    if ((rc = d_gp.alloc(size_t(n) + 1, "hipMalloc gene_ptr"))) return rc;
    if ((rc = d_at.alloc(nnz, "hipMalloc attr_id"))) return rc;
    if ((rc = d_p.alloc(size_t(n), "hipMalloc p_out"))) return rc;
    if ((rc = check_hip(hipMemcpy(d_gp.p, gene_ptr, (size_t(n) + 1) * 4, hipMemcpyHostToDevice), "H2D gene_ptr"))) return rc;
    if (nnz && (rc = check_hip(hipMemcpy(d_at.p, attr_id, nnz * 4, hipMemcpyHostToDevice), "H2D attr_id"))) return rc;
    if ((rc = check_hip(hipMemset(d_p.p, 0, size_t(n) * 8), "memset"))) return rc;
    if ((rc = plan_run_windowed(h.p, d_gp.p, d_at.p, label, d_p.p, nullptr))) return rc;
    if ((rc = check_hip(hipMemcpy(p_out, d_p.p, size_t(n) * 8, hipMemcpyDeviceToHost), "D2H p_out"))) return rc;
    return GECCO_CRF_OK;
}

namespace {
// shared front half of the one-shot whole-contig entry points: plan + CSR upload
struct OneShot {
    gecco_crf_plan h;
    DevBuf<int32_t> d_gp, d_at;
    int32_t n = 0;
    int prepare(const gecco_crf_model *m, int32_t device, const int32_t *contig_ptr, int32_t n_contigs,
                const int32_t *gene_ptr, const int32_t *attr_id) {
        if (!m) return GECCO_CRF_EINVAL;
        int rc = check_device(device);
        if (rc) return rc;
        if ((rc = plan_build(m->m, device, contig_ptr, n_contigs, 1, 1, 1, h.p))) return rc;
        n = h.p.n_genes;
        if (n == 0) return GECCO_CRF_OK;
        if (!gene_ptr) {
            set_error("null buffer");
            return GECCO_CRF_EINVAL;
        }
        const size_t nnz = size_t(gene_ptr[n]);
        for (size_t k = 0; k < nnz; ++k)
            if (attr_id[k] < 0 || attr_id[k] >= m->m.A) {
                set_error("attr_id out of range (unknown attributes must be dropped by the packer)");
                return GECCO_CRF_EINVAL;
            }
        if ((rc = d_gp.alloc(size_t(n) + 1, "hipMalloc gene_ptr"))) return rc;
        if ((rc = d_at.alloc(nnz, "hipMalloc attr_id"))) return rc;
        if ((rc = check_hip(hipMemcpy(d_gp.p, gene_ptr, (size_t(n) + 1) * 4, hipMemcpyHostToDevice), "H2D gene_ptr"))) return rc;
        if (nnz && (rc = check_hip(hipMemcpy(d_at.p, attr_id, nnz * 4, hipMemcpyHostToDevice), "H2D attr_id"))) return rc;
        return GECCO_CRF_OK;
    }
};
}  // namespace

GECCO_API int gecco_crf_marginals_full(const gecco_crf_model *m, int32_t device, const int32_t *contig_ptr,
                                       int32_t n_contigs, const int32_t *gene_ptr, const int32_t *attr_id,
                                       double *marg, double *lognorm) {
    DeviceGuard guard;
    OneShot os;
    int rc = os.prepare(m, device, contig_ptr, n_contigs, gene_ptr, attr_id);
    if (rc) return rc;
    if (n_contigs == 0) return GECCO_CRF_OK;
    const size_t n = size_t(os.n), L = size_t(m->m.L);
    DevBuf<double> d_m, d_ln;
    if ((rc = d_m.alloc(n * L, "hipMalloc marginals"))) return rc;
    if ((rc = d_ln.alloc(size_t(n_contigs), "hipMalloc lognorm"))) return rc;
    if ((rc = plan_run_marginals_full(os.h.p, os.d_gp.p, os.d_at.p, d_m.p, d_ln.p, nullptr))) return rc;
    if (n && marg && (rc = check_hip(hipMemcpy(marg, d_m.p, n * L * 8, hipMemcpyDeviceToHost), "D2H marginals"))) return rc;
    if (lognorm && (rc = check_hip(hipMemcpy(lognorm, d_ln.p, size_t(n_contigs) * 8, hipMemcpyDeviceToHost), "D2H lognorm")))
        return rc;
    return check_hip(hipDeviceSynchronize(), "sync");
}

GECCO_API int gecco_crf_viterbi(const gecco_crf_model *m, int32_t device, const int32_t *contig_ptr, int32_t n_contigs,
                                const int32_t *gene_ptr, const int32_t *attr_id, int8_t *y_out, double *score) {
    DeviceGuard guard;
    OneShot os;
    int rc = os.prepare(m, device, contig_ptr, n_contigs, gene_ptr, attr_id);
    if (rc) return rc;
    if (n_contigs == 0) return GECCO_CRF_OK;
    const size_t n = size_t(os.n);
    DevBuf<int8_t> d_y;
    DevBuf<double> d_sc;
    if ((rc = d_y.alloc(n, "hipMalloc labels"))) return rc;
    if (score && (rc = d_sc.alloc(size_t(n_contigs), "hipMalloc scores"))) return rc;
    if ((rc = plan_run_viterbi(os.h.p, os.d_gp.p, os.d_at.p, d_y.p, score ? d_sc.p : nullptr, nullptr))) return rc;
    if (n && y_out && (rc = check_hip(hipMemcpy(y_out, d_y.p, n, hipMemcpyDeviceToHost), "D2H labels"))) return rc;
    if (score && (rc = check_hip(hipMemcpy(score, d_sc.p, size_t(n_contigs) * 8, hipMemcpyDeviceToHost), "D2H scores")))
        return rc;
    return check_hip(hipDeviceSynchronize(), "sync");
}
GECCO_API int gecco_crf_segment(int32_t device, const double *p, const uint8_t *annotated, const int32_t *contig_ptr,
                                int32_t n_contigs, double threshold, int32_t n_cds, int32_t edge_distance,
                                int32_t trim, int32_t *seg_out, int32_t max_seg, int32_t *n_seg) {
    if (!n_seg || n_contigs < 0 || (n_contigs > 0 && (!contig_ptr || !p || !annotated)) || max_seg < 0 ||
        (max_seg > 0 && !seg_out)) {
        set_error("gecco_crf_segment: bad arguments");
        return GECCO_CRF_EINVAL;
    }
    *n_seg = 0;
    int rc = check_device(device);
    if (rc) return rc;
    if (n_contigs == 0) return GECCO_CRF_OK;
    DeviceGuard guard;
    if ((rc = check_hip(hipSetDevice(device), "hipSetDevice"))) return rc;
    const size_t n = size_t(contig_ptr[n_contigs]), nc = size_t(n_contigs);
    DevBuf<double> d_p;
    DevBuf<uint8_t> d_a;
    DevBuf<int32_t> d_c, d_seg, d_work, d_total;
    if ((rc = d_p.alloc(n, "hipMalloc p"))) return rc;
    if ((rc = d_a.alloc(n, "hipMalloc annotated"))) return rc;
    if ((rc = d_c.alloc(nc + 1, "hipMalloc contig_ptr"))) return rc;
    if ((rc = d_seg.alloc(size_t(max_seg) * 4, "hipMalloc segments"))) return rc;
    if ((rc = d_work.alloc(nc * 3 + 4, "hipMalloc work"))) return rc;
    if ((rc = d_total.alloc(1, "hipMalloc total"))) return rc;
    if (n && (rc = check_hip(hipMemcpy(d_p.p, p, n * 8, hipMemcpyHostToDevice), "H2D p"))) return rc;
    if (n && (rc = check_hip(hipMemcpy(d_a.p, annotated, n, hipMemcpyHostToDevice), "H2D annotated"))) return rc;
    if ((rc = check_hip(hipMemcpy(d_c.p, contig_ptr, (nc + 1) * 4, hipMemcpyHostToDevice), "H2D contig_ptr"))) return rc;
    if ((rc = check_hip(launch_segment(d_p.p, d_a.p, d_c.p, n_contigs, threshold, n_cds, edge_distance, trim, d_seg.p, max_seg,
                                       d_work.p, d_total.p, nullptr), "segment launch")))
        return rc;
    int32_t total = 0;
    if ((rc = check_hip(hipMemcpy(&total, d_total.p, 4, hipMemcpyDeviceToHost), "D2H count"))) return rc;
    *n_seg = total;
    if (total > max_seg) {
        set_error("gecco_crf_segment: seg_out too small");
        return GECCO_CRF_EINVAL;
    }
    if (total && (rc = check_hip(hipMemcpy(seg_out, d_seg.p, size_t(total) * 16, hipMemcpyDeviceToHost), "D2H segments")))
        return rc;
    return GECCO_CRF_OK;
}

GECCO_API int gecco_crf_domain_composition(int32_t device, const int32_t *seg, int32_t n_seg, const int32_t *dom_ptr,
                                           int32_t n_genes, const int32_t *dom_col, const double *dom_weight,
                                           int32_t n_cols, int32_t normalize, double *comp_out) {
    if (n_seg < 0 || n_genes < 0 || n_cols < 0 || (n_seg > 0 && (!seg || !dom_ptr || (n_cols > 0 && !comp_out)))) {
        set_error("gecco_crf_domain_composition: bad arguments");
        return GECCO_CRF_EINVAL;
    }
    int rc = check_device(device);
    if (rc) return rc;
    if (n_seg == 0 || n_cols == 0) return GECCO_CRF_OK;
    const size_t rows = size_t(dom_ptr[n_genes]);
    if (rows && (!dom_col || !dom_weight)) {
        set_error("gecco_crf_domain_composition: null domain arrays");
        return GECCO_CRF_EINVAL;
    }
    for (int32_t k = 0; k < n_seg; ++k) {
        const int32_t a = seg[4 * k + 2], b = seg[4 * k + 3];
        if (a < 0 || b < a || b > n_genes) {
            set_error("gecco_crf_domain_composition: segment outside the gene range");
            return GECCO_CRF_EINVAL;
        }
    }
    DeviceGuard guard;
    if ((rc = check_hip(hipSetDevice(device), "hipSetDevice"))) return rc;
    DevBuf<int32_t> d_seg, d_ptr, d_col;
    DevBuf<double> d_w, d_tmp, d_out;
    const size_t out_n = size_t(n_seg) * size_t(n_cols);
    if ((rc = d_seg.alloc(size_t(n_seg) * 4, "hipMalloc segments"))) return rc;
    if ((rc = d_ptr.alloc(size_t(n_genes) + 1, "hipMalloc dom_ptr"))) return rc;
    if ((rc = d_col.alloc(rows, "hipMalloc dom_col"))) return rc;
    if ((rc = d_w.alloc(rows, "hipMalloc dom_weight"))) return rc;
    if ((rc = d_tmp.alloc(rows, "hipMalloc scratch"))) return rc;
    if ((rc = d_out.alloc(out_n, "hipMalloc compositions"))) return rc;
    if ((rc = check_hip(hipMemcpy(d_seg.p, seg, size_t(n_seg) * 16, hipMemcpyHostToDevice), "H2D segments"))) return rc;
    if ((rc = check_hip(hipMemcpy(d_ptr.p, dom_ptr, (size_t(n_genes) + 1) * 4, hipMemcpyHostToDevice), "H2D dom_ptr"))) return rc;
    if (rows && (rc = check_hip(hipMemcpy(d_col.p, dom_col, rows * 4, hipMemcpyHostToDevice), "H2D dom_col"))) return rc;
    if (rows && (rc = check_hip(hipMemcpy(d_w.p, dom_weight, rows * 8, hipMemcpyHostToDevice), "H2D dom_weight"))) return rc;
    if ((rc = check_hip(launch_composition(d_seg.p, n_seg, d_ptr.p, d_col.p, d_w.p, d_tmp.p, n_cols, normalize ? 1 : 0,
                                           d_out.p, nullptr), "composition launch")))
        return rc;
    return check_hip(hipMemcpy(comp_out, d_out.p, out_n * 8, hipMemcpyDeviceToHost), "D2H compositions");
}
