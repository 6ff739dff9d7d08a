// extern "C" surface declared in include/gecco_crf.h.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <new>
#include <string>

#include "../../include/gecco_crf.h"
#include "crf_exact_exp.hpp"
#include "crf_model.hpp"
#include "crf_plan.hpp"
#include "crf_session.hpp"
#include "crf_tables.hpp"

using namespace gecco;

struct gecco_crf_model {
    Model m;
};
struct gecco_crf_plan {
    Plan p;
};
struct gecco_crf_packed {
    Packed p;
};
struct gecco_crf_cluster_rows {
    ClusterRows r;
};
struct gecco_crf_session {
    Session *s = nullptr;
    ~gecco_crf_session() { session_destroy(s); }
};

// No C++ exception may cross the C boundary (ctypes would std::terminate the interpreter).
#define GECCO_GUARD_BEGIN try {
#define GECCO_GUARD_END                                                     \
    }                                                                       \
    catch (const std::bad_alloc &) {                                        \
        set_error("out of host memory");                                    \
        return GECCO_CRF_ENOMEM;                                            \
    }                                                                       \
    catch (const std::exception &e) {                                       \
        set_error(std::string("internal error: ") + e.what());              \
        return GECCO_CRF_EHIP;                                              \
    }                                                                       \
    catch (...) {                                                           \
        set_error("internal error");                                        \
        return GECCO_CRF_EHIP;                                              \
    }

#define GECCO_API extern "C" __attribute__((visibility("default")))

namespace {
struct DeviceGuard {  // restores the caller's current device
    int prev = -1;
    DeviceGuard() {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && (hipGetDevice(&cur) != hipSuccess || cur != prev)) (void)hipSetDevice(prev);
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
GECCO_API int gecco_crf_version(void) { return 221; }

GECCO_API int gecco_crf_model_load(const uint8_t *lcrf, size_t n_bytes, gecco_crf_model **out) {
    if (!out) return GECCO_CRF_EINVAL;
    *out = nullptr;
    GECCO_GUARD_BEGIN
    std::unique_ptr<gecco_crf_model> h(new gecco_crf_model());
    int rc = parse_lcrf(lcrf, n_bytes, h->m);
    if (rc) return rc;
    *out = h.release();
    return GECCO_CRF_OK;
    GECCO_GUARD_END
}

GECCO_API int gecco_crf_model_from_tables(const double *state, const double *trans, int32_t num_attrs,
                                          int32_t num_labels, gecco_crf_model **out) {
    if (!out) return GECCO_CRF_EINVAL;
    *out = nullptr;
    GECCO_GUARD_BEGIN
    std::unique_ptr<gecco_crf_model> h(new gecco_crf_model());
    int rc = model_from_tables(state, trans, num_attrs, num_labels, h->m);
    if (rc) return rc;
    *out = h.release();
    return GECCO_CRF_OK;
    GECCO_GUARD_END
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
    GECCO_GUARD_BEGIN
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
    GECCO_GUARD_END
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
    if (n_contigs > 0 && contig_ptr && contig_ptr[0] != 0) {
        set_error("contig_ptr[0] must be 0");
        return GECCO_CRF_EINVAL;
    }
    DeviceGuard guard;
    GECCO_GUARD_BEGIN
    std::unique_ptr<gecco_crf_plan> h(new gecco_crf_plan());
    int rc = plan_build(m->m, device, contig_ptr, n_contigs, window, step, pad, h->p);
    if (rc) return rc;
    *out = h.release();
    return GECCO_CRF_OK;
    GECCO_GUARD_END
}
GECCO_API void gecco_crf_plan_free(gecco_crf_plan *p) {
    DeviceGuard guard;
    delete p;
}
GECCO_API int32_t gecco_crf_plan_num_genes(const gecco_crf_plan *p) { return p ? p->p.n_genes : 0; }
GECCO_API int64_t gecco_crf_plan_num_windows(const gecco_crf_plan *p) { return p ? p->p.n_windows : 0; }
GECCO_API int32_t gecco_crf_plan_num_tiles(const gecco_crf_plan *p) { return p ? p->p.ntiles : 0; }
GECCO_API int32_t gecco_crf_plan_tile_out(const gecco_crf_plan *p) { return p ? p->p.tile_out : 0; }
GECCO_API const char *gecco_crf_plan_kernel_name(const gecco_crf_plan *p) { return p ? p->p.kernel_name.c_str() : ""; }

GECCO_API int gecco_crf_plan_run_windowed(gecco_crf_plan *p, const int32_t *d_gene_ptr, const int32_t *d_attr_id,
                                          int32_t label, double *d_p_out, void *stream) {
    if (!p) return GECCO_CRF_EINVAL;
    DeviceGuard guard;
    GECCO_GUARD_BEGIN
    return plan_run_windowed(p->p, d_gene_ptr, d_attr_id, label, d_p_out, static_cast<hipStream_t>(stream));
    GECCO_GUARD_END
}

GECCO_API int gecco_crf_plan_time_windowed(gecco_crf_plan *p, const int32_t *d_gene_ptr, const int32_t *d_attr_id,
                                           int32_t label, double *d_p_out, void *stream, int32_t warmup,
                                           int32_t iters, float *ms_per_launch) {
    if (!p || !ms_per_launch || iters <= 0) return GECCO_CRF_EINVAL;
    DeviceGuard guard;
    GECCO_GUARD_BEGIN
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
    GECCO_GUARD_END
}

GECCO_API int gecco_crf_plan_time_decode_pipelined(gecco_crf_plan *p, const int32_t *d_gene_ptr, const int32_t *d_attr_id,
                                                   int32_t label, double *d_p_out, int8_t *d_y, void *stream, int32_t warmup,
                                                   int32_t iters, float *ms_per_launch) {
    if (!p || !ms_per_launch || iters <= 0) return GECCO_CRF_EINVAL;
    DeviceGuard guard;
    GECCO_GUARD_BEGIN
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc;
    // the plan follows itself: call 0 primes (tiles only), every later call is one launch of tiles + Viterbi workgroups
    if ((rc = plan_run_decode_pipelined(&p->p, d_gene_ptr, d_attr_id, label, d_p_out, nullptr, nullptr, s))) return rc;
    for (int i = 0; i < warmup; ++i)
        if ((rc = plan_run_decode_pipelined(&p->p, d_gene_ptr, d_attr_id, label, d_p_out, &p->p, d_y, s))) return rc;
    hipEvent_t e0, e1;
    if ((rc = check_hip(hipEventCreate(&e0), "hipEventCreate"))) return rc;
    if ((rc = check_hip(hipEventCreate(&e1), "hipEventCreate"))) return rc;
    rc = check_hip(hipEventRecord(e0, s), "hipEventRecord");
    for (int i = 0; i < iters && !rc; ++i) rc = plan_run_decode_pipelined(&p->p, d_gene_ptr, d_attr_id, label, d_p_out, &p->p, d_y, s);
    if (!rc) rc = check_hip(hipEventRecord(e1, s), "hipEventRecord");
    if (!rc) rc = check_hip(hipEventSynchronize(e1), "hipEventSynchronize");
    float ms = 0.f;
    if (!rc) rc = check_hip(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime");
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (!rc) rc = plan_run_decode_pipelined(nullptr, nullptr, nullptr, label, nullptr, &p->p, d_y, s);  // (flush)
    *ms_per_launch = ms / float(iters);
    return rc;
    GECCO_GUARD_END
}

GECCO_API int gecco_crf_plan_viterbi_stats(gecco_crf_plan *p, int64_t out[4], int32_t reset) {
    if (!p || !out) return GECCO_CRF_EINVAL;
    DeviceGuard guard;
    GECCO_GUARD_BEGIN
    return plan_viterbi_stats(p->p, out, reset != 0);
    GECCO_GUARD_END
}

GECCO_API int gecco_crf_plan_run_decode(gecco_crf_plan *p, const int32_t *d_gene_ptr, const int32_t *d_attr_id,
                                        int32_t label, double *d_p_out, int8_t *d_y, double *d_score, void *stream) {
    if (!p) return GECCO_CRF_EINVAL;
    DeviceGuard guard;
    GECCO_GUARD_BEGIN
    return plan_run_decode(p->p, d_gene_ptr, d_attr_id, label, d_p_out, d_y, d_score, static_cast<hipStream_t>(stream));
    GECCO_GUARD_END
}
GECCO_API int gecco_crf_plan_run_decode_pipelined(gecco_crf_plan *p, const int32_t *d_gene_ptr, const int32_t *d_attr_id, int32_t label,
                                                  double *d_p_out, gecco_crf_plan *prev, int8_t *d_prev_y, void *stream) {
    if (!p && !prev) return GECCO_CRF_EINVAL;
    DeviceGuard guard;
    GECCO_GUARD_BEGIN
    return plan_run_decode_pipelined(p ? &p->p : nullptr, d_gene_ptr, d_attr_id, label, d_p_out, prev ? &prev->p : nullptr, d_prev_y,
                                     static_cast<hipStream_t>(stream));
    GECCO_GUARD_END
}
GECCO_API int gecco_crf_plan_run_marginals_full(gecco_crf_plan *p, const int32_t *d_gene_ptr, const int32_t *d_attr_id,
                                                double *d_marg, double *d_lognorm, void *stream) {
    if (!p) return GECCO_CRF_EINVAL;
    DeviceGuard guard;
    GECCO_GUARD_BEGIN
    return plan_run_marginals_full(p->p, d_gene_ptr, d_attr_id, d_marg, d_lognorm, static_cast<hipStream_t>(stream));
    GECCO_GUARD_END
}
GECCO_API int gecco_crf_plan_run_viterbi(gecco_crf_plan *p, const int32_t *d_gene_ptr, const int32_t *d_attr_id,
                                         int8_t *d_y, double *d_score, void *stream) {
    if (!p) return GECCO_CRF_EINVAL;
    DeviceGuard guard;
    GECCO_GUARD_BEGIN
    return plan_run_viterbi(p->p, d_gene_ptr, d_attr_id, d_y, d_score, static_cast<hipStream_t>(stream));
    GECCO_GUARD_END
}

// ---- pinned host memory -----------------------------------------------------------------------
GECCO_API int gecco_crf_host_alloc(size_t n_bytes, void **out) {
    if (!out) return GECCO_CRF_EINVAL;
    *out = nullptr;
    int32_t n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device available (pinned memory needs the HIP runtime)");
        return GECCO_CRF_ENODEV;
    }
    return check_hip(hipHostMalloc(out, n_bytes ? n_bytes : 1, hipHostMallocPortable), "hipHostMalloc");
}
GECCO_API void gecco_crf_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

// ---- batch driver -------------------------------------------------------------------------------
GECCO_API int gecco_crf_session_create(const gecco_crf_model *m, const int32_t *devices, int32_t n_devices,
                                       gecco_crf_session **out) {
    if (!m || !out) return GECCO_CRF_EINVAL;
    *out = nullptr;
    DeviceGuard guard;
    GECCO_GUARD_BEGIN
    std::unique_ptr<gecco_crf_session> h(new gecco_crf_session());
    int rc = session_create(m->m, devices, n_devices, &h->s);
    if (rc) return rc;
    *out = h.release();
    return GECCO_CRF_OK;
    GECCO_GUARD_END
}
GECCO_API void gecco_crf_session_free(gecco_crf_session *s) {
    DeviceGuard guard;
    delete s;
}
GECCO_API int gecco_crf_session_set_chunk_genes(gecco_crf_session *s, int32_t genes) {
    if (!s || genes <= 0) return GECCO_CRF_EINVAL;
    session_set_chunk_genes(*s->s, genes);
    return GECCO_CRF_OK;
}
GECCO_API int gecco_crf_session_set_direct_genes(gecco_crf_session *s, int32_t genes) {
    if (!s || genes < -1) return GECCO_CRF_EINVAL;  // (-1: back to the defaults, as the header documents)
    session_set_direct_genes(*s->s, genes);
    return GECCO_CRF_OK;
}
GECCO_API int gecco_crf_session_set_reference_bits(gecco_crf_session *s, int32_t on) {
    if (!s) return GECCO_CRF_EINVAL;
    session_set_reference_bits(*s->s, on != 0);
    return GECCO_CRF_OK;
}
GECCO_API int gecco_crf_exp_correctly_rounded(const double *x, int64_t n, double *out) {
    if (n < 0 || (n > 0 && (!x || !out))) return GECCO_CRF_EINVAL;
    for (int64_t i = 0; i < n; ++i) out[i] = gecco::ddx::exp_correctly_rounded(x[i]);
    return GECCO_CRF_OK;
}
GECCO_API int gecco_crf_session_stats(const gecco_crf_session *s, int32_t *n_chunks, int64_t *h2d_bytes, int64_t *d2h_bytes,
                                      double *host_plan_seconds, double *wall_seconds) {
    if (!s) return GECCO_CRF_EINVAL;
    const SessionStats st = session_stats(*s->s);
    if (n_chunks) *n_chunks = st.n_chunks;
    if (h2d_bytes) *h2d_bytes = st.h2d_bytes;
    if (d2h_bytes) *d2h_bytes = st.d2h_bytes;
    if (host_plan_seconds) *host_plan_seconds = st.host_plan_seconds;
    if (wall_seconds) *wall_seconds = st.wall_seconds;
    return GECCO_CRF_OK;
}

GECCO_API int gecco_crf_session_stats_ex(const gecco_crf_session *s, gecco_crf_session_stats_t *out) {
    if (!s || !out) return GECCO_CRF_EINVAL;
    const SessionStats st = session_stats(*s->s);
    out->n_chunks = st.n_chunks;
    out->n_devices = st.n_devices;
    out->direct = st.direct;
    out->host_threads = st.host_threads;
    out->h2d_bytes = st.h2d_bytes;
    out->d2h_bytes = st.d2h_bytes;
    out->host_plan_seconds = st.host_plan_seconds;
    out->host_issue_seconds = st.host_issue_seconds;
    out->wall_seconds = st.wall_seconds;
    return GECCO_CRF_OK;
}

namespace {
int run_guarded(Session &s, const BatchRequest &r) {
    GECCO_GUARD_BEGIN
    return session_run(s, r);
    GECCO_GUARD_END
}
BatchRequest csr_request(const int32_t *contig_ptr, int32_t n_contigs, const int32_t *gene_ptr, const int32_t *attr_id) {
    BatchRequest r;
    r.contig_ptr = contig_ptr;
    r.n_contigs = n_contigs;
    r.gene_ptr = gene_ptr;
    r.attr_id = attr_id;
    return r;
}
}  // namespace

GECCO_API int gecco_crf_session_windowed(gecco_crf_session *s, const int32_t *contig_ptr, int32_t n_contigs,
                                         const int32_t *gene_ptr, const int32_t *attr_id, int32_t window, int32_t step,
                                         int32_t label, int32_t pad, double *p_out) {
    if (!s || !p_out) return GECCO_CRF_EINVAL;
    BatchRequest r = csr_request(contig_ptr, n_contigs, gene_ptr, attr_id);
    r.window = window;
    r.step = step;
    r.label = label;
    r.pad = pad;
    r.p_out = p_out;
    return run_guarded(*s->s, r);
}
GECCO_API int gecco_crf_session_windowed_degrees(gecco_crf_session *s, const int32_t *contig_ptr, int32_t n_contigs,
                                                 const int32_t *gene_ptr, const uint8_t *degree, const int32_t *attr_id,
                                                 int32_t window, int32_t step, int32_t label, int32_t pad, double *p_out) {
    if (!s || !p_out || !degree) return GECCO_CRF_EINVAL;
    BatchRequest r = csr_request(contig_ptr, n_contigs, gene_ptr, attr_id);
    r.degree = degree;
    r.window = window;
    r.step = step;
    r.label = label;
    r.pad = pad;
    r.p_out = p_out;
    return run_guarded(*s->s, r);
}
GECCO_API int gecco_crf_session_decode(gecco_crf_session *s, const int32_t *contig_ptr, int32_t n_contigs,
                                       const int32_t *gene_ptr, const int32_t *attr_id, int32_t window, int32_t step,
                                       int32_t label, int32_t pad, double *p_out, int8_t *y_out) {
    if (!s || !p_out || !y_out) return GECCO_CRF_EINVAL;
    BatchRequest r = csr_request(contig_ptr, n_contigs, gene_ptr, attr_id);
    r.window = window;
    r.step = step;
    r.label = label;
    r.pad = pad;
    r.p_out = p_out;
    r.y_out = y_out;
    return run_guarded(*s->s, r);
}
namespace {
SegParams seg_params(const gecco_crf_refine_params &q) {
    SegParams sp;
    sp.threshold = q.threshold;
    sp.average_threshold = q.average_threshold;
    sp.criterion = q.criterion;
    sp.n_cds = q.n_cds;
    sp.n_biopfams = q.n_biopfams;
    sp.edge_distance = q.edge_distance;
    sp.trim = q.trim ? 1 : 0;
    sp.carry = q.carry_state ? 1 : 0;
    sp.bio_ptr = q.marker_ptr;
    sp.bio_id = q.marker_id;
    return sp;
}
gecco_crf_refine_params gecco_params(double threshold, int32_t n_cds, int32_t edge_distance, int32_t trim, int32_t carry_state) {
    gecco_crf_refine_params q{};
    q.threshold = threshold;
    q.average_threshold = 0.6;
    q.criterion = 0;
    q.n_cds = n_cds;
    q.n_biopfams = 5;
    q.edge_distance = edge_distance;
    q.trim = trim;
    q.carry_state = carry_state;
    return q;
}
}  // namespace

GECCO_API int gecco_crf_session_decode_wire(gecco_crf_session *s, const int32_t *contig_ptr, int32_t n_contigs,
                                            const int32_t *gene_ptr, const uint8_t *degree, const int32_t *attr_id,
                                            const uint16_t *attr_id16, int32_t window, int32_t step, int32_t label, int32_t pad,
                                            double *p_out, int8_t *y_out) {
    if (!s || !p_out) return GECCO_CRF_EINVAL;
    BatchRequest r = csr_request(contig_ptr, n_contigs, gene_ptr, attr_id);
    r.degree = degree;
    r.attr_id16 = attr_id16;
    r.window = window;
    r.step = step;
    r.label = label;
    r.pad = pad;
    r.p_out = p_out;
    r.y_out = y_out;
    return run_guarded(*s->s, r);
}
GECCO_API int gecco_crf_session_clusters_ex(gecco_crf_session *s, const int32_t *contig_ptr, int32_t n_contigs,
                                            const int32_t *gene_ptr, const int32_t *attr_id, const uint8_t *annotated,
                                            int32_t window, int32_t step, int32_t label, int32_t pad,
                                            const gecco_crf_refine_params *params, double *p_out, int32_t *seg_out,
                                            int32_t max_seg, int32_t *n_seg, double *seg_p_out, int64_t max_seg_genes,
                                            int64_t *seg_off_out) {
    return gecco_crf_session_clusters_degrees(s, contig_ptr, n_contigs, gene_ptr, nullptr, attr_id, annotated, window, step, label, pad,
                                              params, p_out, seg_out, max_seg, n_seg, seg_p_out, max_seg_genes, seg_off_out);
}
GECCO_API int gecco_crf_session_clusters_degrees(gecco_crf_session *s, const int32_t *contig_ptr, int32_t n_contigs,
                                                 const int32_t *gene_ptr, const uint8_t *degree, const int32_t *attr_id,
                                                 const uint8_t *annotated, int32_t window, int32_t step, int32_t label, int32_t pad,
                                                 const gecco_crf_refine_params *params, double *p_out, int32_t *seg_out,
                                                 int32_t max_seg, int32_t *n_seg, double *seg_p_out, int64_t max_seg_genes,
                                                 int64_t *seg_off_out) {
    return gecco_crf_session_clusters_wire(s, contig_ptr, n_contigs, gene_ptr, degree, attr_id, nullptr, annotated, window, step, label,
                                           pad, params, p_out, seg_out, max_seg, n_seg, seg_p_out, max_seg_genes, seg_off_out);
}
GECCO_API int gecco_crf_session_clusters_wire(gecco_crf_session *s, const int32_t *contig_ptr, int32_t n_contigs,
                                              const int32_t *gene_ptr, const uint8_t *degree, const int32_t *attr_id,
                                              const uint16_t *attr_id16, const uint8_t *annotated, int32_t window, int32_t step,
                                              int32_t label, int32_t pad, const gecco_crf_refine_params *params, double *p_out,
                                              int32_t *seg_out, int32_t max_seg, int32_t *n_seg, double *seg_p_out,
                                              int64_t max_seg_genes, int64_t *seg_off_out) {
    if (!s || !params) return GECCO_CRF_EINVAL;
    BatchRequest r = csr_request(contig_ptr, n_contigs, gene_ptr, attr_id);
    r.degree = degree;
    r.attr_id16 = attr_id16;
    r.window = window;
    r.step = step;
    r.label = label;
    r.pad = pad;
    r.p_out = p_out;
    r.want_segments = true;
    r.annotated = annotated;
    r.seg = seg_params(*params);
    r.seg_out = seg_out;
    r.max_seg = max_seg;
    r.n_seg = n_seg;
    r.seg_p_out = seg_p_out;
    r.max_seg_genes = max_seg_genes;
    r.seg_off_out = seg_off_out;
    return run_guarded(*s->s, r);
}

GECCO_API int gecco_crf_session_clusters(gecco_crf_session *s, const int32_t *contig_ptr, int32_t n_contigs,
                                         const int32_t *gene_ptr, const int32_t *attr_id, const uint8_t *annotated,
                                         int32_t window, int32_t step, int32_t label, int32_t pad, double threshold,
                                         int32_t n_cds, int32_t edge_distance, int32_t trim, double *p_out, int32_t *seg_out,
                                         int32_t max_seg, int32_t *n_seg, double *seg_p_out, int64_t max_seg_genes,
                                         int64_t *seg_off_out) {
    const gecco_crf_refine_params q = gecco_params(threshold, n_cds, edge_distance, trim, 0);
    return gecco_crf_session_clusters_ex(s, contig_ptr, n_contigs, gene_ptr, attr_id, annotated, window, step, label, pad, &q, p_out,
                                         seg_out, max_seg, n_seg, seg_p_out, max_seg_genes, seg_off_out);
}

// ---- one-shot host entry points: thin wrappers over the model's own per-device session ------------
namespace {
int default_session(const gecco_crf_model *m, int32_t device, Session **out) {
    int rc = check_device(device);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(m->m.dev_mutex);
    for (auto &e : m->m.sessions)
        if (e.first == device) {
            *out = e.second;
            return GECCO_CRF_OK;
        }
    Session *s = nullptr;
    if ((rc = session_create(m->m, &device, 1, &s))) return rc;
    m->m.sessions.emplace_back(device, s);
    *out = s;
    return GECCO_CRF_OK;
}
}  // namespace

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
    DeviceGuard guard;
    GECCO_GUARD_BEGIN
    Session *s = nullptr;
    int rc = default_session(m, device, &s);
    if (rc) return rc;
    if (n_contigs > 0 && contig_ptr && contig_ptr[n_contigs] > 0 && !p_out) {
        set_error("null buffer");
        return GECCO_CRF_EINVAL;
    }
    BatchRequest r = csr_request(contig_ptr, n_contigs, gene_ptr, attr_id);
    r.window = window;
    r.step = step;
    r.label = label;
    r.pad = pad;
    r.p_out = p_out;
    return session_run(*s, r);
    GECCO_GUARD_END
}

GECCO_API int gecco_crf_marginals_full(const gecco_crf_model *m, int32_t device, const int32_t *contig_ptr,
                                       int32_t n_contigs, const int32_t *gene_ptr, const int32_t *attr_id,
                                       double *marg, double *lognorm) {
    if (!m) return GECCO_CRF_EINVAL;
    DeviceGuard guard;
    GECCO_GUARD_BEGIN
    Session *s = nullptr;
    int rc = default_session(m, device, &s);
    if (rc) return rc;
    if (!marg && !lognorm) return GECCO_CRF_OK;
    BatchRequest r = csr_request(contig_ptr, n_contigs, gene_ptr, attr_id);
    r.marg_out = marg;
    r.lognorm_out = lognorm;
    return session_run(*s, r);
    GECCO_GUARD_END
}

GECCO_API int gecco_crf_viterbi(const gecco_crf_model *m, int32_t device, const int32_t *contig_ptr, int32_t n_contigs,
                                const int32_t *gene_ptr, const int32_t *attr_id, int8_t *y_out, double *score) {
    if (!m) return GECCO_CRF_EINVAL;
    DeviceGuard guard;
    GECCO_GUARD_BEGIN
    Session *s = nullptr;
    int rc = default_session(m, device, &s);
    if (rc) return rc;
    if (!y_out) {
        if (n_contigs > 0 && contig_ptr && contig_ptr[n_contigs] > 0) {
            set_error("null buffer");
            return GECCO_CRF_EINVAL;
        }
        return GECCO_CRF_OK;
    }
    BatchRequest r = csr_request(contig_ptr, n_contigs, gene_ptr, attr_id);
    r.y_out = y_out;
    r.score_out = score;
    return session_run(*s, r);
    GECCO_GUARD_END
}

namespace {
// grow-only scratch of the stand-alone segmenter / composition calls, one set per host thread
struct Scratch {
    int device = -1;
    char *d = nullptr;
    size_t cap = 0;
    ~Scratch() {
        if (d && hipSetDevice(device) == hipSuccess) (void)hipFree(d);
    }
    int reserve(int dev, size_t bytes) {
        if (d && dev == device && bytes <= cap) return GECCO_CRF_OK;
        if (d && hipSetDevice(device) == hipSuccess) (void)hipFree(d);
        d = nullptr;
        cap = 0;
        int rc = check_hip(hipSetDevice(dev), "hipSetDevice");
        if (rc) return rc;
        const size_t want = bytes + bytes / 8 + 256;
        if ((rc = check_hip(hipMalloc(reinterpret_cast<void **>(&d), want), "hipMalloc scratch"))) return rc;
        device = dev;
        cap = want;
        return GECCO_CRF_OK;
    }
};
inline size_t al256(size_t x) { return (x + 255) & ~size_t(255); }
}  // namespace

GECCO_API int gecco_crf_segment_ex(int32_t device, const double *p, const uint8_t *annotated, const int32_t *contig_ptr,
                                   int32_t n_contigs, const gecco_crf_refine_params *params, int32_t *seg_out, int32_t max_seg,
                                   int32_t *n_seg) {
    if (!params || !n_seg || n_contigs < 0 || (n_contigs > 0 && !contig_ptr) || max_seg < 0 || (max_seg > 0 && !seg_out)) {
        set_error("gecco_crf_segment: bad arguments");
        return GECCO_CRF_EINVAL;
    }
    *n_seg = 0;
    if (params->criterion != 0 && params->criterion != 1) {
        set_error("Unknown cluster filtering criterion");  // refine.py:165
        return GECCO_CRF_EINVAL;
    }
    int rc = check_device(device);
    if (rc) return rc;
    if (n_contigs == 0) return GECCO_CRF_OK;
    for (int32_t c = 0; c < n_contigs; ++c)
        if (contig_ptr[c + 1] < contig_ptr[c] || contig_ptr[0] != 0) {
            set_error("contig_ptr must start at 0 and be non-decreasing");
            return GECCO_CRF_EINVAL;
        }
    const size_t n = size_t(contig_ptr[n_contigs]), nc = size_t(n_contigs);
    if (n == 0) return GECCO_CRF_OK;
    if (!p || !annotated) {
        set_error("gecco_crf_segment: bad arguments");
        return GECCO_CRF_EINVAL;
    }
    SegParams sp = seg_params(*params);
    size_t nb = 0;
    if (sp.criterion == 1) {
        if (!sp.bio_ptr || sp.bio_ptr[0] != 0 || sp.bio_ptr[n] < 0 || (sp.bio_ptr[n] > 0 && !sp.bio_id)) {
            set_error("the antismash criterion needs the genes' marker domains (marker_ptr[0] = 0)");
            return GECCO_CRF_EINVAL;
        }
        for (size_t g = 0; g < n; ++g)
            if (sp.bio_ptr[g + 1] < sp.bio_ptr[g]) {
                set_error("marker_ptr must be non-decreasing");
                return GECCO_CRF_EINVAL;
            }
        nb = size_t(sp.bio_ptr[n]);
    }
    DeviceGuard guard;
    GECCO_GUARD_BEGIN
    static thread_local Scratch scratch;
    const size_t o_p = 0, o_a = o_p + al256(n * 8), o_c = o_a + al256(n + 8), o_seg = o_c + al256((nc + 1) * 4),
                 o_tot = o_seg + al256(size_t(max_seg) * 16 + 16), o_bp = o_tot + 256, o_bi = o_bp + al256((n + 1) * 4),
                 o_ws = o_bi + al256((nb + 4) * 4), bytes = o_ws + segment_workspace_bytes(int(n), n_contigs);
    if ((rc = scratch.reserve(device, bytes))) return rc;
    char *d = scratch.d;
    if ((rc = check_hip(hipMemcpy(d + o_p, p, n * 8, hipMemcpyHostToDevice), "H2D p"))) return rc;
    if ((rc = check_hip(hipMemcpy(d + o_a, annotated, n, hipMemcpyHostToDevice), "H2D annotated"))) return rc;
    if ((rc = check_hip(hipMemcpy(d + o_c, contig_ptr, (nc + 1) * 4, hipMemcpyHostToDevice), "H2D contig_ptr"))) return rc;
    if (sp.criterion == 1) {
        if ((rc = check_hip(hipMemcpy(d + o_bp, sp.bio_ptr, (n + 1) * 4, hipMemcpyHostToDevice), "H2D marker_ptr"))) return rc;
        if (nb && (rc = check_hip(hipMemcpy(d + o_bi, sp.bio_id, nb * 4, hipMemcpyHostToDevice), "H2D marker_id"))) return rc;
        sp.bio_ptr = reinterpret_cast<const int32_t *>(d + o_bp);
        sp.bio_id = reinterpret_cast<const int32_t *>(d + o_bi);
    }
    int32_t *d_seg = reinterpret_cast<int32_t *>(d + o_seg), *d_total = reinterpret_cast<int32_t *>(d + o_tot);
    if ((rc = check_hip(launch_segment(reinterpret_cast<const double *>(d + o_p), reinterpret_cast<const uint8_t *>(d + o_a), nullptr,
                                       reinterpret_cast<const int32_t *>(d + o_c), int(n), n_contigs, sp, d_seg, max_seg, nullptr,
                                       d_total, d + o_ws, nullptr),
                        "segment launch")))
        return rc;
    int32_t total = 0;
    if ((rc = check_hip(hipMemcpy(&total, d_total, 4, hipMemcpyDeviceToHost), "D2H count"))) return rc;
    *n_seg = total;
    if (total > max_seg) {
        set_error("gecco_crf_segment: seg_out too small");
        return GECCO_CRF_EINVAL;
    }
    if (total && (rc = check_hip(hipMemcpy(seg_out, d_seg, size_t(total) * 16, hipMemcpyDeviceToHost), "D2H segments")))
        return rc;
    return GECCO_CRF_OK;
    GECCO_GUARD_END
}

GECCO_API int gecco_crf_segment(int32_t device, const double *p, const uint8_t *annotated, const int32_t *contig_ptr,
                                int32_t n_contigs, double threshold, int32_t n_cds, int32_t edge_distance,
                                int32_t trim, int32_t carry_state, int32_t *seg_out, int32_t max_seg, int32_t *n_seg) {
    const gecco_crf_refine_params q = gecco_params(threshold, n_cds, edge_distance, trim, carry_state);
    return gecco_crf_segment_ex(device, p, annotated, contig_ptr, n_contigs, &q, seg_out, max_seg, n_seg);
}

GECCO_API int gecco_crf_plan_run_segment_ex(gecco_crf_plan *p, const double *d_p, const uint8_t *d_annotated,
                                            const gecco_crf_refine_params *params, int32_t *d_seg, int32_t max_seg,
                                            int32_t *d_n_seg, void *stream) {
    if (!p || !params) return GECCO_CRF_EINVAL;
    DeviceGuard guard;
    GECCO_GUARD_BEGIN
    return plan_run_segment(p->p, d_p, d_annotated, seg_params(*params), d_seg, max_seg, nullptr, d_n_seg,
                            static_cast<hipStream_t>(stream));
    GECCO_GUARD_END
}

GECCO_API int gecco_crf_plan_run_segment(gecco_crf_plan *p, const double *d_p, const uint8_t *d_annotated, double threshold,
                                         int32_t n_cds, int32_t edge_distance, int32_t trim, int32_t carry_state,
                                         int32_t *d_seg, int32_t max_seg, int32_t *d_n_seg, void *stream) {
    const gecco_crf_refine_params q = gecco_params(threshold, n_cds, edge_distance, trim, carry_state);
    return gecco_crf_plan_run_segment_ex(p, d_p, d_annotated, &q, d_seg, max_seg, d_n_seg, stream);
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
    GECCO_GUARD_BEGIN
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
    GECCO_GUARD_END
}

// ---- columnar host side ---------------------------------------------------------------------------
GECCO_API int gecco_crf_pack_columns(const gecco_crf_model *m, const gecco_crf_table_columns *t, gecco_crf_packed **out) {
    if (!m || !t || !out) return GECCO_CRF_EINVAL;
    *out = nullptr;
    GECCO_GUARD_BEGIN
    std::unique_ptr<gecco_crf_packed> h(new gecco_crf_packed());
    int rc = pack_columns(m->m, *t, h->p);
    if (rc) return rc;
    *out = h.release();
    return GECCO_CRF_OK;
    GECCO_GUARD_END
}
GECCO_API void gecco_crf_packed_free(gecco_crf_packed *p) { delete p; }
GECCO_API int gecco_crf_packed_info(const gecco_crf_packed *p, int32_t *n_genes, int32_t *n_contigs, int64_t *nnz,
                                    int32_t *n_duplicate_gene_ids, int32_t *n_unlisted_proteins, int32_t *pinned) {
    if (!p) return GECCO_CRF_EINVAL;
    if (n_genes) *n_genes = p->p.n_genes;
    if (n_contigs) *n_contigs = p->p.n_contigs;
    if (nnz) *nnz = p->p.nnz;
    if (n_duplicate_gene_ids) *n_duplicate_gene_ids = p->p.n_duplicate_gene_ids;
    if (n_unlisted_proteins) *n_unlisted_proteins = p->p.n_unlisted_proteins;
    if (pinned) *pinned = p->p.pinned ? 1 : 0;
    return GECCO_CRF_OK;
}
GECCO_API const int32_t *gecco_crf_packed_contig_ptr(const gecco_crf_packed *p) { return p ? p->p.contig_ptr : nullptr; }
GECCO_API const int32_t *gecco_crf_packed_gene_ptr(const gecco_crf_packed *p) { return p ? p->p.gene_ptr : nullptr; }
GECCO_API const int32_t *gecco_crf_packed_attr_id(const gecco_crf_packed *p) { return p ? p->p.attr_id : nullptr; }
GECCO_API const uint8_t *gecco_crf_packed_annotated(const gecco_crf_packed *p) { return p ? p->p.annotated : nullptr; }
GECCO_API const int64_t *gecco_crf_packed_gene_row(const gecco_crf_packed *p) { return p ? p->p.gene_row.data() : nullptr; }
GECCO_API const int32_t *gecco_crf_packed_row_gene(const gecco_crf_packed *p) { return p ? p->p.row_gene.data() : nullptr; }
GECCO_API const int64_t *gecco_crf_packed_row_order(const gecco_crf_packed *p) { return p ? p->p.row_order.data() : nullptr; }
GECCO_API const int64_t *gecco_crf_packed_row_ptr(const gecco_crf_packed *p) { return p ? p->p.row_ptr.data() : nullptr; }
GECCO_API const int32_t *gecco_crf_packed_marker_ptr(const gecco_crf_packed *p) { return p ? p->p.marker_ptr : nullptr; }
GECCO_API const int32_t *gecco_crf_packed_marker_id(const gecco_crf_packed *p) { return p ? p->p.marker_id : nullptr; }

GECCO_API int gecco_crf_cluster_rows_build(const gecco_crf_packed *p, const gecco_crf_table_columns *t, const int64_t *gene_end,
                                           const int64_t *feature_end, const int32_t *seg, int32_t n_seg, const double *seg_p,
                                           const int64_t *seg_off, gecco_crf_cluster_rows **out) {
    if (!p || !t || !out || n_seg < 0 || (n_seg > 0 && (!seg || !seg_p || !seg_off))) return GECCO_CRF_EINVAL;
    *out = nullptr;
    if ((t->n_genes > 0 && !gene_end) || (t->n_rows > 0 && !feature_end)) {
        set_error("cluster_rows: the tables' `end` columns are required");
        return GECCO_CRF_EINVAL;
    }
    GECCO_GUARD_BEGIN
    std::unique_ptr<gecco_crf_cluster_rows> h(new gecco_crf_cluster_rows());
    int rc = cluster_rows(p->p, *t, gene_end, feature_end, seg, n_seg, seg_p, seg_off, h->r);
    if (rc) return rc;
    *out = h.release();
    return GECCO_CRF_OK;
    GECCO_GUARD_END
}
GECCO_API void gecco_crf_cluster_rows_free(gecco_crf_cluster_rows *r) { delete r; }
GECCO_API const int64_t *gecco_crf_cluster_rows_start(const gecco_crf_cluster_rows *r) { return r ? r->r.start.data() : nullptr; }
GECCO_API const int64_t *gecco_crf_cluster_rows_end(const gecco_crf_cluster_rows *r) { return r ? r->r.end.data() : nullptr; }
GECCO_API const double *gecco_crf_cluster_rows_average_p(const gecco_crf_cluster_rows *r) { return r ? r->r.average_p.data() : nullptr; }
GECCO_API const double *gecco_crf_cluster_rows_max_p(const gecco_crf_cluster_rows *r) { return r ? r->r.max_p.data() : nullptr; }
GECCO_API int gecco_crf_cluster_rows_strings(const gecco_crf_cluster_rows *r, int32_t which, const uint8_t **data,
                                             const int64_t **offsets) {
    if (!r || !data || !offsets || which < 0 || which > 3) return GECCO_CRF_EINVAL;
    const StrOut *c = which == 0 ? &r->r.sequence_id : which == 1 ? &r->r.cluster_id : which == 2 ? &r->r.proteins : &r->r.domains;
    *data = c->data.data();
    *offsets = c->offsets.data();
    return GECCO_CRF_OK;
}
GECCO_API double gecco_crf_exact_mean(const double *v, int64_t n) { return (v && n > 0) ? exact_mean(v, n) : std::nan(""); }
GECCO_API int gecco_crf_gather_f64(const double *src, int64_t n_src, const int32_t *idx, int64_t n, double *out) {
    if (n < 0 || n_src < 0 || (n > 0 && (!src || !idx || !out))) return GECCO_CRF_EINVAL;
    GECCO_GUARD_BEGIN
    return gather_f64(src, n_src, idx, n, out);
    GECCO_GUARD_END
}
GECCO_API int gecco_crf_packed_order_info(const gecco_crf_packed *p, const int64_t *gene_start, const int64_t *gene_end, int64_t n_gene_rows,
                                          int32_t *rows_in_order, int32_t *refiner_order_differs) {
    if (!p || !rows_in_order || !refiner_order_differs || n_gene_rows < 0 || (p->p.n_genes > 0 && (!gene_start || !gene_end)))
        return GECCO_CRF_EINVAL;
    GECCO_GUARD_BEGIN
    return order_info(p->p, gene_start, gene_end, n_gene_rows, rows_in_order, refiner_order_differs);
    GECCO_GUARD_END
}

GECCO_API int gecco_crf_tsv_format(int64_t n_rows, int32_t n_cols, const int32_t *kinds, const void *const *data,
                                   const int64_t *const *offsets, const char *header, uint8_t **out, int64_t *out_len) {
    if (n_rows < 0 || n_cols < 0 || !out || !out_len || (n_cols > 0 && (!kinds || !data || !offsets))) return GECCO_CRF_EINVAL;
    *out = nullptr;
    *out_len = 0;
    for (int32_t c = 0; c < n_cols; ++c)
        if (kinds[c] < 0 || kinds[c] > 2 || (n_rows > 0 && !data[c]) || (kinds[c] == 0 && !offsets[c])) {
            set_error("tsv_format: bad column");
            return GECCO_CRF_EINVAL;
        }
    GECCO_GUARD_BEGIN
    return format_tsv(n_rows, n_cols, kinds, data, offsets, header, out, out_len);
    GECCO_GUARD_END
}
GECCO_API void gecco_crf_buffer_free(uint8_t *p) { std::free(p); }
