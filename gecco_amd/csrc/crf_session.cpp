#include "crf_session.hpp"

#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <numeric>

#include "../../include/gecco_crf.h"

namespace gecco {
namespace {

constexpr int kLanes = 4;  // buffers in flight per device: uploading, computing, downloading + one being laid out

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// GECCO_CRF_TRACE=1: host-side time of every step of a submission, to stderr (is a call blocking?)
bool trace_on() {
    static const bool on = [] {
        const char *e = std::getenv("GECCO_CRF_TRACE");
        return e && e[0] == '1';
    }();
    return on;
}
struct TraceMark {
    double t;
    TraceMark() : t(trace_on() ? now_s() : 0.0) {}
    void lap(const char *what, int chunk) {
        if (!trace_on()) return;
        const double n = now_s();
        std::fprintf(stderr, "[gecco_crf] chunk %d %-14s %8.1f us\n", chunk, what, (n - t) * 1e6);
        t = n;
    }
};

struct DevBuf {  // grow-only device block
    char *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes, const char *what) {
        if (p && bytes <= cap) return GECCO_CRF_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 8 + 256;
        int rc = check_hip(hipMalloc(reinterpret_cast<void **>(&p), want), what);
        if (!rc) cap = want;
        return rc;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct HostBuf {  // grow-only pinned, device-visible host block: kernels write the (few) segment rows straight into it
    char *p = nullptr, *dp = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes, const char *what) {
        if (p && bytes <= cap) return GECCO_CRF_OK;
        release();
        const size_t want = bytes + bytes / 8 + 256;
        int rc = check_hip(hipHostMalloc(reinterpret_cast<void **>(&p), want, hipHostMallocMapped | hipHostMallocPortable), what);
        if (rc) return rc;
        if ((rc = check_hip(hipHostGetDevicePointer(reinterpret_cast<void **>(&dp), p, 0), what))) {
            (void)hipHostFree(p);
            p = nullptr;
            return rc;
        }
        cap = want;
        return GECCO_CRF_OK;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = dp = nullptr;
        cap = 0;
    }
};

struct Chunk {
    int32_t c0 = 0, c1 = 0;  // contigs
    int32_t g0 = 0, g1 = 0;  // genes whose results the chunk delivers
    // genes the chunk uploads and scores: [g0, g1) for a chunk of whole contigs; for a PIECE of a contig too long for one
    // chunk (cut_chunks) the W - 1 genes either side as well -- every window that covers a gene of [g0, g1) lies inside
    // [u0, u1), so the piece is scored as a contig of its own and its inner genes get the bits they get in the whole contig
    int32_t u0 = 0, u1 = 0;
    bool piece = false;
    int32_t device_slot = 0;
    // segment rows of the chunk, translated to batch indices, and the probabilities of their genes
    std::vector<int32_t> rows;
    std::vector<double> seg_p;   // (only when the caller's seg_p_out is too small to stage the chunk at its gene offset)
    int64_t seg_p_count = 0;     // probabilities of the chunk's rows, parked at seg_p_out + g0 until the final compaction
};

// Streams are per DIRECTION, not per chunk: all uploads of a device go through `up`, all kernels through
// `comp`, all downloads through `down`, chained by events.  Measured on MI355X (tools/ubench/pipe_test.hip):
// with one stream per chunk (upload, kernel, download in each) the copies of consecutive chunks do not
// overlap at all on this runtime -- 0.66 ms for a 2 M-gene batch however it is cut, the same as no
// pipelining; one stream per direction: 0.56 ms.
struct Lane {
    int device = -1;
    hipStream_t up = nullptr, comp = nullptr, down = nullptr;  // the device's three streams (not owned)
    hipEvent_t ev_up = nullptr, ev_comp = nullptr, done = nullptr;
    int chunk = -1;  // index of the chunk in flight, -1: idle
    Plan plan;
    int up_chunk = -1;  // chunk whose arrays were uploaded ahead of its submission (-1: none)
    int64_t up_a0 = 0, up_nnz = 0, up_b0 = 0;
    DevBuf d_gp, d_at, d_at16, d_p, d_y, d_ann, d_score, d_marg, d_lognorm, d_bp, d_bi, d_seg, d_deg, d_deg_ws, d_segp;
    SessionStats *st = nullptr;  // the device's share of the batch's figures (merged when the batch is done)
    int8_t *y_dst = nullptr;     // where the decoder writes the chunk's labels: the lane's device buffer, or the caller's pinned array
    bool y_to_host = false;      // ... the latter: no download of labels
    HostBuf h_io;   // direct path (small batches): the chunk's arrays and outputs in pinned, device-visible memory
    // direct path: device-visible addresses of the chunk's arrays (the staging block, or the caller's own pinned buffers) and
    // what the host copies out of the staging block once the launch has completed
    const int32_t *x_gp = nullptr, *x_at = nullptr, *x_bp = nullptr, *x_bi = nullptr;
    const uint8_t *x_ann = nullptr;
    double *x_p = nullptr;       // where the kernels write p (null: the lane's device buffer)
    int8_t *x_y = nullptr;
    const double *x_p_host = nullptr;  // ... the same places as the host sees them, when they lie inside h_io: copied to the
    const int8_t *x_y_host = nullptr;  // caller's arrays at retirement (null: the kernels wrote to the caller's arrays)
    HostBuf h_seg;  // [total][pad..][seg_off: cap+1][rows: cap*4]
    bool segp_in_flight = false;  // the download of the rows' probabilities has been issued (retire_begin), not yet waited for
    int32_t seg_cap = 0;
    size_t o_off = 0, o_rows = 0, o_p = 0;
};

struct DeviceCtx {
    int device = -1;
    SessionStats stats;  // this device's share of the last batch (written by the thread that drives the device)
    int rc = 0;          // ... and how its submissions ended
    std::string err;
    hipStream_t up = nullptr, comp = nullptr, down = nullptr;
    bool owns_streams = true;  // (a device listed twice: the later entries use the first one's three streams)
    Lane lanes[kLanes];
    int next_lane = 0;
    // decode calls (marginals + labels): the launch of chunk k carries the Viterbi workgroups of the device's chunk k - 1
    // (plan_run_decode_pipelined), so that chunk's labels are downloaded -- and its lane is `done` -- behind the NEXT launch
    Lane *pending = nullptr;
};

inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

}  // namespace

// One thread per device entry beyond the first, parked between batches: run(n, job) has the calling thread do job(0) and
// worker k do job(k) and returns when all are done.
class Workers {
public:
    ~Workers() { stop(); }
    void run(int n, const std::function<void(int)> &job) {
        {
            std::unique_lock<std::mutex> lock(mu_);
            while (int(th_.size()) < n - 1) {
                const int slot = int(th_.size()) + 1;
                th_.emplace_back([this, slot] { loop(slot); });
            }
            job_ = &job;
            n_ = n;
            pending_ = n - 1;
            ++gen_;
        }
        cv_go_.notify_all();
        job(0);
        std::unique_lock<std::mutex> lock(mu_);
        cv_done_.wait(lock, [this] { return pending_ == 0; });
        job_ = nullptr;
    }
    void stop() {
        {
            std::unique_lock<std::mutex> lock(mu_);
            quit_ = true;
        }
        cv_go_.notify_all();
        for (std::thread &t : th_)
            if (t.joinable()) t.join();
        th_.clear();
    }

private:
    void loop(int slot) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)> *job = nullptr;
            {
                std::unique_lock<std::mutex> lock(mu_);
                cv_go_.wait(lock, [&] { return quit_ || gen_ != seen; });
                if (quit_) return;
                seen = gen_;
                if (slot < n_) job = job_;  // (a batch over fewer entries than there are workers: nothing to do)
            }
            if (!job) continue;
            (*job)(slot);
            {
                std::unique_lock<std::mutex> lock(mu_);
                --pending_;
            }
            cv_done_.notify_one();
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_go_, cv_done_;
    const std::function<void(int)> *job_ = nullptr;
    uint64_t gen_ = 0;
    int n_ = 0, pending_ = 0;
    bool quit_ = false;
};

struct Session {
    const Model *model = nullptr;
    Workers workers;
    std::vector<std::unique_ptr<DeviceCtx>> devs;
    int32_t chunk_genes = 1 << 19;
    std::mutex mu;  // one batch at a time per session
    // Batches of at most this many genes take the DIRECT path: one chunk, no copy commands, no second stream -- the kernels read
    // the arrays from pinned host memory (or from device memory filled by a copy on the compute stream) and write their outputs
    // there, the host waits for the compute stream.  gecco_crf_session_set_direct_genes / GECCO_CRF_DIRECT_GENES
    // override (0: never); -1 = the defaults below.
    // Measured (tools/direct_sweep.py, pinned buffers, us per call direct / ONE chunk through the three streams): marginals
    // 33 / 62 at 2 000 genes, 33 / 80 at 10 000, 51 / 98 at 65 000, 106 / 147 at 200 000, 206 / 237 at 500 000 -- a batch that is
    // one chunk anyway is faster direct; decode calls likewise (206 / 213 at 500 000).  Cluster rows 59 / 74, 62 / 91, 90 / 95,
    // 104 / 99 at 100 000: the refiner's seven short launches sit behind the tiles either way, so cluster calls stop at 65 536.
    // A session over several devices keeps 131 072: beyond, its chunks run on all of them at once.
    int32_t direct_genes = -1;
    int32_t direct_limit(bool want_segments) const {
        if (direct_genes >= 0) return want_segments ? direct_genes / 2 : direct_genes;
        return want_segments ? (1 << 16) : devs.size() > 1 ? (1 << 17) : (1 << 19);
    }
    bool reference_bits = false;  // windowed marginals in CRFsuite's operation order (crf_exact.hip): the reference's bits
    SessionStats stats;
    ~Session();
};

Session::~Session() {
    workers.stop();
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    for (auto &d : devs) {
        if (hipSetDevice(d->device) != hipSuccess) continue;
        (void)hipDeviceSynchronize();
        for (Lane &ln : d->lanes) {
            for (DevBuf *b : {&ln.d_gp, &ln.d_at, &ln.d_at16, &ln.d_p, &ln.d_y, &ln.d_ann, &ln.d_score, &ln.d_marg, &ln.d_lognorm, &ln.d_bp, &ln.d_bi, &ln.d_seg, &ln.d_deg, &ln.d_deg_ws, &ln.d_segp}) b->release();
            ln.h_seg.release();
            ln.h_io.release();
            for (hipEvent_t e : {ln.ev_up, ln.ev_comp, ln.done})
                if (e) (void)hipEventDestroy(e);
        }
        if (d->owns_streams)
            for (hipStream_t st : {d->up, d->comp, d->down})
                if (st) (void)hipStreamDestroy(st);
    }
    devs.clear();  // plans free their blocks under their own device guard
    if (prev >= 0) (void)hipSetDevice(prev);
}

int session_create(const Model &m, const int32_t *devices, int32_t n_devices, Session **out) {
    if (!out || n_devices <= 0 || !devices) {
        set_error("session: no devices given");
        return GECCO_CRF_EINVAL;
    }
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        set_error("no HIP device available (this library has no CPU fallback)");
        return GECCO_CRF_ENODEV;
    }
    std::unique_ptr<Session> s(new Session());
    s->model = &m;
    if (const char *env = std::getenv("GECCO_CRF_CHUNK_GENES")) {
        const long v = std::atol(env);
        if (v >= 1024) s->chunk_genes = int32_t(std::min<long>(v, 1 << 28));
    }
    if (const char *env = std::getenv("GECCO_CRF_DIRECT_GENES")) s->direct_genes = int32_t(std::max<long>(0, std::min<long>(std::atol(env), 1 << 24)));
    for (int32_t i = 0; i < n_devices; ++i) {
        if (devices[i] < 0 || devices[i] >= count) {
            set_error("device index out of range");
            return GECCO_CRF_ENODEV;
        }
        // a device may be listed more than once: every entry gets its own ring of lanes and its own submitting thread, but a
        // physical device has ONE stream per direction (copies of one direction queue up behind each other anyway; three
        // streams per entry only multiplied the hardware queues the device has to poll)
        int rc = check_hip(hipSetDevice(devices[i]), "hipSetDevice");
        if (rc) return rc;
        std::unique_ptr<DeviceCtx> d(new DeviceCtx());
        d->device = devices[i];
        s->devs.push_back(std::move(d));  // from here on the session's destructor cleans up
        DeviceCtx &D = *s->devs.back();
        const DeviceCtx *first = nullptr;
        for (size_t k = 0; k + 1 < s->devs.size() && !first; ++k)
            if (s->devs[k]->device == devices[i]) first = s->devs[k].get();
        if (first) {
            D.up = first->up;
            D.comp = first->comp;
            D.down = first->down;
            D.owns_streams = false;
        } else {
            for (hipStream_t *st : {&D.up, &D.comp, &D.down})
                if ((rc = check_hip(hipStreamCreateWithFlags(st, hipStreamNonBlocking), "hipStreamCreate"))) return rc;
        }
        for (Lane &ln : D.lanes) {
            ln.device = devices[i];
            ln.up = D.up;
            ln.comp = D.comp;
            ln.down = D.down;
            ln.plan.async_tables = true;
            ln.st = &D.stats;
            for (hipEvent_t *e : {&ln.ev_up, &ln.ev_comp, &ln.done})
                if ((rc = check_hip(hipEventCreateWithFlags(e, hipEventDisableTiming), "hipEventCreate"))) return rc;
        }
    }
    *out = s.release();
    return GECCO_CRF_OK;
}

void session_destroy(Session *s) { delete s; }
void session_set_chunk_genes(Session &s, int32_t genes) {
    std::lock_guard<std::mutex> lock(s.mu);
    s.chunk_genes = std::max(1024, genes);
}
void session_set_reference_bits(Session &s, bool on) {
    std::lock_guard<std::mutex> lock(s.mu);
    s.reference_bits = on;
}
void session_set_direct_genes(Session &s, int32_t genes) {
    std::lock_guard<std::mutex> lock(s.mu);
    s.direct_genes = genes < 0 ? -1 : genes;  // (-1: the defaults)
}
SessionStats session_stats(const Session &s) { return s.stats; }

namespace {

// Cut the batch into chunks of about `target` genes.  Chunks end at contig boundaries; a contig much longer than a chunk is
// cut into PIECES when the call allows it (`split_window` > 0: windowed marginals only -- every window is independent,
// /root/reference/gecco/crf/__init__.py:251-256, so a piece plus the W - 1 genes either side of it is scored as a contig of its
// own; whole-contig recursions and the refiner need the contig in one place).  With several devices there are at
// least two chunks per device when the batch is large enough to make that worthwhile.
void cut_chunks(const BatchRequest &r, int32_t target, int n_devices, std::vector<Chunk> &chunks, int32_t split_window = 0,
                int32_t split_step = 1) {
    chunks.clear();
    const int32_t nc = r.n_contigs;
    if (nc <= 0) return;
    const int64_t n = int64_t(r.contig_ptr[nc]) - r.contig_ptr[0];
    int64_t want = std::max<int64_t>(1, (n + target / 2) / target);
    if (n_devices > 1) want = std::max<int64_t>(want, std::min<int64_t>(2 * n_devices, std::max<int64_t>(1, n / 65536)));
    const double per = double(n) / double(std::max<int64_t>(want, 1));
    // a contig is "long" when it alone would make a chunk half as large again as the others
    const int64_t long_genes = std::max<int64_t>(int64_t(per * 1.5), 4 * int64_t(split_window) + 1024);
    bool any_long = false;
    if (split_window > 0)
        for (int32_t c = 0; c < nc && !any_long; ++c) any_long = int64_t(r.contig_ptr[c + 1]) - r.contig_ptr[c] > long_genes;
    if (!any_long) {
        want = std::min<int64_t>(want, nc);
        // (a half-sized first and last chunk -- the first download starts and the last one ends half a chunk earlier, for one
        // chunk more -- was measured in round 5: 0.63 against 0.60 ms per C3 call)
        const double per_c = double(n) / double(want);
        int32_t c = 0;
        for (int64_t k = 0; k < want && c < nc; ++k) {
            Chunk ck;
            ck.c0 = c;
            const int64_t goal = r.contig_ptr[0] + int64_t(per_c * double(k + 1) + 0.5);
            if (k == want - 1) {
                c = nc;
            } else {
                // first contig boundary at or past the goal, leaving at least one contig per remaining chunk
                const int32_t *e = std::lower_bound(r.contig_ptr + c + 1, r.contig_ptr + nc, int32_t(std::min<int64_t>(goal, INT32_MAX)));
                c = int32_t(e - r.contig_ptr);
                c = std::min<int32_t>(c, nc - int32_t(want - 1 - k));
                c = std::max<int32_t>(c, ck.c0 + 1);
            }
            ck.c1 = c;
            ck.g0 = ck.u0 = r.contig_ptr[ck.c0];
            ck.g1 = ck.u1 = r.contig_ptr[ck.c1];
            chunks.push_back(std::move(ck));
        }
        return;
    }
    // some contig is long: runs of whole contigs of about `per` genes, long contigs in pieces of about `per` genes
    const int32_t halo = split_window - 1;
    Chunk cur;
    bool open = false;
    auto close = [&](int32_t c_end) {
        if (!open) return;
        cur.c1 = c_end;
        cur.g1 = cur.u1 = r.contig_ptr[c_end];
        chunks.push_back(cur);
        open = false;
    };
    for (int32_t c = 0; c < nc; ++c) {
        const int32_t s0 = r.contig_ptr[c], s1 = r.contig_ptr[c + 1];
        const int64_t len = int64_t(s1) - s0;
        if (len > long_genes) {
            close(c);
            const int64_t pieces = std::max<int64_t>(2, (len + int64_t(per) / 2) / std::max<int64_t>(int64_t(per), 1));
            for (int64_t k = 0; k < pieces; ++k) {
                Chunk pk;
                pk.piece = true;
                pk.c0 = c;
                pk.c1 = c + 1;
                pk.g0 = int32_t(s0 + len * k / pieces);
                pk.g1 = int32_t(s0 + len * (k + 1) / pieces);
                // the halo: W - 1 genes either side, the left end moved down to a window start of the contig (step > 1)
                int64_t lo = std::max<int64_t>(s0, int64_t(pk.g0) - halo);
                lo -= (lo - s0) % split_step;
                pk.u0 = int32_t(lo);
                pk.u1 = int32_t(std::min<int64_t>(s1, int64_t(pk.g1) + halo));
                chunks.push_back(pk);
            }
            continue;
        }
        if (!open) {
            cur = Chunk{};
            cur.c0 = c;
            cur.g0 = cur.u0 = s0;
            open = true;
        }
        if (int64_t(s1) - cur.g0 >= int64_t(per)) close(c + 1);
    }
    close(nc);
}

// Chunks to devices, longest first onto the least loaded device (the greedy partition of SURVEY.md
// 8e / gecco_amd/sharding.py, at chunk granularity: every chunk is a contiguous slice of the caller's
// arrays, so nothing is gathered on the host).
void deal_chunks(std::vector<Chunk> &chunks, int n_devices) {
    std::vector<int> order(chunks.size());
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        return (chunks[a].g1 - chunks[a].g0) > (chunks[b].g1 - chunks[b].g0);
    });
    std::vector<int64_t> load(size_t(n_devices), 0);
    for (int i : order) {
        const int d = int(std::min_element(load.begin(), load.end()) - load.begin());
        chunks[i].device_slot = d;
        load[d] += chunks[i].g1 - chunks[i].g0;
    }
}

// sum of n bytes: psadbw adds sixteen at a time into two 64-bit lanes (SSE2: the x86-64 baseline)
uint64_t byte_sum(const uint8_t *p, size_t n) {
#if defined(__SSE2__)
    const __m128i zero = _mm_setzero_si128();
    __m128i a0 = zero, a1 = zero, a2 = zero, a3 = zero;
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
        a0 = _mm_add_epi64(a0, _mm_sad_epu8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(p + i)), zero));
        a1 = _mm_add_epi64(a1, _mm_sad_epu8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(p + i + 16)), zero));
        a2 = _mm_add_epi64(a2, _mm_sad_epu8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(p + i + 32)), zero));
        a3 = _mm_add_epi64(a3, _mm_sad_epu8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(p + i + 48)), zero));
    }
    const __m128i acc = _mm_add_epi64(_mm_add_epi64(a0, a1), _mm_add_epi64(a2, a3));
    uint64_t lanes[2];
    _mm_storeu_si128(reinterpret_cast<__m128i *>(lanes), acc);
    uint64_t total = lanes[0] + lanes[1];
#else
    uint64_t total = 0;
    size_t i = 0;
#endif
    for (; i < n; ++i) total += p[i];
    return total;
}

struct RunCtx {
    Session &S;
    const BatchRequest &r;
    std::vector<Chunk> &chunks;
    bool windowed, viterbi, full;
    int32_t W, step, pad;
    bool direct = false;  // small batch: one chunk, kernels on pinned host memory, no copy commands (submit_direct_arrays)
};

// labels of the device's pending chunk: its Viterbi workgroups have just been launched on `comp` and `down` waits for them
int finish_pending(RunCtx &X, DeviceCtx &D) {
    Lane &pl = *D.pending;
    D.pending = nullptr;
    const Chunk &pk = X.chunks[pl.chunk];
    const size_t ng = size_t(pk.g1 - pk.g0);
    int rc = GECCO_CRF_OK;
    if (!pl.y_to_host) {  // (labels written into the caller's pinned array by the decoder itself need no copy)
        pl.st->d2h_bytes += int64_t(ng);
        rc = check_hip(hipMemcpyAsync(X.r.y_out + pk.g0, pl.d_y.p, ng, hipMemcpyDeviceToHost, pl.down), "D2H labels");
        if (rc) return rc;
    }
    return check_hip(hipEventRecord(pl.done, pl.down), "hipEventRecord");
}

// the labels of the last chunk a device scored in a decode call: a launch of Viterbi workgroups only
int flush_pending(RunCtx &X, DeviceCtx &D) {
    if (!D.pending) return GECCO_CRF_OK;
    Lane &pl = *D.pending;
    int rc = check_hip(hipSetDevice(pl.device), "hipSetDevice");
    if (rc) return rc;
    if ((rc = plan_run_decode_pipelined(nullptr, nullptr, nullptr, X.r.label, nullptr, &pl.plan, pl.y_dst, pl.comp)))
        return rc;
    if ((rc = check_hip(hipEventRecord(pl.ev_comp, pl.comp), "hipEventRecord"))) return rc;
    if ((rc = check_hip(hipStreamWaitEvent(pl.down, pl.ev_comp, 0), "hipStreamWaitEvent"))) return rc;
    return finish_pending(X, D);
}

// the bulk uploads of a chunk into an idle lane.  Issued one chunk ahead (session_run): the copy engine then goes from one
// chunk's arrays to the next one's while the host still lays out and launches the first
int submit_uploads(RunCtx &X, Lane &ln, int chunk_index) {
    const BatchRequest &r = X.r;
    Chunk &ck = X.chunks[chunk_index];
    int rc = check_hip(hipSetDevice(ln.device), "hipSetDevice");
    if (rc) return rc;
    const int32_t ng = ck.u1 - ck.u0;
    TraceMark tm;
    const int64_t a0 = ng ? r.gene_ptr[ck.u0] : 0, a1 = ng ? r.gene_ptr[ck.u1] : 0;
    if (a0 < 0 || a1 < a0) {
        set_error("gene_ptr must be non-decreasing and start at a non-negative offset");
        return GECCO_CRF_EINVAL;
    }
    const size_t nnz = size_t(a1 - a0);
    int64_t b0 = 0;  // first marker offset of the chunk (antismash criterion)
    ln.up_chunk = chunk_index;
    ln.up_a0 = a0;
    ln.up_nnz = int64_t(nnz);
    if (ng) {
        if (r.degree) {  // degree bytes cross PCIe; the row pointers are rebuilt on the device (below, on the compute stream)
            if ((rc = ln.d_gp.reserve((size_t(ng) + 1) * 4, "hipMalloc gene_ptr"))) return rc;
            if ((rc = ln.d_deg.reserve(size_t(ng) + 32, "hipMalloc degrees"))) return rc;
            if ((rc = ln.d_deg_ws.reserve(degree_scratch_bytes(ng), "hipMalloc degree scan"))) return rc;
            if ((rc = check_hip(hipMemcpyAsync(ln.d_deg.p, r.degree + ck.u0, size_t(ng), hipMemcpyHostToDevice, ln.up), "H2D degrees")))
                return rc;
            ln.st->h2d_bytes += int64_t(ng);
        } else {
            if ((rc = ln.d_gp.reserve((size_t(ng) + 1) * 4, "hipMalloc gene_ptr"))) return rc;
            if ((rc = check_hip(hipMemcpyAsync(ln.d_gp.p, r.gene_ptr + ck.u0, (size_t(ng) + 1) * 4, hipMemcpyHostToDevice, ln.up), "H2D gene_ptr")))
                return rc;
            ln.st->h2d_bytes += int64_t((size_t(ng) + 1) * 4);
        }
        if ((rc = ln.d_at.reserve((nnz + 8) * 4, "hipMalloc attr_id"))) return rc;
        if (r.attr_id16) {  // 16-bit indices cross PCIe; widened on the compute stream (below)
            if ((rc = ln.d_at16.reserve((nnz + 8) * 2, "hipMalloc attr_id16"))) return rc;
            if (nnz && (rc = check_hip(hipMemcpyAsync(ln.d_at16.p, r.attr_id16 + a0, nnz * 2, hipMemcpyHostToDevice, ln.up), "H2D attr_id16")))
                return rc;
            ln.st->h2d_bytes += int64_t(nnz * 2);
        } else {
            if (nnz && (rc = check_hip(hipMemcpyAsync(ln.d_at.p, r.attr_id + a0, nnz * 4, hipMemcpyHostToDevice, ln.up), "H2D attr_id")))
                return rc;
            ln.st->h2d_bytes += int64_t(nnz * 4);
        }
        if (r.want_segments) {
            if (r.annotated) {  // (null: a gene is annotated iff it has a domain the model knows -- the degree bytes themselves)
                if ((rc = ln.d_ann.reserve(size_t(ng) + 8, "hipMalloc annotated"))) return rc;
                if ((rc = check_hip(hipMemcpyAsync(ln.d_ann.p, r.annotated + ck.u0, size_t(ng), hipMemcpyHostToDevice, ln.up), "H2D annotated")))
                    return rc;
                ln.st->h2d_bytes += ng;
            }
            if (r.seg.criterion == 1) {  // the genes' marker domains, offsets kept as the caller's (like gene_ptr)
                b0 = r.seg.bio_ptr[ck.u0];
                const int64_t b1 = r.seg.bio_ptr[ck.u1];
                if (b0 < 0 || b1 < b0) {
                    set_error("marker_ptr must be non-decreasing and start at a non-negative offset");
                    return GECCO_CRF_EINVAL;
                }
                const size_t nb = size_t(b1 - b0);
                if ((rc = ln.d_bp.reserve((size_t(ng) + 1) * 4, "hipMalloc marker_ptr"))) return rc;
                if ((rc = ln.d_bi.reserve((nb + 4) * 4, "hipMalloc marker_id"))) return rc;
                if ((rc = check_hip(hipMemcpyAsync(ln.d_bp.p, r.seg.bio_ptr + ck.u0, (size_t(ng) + 1) * 4, hipMemcpyHostToDevice, ln.up), "H2D marker_ptr")))
                    return rc;
                if (nb && (rc = check_hip(hipMemcpyAsync(ln.d_bi.p, r.seg.bio_id + b0, nb * 4, hipMemcpyHostToDevice, ln.up), "H2D marker_id")))
                    return rc;
                ln.st->h2d_bytes += int64_t((size_t(ng) + 1 + nb) * 4);
            }
        }
    }
    ln.up_b0 = b0;
    // the device derives the rows from the degree bytes, the host takes the chunk's base offset from gene_ptr: they must agree
    // (checked while the copies above are under way)
    if (ng && r.degree && byte_sum(r.degree + ck.u0, size_t(ng)) != uint64_t(a1 - a0)) {
        set_error("degree bytes do not add up to gene_ptr over a chunk (degree must equal diff(gene_ptr))");
        return GECCO_CRF_EINVAL;
    }
    tm.lap("h2d csr", chunk_index);
    return GECCO_CRF_OK;
}

// Device-visible address of caller memory that is pinned (gecco_crf_host_alloc / hipHostMalloc / hipHostRegister), or null.
// Only asked for arrays large enough for the question to cost less than the copy it may save.
template <class T>
T *mapped_or_null(T *host) {
    if (!host) return nullptr;
    void *dv = nullptr;
    if (hipHostGetDevicePointer(&dv, const_cast<void *>(static_cast<const void *>(host)), 0) != hipSuccess) {
        (void)hipGetLastError();  // (not pinned: not an error of this call)
        return nullptr;
    }
    return static_cast<T *>(dv);
}

// Direct path: what submit_uploads does on the upload stream, done by the host and by address.  The chunk (= the whole batch)
// gets device-visible addresses for its arrays: a copy in the lane's pinned staging block (a 50-gene contig: 600 bytes) that
// the kernels read in place; for an array of 256 KB and more, device memory filled by a copy command on the COMPUTE stream.
// Outputs are written by the kernels into pinned host memory: the caller's own buffer where it is pinned and large enough to
// be worth asking, the staging block otherwise.  The compact wire format is undone on the way (row pointers are the caller's
// gene_ptr; 16-bit indices are widened by the staging copy), so no launch has to.
int submit_direct_arrays(RunCtx &X, Lane &ln, int chunk_index) {
    const BatchRequest &r = X.r;
    Chunk &ck = X.chunks[chunk_index];
    int rc = check_hip(hipSetDevice(ln.device), "hipSetDevice");
    if (rc) return rc;
    const size_t ng = size_t(ck.g1 - ck.g0);
    const int64_t a0 = r.gene_ptr[ck.g0], a1 = r.gene_ptr[ck.g1];
    if (a0 < 0 || a1 < a0) {
        set_error("gene_ptr must be non-decreasing and start at a non-negative offset");
        return GECCO_CRF_EINVAL;
    }
    const size_t nnz = size_t(a1 - a0);
    if (r.degree && byte_sum(r.degree + ck.g0, ng) != uint64_t(nnz)) {
        set_error("degree bytes do not add up to gene_ptr over a chunk (degree must equal diff(gene_ptr))");
        return GECCO_CRF_EINVAL;
    }
    int64_t b0 = 0;
    size_t nb = 0;
    const bool markers = r.want_segments && r.seg.criterion == 1;
    if (markers) {
        b0 = r.seg.bio_ptr[ck.g0];
        const int64_t b1 = r.seg.bio_ptr[ck.g1];
        if (b0 < 0 || b1 < b0) {
            set_error("marker_ptr must be non-decreasing and start at a non-negative offset");
            return GECCO_CRF_EINVAL;
        }
        nb = size_t(b1 - b0);
    }
    constexpr size_t kAsk = 32768;  // bytes from which a caller's OUTPUT buffer is asked whether it is pinned
    // An input array of kCopy bytes and more goes to device memory by a copy command on the compute stream (~8 us of engine
    // turnaround + the transfer, against a PCIe round trip for every tile that reads it in place); smaller ones are read from
    // the staging block.  tools/direct_sweep.py, us per windowed call at 20 000 / 40 000 / 65 000 genes: copy from 64 KB on
    // 47.6 / 53.5 / 61.1, from 256 KB 34.0 / 44.2 / 49.7, from 1 MB 34.6 / 42.0 / 55.5.  (GECCO_CRF_DIRECT_COPY_BYTES: A/B runs)
    static const size_t kCopy = [] {
        const char *e = std::getenv("GECCO_CRF_DIRECT_COPY_BYTES");
        return e ? size_t(std::atoll(e)) : size_t(262144);
    }();
    const bool c_gp = (ng + 1) * 4 >= kCopy, c_at = nnz * 4 >= kCopy && !r.attr_id16;
    if (c_gp) {
        if ((rc = ln.d_gp.reserve((ng + 1) * 4, "hipMalloc gene_ptr"))) return rc;
        if ((rc = check_hip(hipMemcpyAsync(ln.d_gp.p, r.gene_ptr + ck.g0, (ng + 1) * 4, hipMemcpyHostToDevice, ln.comp), "H2D gene_ptr"))) return rc;
        ln.st->h2d_bytes += int64_t((ng + 1) * 4);
    }
    if (c_at) {
        if ((rc = ln.d_at.reserve((nnz + 8) * 4, "hipMalloc attr_id"))) return rc;
        if ((rc = check_hip(hipMemcpyAsync(ln.d_at.p, r.attr_id + a0, nnz * 4, hipMemcpyHostToDevice, ln.comp), "H2D attr_id"))) return rc;
        ln.st->h2d_bytes += int64_t(nnz * 4);
    }
    const bool c_at16 = nnz * 4 >= kCopy && r.attr_id16;  // (16-bit indices: half the bytes cross, a launch widens them)
    if (c_at16) {
        if ((rc = ln.d_at.reserve((nnz + 8) * 4, "hipMalloc attr_id"))) return rc;
        if ((rc = ln.d_at16.reserve((nnz + 8) * 2, "hipMalloc attr_id16"))) return rc;
        if ((rc = check_hip(hipMemcpyAsync(ln.d_at16.p, r.attr_id16 + a0, nnz * 2, hipMemcpyHostToDevice, ln.comp), "H2D attr_id16"))) return rc;
        ln.st->h2d_bytes += int64_t(nnz * 2);
        if ((rc = check_hip(launch_wire_format(nullptr, 0, 0, nullptr, nullptr, reinterpret_cast<const uint16_t *>(ln.d_at16.p), int64_t(nnz),
                                               reinterpret_cast<int32_t *>(ln.d_at.p), ln.comp), "wire format launch")))
            return rc;
    }
    const int32_t *m_gp = c_gp ? reinterpret_cast<const int32_t *>(ln.d_gp.p) : nullptr;
    const int32_t *m_at = (c_at || c_at16) ? reinterpret_cast<const int32_t *>(ln.d_at.p) : nullptr;
    // the kernels write p / y straight into host memory only where nothing reads them again on the device (no refiner behind
    // them) and the window kernel stores every gene exactly once (the 2-label register kernel: decided in submit)
    double *m_p = (r.p_out && !r.want_segments && ng * 8 >= kAsk) ? mapped_or_null(r.p_out + ck.g0) : nullptr;
    int8_t *m_y = (r.y_out && ng >= kAsk) ? mapped_or_null(r.y_out + ck.g0) : nullptr;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += align256(bytes + 32);
        return o;
    };
    const size_t o_gp = take(m_gp ? 0 : (ng + 1) * 4), o_at = take(m_at ? 0 : (nnz + 8) * 4),
                 o_ann = take(r.want_segments ? ng + 32 : 0), o_bp = take(markers ? (ng + 1) * 4 : 0), o_bi = take(markers ? (nb + 4) * 4 : 0),
                 o_p = take((r.p_out && !r.want_segments && !m_p) ? ng * 8 : 0), o_y = take((r.y_out && !m_y) ? ng + 8 : 0);
    if ((rc = ln.h_io.reserve(off, "hipHostMalloc staging"))) return rc;
    char *h = ln.h_io.p, *d = ln.h_io.dp;
    ln.x_gp = m_gp;
    if (!m_gp) {
        std::memcpy(h + o_gp, r.gene_ptr + ck.g0, (ng + 1) * 4);
        ln.x_gp = reinterpret_cast<const int32_t *>(d + o_gp);
    }
    ln.x_at = m_at;
    if (!m_at) {
        int32_t *dst = reinterpret_cast<int32_t *>(h + o_at);
        if (r.attr_id16) {
            const uint16_t *src = r.attr_id16 + a0;
            for (size_t i = 0; i < nnz; ++i) dst[i] = src[i];
        } else if (nnz) {
            std::memcpy(dst, r.attr_id + a0, nnz * 4);
        }
        ln.x_at = reinterpret_cast<const int32_t *>(d + o_at);
    }
    ln.x_ann = nullptr;
    if (r.want_segments) {  // (annotated, or the degree bytes standing in for it)
        std::memcpy(h + o_ann, (r.annotated ? r.annotated : r.degree) + ck.g0, ng);
        ln.x_ann = reinterpret_cast<const uint8_t *>(d + o_ann);
    }
    ln.x_bp = ln.x_bi = nullptr;
    if (markers) {
        std::memcpy(h + o_bp, r.seg.bio_ptr + ck.g0, (ng + 1) * 4);
        if (nb) std::memcpy(h + o_bi, r.seg.bio_id + b0, nb * 4);
        ln.x_bp = reinterpret_cast<const int32_t *>(d + o_bp);
        ln.x_bi = reinterpret_cast<const int32_t *>(d + o_bi);
    }
    ln.x_p = m_p;
    ln.x_p_host = nullptr;
    if (r.p_out && !r.want_segments && !m_p) {
        ln.x_p = reinterpret_cast<double *>(d + o_p);
        ln.x_p_host = reinterpret_cast<const double *>(h + o_p);
    }
    ln.x_y = m_y;
    ln.x_y_host = nullptr;
    if (r.y_out && !m_y) {
        ln.x_y = reinterpret_cast<int8_t *>(d + o_y);
        ln.x_y_host = reinterpret_cast<const int8_t *>(h + o_y);
    }
    ln.up_chunk = chunk_index;
    ln.up_a0 = a0;
    ln.up_nnz = int64_t(nnz);
    ln.up_b0 = b0;
    return GECCO_CRF_OK;
}

// the chunk's layout (host) and its tables' upload, behind the chunk's arrays on the upload stream
int submit_plan(RunCtx &X, Lane &ln, int chunk_index) {
    Session &S = X.S;
    const BatchRequest &r = X.r;
    Chunk &ck = X.chunks[chunk_index];
    const Model &m = *S.model;
    int rc = check_hip(hipSetDevice(ln.device), "hipSetDevice");
    if (rc) return rc;
    const int32_t nc = ck.c1 - ck.c0;
    TraceMark tm;
    const double t0 = now_s();
    {
        // marginals (and labels) only: the window kernel reads the plan tables (~0.1 MB per chunk, each word once) from the
        // plan's pinned block itself -- one copy and one ~10 us gap between copies less per chunk.  The refiner and the
        // whole-contig marginals search the contig table many times: for them it is copied.  GECCO_CRF_TABLES_COPY=1: always copy
        static const bool always_copy = [] {
            const char *env = std::getenv("GECCO_CRF_TABLES_COPY");
            return env && env[0] == '1';
        }();
        ln.plan.tables_in_host_memory = !always_copy && !r.want_segments && !X.full && !r.score_out;
        // copied tables are fetched by a small launch at the head of the chunk's kernels, not by the copy engine: the upload
        // stream then carries nothing but the chunks' arrays, back to back (a copy issued behind the next chunk's arrays
        // would hold this chunk's kernels back until those have crossed; every copy also costs ~10 us of engine turnaround)
        ln.plan.tables_by_kernel = true;
        // small batches: the decoder's tables and flag bytes too are read where the host wrote them
        ln.plan.seq_in_host_memory = X.direct && !r.want_segments && !X.full && !r.score_out;
        ln.plan.reference_bits = X.S.reference_bits && X.windowed;
        ln.plan.windowed_use = X.windowed;
    }
    if (ck.piece) {  // a stretch of ONE long contig, scored as a contig of its own
        const int32_t span[2] = {ck.u0, ck.u1};
        if ((rc = plan_build(m, ln.device, span, 1, X.W, X.step, X.pad, ln.plan, ln.comp, false))) return rc;
    } else if ((rc = plan_build(m, ln.device, r.contig_ptr + ck.c0, nc, X.W, X.step, X.pad, ln.plan, ln.comp, false))) {
        return rc;
    }
    // (the whole-contig tables too are fetched by a launch on the compute stream; the refiner builds its contig flags on the device)
    if ((X.viterbi || X.full) && (rc = plan_ensure_seq(ln.plan, ln.comp, false))) return rc;
    ln.st->host_plan_seconds += now_s() - t0;
    tm.lap("plan_build", chunk_index);
    if (X.direct) {  // (nothing is on the upload stream; the host has read the CSR's extent itself)
        ln.plan.csr_begin = ln.up_a0;
        ln.plan.csr_end = ln.up_a0 + ln.up_nnz;
        return GECCO_CRF_OK;
    }
    return check_hip(hipEventRecord(ln.ev_up, ln.up), "hipEventRecord");
}

// Direct path: every launch of the (one) chunk on the compute stream, on the addresses submit_direct_arrays chose; the host then
// waits for that stream (retire_begin) -- no copy command, no event, no second stream.  A 50-gene contig: one launch.
int submit_direct(RunCtx &X, DeviceCtx &D, Lane &ln, int chunk_index) {
    const BatchRequest &r = X.r;
    Chunk &ck = X.chunks[chunk_index];
    int rc;
    const int32_t ng = ck.g1 - ck.g0;
    const int32_t *d_gp = ln.x_gp, *d_at = ln.x_at - ln.up_a0;  // (gene_ptr keeps the caller's offsets)
    // the register-resident 2-label kernel and the reference-bits kernel store every gene's probability exactly once: they may
    // write to host memory.  The other window kernels accumulate with atomic maxima on a zeroed array: device memory, copied
    // out on the same stream.
    const bool p_to_host = ln.x_p && !ln.plan.general && (ln.plan.fast_ok || ln.plan.reference_now);
    double *d_p = nullptr;
    if (X.windowed) {
        if (p_to_host) {
            d_p = ln.x_p;
        } else {
            if ((rc = ln.d_p.reserve(size_t(ng) * 8, "hipMalloc p"))) return rc;
            d_p = reinterpret_cast<double *>(ln.d_p.p);
        }
    }
    if (X.windowed && X.viterbi) {
        // marginals, then the labels from the score differences the tiles left behind (the pipelined pair on one batch)
        if ((rc = plan_run_decode_pipelined(&ln.plan, d_gp, d_at, r.label, d_p, nullptr, nullptr, ln.comp))) return rc;
        if ((rc = plan_run_decode_pipelined(nullptr, nullptr, nullptr, r.label, nullptr, &ln.plan, ln.x_y, ln.comp))) return rc;
    } else if (X.windowed) {
        if ((rc = plan_run_windowed(ln.plan, d_gp, d_at, r.label, d_p, ln.comp))) return rc;
    } else if (X.viterbi) {
        if ((rc = plan_run_viterbi(ln.plan, d_gp, d_at, ln.x_y, nullptr, ln.comp))) return rc;
    }
    if (r.want_segments) {
        const int32_t nc = ck.c1 - ck.c0;
        const size_t cap = std::min<size_t>(size_t(ng), size_t(ng) / 2 + size_t(nc)) + 1;
        ln.seg_cap = int32_t(cap);
        ln.o_off = 256;
        ln.o_rows = ln.o_off + align256((cap + 1) * 4);
        ln.o_p = ln.o_rows + align256(cap * 16);
        if ((rc = ln.h_seg.reserve(ln.o_p + 256, "hipHostMalloc segments"))) return rc;
        if (r.seg_p_out && (rc = ln.d_segp.reserve(size_t(ng) * 8 + 8, "hipMalloc cluster probabilities"))) return rc;
        char *dp = ln.h_seg.dp;
        SegParams sp = r.seg;
        sp.carry = 0;
        sp.row_contig0 = ck.c0;
        sp.row_gene0 = ck.g0;
        sp.bio_ptr = sp.bio_id = nullptr;
        if (sp.criterion == 1) {
            sp.bio_ptr = ln.x_bp;
            sp.bio_id = ln.x_bi - ln.up_b0;
        }
        if ((rc = plan_run_segment(ln.plan, d_p, ln.x_ann, sp, reinterpret_cast<int32_t *>(dp + ln.o_rows), int32_t(cap),
                                   reinterpret_cast<int32_t *>(dp + ln.o_off), reinterpret_cast<int32_t *>(dp), ln.comp,
                                   r.seg_p_out ? reinterpret_cast<double *>(ln.d_segp.p) : nullptr, ng)))
            return rc;
    }
    if (r.p_out && !p_to_host) {  // (p lives in device memory: the refiner reads it, or a window kernel with atomic maxima wrote it)
        ln.st->d2h_bytes += int64_t(ng) * 8;
        if ((rc = check_hip(hipMemcpyAsync(r.p_out + ck.g0, d_p, size_t(ng) * 8, hipMemcpyDeviceToHost, ln.comp), "D2H p"))) return rc;
        ln.x_p_host = nullptr;
    }
    (void)D;
    return GECCO_CRF_OK;
}

int submit(RunCtx &X, DeviceCtx &D, Lane &ln, int chunk_index) {
    Session &S = X.S;
    const BatchRequest &r = X.r;
    Chunk &ck = X.chunks[chunk_index];
    const Model &m = *S.model;
    int rc = check_hip(hipSetDevice(ln.device), "hipSetDevice");
    if (rc) return rc;
    const int32_t nc = ck.c1 - ck.c0, ng = ck.u1 - ck.u0;  // (ng: the genes the chunk scores -- a piece's halo included)
    TraceMark tm;
    ln.chunk = chunk_index;
    ln.up_chunk = -1;
    const int64_t a0 = ln.up_a0, b0 = ln.up_b0;
    const size_t nnz = size_t(ln.up_nnz), L = size_t(m.L);
    if (X.direct) return submit_direct(X, D, ln, chunk_index);
    if ((rc = check_hip(hipStreamWaitEvent(ln.comp, ln.ev_up, 0), "hipStreamWaitEvent"))) return rc;
    if (ng == 0) return check_hip(hipEventRecord(ln.done, ln.comp), "hipEventRecord");
    // the wire format's two launches: degree bytes -> row pointers, 16-bit attribute indices -> 32-bit ones
    if ((r.degree || r.attr_id16) &&
        (rc = check_hip(launch_wire_format(r.degree ? reinterpret_cast<const uint8_t *>(ln.d_deg.p) : nullptr, int(ng), int32_t(a0),
                                           reinterpret_cast<int32_t *>(ln.d_gp.p), reinterpret_cast<int32_t *>(ln.d_deg_ws.p),
                                           r.attr_id16 ? reinterpret_cast<const uint16_t *>(ln.d_at16.p) : nullptr, int64_t(nnz),
                                           reinterpret_cast<int32_t *>(ln.d_at.p), ln.comp), "wire format launch")))
        return rc;
    // gene_ptr keeps the caller's offsets: the attribute array is addressed from where its element 0 would be
    const int32_t *d_gp = reinterpret_cast<const int32_t *>(ln.d_gp.p);
    const int32_t *d_at = reinterpret_cast<const int32_t *>(ln.d_at.p) - a0;
    double *d_p = nullptr, *d_score = nullptr;
    int8_t *d_y = nullptr;
    // Where the tiles write p: the lane's device buffer, downloaded by a copy.  GECCO_CRF_P_TO_HOST=1 (experiment, round 5): into
    // the caller's own array when it is pinned and nothing on the device reads p again -- the probabilities then cross PCIe as
    // the tiles' posted writes and the chunk has no download of 8 bytes per gene.  Not faster: the tiles' stores share the link
    // with the next chunk's upload less gracefully than the copy engine does (tools/ubench/pcie_bw.hip: 19 MB up + 16 MB down
    // at once 354 us by copies, 432 us with the download done by a kernel's stores).
    bool p_to_host = false;
    if (X.windowed) {
        static const bool allowed = [] {  // (off by default: measured on C3, pinned buffers: 0.619 ms against 0.609 through a copy)
            const char *env = std::getenv("GECCO_CRF_P_TO_HOST");
            return env && env[0] == '1';
        }();
        double *m_p = nullptr;
        if (allowed && r.p_out && !r.want_segments && !ck.piece && !ln.plan.general && ln.plan.fast_ok) m_p = mapped_or_null(r.p_out + ck.g0);
        if (m_p) {
            d_p = m_p;
            p_to_host = true;
        } else {
            if ((rc = ln.d_p.reserve(size_t(ng) * 8, "hipMalloc p"))) return rc;
            d_p = reinterpret_cast<double *>(ln.d_p.p);
        }
    }
    if (X.viterbi) {
        // Labels (a byte per gene) go straight into the caller's array when it is pinned: a chunk's 0.5 MB of them cost a copy
        // command and its ~12 us of engine turnaround each -- on the download stream, which is the critical path of a decode
        // call on the compact wire format -- and nothing as stores of the decoder.  GECCO_CRF_Y_TO_HOST=0: always a copy.
        static const bool y_allowed = [] {
            const char *env = std::getenv("GECCO_CRF_Y_TO_HOST");
            return !(env && env[0] == '0');
        }();
        int8_t *m_y = (y_allowed && r.y_out && !ck.piece) ? mapped_or_null(r.y_out + ck.g0) : nullptr;
        ln.y_to_host = m_y != nullptr;
        if (m_y) {
            d_y = m_y;
        } else {
            if ((rc = ln.d_y.reserve(size_t(ng) + 8, "hipMalloc labels"))) return rc;
            d_y = reinterpret_cast<int8_t *>(ln.d_y.p);
        }
        ln.y_dst = d_y;
        if (r.score_out) {
            if ((rc = ln.d_score.reserve(size_t(nc) * 8, "hipMalloc scores"))) return rc;
            d_score = reinterpret_cast<double *>(ln.d_score.p);
        }
    }
    const bool piped = X.windowed && X.viterbi && !r.score_out && !X.full && !r.want_segments;
    if (piped) {
        // ONE launch per chunk: this chunk's window tiles + the Viterbi workgroups of the chunk this device scored before
        Lane *prev = D.pending;
        rc = plan_run_decode_pipelined(&ln.plan, d_gp, d_at, r.label, d_p, prev ? &prev->plan : nullptr,
                                       prev ? prev->y_dst : nullptr, ln.comp);
        if (rc) return rc;
        if ((rc = check_hip(hipEventRecord(ln.ev_comp, ln.comp), "hipEventRecord"))) return rc;
        if ((rc = check_hip(hipStreamWaitEvent(ln.down, ln.ev_comp, 0), "hipStreamWaitEvent"))) return rc;
        if (prev && (rc = finish_pending(X, D))) return rc;
        if (!p_to_host) {
            ln.st->d2h_bytes += int64_t(ng) * 8;
            if ((rc = check_hip(hipMemcpyAsync(r.p_out + ck.g0, d_p, size_t(ng) * 8, hipMemcpyDeviceToHost, ln.down), "D2H p"))) return rc;
        }
        D.pending = &ln;  // (its labels and its `done` come with the device's next launch, or with the flush)
        tm.lap("launch", chunk_index);
        return GECCO_CRF_OK;
    }
    if (X.windowed && X.viterbi) {
        rc = plan_run_decode(ln.plan, d_gp, d_at, r.label, d_p, d_y, d_score, ln.comp);
    } else if (X.windowed) {
        rc = plan_run_windowed(ln.plan, d_gp, d_at, r.label, d_p, ln.comp);
    } else if (X.viterbi) {
        rc = plan_run_viterbi(ln.plan, d_gp, d_at, d_y, d_score, ln.comp);
    }
    if (rc) return rc;
    double *d_marg = nullptr, *d_lognorm = nullptr;
    if (X.full) {
        if ((rc = ln.d_marg.reserve(size_t(ng) * L * 8, "hipMalloc marginals"))) return rc;
        d_marg = reinterpret_cast<double *>(ln.d_marg.p);
        if ((rc = ln.d_lognorm.reserve(size_t(nc) * 8, "hipMalloc lognorm"))) return rc;
        d_lognorm = reinterpret_cast<double *>(ln.d_lognorm.p);
        if ((rc = plan_run_marginals_full(ln.plan, d_gp, d_at, d_marg, d_lognorm, ln.comp))) return rc;
    }
    if (r.want_segments) {
        // rows, their offsets and the probabilities of their genes are written by the kernels into pinned
        // host memory: nothing but those few bytes crosses PCIe for a cluster call
        const size_t cap = std::min<size_t>(size_t(ng), size_t(ng) / 2 + size_t(nc)) + 1;
        ln.seg_cap = int32_t(cap);
        ln.o_off = 256;
        ln.o_rows = ln.o_off + align256((cap + 1) * 4);
        ln.o_p = ln.o_rows + align256(cap * 16);
        const size_t bytes = ln.o_p + 256;
        if ((rc = ln.h_seg.reserve(bytes, "hipHostMalloc segments"))) return rc;
        // the segmenter's last launch writes rows, offsets and count straight into the pinned block (written across PCIe,
        // never read back across it); the probabilities of the rows' genes -- under SURVEY.md 8d's weight law nine genes in
        // ten -- are gathered in device memory and downloaded by the copy engine once the host knows how many there are
        // (read back from the pinned block by the host they came at 4 GB/s: 3.5 ms per C3 batch)
        if (r.seg_p_out && (rc = ln.d_segp.reserve(size_t(ng) * 8 + 8, "hipMalloc cluster probabilities"))) return rc;
        char *dp = ln.h_seg.dp;
        int32_t *d_total = reinterpret_cast<int32_t *>(dp), *d_off = reinterpret_cast<int32_t *>(dp + ln.o_off),
                *d_rows = reinterpret_cast<int32_t *>(dp + ln.o_rows);
        SegParams sp = r.seg;
        sp.carry = 0;
        sp.row_contig0 = ck.c0;  // (rows arrive in the batch's own indices)
        sp.row_gene0 = ck.g0;
        sp.bio_ptr = sp.bio_id = nullptr;
        if (sp.criterion == 1) {
            sp.bio_ptr = reinterpret_cast<const int32_t *>(ln.d_bp.p);
            sp.bio_id = reinterpret_cast<const int32_t *>(ln.d_bi.p) - b0;
        }
        if ((rc = plan_run_segment(ln.plan, d_p, reinterpret_cast<const uint8_t *>(r.annotated ? ln.d_ann.p : ln.d_deg.p), sp, d_rows, int32_t(cap), d_off, d_total,
                                   ln.comp, r.seg_p_out ? reinterpret_cast<double *>(ln.d_segp.p) : nullptr, ng)))
            return rc;
    }
    tm.lap("launch", chunk_index);
    const bool c_p = r.p_out && !p_to_host, c_y = r.y_out && !ln.y_to_host, c_score = r.score_out, c_marg = r.marg_out, c_ln = r.lognorm_out;
    if (!(c_p || c_y || c_score || c_marg || c_ln)) return check_hip(hipEventRecord(ln.done, ln.comp), "hipEventRecord");
    if ((rc = check_hip(hipEventRecord(ln.ev_comp, ln.comp), "hipEventRecord"))) return rc;
    if ((rc = check_hip(hipStreamWaitEvent(ln.down, ln.ev_comp, 0), "hipStreamWaitEvent"))) return rc;
    auto d2h = [&](void *dst, const void *src, size_t bytes, const char *what) {
        ln.st->d2h_bytes += int64_t(bytes);
        return bytes ? check_hip(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ln.down), what) : GECCO_CRF_OK;
    };
    if (c_p && (rc = d2h(r.p_out + ck.g0, d_p + (ck.g0 - ck.u0), size_t(ck.g1 - ck.g0) * 8, "D2H p"))) return rc;  // (a piece keeps its inner genes)
    if (c_y && (rc = d2h(r.y_out + ck.g0, d_y, size_t(ng), "D2H labels"))) return rc;
    if (c_score && (rc = d2h(r.score_out + ck.c0, d_score, size_t(nc) * 8, "D2H scores"))) return rc;
    if (c_marg && (rc = d2h(r.marg_out + size_t(ck.g0) * L, d_marg, size_t(ng) * L * 8, "D2H marginals"))) return rc;
    if (c_ln && (rc = d2h(r.lognorm_out + ck.c0, d_lognorm, size_t(nc) * 8, "D2H lognorm"))) return rc;
    tm.lap("d2h", chunk_index);
    rc = check_hip(hipEventRecord(ln.done, ln.down), "hipEventRecord");
    tm.lap("event", chunk_index);
    return rc;
}

// wait for the lane's chunk and take its segment rows over (translated to batch indices); the download of the rows'
// probabilities is ISSUED here (its size is only known now) and waited for in retire_end
int retire_begin(RunCtx &X, Lane &ln) {
    if (ln.chunk < 0) return GECCO_CRF_OK;
    TraceMark tm;
    int rc;
    Chunk &ck = X.chunks[ln.chunk];
    if (X.direct) {
        // everything of the chunk went to the compute stream; outputs the kernels left in the staging block go to the caller
        rc = check_hip(hipStreamSynchronize(ln.comp), "chunk completion");
        const size_t ng = size_t(ck.g1 - ck.g0);
        if (!rc && ln.x_p_host && X.r.p_out) std::memcpy(X.r.p_out + ck.g0, ln.x_p_host, ng * 8);
        if (!rc && ln.x_y_host && X.r.y_out) std::memcpy(X.r.y_out + ck.g0, ln.x_y_host, ng);
    } else {
        rc = check_hip(hipEventSynchronize(ln.done), "chunk completion");
    }
    tm.lap("wait", ln.chunk);
    if (rc) {
        ln.chunk = -1;
        return rc;
    }
    if (X.r.want_segments && ck.g1 > ck.g0) {
        const char *hp = ln.h_seg.p;
        const int32_t total = *reinterpret_cast<const int32_t *>(hp);
        if (total < 0 || total > ln.seg_cap) {
            ln.chunk = -1;
            set_error("segmenter returned an impossible row count");
            return GECCO_CRF_EHIP;
        }
        const int32_t *rows = reinterpret_cast<const int32_t *>(hp + ln.o_rows), *off = reinterpret_cast<const int32_t *>(hp + ln.o_off);
        ck.rows.assign(rows, rows + size_t(total) * 4);  // (written in batch indices by the segmenter's last launch)
        if (X.r.seg_p_out) {
            // straight into the caller's array at the chunk's gene offset (never behind their final place: the chunks before
            // hold at least as many genes as their rows do), closed up at the end of the call
            ck.seg_p_count = off[total];
            double *dst = X.r.seg_p_out + ck.g0;
            if (X.r.max_seg_genes < int64_t(ck.g0) + ck.seg_p_count) {  // (a caller array smaller than the batch: staged)
                ck.seg_p.resize(size_t(ck.seg_p_count));
                dst = ck.seg_p.data();
            }
            if (ck.seg_p_count) {
                if ((rc = check_hip(hipSetDevice(ln.device), "hipSetDevice"))) return rc;
                ln.st->d2h_bytes += ck.seg_p_count * 8;
                if ((rc = check_hip(hipMemcpyAsync(dst, ln.d_segp.p, size_t(ck.seg_p_count) * 8, hipMemcpyDeviceToHost, ln.down), "D2H cluster probabilities")))
                    return rc;
                if ((rc = check_hip(hipEventRecord(ln.done, ln.down), "hipEventRecord"))) return rc;
                ln.segp_in_flight = true;
            }
        }
    }
    return GECCO_CRF_OK;
}
int retire_end(RunCtx &X, Lane &ln) {
    if (ln.chunk < 0) return GECCO_CRF_OK;
    ln.chunk = -1;
    if (!ln.segp_in_flight) return GECCO_CRF_OK;
    ln.segp_in_flight = false;
    return check_hip(hipEventSynchronize(ln.done), "cluster probabilities");
}
int retire(RunCtx &X, Lane &ln) {
    const int rc = retire_begin(X, ln);
    const int rc2 = retire_end(X, ln);
    return rc ? rc : rc2;
}

}  // namespace

int session_run(Session &S, const BatchRequest &r) {
    const Model &m = *S.model;
    const bool windowed = r.p_out || r.want_segments, viterbi = r.y_out != nullptr,
               full = r.marg_out != nullptr || r.lognorm_out != nullptr;
    // argument errors first, so that they surface even on a box without a GPU (texts: gecco/_meta.py:127-130)
    if (windowed) {
        if (r.window <= 0) {
            set_error("Window size must be strictly positive");
            return GECCO_CRF_EINVAL;
        }
        if (r.step <= 0 || r.step > r.window) {
            set_error("Window step must be strictly positive and under `window_size`");
            return GECCO_CRF_EINVAL;
        }
        if (r.label < 0 || r.label >= m.L) {
            set_error("label out of range");
            return GECCO_CRF_EINVAL;
        }
    }
    if (r.n_contigs < 0 || (r.n_contigs > 0 && !r.contig_ptr)) {
        set_error("bad contig_ptr");
        return GECCO_CRF_EINVAL;
    }
    if (r.n_contigs > 0 && r.contig_ptr[0] != 0) {
        set_error("contig_ptr[0] must be 0");
        return GECCO_CRF_EINVAL;
    }
    if (r.score_out && !viterbi) {
        set_error("path scores come with the labels: y_out is required");
        return GECCO_CRF_EINVAL;
    }
    if (r.want_segments && (!r.n_seg || r.max_seg < 0 || (r.max_seg > 0 && !r.seg_out) || (r.seg_p_out && !r.seg_off_out))) {
        set_error("segments: bad output arguments");
        return GECCO_CRF_EINVAL;
    }
    if (r.want_segments && r.seg.criterion != 0 && r.seg.criterion != 1) {
        set_error("Unknown cluster filtering criterion");  // refine.py:165
        return GECCO_CRF_EINVAL;
    }
    if (r.n_seg) *r.n_seg = 0;
    for (int32_t c = 0; c < r.n_contigs; ++c)
        if (r.contig_ptr[c + 1] < r.contig_ptr[c]) {
            set_error("contig_ptr must be non-decreasing");
            return GECCO_CRF_EINVAL;
        }
    const int64_t n_genes = r.n_contigs ? r.contig_ptr[r.n_contigs] : 0;
    if (n_genes > 0 && (!r.gene_ptr || (r.want_segments && !r.annotated && !r.degree))) {  // (annotated may be left to the degree bytes)
        set_error("null buffer");
        return GECCO_CRF_EINVAL;
    }
    if (n_genes > 0 && r.want_segments && r.seg.criterion == 1 &&
        (!r.seg.bio_ptr || (r.seg.bio_ptr[n_genes] > r.seg.bio_ptr[0] && !r.seg.bio_id))) {
        set_error("the antismash criterion needs the genes' marker domains");
        return GECCO_CRF_EINVAL;
    }
    if (n_genes > 0 && r.gene_ptr[n_genes] > r.gene_ptr[0] && !r.attr_id && !r.attr_id16) {
        set_error("null buffer");
        return GECCO_CRF_EINVAL;
    }
    if (r.attr_id16 && S.model->A > 65536) {
        set_error("16-bit attribute indices need a model with at most 65536 attributes");
        return GECCO_CRF_EINVAL;
    }
    std::lock_guard<std::mutex> lock(S.mu);
    const double t_start = now_s();
    S.stats = SessionStats{};
    S.stats.n_devices = int32_t(S.devs.size());
    if (r.seg_off_out) r.seg_off_out[0] = 0;
    if (!windowed && !viterbi && !full) return GECCO_CRF_OK;
    int prev_device = -1;
    if (hipGetDevice(&prev_device) != hipSuccess) prev_device = -1;

    std::vector<Chunk> chunks;
    // a small batch is ONE chunk on the first device, its kernels working on pinned host memory (no copy commands)
    // (a caller who asked for chunks smaller than the batch gets chunks)
    const bool direct = n_genes > 0 && n_genes <= std::min(S.direct_limit(r.want_segments), S.chunk_genes) && !full && !r.score_out;
    if (direct) {
        Chunk ck;
        ck.c1 = r.n_contigs;
        ck.g1 = ck.u1 = int32_t(n_genes);
        chunks.push_back(std::move(ck));
    } else {
        // cluster calls launch eight more (short) kernels per chunk and download next to nothing: twice the chunk size
        // (a contig too long for one chunk is cut into pieces with a W - 1 gene halo when only its windowed marginals are asked for)
        const bool may_split = windowed && !viterbi && !full && !r.want_segments;
        cut_chunks(r, r.want_segments ? int32_t(std::min<int64_t>(2 * int64_t(S.chunk_genes), 1 << 28)) : S.chunk_genes, int(S.devs.size()),
                   chunks, may_split ? r.window : 0, may_split ? r.step : 1);
        deal_chunks(chunks, int(S.devs.size()));
    }
    S.stats.n_chunks = int32_t(chunks.size());
    S.stats.direct = direct ? 1 : 0;
    RunCtx X{S, r, chunks, windowed, viterbi, full, windowed ? r.window : 1, windowed ? r.step : 1, windowed ? r.pad : 1, direct};
    // (every lane is idle here: start from lane 0 again, so that calls of a chunk or two keep to the lanes whose buffers and
    // workspaces exist already instead of walking the ring and allocating in each of its lanes in turn)
    for (auto &d : S.devs) {
        d->next_lane = 0;
        for (Lane &ln : d->lanes) ln.up_chunk = -1;
    }
    // per-device queues in batch order; devices are fed round-robin so that all of them start at once
    // A chunk without genes (contigs of length 0 only) has nothing to upload, launch or download and never takes a lane: in a
    // decode call its lane would otherwise come round to the one whose labels are still pending without ever recording an
    // event for it.  Per-contig outputs of such contigs are what an empty sequence gives: score 0, log Z 0.
    std::vector<std::vector<int>> queue(S.devs.size());
    for (size_t i = 0; i < chunks.size(); ++i) {
        const Chunk &ck = chunks[i];
        if (ck.g1 > ck.g0) {
            queue[size_t(ck.device_slot)].push_back(int(i));
            continue;
        }
        for (int32_t c = ck.c0; c < ck.c1; ++c) {
            if (r.score_out) r.score_out[c] = 0.0;
            if (r.lognorm_out) r.lognorm_out[c] = 0.0;
        }
    }
    // Every device entry is driven by a thread of its own (the calling thread takes the first, the session's workers the
    // others): a 2^19-gene chunk is ~90 us of PCIe, its submission a dozen HIP API calls and a layout -- one thread feeding
    // eight devices round-robin would be their limiter (crf/__init__.py:244: contigs are independent, so are the queues).
    auto device_job = [&](int d) {
        DeviceCtx &D = *S.devs[size_t(d)];
        const std::vector<int> &q = queue[size_t(d)];
        D.stats = SessionStats{};
        D.err.clear();
        int rc = GECCO_CRF_OK;
        double issue = 0.0;
        for (size_t head = 0; head < q.size() && !rc; ++head) {
            Lane &ln = D.lanes[D.next_lane];
            D.next_lane = (D.next_lane + 1) % kLanes;
            if ((rc = retire(X, ln))) break;  // (waits for the lane's previous chunk: not counted as issue time)
            const double t_issue = now_s();
            const int mine = q[head];
            if (X.direct) {
                if ((rc = submit_direct_arrays(X, ln, mine))) break;
            } else if (ln.up_chunk != mine && (rc = submit_uploads(X, ln, mine))) {
                break;
            }
            if ((rc = submit_plan(X, ln, mine))) break;
            // the next chunk's arrays follow this one's arrays and tables through the copy engine, if the lane they go to is
            // idle -- or has finished meanwhile: it is then retired here instead of at its own turn --: they are on their way
            // while the host launches this chunk and lays out the next
            Lane &nl = D.lanes[D.next_lane];
            if (head + 1 < q.size() && &nl != &ln) {
                if (nl.chunk >= 0 && &nl != D.pending && hipEventQuery(nl.done) == hipSuccess && (rc = retire(X, nl))) break;
                (void)hipGetLastError();  // (hipErrorNotReady of the query is not an error of this call)
                if (nl.chunk < 0 && (rc = submit_uploads(X, nl, q[head + 1]))) break;
            }
            rc = submit(X, D, ln, mine);
            issue += now_s() - t_issue;
        }
        if (!rc) rc = flush_pending(X, D);
        D.pending = nullptr;
        // drain (also after an error: nothing of this call may still be in flight when it returns)
        for (int pass = 0; pass < 2; ++pass)
            for (Lane &ln : D.lanes) {
                if (rc) {  // the failed submission may have left work behind an unrecorded event
                    const std::string keep = last_error();
                    if (hipSetDevice(ln.device) == hipSuccess) (void)hipDeviceSynchronize();
                    ln.chunk = -1;
                    ln.segp_in_flight = false;
                    set_error(keep);
                    continue;
                }
                rc = pass == 0 ? retire_begin(X, ln) : retire_end(X, ln);  // (all downloads are issued before any is waited for)
            }
        D.stats.host_issue_seconds = issue;
        D.rc = rc;
        if (rc) D.err = last_error();  // (the error text is thread-local: handed to the calling thread below)
    };
    const int n_dev = int(S.devs.size());
    int busy = 0;
    for (int d = 0; d < n_dev; ++d) busy += queue[size_t(d)].empty() ? 0 : 1;
    if (busy <= 1 || n_dev == 1) {
        for (int d = 0; d < n_dev; ++d) device_job(d);
        S.stats.host_threads = 1;
    } else {
        S.workers.run(n_dev, device_job);
        S.stats.host_threads = n_dev;
    }
    int rc = GECCO_CRF_OK;
    for (auto &d : S.devs) {
        S.stats.h2d_bytes += d->stats.h2d_bytes;
        S.stats.d2h_bytes += d->stats.d2h_bytes;
        S.stats.host_plan_seconds += d->stats.host_plan_seconds;
        S.stats.host_issue_seconds += d->stats.host_issue_seconds;
        if (d->rc && !rc) {
            rc = d->rc;
            set_error(d->err);
        }
    }
    if (prev_device >= 0) (void)hipSetDevice(prev_device);
    if (rc) return rc;

    if (r.want_segments) {
        int64_t total = 0, genes = 0;
        for (const Chunk &ck : chunks) total += int64_t(ck.rows.size() / 4);
        *r.n_seg = int32_t(std::min<int64_t>(total, INT32_MAX));
        if (total > r.max_seg) {
            set_error("segments: seg_out too small");
            return GECCO_CRF_EINVAL;
        }
        int64_t row = 0;
        for (const Chunk &ck : chunks) {
            const int64_t k = int64_t(ck.rows.size() / 4);
            if (k) std::memcpy(r.seg_out + 4 * row, ck.rows.data(), size_t(k) * 16);
            if (r.seg_p_out) {
                if (genes + ck.seg_p_count > r.max_seg_genes) {
                    set_error("segments: seg_p_out too small");
                    return GECCO_CRF_EINVAL;
                }
                if (!ck.seg_p.empty())
                    std::memcpy(r.seg_p_out + genes, ck.seg_p.data(), size_t(ck.seg_p_count) * 8);
                else if (genes != ck.g0 && ck.seg_p_count)
                    std::memmove(r.seg_p_out + genes, r.seg_p_out + ck.g0, size_t(ck.seg_p_count) * 8);
                for (int64_t i = 0; i < k; ++i) {
                    genes += ck.rows[4 * size_t(i) + 3] - ck.rows[4 * size_t(i) + 2];
                    r.seg_off_out[row + i + 1] = genes;
                }
            }
            row += k;
        }
    }
    S.stats.wall_seconds = now_s() - t_start;
    return GECCO_CRF_OK;
}

}  // namespace gecco
