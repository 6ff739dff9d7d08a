#include "crf_plan.hpp"

#include <algorithm>
#include <functional>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/gecco_crf.h"
#include "crf_session.hpp"

namespace gecco {

int check_hip(hipError_t e, const char *what) {
    if (e == hipSuccess) return GECCO_CRF_OK;
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return GECCO_CRF_ENODEV;
    if (e == hipErrorOutOfMemory) return GECCO_CRF_ENOMEM;
    return GECCO_CRF_EHIP;
}

namespace {
template <class T>
int upload(T **dst, const T *src, size_t n, const char *what) {
    *dst = nullptr;
    if (n == 0) n = 1;  // keep pointers valid
    int rc = check_hip(hipMalloc(reinterpret_cast<void **>(dst), n * sizeof(T)), what);
    if (rc) return rc;
    if (src) rc = check_hip(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice), what);
    return rc;
}
}  // namespace

Model::~Model() {
    for (auto &e : sessions) session_destroy(e.second);  // before the tables their plans point at
    for (DeviceTables *t : dev_tables) {
        if (!t) continue;
        int prev = 0;
        if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(t->device) == hipSuccess) {
            (void)hipFree(t->wtab);
            (void)hipFree(t->wtab2[0]);
            (void)hipFree(t->wtab2[1]);
            (void)hipFree(t->exp_trans);
            (void)hipFree(t->trans);
            (void)hipFree(t->rtab[0]);
            (void)hipFree(t->rtab[1]);
            (void)hipSetDevice(prev);
        }
        delete t;  // without a usable device the allocations die with the context
    }
}

static void free_device_tables(DeviceTables *t) {
    (void)hipFree(t->wtab);
    (void)hipFree(t->wtab2[0]);
    (void)hipFree(t->wtab2[1]);
    (void)hipFree(t->exp_trans);
    (void)hipFree(t->trans);
    (void)hipFree(t->rtab[0]);
    (void)hipFree(t->rtab[1]);
    delete t;
}

int get_device_tables(const Model &m, int device, const DeviceTables **out) {
    std::lock_guard<std::mutex> lock(m.dev_mutex);
    for (DeviceTables *t : m.dev_tables)
        if (t && t->device == device) {
            *out = t;
            return GECCO_CRF_OK;
        }
    int rc = check_hip(hipSetDevice(device), "hipSetDevice");
    if (rc) return rc;
    auto *t = new DeviceTables();
    t->device = device;
    const size_t A = size_t(m.A), L = size_t(m.L);
    std::vector<double> et(L * L);
    for (size_t i = 0; i < L * L; ++i) et[i] = std::exp(m.trans[i]);
    for (double v : m.state) t->wmax_abs = std::max(t->wmax_abs, std::fabs(v));
    for (double v : m.trans) t->tmax_abs = std::max(t->tmax_abs, std::fabs(v));
    rc = upload(&t->wtab, m.state.data(), A * L, "upload state weights");
    if (!rc) rc = upload(&t->exp_trans, et.data(), L * L, "upload transitions");
    if (!rc) rc = upload(&t->trans, m.trans.data(), L * L, "upload transitions");
    if (!rc && m.L == 2) {
        std::vector<double2> w2(A ? A : 1);
        for (int label = 0; label < 2 && !rc; ++label) {
            for (size_t a = 0; a < A; ++a) w2[a] = make_double2(m.state[a * 2 + (1 - label)], m.state[a * 2 + label]);
            rc = upload(&t->wtab2[label], w2.data(), A, "upload state weight pairs");
        }
        // r = mu01 exp(d) in the window kernel (mu_exp_tab): exp(d) = 2^e 2^(j/32) exp(r'), |r'| <= ln2/64; the table holds
        // mu01 2^(j/32), rounded once from extended precision
        for (int label = 0; label < 2 && !rc; ++label) {
            const int o = 1 - label;
            auto T = [&](int i, int j) { return (long double)m.trans[size_t(i) * 2 + j]; };
            const long double lmu = T(o, label) + T(label, o) - 2.0L * T(o, o);
            double tab[32];
            for (int j = 0; j < 32; ++j) tab[j] = double(expl(lmu + (long double)j * 0.693147180559945309417232121458L / 32.0L));
            rc = upload(&t->rtab[label], tab, 32, "upload exp table");
        }
    }
    if (rc) {
        free_device_tables(t);
        return rc;
    }
    if (m.L == 2) {
        for (int label = 0; label < 2; ++label) {
            DeviceTables::WinConsts &c = t->win[label];
            // exp() of differences only: every constant is a ratio of transition weights
            const int o = 1 - label;
            auto T = [&](int i, int j) { return m.trans[size_t(i) * 2 + j]; };
            c.mu01 = std::exp(T(o, label) + T(label, o) - 2.0 * T(o, o));
            c.rho = std::exp(T(label, label) + T(o, o) - T(o, label) - T(label, o));  // mu11 / mu01
            const double kappa = std::exp(T(label, o) - T(o, o));
            c.kappa_over_mu01 = std::exp(T(o, o) - T(o, label));                       // kappa / mu01
            c.inv_kappa = 1.0 / kappa;
            const double mx = *std::max_element(m.trans.begin(), m.trans.end());
            auto G = [&](int i, int j) { return std::exp(m.trans[size_t(i) * 2 + j] - mx); };
            c.g00 = G(o, o);
            c.g01 = G(o, label);
            c.g10 = G(label, o);
            c.g11 = G(label, label);
            fill_exp_coefficients(c.expc);
            const char *env = std::getenv("GECCO_CRF_RATIO");
            c.ratio_zmax = (env && env[0] == '0') ? -1.0 : 1.0e250;
        }
        DeviceTables::SeqConsts &q = t->seq;
        q.mx = *std::max_element(m.trans.begin(), m.trans.end());
        q.m00 = std::exp(m.trans[0] - q.mx);
        q.m01 = std::exp(m.trans[1] - q.mx);
        q.m10 = std::exp(m.trans[2] - q.mx);
        q.m11 = std::exp(m.trans[3] - q.mx);
        const double lo = *std::min_element(m.trans.begin(), m.trans.end());
        q.raw_fold = (q.mx - lo) * double(kSeqGenesPerLane) < 600.0 ? 1 : 0;
        q.v_lo = m.trans[1] - m.trans[3];
        q.v_hi = m.trans[0] - m.trans[2];
        q.v_k = m.trans[3] - m.trans[0];
        const char *env = std::getenv("GECCO_CRF_VD_EXACT");
        q.v_exact = (env && env[0] == '0') ? 0 : 1;
        fill_exp_coefficients(q.expc);
        t->consts_ok = true;
    }
    m.dev_tables.push_back(t);
    *out = t;
    return GECCO_CRF_OK;
}

// hipSetDevice only when the calling thread is on another device (the C ABI's guard restores the caller's device afterwards):
// a launch-bound step used to make four runtime calls for this
static inline int use_device(int device) {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == device) return GECCO_CRF_OK;
    return check_hip(hipSetDevice(device), "hipSetDevice");
}

int Arena::reserve(size_t bytes, const char *what) {
    if (bytes <= cap && h) return GECCO_CRF_OK;
    release();
    const size_t want = std::max<size_t>(bytes + bytes / 4, 4096);  // head room: chunks of one batch differ a little
    // (mapped + portable: the kernels of ANY device of a session may read the block where the host wrote it)
    int rc = check_hip(hipHostMalloc(reinterpret_cast<void **>(&h), want, hipHostMallocMapped | hipHostMallocPortable), what);
    if (rc) return rc;
    if ((rc = check_hip(hipMalloc(reinterpret_cast<void **>(&d), want), what))) {
        (void)hipHostFree(h);
        h = nullptr;
        return rc;
    }
    cap = want;
    return GECCO_CRF_OK;
}

void Arena::release() {
    if (h) (void)hipHostFree(h);
    if (d) (void)hipFree(d);
    h = d = nullptr;
    cap = 0;
}

Plan::~Plan() {
    if (device >= 0) {
        int prev = 0;
        if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(device) == hipSuccess) {
            tables.release();
            seq.release();
            (void)hipFree(d_seq_ws);
            (void)hipFree(d_win_scratch);
            (void)hipFree(d_gen_ws);
            (void)hipFree(gen_tab[0].d);
            (void)hipFree(gen_tab[1].d);
            (void)hipFree(d_seg_ws);
            if (side_stream) (void)hipStreamDestroy(side_stream);
            if (ev_fork) (void)hipEventDestroy(ev_fork);
            if (ev_join) (void)hipEventDestroy(ev_join);
            (void)hipSetDevice(prev);
        }
    }
}

// Steps after which the un-normalised DP vectors are rescaled by a power of two.
// The kernel keeps alpha/beta un-normalised in a transformed basis (crf_kernels.hip):
// emissions max-normalised to (0,1], transitions divided by m00.  With
//   bits  = log2(max(M)/min(M)),  cbits = log2(max(M)/m00)      (M = exp(trans))
// the larger component of alpha~ after k steps lies in [2^-(bits(k+1)), 2^(k(1+cbits)+bits)],
// that of beta~ in [2^-(bits(2k+1)), 2^(k(1+cbits)+bits)] with min/max >= 2^-(2 bits), so a
// candidate pair has max(x,y) in [2^-(bits(3k+4)), 2^(2k(1+cbits)+2bits)].  Comparing two
// candidates multiplies two such numbers; both products stay inside fp64's normal range if
//   bits*(3P+4) <= 500   and   2P(1+cbits) + 2 bits <= 500
// for a rescale period P.  GECCO's embedded model: bits = 7.6, cbits <= 0.15 -> P = 20 >=
// W-1, i.e. no rescale inside a 20-gene window (mask = 0).
static bool rescale_mask_for(const Model &m, int W, uint32_t *mask) {
    *mask = 0;
    double lo = m.trans[0], hi = m.trans[0];
    for (double t : m.trans) {
        lo = std::min(lo, t);
        hi = std::max(hi, t);
    }
    if (!std::isfinite(lo) || !std::isfinite(hi)) return false;
    const double ln2 = std::log(2.0);
    const double bits = (hi - lo) / ln2;
    // m00 is trans[other][other]; `other` is 0 or 1 depending on the queried label
    const double cbits = (hi - std::min(m.trans[0], m.trans[3])) / ln2;
    double pf = (500.0 - 2.0 * bits) / (2.0 * (1.0 + cbits));
    if (bits > 0.0) pf = std::min(pf, (500.0 / bits - 4.0) / 3.0);
    pf = std::floor(pf);
    if (pf < 1.0) return false;
    if (pf >= double(W)) return true;
    const int P = int(pf);
    for (int k = P; k < W && k < 32; k += P) *mask |= (1u << k);
    return true;
}

// bits [lo, hi) of a bit array
static inline void set_bit_range(uint64_t *bits, int64_t lo, int64_t hi) {
    if (lo >= hi) return;
    const int64_t w0 = lo >> 6, w1 = (hi - 1) >> 6;
    const uint64_t m0 = ~0ull << (lo & 63), m1 = ~0ull >> (63 - ((hi - 1) & 63));
    if (w0 == w1) {
        bits[w0] |= m0 & m1;
        return;
    }
    bits[w0] |= m0;
    for (int64_t w = w0 + 1; w < w1; ++w) bits[w] = ~0ull;
    bits[w1] |= m1;
}

namespace {
inline size_t align256p(size_t x) { return (x + 255) & ~size_t(255); }

// grow-only device workspace (hipFree waits for the launches that may still use the old block)
template <class T>
int grow_ws(T *&ptr, size_t &cap, size_t bytes, const char *what) {
    if (ptr && bytes <= cap) return GECCO_CRF_OK;
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
    const size_t want = bytes + bytes / 8 + 256;
    int rc = check_hip(hipMalloc(reinterpret_cast<void **>(&ptr), want), what);
    if (rc) return rc;
    cap = want;
    return GECCO_CRF_OK;
}
}  // namespace

int plan_build(const Model &m, int device, const int32_t *contig_ptr, int32_t n_contigs, int32_t W, int32_t step,
               int32_t pad, Plan &p, hipStream_t upload_stream, bool sync) {
    // same checks, same order as gecco/_meta.py:127-130
    if (W <= 0) {
        set_error("Window size must be strictly positive");
        return GECCO_CRF_EINVAL;
    }
    if (step <= 0 || step > W) {
        set_error("Window step must be strictly positive and under `window_size`");
        return GECCO_CRF_EINVAL;
    }
    if (n_contigs < 0 || (n_contigs > 0 && !contig_ptr)) {
        set_error("bad contig_ptr");
        return GECCO_CRF_EINVAL;
    }
    if (p.device >= 0 && p.device != device) {
        set_error("a plan cannot move to another device");
        return GECCO_CRF_EINVAL;
    }
    p.gen_small = false;
    p.gen_tab[0].chunk = p.gen_tab[1].chunk = 0;  // (chunk tables of the previous contigs)
    p.gen_wave_tmax = p.gen_wave_tmax_f = INT32_MIN;
    p.pipe = Plan::Pipe{};  // (score differences and CSR pointers of the previous layout)
    p.csr_begin = p.csr_end = -1;  // (the owner sets them after the build, for the batch at hand)
    p.model = &m;
    p.device = device;
    p.W = W;
    p.step = step;
    p.pad = pad ? 1 : 0;
    p.n_contigs = n_contigs;
    // a slice of a larger batch is accepted: gene offsets are taken relative to its first contig
    const int64_t g_base = n_contigs ? contig_ptr[0] : 0;
    {
        const int64_t ng = n_contigs ? int64_t(contig_ptr[n_contigs]) - g_base : 0;
        if (ng < 0) {
            set_error("contig_ptr must be non-decreasing");
            return GECCO_CRF_EINVAL;
        }
        p.n_genes = int32_t(ng);
    }
    p.contig_ptr.resize(n_contigs ? size_t(n_contigs) + 1 : 0);
    p.c_slot.clear();
    p.c_gene.clear();
    p.c_n.clear();
    p.skipped.clear();
    p.seq_ready = false;
    p.n_max = 0;
    int64_t S = 0;
    p.n_windows = 0;
    if (n_contigs) p.contig_ptr[0] = 0;
    for (int32_t c = 0; c < n_contigs; ++c) {
        const int32_t g0 = int32_t(contig_ptr[c] - g_base), n = contig_ptr[c + 1] - contig_ptr[c];
        p.contig_ptr[size_t(c) + 1] = g0 + n;
        p.n_max = std::max(p.n_max, n);
        if (n < 0) {
            set_error("contig_ptr must be non-decreasing");
            return GECCO_CRF_EINVAL;
        }
        if (n == 0) continue;  // cannot occur in the reference (groupby never yields empty groups)
        int32_t np = n;
        if (n < W) {
            if (!p.pad) {  // crf/__init__.py:228-234: contig skipped, genes keep "no prediction"
                p.skipped.push_back(make_int2(g0, g0 + n));
                continue;
            }
            np = W;  // :226-227
        }
        p.c_slot.push_back(int32_t(S));
        p.c_gene.push_back(g0);
        p.c_n.push_back(n);
        S += np;
        p.n_windows += np - W + 1;  // :239 (ignores step, like the reference)
        if (S > INT32_MAX - 4096) {
            set_error("batch too large: more than 2^31 padded gene slots");
            return GECCO_CRF_EUNSUPPORTED;
        }
    }
    p.K = int32_t(p.c_gene.size());
    p.S = int32_t(S);
    p.c_slot.push_back(p.S);

    if (m.L < 1 || m.L > kGenMaxL) {
        set_error("models with more than 32 labels are not supported");
        return GECCO_CRF_EUNSUPPORTED;
    }
    {
        const char *env = std::getenv("GECCO_CRF_FORCE_GENERAL");
        p.general = m.L != 2 || (env && env[0] == '1');
    }
    p.fast_ok = (!p.general && W <= kWinMaxW && rescale_mask_for(m, W, &p.rescale_mask));
    {
        static const bool env_reference = [] {
            const char *env = std::getenv("GECCO_CRF_REFERENCE_BITS");
            return env && env[0] == '1';
        }();
        // (the environment switch is for windowed marginals of 2-label models: a Viterbi-only or whole-contig layout, or another
        // label count, keeps its kernels; an explicit request -- reference_bits -- for an unsupported shape is an error)
        p.reference_now = p.reference_bits || (env_reference && p.windowed_use && m.L == 2 && reference_bits_ok(m.L, W));
        if (p.reference_now) {
            if (!reference_bits_ok(m.L, W)) {
                set_error("reference-bits mode serves 2-label models and windows of at most 32 genes");
                return GECCO_CRF_EUNSUPPORTED;
            }
            // (the fast kernels' by-products -- score differences for the decoder, p in host memory -- are theirs alone: the
            // decoder sums its state scores itself, p goes through device memory)
            p.general = false;
            p.fast_ok = false;
        }
    }
    {
        const char *env = std::getenv("GECCO_CRF_FORCE_GENERIC");
        p.force_generic = env && env[0] == '1';
    }
    if (p.force_generic) p.fast_ok = false;
    {
        const char *env = std::getenv("GECCO_CRF_TILES_PER_WG");
        p.tiles_per_wg = (env && env[0] >= '1' && env[0] <= '3' && !env[1]) ? env[0] - '0' : kWinTilesPerWg;
        // A batch that leaves most wave slots of the chip empty runs ONE tile per workgroup: the two tiles of a workgroup are a
        // chain, and with a wave or two per SIMD nothing else hides a tile's latency (C2, 0.2 M genes: window kernel 10.0 -> 7.9 us).
        // A batch of one workgroup keeps two (crf_windowed_small_l2).  GECCO_CRF_TILES1_MAX_SLOTS moves the limit (tests / A/B).
        static const int64_t tiles1_max = [] {
            const char *e = std::getenv("GECCO_CRF_TILES1_MAX_SLOTS");
            return e ? std::atoll(e) : int64_t(kWinTiles1MaxSlots);
        }();
        if (!env && !p.general && W == 20 && p.S > 2 * (kWinThreads - (W - 1)) && p.S <= tiles1_max) p.tiles_per_wg = 1;
        // windows other than GECCO's 20 take the dynamic-W instantiation, whose two-tile form spills scalar registers: one tile
        // (tools/window_size_sweep.py, 1 M genes, two / one tile: W = 5 19.9 / 17.1 us, W = 10 22.9 / 19.2, W = 32 76 / 69)
        if (!env && !p.general && W != 20) p.tiles_per_wg = 1;
    }
    // window-start flags per slot (_meta.py:131: starts at 0, step, 2*step, ... <= n' - W)
    // one zero word in front (slots -64 .. -1: the lead-in of the first workgroup) and kStartBitsTail behind (the reach of
    // the last one): the window kernels index the array with any slot of their reach, unclamped
    constexpr size_t kStartBitsTail = 24;
    p.start_bits.assign(1 + size_t(p.S) / 64 + 2 + kStartBitsTail, 0);
    uint64_t *const start_bits = p.start_bits.data() + 1;  // word of slot 0
    // irregular[k]: contig k is padded, or a skipped contig lies between it and contig k+1
    std::vector<int32_t> &irr_prefix = p.irr_prefix;
    irr_prefix.assign(size_t(p.K) + 1, 0);
    for (int32_t k = 0; k < p.K; ++k) {
        const int32_t s0 = p.c_slot[k], np = p.c_slot[k + 1] - s0;
        if (step == 1) {
            set_bit_range(start_bits, s0, int64_t(s0) + np - W + 1);
        } else {
            for (int32_t pos = 0; pos + W <= np; pos += step) {
                const int64_t q = int64_t(s0) + pos;
                start_bits[size_t(q >> 6)] |= 1ull << (q & 63);
            }
        }
        const bool padded = np != p.c_n[k];
        const bool gap_after = k + 1 < p.K && p.c_gene[k + 1] != p.c_gene[k] + p.c_n[k];
        irr_prefix[k + 1] = irr_prefix[k] + ((padded || gap_after) ? 1 : 0);
    }
    // slot space = gene space everywhere: no contig padded, none skipped (empty contigs take no slots and no genes)
    p.all_regular = p.skipped.empty() && p.S == p.n_genes && irr_prefix[size_t(p.K)] == 0;
    if (p.general && m.L >= 3 && gen_small_ok(m.L, W, m.trans.data()) && !std::getenv("GECCO_CRF_GENERAL_GROUPS")) {
        // a handful of labels: one lane per window start (GECCO_CRF_GENERAL_GROUPS=1: the lane-group kernel, tests / A/B)
        p.kernel_name = m.L > 8 ? "gl_windowed_mfma" : "gl_windowed_small";  // (9 to 32 labels: sixteen windows per wave on the matrix cores)
        p.gen_small = true;
        p.tile_out = gen_small_tile_out(W);
    } else {
        p.kernel_name = p.reference_now ? "crf_windowed_reference_l2" : p.general ? "gl_windowed" : windowed_kernel_name(W, m.L, p.fast_ok);
        p.tile_out = windowed_tile_out(W, m.L, p.tiles_per_wg);
    }
    p.ntiles = p.S > 0 ? (p.S + p.tile_out - 1) / p.tile_out : 0;
    p.tile_desc.resize(p.ntiles);
    {
        // contigs in reach of a workgroup: both ends of the reach only move forward from tile to tile
        int first = 0, last = 0;
        for (int32_t b = 0; b < p.ntiles; ++b) {
            const int64_t q0 = int64_t(b) * p.tile_out - (W - 1);
            const int64_t q_lo = std::max<int64_t>(q0, 0);
            const int64_t q_hi = std::min<int64_t>(q0 + p.tile_out + 2 * (W - 1) - 1, p.S - 1);
            while (first + 1 < p.K && p.c_slot[first + 1] <= q_lo) ++first;  // largest k with c_slot[k] <= q_lo
            if (last < first) last = first;
            while (last + 1 < p.K && p.c_slot[last + 1] <= q_hi) ++last;
            // regular: no padded contig in reach and no skipped contig between the contigs in reach
            // (a gap after the last contig is harmless)
            const bool padded_or_gap =
                (irr_prefix[last] - irr_prefix[first]) != 0 || (p.c_slot[last + 1] - p.c_slot[last] != p.c_n[last]);
            const int shift = p.c_gene[first] - p.c_slot[first];
            p.tile_desc[b] = make_int4(shift, first, last, padded_or_gap ? 0 : 1);
        }
    }
    if (device < 0) return GECCO_CRF_OK;

    int rc = check_hip(hipSetDevice(device), "hipSetDevice");
    if (rc) return rc;
    if ((rc = get_device_tables(m, device, &p.tables_model))) return rc;
    // one pinned block -> one device block, one copy
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t r = off;
        off += align256p(bytes ? bytes : 1);
        return r;
    };
    const size_t o_slot = take(p.c_slot.size() * 4), o_gene = take(p.c_gene.size() * 4), o_n = take(p.c_n.size() * 4),
                 o_tile = take(p.tile_desc.size() * sizeof(int4)), o_bits = take(p.start_bits.size() * 8),
                 o_skip = take(p.skipped.size() * sizeof(int2)), o_cptr = take(p.contig_ptr.size() * 4);
    if ((rc = p.tables.reserve(off, "plan tables"))) return rc;
    char *h = p.tables.h, *d = p.tables.d;
    if (p.tables_in_host_memory) {
        void *dv = nullptr;
        if ((rc = check_hip(hipHostGetDevicePointer(&dv, h, 0), "hipHostGetDevicePointer"))) return rc;
        d = static_cast<char *>(dv);
    }
    auto put = [&](size_t o, const void *src, size_t bytes) {
        if (bytes) std::memcpy(h + o, src, bytes);
    };
    put(o_slot, p.c_slot.data(), p.c_slot.size() * 4);
    put(o_gene, p.c_gene.data(), p.c_gene.size() * 4);
    put(o_n, p.c_n.data(), p.c_n.size() * 4);
    put(o_tile, p.tile_desc.data(), p.tile_desc.size() * sizeof(int4));
    put(o_bits, p.start_bits.data(), p.start_bits.size() * 8);
    put(o_skip, p.skipped.data(), p.skipped.size() * sizeof(int2));
    put(o_cptr, p.contig_ptr.data(), p.contig_ptr.size() * 4);
    p.d_c_slot = reinterpret_cast<int32_t *>(d + o_slot);
    p.d_c_gene = reinterpret_cast<int32_t *>(d + o_gene);
    p.d_c_n = reinterpret_cast<int32_t *>(d + o_n);
    p.d_tile_desc = reinterpret_cast<int4 *>(d + o_tile);
    p.d_start_bits = reinterpret_cast<uint64_t *>(d + o_bits) + 1;  // (the word of slot 0)
    p.d_skipped = reinterpret_cast<int2 *>(d + o_skip);
    p.d_contig_ptr = reinterpret_cast<int32_t *>(d + o_cptr);
    if (p.tables_in_host_memory) return GECCO_CRF_OK;
    if (p.tables_by_kernel && !sync) {
        void *dv = nullptr;
        if ((rc = check_hip(hipHostGetDevicePointer(&dv, h, 0), "hipHostGetDevicePointer"))) return rc;
        return check_hip(launch_copy_block(dv, d, off, upload_stream), "plan tables launch");
    }
    if ((rc = check_hip(hipMemcpyAsync(d, h, off, hipMemcpyHostToDevice, upload_stream), "upload plan tables"))) return rc;
    if (sync && (rc = check_hip(hipStreamSynchronize(upload_stream), "upload plan tables"))) return rc;
    return GECCO_CRF_OK;
}

// ---- any number of labels (crf_general.hip) ------------------------------------------------
namespace {
inline size_t align256g(size_t x) { return (x + 255) & ~size_t(255); }

// A contig longer than this sends the whole batch through the chunked whole-contig kernels (crf_general.hip):
// below it, one group of lanes per contig walking it sequentially is the cheaper arrangement.
constexpr int32_t kGenLongContig = 2048;

// chunk_min_len >= 0: only contigs LONGER than that get chunks (the second table set); -1: every contig
int fill_gen_args(Plan &p, const int32_t *d_gene_ptr, const int32_t *d_attr_id, GenArgs &a, bool whole_contig = false,
                  hipStream_t stream = nullptr, int32_t chunk_min_len = -1) {
    const Model &m = *p.model;
    const size_t n = size_t(p.n_genes), L = size_t(m.L);
    const size_t b_vec = align256g(n * L * 8 + 8), b_one = align256g(n * 8 + 8), b_back = align256g(n * L + 8);
    // Whole-contig recursions go through the chunked kernels for EVERY batch (round 3): a contig-sequential group of lanes
    // is a dependent chain as long as the contig, and a batch of 200-gene contigs has three to seven chunks' worth of
    // parallelism inside every contig (1 000 contigs x 200 genes, L = 3: Viterbi 0.40 -> 1.5 G genes/s, marginals 0.17 ->
    // 1.1 G).  Chunks of 32 genes for batches of short contigs, 64 when a contig is longer than kGenLongContig (the walk
    // over a contig's chunks is the serial part).  GECCO_CRF_GENERAL_CHUNKED=0: the contig-sequential kernels (tests).
    bool chunked = whole_contig;
    if (whole_contig)
        if (const char *env = std::getenv("GECCO_CRF_GENERAL_CHUNKED")) chunked = env[0] == '1';  // tests: force either path
    size_t nch = 0;
    if (chunked) {
        // the chunk tables depend on the plan's contigs and the chunk length only: built and uploaded once
        bool has_long = false;
        for (int32_t c = 0; c < p.n_contigs && !has_long; ++c) has_long = p.contig_ptr[c + 1] - p.contig_ptr[c] > kGenLongContig;
        int32_t C = has_long ? gen_chunk_genes() : gen_chunk_genes() / 2;
        if (const char *env = std::getenv("GECCO_CRF_GENERAL_CHUNK")) {
            const int v = std::atoi(env);
            if (v >= 4 && v <= 4096) C = v;
        }
        std::lock_guard<std::mutex> lock(p.ws_mutex);
        Plan::GenTab &tab = p.gen_tab[chunk_min_len >= 0 ? 1 : 0];
        if (!tab.d || tab.chunk != C || tab.min_len != chunk_min_len) {
            std::vector<int32_t> ch_g0, ch_contig, cc_ptr;
            cc_ptr.push_back(0);
            for (int32_t c = 0; c < p.n_contigs; ++c) {
                if (p.contig_ptr[c + 1] - p.contig_ptr[c] > chunk_min_len) {
                    for (int32_t g = p.contig_ptr[c]; g < p.contig_ptr[c + 1]; g += C) {
                        ch_g0.push_back(g);
                        ch_contig.push_back(c);
                    }
                }
                cc_ptr.push_back(int32_t(ch_g0.size()));
            }
            ch_g0.push_back(p.n_genes);  // (a chunk ends where the next one starts or where its contig does: gl_chunk_end)
            const size_t n_ch = ch_contig.size();
            const size_t o1 = align256g(ch_g0.size() * 4), o2 = o1 + align256g(n_ch * 4 + 4), total = o2 + align256g(cc_ptr.size() * 4);
            if (tab.d) (void)hipFree(tab.d);
            tab.d = nullptr;
            tab.chunk = 0;
            int rc = check_hip(hipMalloc(reinterpret_cast<void **>(&tab.d), total), "hipMalloc chunk tables");
            if (rc) return rc;
            std::vector<char> img(total, 0);
            std::memcpy(img.data(), ch_g0.data(), ch_g0.size() * 4);
            std::memcpy(img.data() + o1, ch_contig.data(), n_ch * 4);
            std::memcpy(img.data() + o2, cc_ptr.data(), cc_ptr.size() * 4);
            if ((rc = check_hip(hipMemcpy(tab.d, img.data(), total, hipMemcpyHostToDevice), "upload chunk tables"))) return rc;
            tab.chunk = C;
            tab.min_len = chunk_min_len;
            tab.nch = n_ch;
            tab.off1 = o1;
            tab.off2 = o2;
        }
        nch = tab.nch;
    }
    const size_t b_chM = align256g(nch * L * L * 8 + 8), b_chv = align256g(nch * L * 8 + 8), b_chs = align256g(nch * 8 + 8);
    const size_t b_chunks = chunked ? b_chM + 3 * b_chv + 2 * b_chs + 2 * align256g(nch * L + 8) : 0;
    {
        std::lock_guard<std::mutex> lock(p.ws_mutex);
        int rc = grow_ws(p.d_gen_ws, p.gen_ws_cap, 3 * b_vec + 2 * b_one + b_back + b_chunks, "hipMalloc general-L workspace");
        if (rc) return rc;
    }
    char *w = p.d_gen_ws;
    a = GenArgs{};
    a.gene_ptr = d_gene_ptr;
    a.attr_id = d_attr_id;
    a.wtab = p.tables_model->wtab;
    a.exp_trans = p.tables_model->exp_trans;
    a.trans = p.tables_model->trans;
    a.contig_ptr = p.d_contig_ptr;
    a.L = m.L;
    a.A = m.A;
    a.n_genes = p.n_genes;
    a.n_contigs = p.n_contigs;
    a.rows_rescale_period = 4.0 * p.tables_model->tmax_abs < 600.0 ? 4 : 1;
    a.state = reinterpret_cast<double *>(w);
    a.E = reinterpret_cast<double *>(w + b_vec);
    a.alpha = reinterpret_cast<double *>(w + 2 * b_vec);
    a.smax = reinterpret_cast<double *>(w + 3 * b_vec);
    a.scale = reinterpret_cast<double *>(w + 3 * b_vec + b_one);
    a.back = reinterpret_cast<uint8_t *>(w + 3 * b_vec + 2 * b_one);
    if (chunked && nch) {
        char *q = w + 3 * b_vec + 2 * b_one + b_back;
        const Plan::GenTab &tab = p.gen_tab[chunk_min_len >= 0 ? 1 : 0];
        a.ch_g0 = reinterpret_cast<const int32_t *>(tab.d);
        a.ch_contig = reinterpret_cast<const int32_t *>(tab.d + tab.off1);
        a.cc_ptr = reinterpret_cast<const int32_t *>(tab.d + tab.off2);
        a.n_chunks = int32_t(nch);
        a.chM = reinterpret_cast<double *>(q);
        q += b_chM;
        a.chV = reinterpret_cast<double *>(q);
        q += b_chv;
        a.chB = reinterpret_cast<double *>(q);
        q += b_chv;
        a.chEx = reinterpret_cast<int32_t *>(q);
        q += b_chv;
        a.chZ = reinterpret_cast<double *>(q);
        q += 2 * b_chs;
        a.chMap = reinterpret_cast<uint8_t *>(q);
        q += align256g(nch * L + 8);
        a.chY = reinterpret_cast<int8_t *>(q);
    }
    a.c_slot = p.d_c_slot;
    a.c_gene = p.d_c_gene;
    a.c_n = p.d_c_n;
    a.start_bits = p.d_start_bits;
    a.K = p.K;
    a.S = p.S;
    a.W = p.W;
    return GECCO_CRF_OK;
}

// Whole-contig recursions of 9 to 32 labels: which contigs go to the wave-per-contig kernel (crf_general.hip).  The waves take as
// long as the longest contig they are given (t_step us per gene: a lone wave issues an instruction every four cycles); the
// chunked kernels pay L x the arithmetic for every gene they are given (t_gene us), a walk over the chunks of their longest
// contig (t_walk us per gene) and a chain of launches (t_launches us).  So the batch is SPLIT: the k longest contigs -- the
// tail of a metagenome's length distribution -- go through the chunked kernels on the plan's side stream, NEXT TO the waves
// of the others, with k minimising max(t_waves(longest of the others), t_chunked(the k longest)).
// Returns -1: chunked kernels for everything; 0: waves for everything; > 0: waves for contigs up to this length.
// `env`: wave | chunked | split (k = 1) forces; `automatic`: whether the model's label count is one the choice is made for;
// `cache`: the choice for this layout (INT32_MIN: not made yet).
int32_t choose_wave_split(const Plan &p, const char *env, bool automatic, double t_step, double t_gene, double t_walk, double t_launches,
                          int32_t &cache) {
    const int forced = !env ? -1 : env[0] == 'w' ? 1 : env[0] == 'c' ? 0 : env[0] == 's' ? 2 : -1;
    if (forced == 1) return 0;
    if (forced == 0 || (forced < 0 && !automatic)) return -1;
    if (forced < 0 && cache != INT32_MIN) return cache;  // (the sort below is the host's only loop over the contigs)
    int32_t wave_tmax = -1;
    std::vector<int32_t> len(size_t(p.n_contigs));
    for (int32_t c = 0; c < p.n_contigs; ++c) len[size_t(c)] = p.contig_ptr[c + 1] - p.contig_ptr[c];
    std::sort(len.begin(), len.end(), std::greater<int32_t>());
    const double all_chunked = double(p.n_genes) * t_gene + double(len[0]) * t_walk + t_launches;
    double best = all_chunked;
    int64_t tail = 0;
    for (int32_t k = 0; k < p.n_contigs; ++k) {  // the k longest contigs chunked NEXT TO the waves of the others
        const double t_tail = k ? double(tail) * t_gene + double(len[0]) * t_walk + t_launches : 0.0;
        const double t = std::max(double(len[size_t(k)]) * t_step, t_tail);
        if (t < best || (forced == 2 && k == 1)) {
            best = t;
            wave_tmax = k ? len[size_t(k)] : 0;  // (contigs as long as the k-th longest stay with the waves)
            if (forced == 2 && k == 1) break;
        }
        tail += len[size_t(k)];
        if (double(tail) * t_gene + t_launches > best) break;  // (cannot get better from here)
    }
    if (wave_tmax > 0 && len[0] <= wave_tmax) wave_tmax = 0;  // (nothing is longer: no tail)
    if (forced < 0) cache = wave_tmax;
    return wave_tmax;
}

// fork the chunked kernels of a batch's long tail onto the plan's side stream (behind what `stream` holds so far); join_tail
// makes `stream` wait for them
int fork_tail(Plan &p, hipStream_t stream) {
    int rc;
    if (!p.side_stream) {
        if ((rc = check_hip(hipStreamCreateWithFlags(&p.side_stream, hipStreamNonBlocking), "hipStreamCreate"))) return rc;
        if ((rc = check_hip(hipEventCreateWithFlags(&p.ev_fork, hipEventDisableTiming), "hipEventCreate"))) return rc;
        if ((rc = check_hip(hipEventCreateWithFlags(&p.ev_join, hipEventDisableTiming), "hipEventCreate"))) return rc;
    }
    if ((rc = check_hip(hipEventRecord(p.ev_fork, stream), "hipEventRecord"))) return rc;
    return check_hip(hipStreamWaitEvent(p.side_stream, p.ev_fork, 0), "hipStreamWaitEvent");
}
int join_tail(Plan &p, hipStream_t stream) {
    int rc = check_hip(hipEventRecord(p.ev_join, p.side_stream), "hipEventRecord");
    if (rc) return rc;
    return check_hip(hipStreamWaitEvent(stream, p.ev_join, 0), "hipStreamWaitEvent");
}

int run_windowed_general(Plan &p, const int32_t *d_gene_ptr, const int32_t *d_attr_id, int32_t label, double *d_p_out,
                         hipStream_t stream) {
    if (p.W > kGenMaxW) {
        set_error("window too long for the any-L kernel (alpha of a whole window is LDS-resident: W <= 48)");
        return GECCO_CRF_EUNSUPPORTED;
    }
    GenArgs a;
    int rc = fill_gen_args(p, d_gene_ptr, d_attr_id, a);
    if (rc) return rc;
    a.p_out = d_p_out;
    a.label = label;
    // atomic-max accumulation starts from 0.0 (numpy.zeros, crf/__init__.py:251)
    if ((rc = check_hip(hipMemsetAsync(d_p_out, 0, size_t(p.n_genes) * 8, stream), "memset p"))) return rc;
    if (!p.skipped.empty())
        if ((rc = check_hip(launch_fill_nan(d_p_out, p.d_skipped, int(p.skipped.size()), stream), "fill_nan launch")))
            return rc;
    double *keep_state = a.state;
    a.state = nullptr;  // marginals only need exp(state - max)
    if ((rc = check_hip(launch_gen_state(a, stream), "state score launch"))) return rc;
    a.state = keep_state;
    if (p.gen_small)
        return check_hip(launch_gen_windowed_small(a, p.model->trans.data(), p.d_tile_desc, p.ntiles, stream), "windowed launch");
    return check_hip(launch_gen_windowed(a, stream), "windowed launch");
}
}  // namespace

// the launch also carries the Viterbi workgroups of `seq`, which belongs to the PREVIOUS batch (crf_decode_pipelined: nothing
// is exchanged inside the launch)
struct PipelinedLaunch {
    const SeqArgs *seq;
    bool *took = nullptr;  // set when the one launch was made (otherwise the caller launches the Viterbi side itself)
};
static int run_windowed_impl(Plan &p, const int32_t *d_gene_ptr, const int32_t *d_attr_id, int32_t label, double *d_p_out,
                             double2 *d_state_out, double *d_dstate_out, hipStream_t stream, const PipelinedLaunch *piped = nullptr);

int plan_run_windowed(Plan &p, const int32_t *d_gene_ptr, const int32_t *d_attr_id, int32_t label, double *d_p_out,
                      hipStream_t stream) {
    return run_windowed_impl(p, d_gene_ptr, d_attr_id, label, d_p_out, nullptr, nullptr, stream);
}

static int run_windowed_impl(Plan &p, const int32_t *d_gene_ptr, const int32_t *d_attr_id, int32_t label, double *d_p_out,
                             double2 *d_state_out, double *d_dstate_out, hipStream_t stream, const PipelinedLaunch *piped) {
    if (p.device < 0) {
        set_error("host-only plan: no HIP device bound (there is no CPU fallback)");
        return GECCO_CRF_ENODEV;
    }
    if (label < 0 || label >= p.model->L) {
        set_error("label out of range");
        return GECCO_CRF_EINVAL;
    }
    if (p.n_genes == 0) return GECCO_CRF_OK;
    if (!d_gene_ptr || !d_p_out) {
        set_error("null device buffer");
        return GECCO_CRF_EINVAL;
    }
    int rc = use_device(p.device);
    if (rc) return rc;
    const Model &m = *p.model;
    if (p.general) return run_windowed_general(p, d_gene_ptr, d_attr_id, label, d_p_out, stream);
    WinArgs a{};
    a.gene_ptr = d_gene_ptr;
    a.attr_id = d_attr_id;
    a.wtab = p.tables_model->wtab;
    a.wtab2 = p.tables_model->wtab2[label];
    a.exp_trans = p.tables_model->exp_trans;
    a.rtab = p.tables_model->rtab[label];
    a.c_slot = p.d_c_slot;
    a.c_gene = p.d_c_gene;
    a.c_n = p.d_c_n;
    a.tile_desc = p.d_tile_desc;
    a.start_bits = p.d_start_bits;
    a.p_out = d_p_out;
    a.state_out = d_state_out;
    a.dstate_out = d_dstate_out;
    a.K = p.K;
    a.S = p.S;
    a.ntiles = p.ntiles;
    a.W = p.W;
    a.step = p.step;
    a.L = m.L;
    a.label = label;
    a.n_genes = p.n_genes;
    a.A = m.A;
    a.tiles_per_wg = p.tiles_per_wg;
    a.all_regular = p.all_regular ? 1 : 0;
    a.rescale_mask = p.rescale_mask;
    {
        const DeviceTables::WinConsts &c = p.tables_model->win[label];  // (model constants, computed once per device: get_device_tables)
        a.mu01 = c.mu01;
        a.rho = c.rho;
        a.kappa_over_mu01 = c.kappa_over_mu01;
        a.inv_kappa = c.inv_kappa;
        a.g00 = c.g00;
        a.g01 = c.g01;
        a.g10 = c.g10;
        a.g11 = c.g11;
        std::memcpy(a.expc, c.expc, sizeof(a.expc));
        a.ratio_zmax = c.ratio_zmax;
    }
    a.csr_begin = int32_t(p.csr_begin);  // (row pointers are 32-bit)
    a.csr_end = int32_t(p.csr_end);
    if (p.reference_now) {
        {
            std::lock_guard<std::mutex> lock(p.ws_mutex);
            if ((rc = grow_ws(p.d_win_scratch, p.win_scratch_cap, reference_scratch_bytes(p.n_genes), "hipMalloc reference scratch"))) return rc;
        }
        // (every gene of slot space is stored once by the tile that owns its slot: nothing to zero)
        if (!p.skipped.empty())
            if ((rc = check_hip(launch_fill_nan(d_p_out, p.d_skipped, int(p.skipped.size()), stream), "fill_nan launch"))) return rc;
        double et[4];
        for (int i = 0; i < 4; ++i) et[i] = std::exp(m.trans[size_t(i)]);  // (the host's libm, as the reference's CRFsuite calls it)
        return check_hip(launch_windowed_reference(a, p.tables_model->wtab2[1], et, p.d_win_scratch, stream), "reference-bits launch");
    }
    a.generic = p.fast_ok ? 0 : 1;
    if (a.generic) {
        {
            std::lock_guard<std::mutex> lock(p.ws_mutex);
            if ((rc = grow_ws(p.d_win_scratch, p.win_scratch_cap, size_t(p.S) * size_t(p.W) * 16 + 16, "hipMalloc window scratch")))
                return rc;
        }
        a.scratch = p.d_win_scratch;
        // atomic-max accumulation starts from 0.0 (numpy.zeros, crf/__init__.py:251)
        if ((rc = check_hip(hipMemsetAsync(d_p_out, 0, size_t(p.n_genes) * 8, stream), "memset p"))) return rc;
    }
    if (!p.skipped.empty())
        if ((rc = check_hip(launch_fill_nan(d_p_out, p.d_skipped, int(p.skipped.size()), stream), "fill_nan launch")))
            return rc;
    if (piped && !a.generic && decode_pipelined_ok(a, *piped->seq)) {
        *piped->took = true;
        return check_hip(launch_decode_pipelined(a, *piped->seq, stream), "pipelined decode launch");
    }
    return check_hip(launch_windowed(a, stream), "windowed launch");
}

// ---- whole-contig scans (rows F, V) ------------------------------------------------------
namespace {
inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

struct SeqLayout {
    size_t lanes, blocks, bytes;
    size_t off_state, off_alpha, off_tmp, off_vlane, off_vblock, off_vmaps, off_vlanemap, off_vblockmap, off_flane,
        off_fblock, off_flanesuf, off_fblocksuf, off_cand, off_stats;
};
SeqLayout seq_layout(size_t n) {
    SeqLayout l{};
    l.lanes = (n + kSeqGenesPerLane - 1) / kSeqGenesPerLane;
    l.blocks = (l.lanes + 255) / 256;
    const size_t lanes_pad = l.blocks * 256;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t r = o;
        o += align256(bytes);
        return r;
    };
    l.off_stats = take(64);  // (first: the decoder's counters keep their place when the batch size changes)
    l.off_state = take((n + kSeqGenesPerLane) * 16);
    l.off_alpha = take(n * 16);
    l.off_tmp = take(n * 16);
    l.off_vlane = take(lanes_pad * sizeof(VE));
    l.off_vblock = take(l.blocks * sizeof(VE));
    l.off_vmaps = take(lanes_pad * 4);
    l.off_vlanemap = take(lanes_pad * 4);
    l.off_vblockmap = take(l.blocks * 4);
    l.off_flane = take(lanes_pad * sizeof(FE));
    l.off_fblock = take(l.blocks * sizeof(FE));
    l.off_flanesuf = take(lanes_pad * sizeof(FE));
    l.off_fblocksuf = take(l.blocks * sizeof(FE));
    l.off_cand = take((lanes_pad + 4 * l.blocks + 16) * sizeof(SeqArgs::VdCand));
    l.bytes = o + 256;
    return l;
}
// (the per-contig "decode again" flags of the long-contig Viterbi path live behind the scan workspace)

}  // namespace

int plan_ensure_seq(Plan &p, hipStream_t stream, bool sync) {
    std::lock_guard<std::mutex> lock(p.ws_mutex);
    if (p.seq_ready) return GECCO_CRF_OK;
    if (p.device < 0) {
        set_error("host-only plan: no HIP device bound (there is no CPU fallback)");
        return GECCO_CRF_ENODEV;
    }
    int rc = use_device(p.device);
    if (rc) return rc;
    const size_t n = size_t(p.n_genes);
    // short contigs (none longer than a scan block): whole contigs are packed into workgroups of <= 2048 genes
    p.seq_short = true;
    for (int32_t c = 0; c < p.n_contigs && p.seq_short; ++c)
        if (p.contig_ptr[c + 1] - p.contig_ptr[c] > kSeqBlockGenes) p.seq_short = false;
    std::vector<int32_t> &cblk = p.irr_prefix;  // scratch
    cblk.clear();
    if (p.seq_short && n) {
        int32_t start = 0;
        cblk.push_back(0);
        for (int32_t c = 0; c < p.n_contigs; ++c) {
            const int32_t g1 = p.contig_ptr[c + 1];
            if (g1 - start > kSeqBlockGenes) {  // contig c does not fit any more: close the workgroup before it
                start = p.contig_ptr[c];
                cblk.push_back(start);
            }
        }
        cblk.push_back(int32_t(n));
    }
    p.n_cblocks = cblk.empty() ? 0 : int32_t(cblk.size()) - 1;
    // non-empty contigs: their indices, and how many of them precede every workgroup (log Z is written per contig)
    std::vector<int32_t> ne, rank;
    p.n_empty_contigs = 0;
    if (p.n_cblocks) {
        size_t b = 0;
        rank.assign(size_t(p.n_cblocks), 0);
        for (int32_t c = 0; c < p.n_contigs; ++c) {
            if (p.contig_ptr[c + 1] == p.contig_ptr[c]) {
                ++p.n_empty_contigs;
                continue;
            }
            while (b < size_t(p.n_cblocks) && cblk[b] <= p.contig_ptr[c]) {  // workgroups starting at or before this contig
                if (cblk[b] == p.contig_ptr[c]) rank[b] = int32_t(ne.size());
                ++b;
            }
            ne.push_back(c);
        }
    }
    const size_t n_lane_bits = size_t(p.n_cblocks) * 256;  // short contigs: per lane of every workgroup, which of its 8 genes start / end a contig
    // uploaded: the workgroup tables, the lane bits and a copy of the contig table; NOT uploaded: the byte per gene that
    // says "first / last gene of its contig" -- a launch behind the copy derives it from the contig table on the device
    // (two thirds of the block's bytes: 0.5 of 0.77 MB per half-million-gene chunk of the batch driver)
    const size_t o_blk = 0, o_rank = o_blk + align256p((cblk.size() + 1) * 4),
                 o_ne = o_rank + align256p((rank.size() + 1) * 4), o_lb = o_ne + align256p((ne.size() + 1) * 4),
                 o_fb = o_lb + align256p(n_lane_bits * 2 + 2), n_flat_bits = ((n + kSeqBlockGenes - 1) / kSeqBlockGenes) * 256,
                 o_cp = o_fb + align256p(n_flat_bits * 2 + 2), bytes = o_cp + align256p((size_t(p.n_contigs) + 1) * 4),
                 o_flags = bytes, all_bytes = o_flags + align256p(n + 16);
    if ((rc = p.seq.reserve(all_bytes, "contig flags"))) return rc;
    if (p.n_contigs) std::memcpy(p.seq.h + o_cp, p.contig_ptr.data(), (size_t(p.n_contigs) + 1) * 4);
    if (n_flat_bits) {
        // the flat layout (long contigs): lane l of the batch owns genes 8 l .. 8 l + 7
        uint16_t *fb = reinterpret_cast<uint16_t *>(p.seq.h + o_fb);
        std::memset(fb, 0, n_flat_bits * 2);
        for (int32_t c = 0; c < p.n_contigs; ++c) {
            const int32_t g0 = p.contig_ptr[c], g1 = p.contig_ptr[c + 1];
            if (g1 <= g0) continue;
            fb[size_t(g0 / kSeqGenesPerLane)] |= uint16_t(1u << (g0 % kSeqGenesPerLane));
            fb[size_t((g1 - 1) / kSeqGenesPerLane)] |= uint16_t(0x100u << ((g1 - 1) % kSeqGenesPerLane));
        }
        size_t l = n;  // positions past the last gene: one-gene contigs
        for (; l < n_flat_bits * kSeqGenesPerLane && l % kSeqGenesPerLane; ++l)
            fb[l / kSeqGenesPerLane] |= uint16_t(0x101u << (l % kSeqGenesPerLane));
        for (; l < n_flat_bits * kSeqGenesPerLane; l += kSeqGenesPerLane) fb[l / kSeqGenesPerLane] = 0xffff;
    }
    if (n_lane_bits) {
        // bit k of the low byte: gene k of the lane is the first of its contig; of the high byte: the last
        uint16_t *lb = reinterpret_cast<uint16_t *>(p.seq.h + o_lb);
        std::memset(lb, 0, n_lane_bits * 2);
        size_t b = 0;
        for (int32_t c = 0; c < p.n_contigs; ++c) {
            const int32_t g0 = p.contig_ptr[c], g1 = p.contig_ptr[c + 1];
            if (g1 <= g0) continue;
            while (cblk[b + 1] <= g0) ++b;  // the workgroup that holds the (whole) contig
            const int32_t l0 = g0 - cblk[b], l1 = g1 - 1 - cblk[b];
            lb[b * 256 + size_t(l0 / kSeqGenesPerLane)] |= uint16_t(1u << (l0 % kSeqGenesPerLane));
            lb[b * 256 + size_t(l1 / kSeqGenesPerLane)] |= uint16_t(0x100u << (l1 % kSeqGenesPerLane));
        }
        // positions past a workgroup's last gene: one-gene contigs (the kernels give them a score difference that decides
        // nothing), so that the hot loops run over all 8 positions of every lane without testing
        for (int32_t blk = 0; blk < p.n_cblocks; ++blk) {
            const int32_t nb = cblk[size_t(blk) + 1] - cblk[size_t(blk)];
            int32_t l = nb;
            for (; l < kSeqBlockGenes && l % kSeqGenesPerLane; ++l)
                lb[size_t(blk) * 256 + size_t(l / kSeqGenesPerLane)] |= uint16_t(0x101u << (l % kSeqGenesPerLane));
            for (; l < kSeqBlockGenes; l += kSeqGenesPerLane) lb[size_t(blk) * 256 + size_t(l / kSeqGenesPerLane)] = 0xffff;
        }
    }
    if (!cblk.empty()) std::memcpy(p.seq.h + o_blk, cblk.data(), cblk.size() * 4);
    if (!rank.empty()) std::memcpy(p.seq.h + o_rank, rank.data(), rank.size() * 4);
    if (!ne.empty()) std::memcpy(p.seq.h + o_ne, ne.data(), ne.size() * 4);
    char *base = p.seq.d;
    if (p.seq_in_host_memory) {
        // a small batch: the decoder reads its tables (a few hundred bytes) from the pinned block itself and the host writes the
        // flag bytes -- nothing is copied, nothing is launched in front of the decoder
        uint8_t *fl = reinterpret_cast<uint8_t *>(p.seq.h + o_flags);
        std::memset(fl, 0, n + 16);
        for (int32_t c = 0; c < p.n_contigs; ++c) {
            const int32_t g0 = p.contig_ptr[c], g1 = p.contig_ptr[c + 1];
            if (g1 <= g0) continue;
            fl[g0] |= 1u;
            fl[g1 - 1] |= 2u;
        }
        void *dv = nullptr;
        if ((rc = check_hip(hipHostGetDevicePointer(&dv, p.seq.h, 0), "hipHostGetDevicePointer"))) return rc;
        base = static_cast<char *>(dv);
    }
    p.d_seq_flags = reinterpret_cast<uint8_t *>(base + o_flags);
    p.d_seq_cblk = reinterpret_cast<int32_t *>(base + o_blk);
    p.d_seq_cblk_rank = reinterpret_cast<int32_t *>(base + o_rank);
    p.d_seq_ne_contig = reinterpret_cast<int32_t *>(base + o_ne);
    p.d_seq_lane_bits = reinterpret_cast<const uint16_t *>(base + o_lb);
    p.d_seq_flat_bits = reinterpret_cast<const uint16_t *>(base + o_fb);
    if (p.seq_in_host_memory) {
        p.seq_ready = true;
        return GECCO_CRF_OK;
    }
    // launches that read the tables must be ordered behind this copy: `sync` (any stream may follow), or the
    // caller keeps to `stream` (the batch driver)
    if (p.tables_by_kernel && !sync) {  // (batch driver: fetched by a launch, the copy engine keeps to the chunks' arrays)
        void *dv = nullptr;
        if ((rc = check_hip(hipHostGetDevicePointer(&dv, p.seq.h, 0), "hipHostGetDevicePointer"))) return rc;
        if ((rc = check_hip(launch_copy_block(dv, p.seq.d, bytes, stream), "contig tables launch"))) return rc;
    } else if ((rc = check_hip(hipMemcpyAsync(p.seq.d, p.seq.h, bytes, hipMemcpyHostToDevice, stream), "upload contig flags"))) {
        return rc;
    }
    if ((rc = check_hip(launch_contig_flags(reinterpret_cast<const int32_t *>(p.seq.d + o_cp), p.n_contigs, p.n_genes, p.d_seq_flags, stream),
                        "contig flags launch")))
        return rc;
    if (sync && (rc = check_hip(hipStreamSynchronize(stream), "upload contig flags"))) return rc;
    p.seq_ready = true;
    return GECCO_CRF_OK;
}

namespace {
int ensure_seq(Plan &p, hipStream_t stream) {
    int rc = plan_ensure_seq(p, stream, !p.async_tables);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(p.ws_mutex);
    const SeqLayout l = seq_layout(size_t(p.n_genes));
    const bool fresh = !p.d_seq_ws || l.bytes + align256(size_t(p.n_contigs) + 24) > p.seq_ws_cap;
    rc = grow_ws(p.d_seq_ws, p.seq_ws_cap, l.bytes + align256(size_t(p.n_contigs) + 24), "hipMalloc scan workspace");
    // (the decoder's counters accumulate over launches: they start from zero in a new block)
    if (!rc && fresh) rc = check_hip(hipMemsetAsync(p.d_seq_ws + l.off_stats, 0, 64, stream), "memset decoder counters");
    // the exactness test's bound, candidate counter and contig flags (behind the layout): zero before the first decode on
    // this layout; every decode leaves them zero again (vd_exact_fix)
    const size_t sig = l.bytes * 1000003u + size_t(p.n_contigs) + 1;
    if (!rc && (fresh || sig != p.vbound_sig)) {
        rc = check_hip(hipMemsetAsync(p.d_seq_ws + l.bytes, 0, 16 + size_t(p.n_contigs), stream), "memset contig flags");
        if (!rc) p.vbound_sig = sig;
    }
    return rc;
}

int fill_seq_args(Plan &p, SeqArgs &a, hipStream_t stream) {
    if (p.device < 0) {
        set_error("host-only plan: no HIP device bound (there is no CPU fallback)");
        return GECCO_CRF_ENODEV;
    }
    int rc = use_device(p.device);
    if (rc) return rc;
    if (p.general) return GECCO_CRF_OK;  // the any-L path has its own workspace
    if ((rc = ensure_seq(p, stream))) return rc;
    const Model &m = *p.model;
    const SeqLayout l = seq_layout(size_t(p.n_genes));
    char *w = p.d_seq_ws;
    a = SeqArgs{};
    a.state = reinterpret_cast<double2 *>(w + l.off_state);
    a.alpha = reinterpret_cast<double2 *>(w + l.off_alpha);
    a.contigTmp = reinterpret_cast<double2 *>(w + l.off_tmp);
    a.vLane = reinterpret_cast<VE *>(w + l.off_vlane);
    a.vBlock = reinterpret_cast<VE *>(w + l.off_vblock);
    a.vMaps = reinterpret_cast<uint32_t *>(w + l.off_vmaps);
    a.vLaneMap = reinterpret_cast<uint32_t *>(w + l.off_vlanemap);
    a.vBlockMap = reinterpret_cast<uint32_t *>(w + l.off_vblockmap);
    a.fLane = reinterpret_cast<FE *>(w + l.off_flane);
    a.fBlock = reinterpret_cast<FE *>(w + l.off_fblock);
    a.fLaneSuf = reinterpret_cast<FE *>(w + l.off_flanesuf);
    a.fBlockSuf = reinterpret_cast<FE *>(w + l.off_fblocksuf);
    a.vCand = reinterpret_cast<SeqArgs::VdCand *>(w + l.off_cand);
    a.vd_stats = reinterpret_cast<uint32_t *>(w + l.off_stats);
    a.vBound = reinterpret_cast<unsigned long long *>(w + l.bytes);  // (16 bytes in front of the flags: one memset clears both)
    a.fix_flag = reinterpret_cast<uint8_t *>(w + l.bytes + 16);
    a.v_wmax2 = 2.0 * p.tables_model->wmax_abs;
    a.v_tmax = p.tables_model->tmax_abs;
    a.v_nmax = p.n_max;
    a.flags = p.d_seq_flags;
    a.lane_bits = p.d_seq_lane_bits;
    a.flat_bits = p.d_seq_flat_bits;
    a.cblk = p.d_seq_cblk;
    a.n_cblocks = p.n_cblocks;
    a.cblk_rank = p.d_seq_cblk_rank;
    a.ne_contig = p.d_seq_ne_contig;
    a.contig_ptr = p.d_contig_ptr;
    a.short_contigs = p.seq_short ? 1 : 0;
    a.n_contigs = p.n_contigs;
    a.n_genes = p.n_genes;
    const DeviceTables::SeqConsts &q = p.tables_model->seq;  // (model constants, computed once per device)
    a.mx = q.mx;
    a.t00 = m.trans[0];
    a.t01 = m.trans[1];
    a.t10 = m.trans[2];
    a.t11 = m.trans[3];
    a.m00 = q.m00;
    a.m01 = q.m01;
    a.m10 = q.m10;
    a.m11 = q.m11;
    a.dstate = reinterpret_cast<const double *>(a.state);  // same workspace, one of the two forms per call
    a.raw_fold = q.raw_fold;
    a.v_lo = q.v_lo;
    a.v_hi = q.v_hi;
    a.v_k = q.v_k;
    a.v_exact = q.v_exact;
    std::memcpy(a.expc, q.expc, sizeof(a.expc));
    return GECCO_CRF_OK;
}
}  // namespace

// Labels without path scores from a 2-label model whose transitions satisfy lo <= hi take the
// difference form (8 B/gene, 24-B scan elements); GECCO_CRF_VITERBI=matrix forces the general form.
static bool viterbi_delta_ok(const SeqArgs &a, const double *d_score) {
    const char *env = std::getenv("GECCO_CRF_VITERBI");
    if (env && env[0] == 'm') return false;
    return !d_score && a.v_lo <= a.v_hi && std::isfinite(a.v_lo) && std::isfinite(a.v_hi);
}

int plan_run_marginals_full(Plan &p, const int32_t *d_gene_ptr, const int32_t *d_attr_id, double *d_marg,
                            double *d_lognorm, hipStream_t stream) {
    p.pipe.pending = false;  // (the workspace of pipelined decode calls is written below)
    SeqArgs a;
    int rc = fill_seq_args(p, a, stream);
    if (rc) return rc;
    if (p.n_contigs == 0) return GECCO_CRF_OK;
    if (p.n_genes > 0 && (!d_gene_ptr || !d_marg)) {
        set_error("null device buffer");
        return GECCO_CRF_EINVAL;
    }
    if (p.general) {
        // 17 to 32 labels: the split of plan_run_viterbi for the forward-backward recursion (gl_marginals_wave; a step of it is
        // ~0.5 us; the chunked kernels -- transfer matrices on the matrix cores -- run at 2.6 / 2.2 ns per gene at L = 32 / 24).
        // Measured on 1 000 contigs / 0.22 M genes, longest 1 519 (all chunked -> all waves -> split): L = 32 0.60 -> 0.77 -> 0.44 ms.
        // GECCO_CRF_GENERAL_MARGINALS=wave|chunked|split forces for 9 <= L <= 32 (tests, A/B).
        const int L = p.model->L;
        int32_t wave_tmax = -1;
        if (L > 8 && p.n_contigs > 0)
            wave_tmax = choose_wave_split(p, std::getenv("GECCO_CRF_GENERAL_MARGINALS"), L > 16, L > 16 ? 0.5 : 0.38,
                                          L >= 28 ? 2.6e-3 : L > 16 ? 2.2e-3 : 1.0e-3, 0.1, 200.0, p.gen_wave_tmax_f);
        const bool wave = wave_tmax >= 0, tail_chunked = wave_tmax > 0;
        GenArgs g;
        if ((rc = fill_gen_args(p, d_gene_ptr, d_attr_id, g, !wave || tail_chunked, stream, tail_chunked ? wave_tmax : -1))) return rc;
        g.marg = d_marg;
        g.lognorm = d_lognorm;
        g.state = nullptr;
        g.wave_tmax = tail_chunked ? wave_tmax : 0;
        if ((rc = check_hip(launch_gen_state(g, stream), "state score launch"))) return rc;
        if (!wave) return check_hip(launch_gen_marginals(g, stream), "marginals launch");
        if (!tail_chunked || g.n_chunks <= 0) return check_hip(launch_gen_marginals_wave(g, stream), "marginals launch");
        if ((rc = fork_tail(p, stream))) return rc;
        if ((rc = check_hip(launch_gen_marginals(g, p.side_stream), "marginals launch"))) return rc;
        if ((rc = check_hip(launch_gen_marginals_wave(g, stream), "marginals launch"))) return rc;
        return join_tail(p, stream);
    }
    a.marg = d_marg;
    a.lognorm = d_lognorm;
    static const bool general_path = [] {  // GECCO_CRF_MARGINALS=general: the three-pass path on 16-byte states (A/B runs, tests)
        const char *env = std::getenv("GECCO_CRF_MARGINALS");
        return env && env[0] == 'g';
    }();
    if (p.seq_short || !general_path) {
        // 8-byte inputs, alpha in registers: one fused kernel when workgroups own whole contigs, the workgroups' products
        // + the fused kernel (look-back / look-ahead over them) for contigs of any length
        a.smax = reinterpret_cast<const double *>(a.alpha);
        // short contigs: log Z is written by the kernel at every contig's last gene; contigs without genes get their 0 here
        if (p.seq_short && d_lognorm && p.n_empty_contigs &&
            (rc = check_hip(hipMemsetAsync(d_lognorm, 0, size_t(p.n_contigs) * 8, stream), "memset lognorm")))
            return rc;
        return check_hip(launch_seq_marginals_short(a, d_gene_ptr, d_attr_id, p.tables_model->wtab2[1], p.model->A, p.d_contig_ptr,
                                                    stream), "marginals launch");
    }
    // wtab2[1] holds (w[a][0], w[a][1]) = (other, label) pairs for label 1
    if ((rc = check_hip(launch_seq_state(d_gene_ptr, d_attr_id, p.tables_model->wtab2[1], p.model->A, p.n_genes,
                                         const_cast<double2 *>(a.state), stream), "state score launch")))
        return rc;
    return check_hip(launch_seq_marginals(a, p.d_contig_ptr, stream), "marginals launch");
}

int plan_run_viterbi(Plan &p, const int32_t *d_gene_ptr, const int32_t *d_attr_id, int8_t *d_y, double *d_score,
                     hipStream_t stream) {
    p.pipe.pending = false;  // (the workspace of pipelined decode calls is written below)
    SeqArgs a;
    int rc = fill_seq_args(p, a, stream);
    if (rc) return rc;
    if (p.n_contigs == 0) return GECCO_CRF_OK;
    if (p.n_genes > 0 && (!d_gene_ptr || !d_y)) {
        set_error("null device buffer");
        return GECCO_CRF_EINVAL;
    }
    if (p.general) {
        // 13 to 32 labels: a wave per contig (gl_viterbi_wave) where the batch has its parallelism in its contigs.  The wave
        // kernel takes as long as the longest contig it is given (~0.36 / 0.25 us per gene above / up to 16 labels: a lone wave
        // issues an instruction every four cycles); the chunked kernels pay L x the arithmetic for every gene they are given
        // plus a chain of six launches and the walk over their longest contig's chunks.  So the batch is SPLIT: the k longest
        // contigs -- the tail of a metagenome's length distribution -- go through the chunked kernels on the plan's side
        // stream, NEXT TO the waves of the others, with k minimising  max(t_wave(longest of the others), t_chunked(the k longest)).
        // Measured on 1 000 contigs / 0.22 M genes, longest 1 519, next 762 (all chunked -> all waves -> split):
        // L = 32 1.42 -> 0.55 -> 0.32 ms, L = 24 0.96 -> 0.55 -> 0.32 ms, L = 16 0.28 -> 0.37 -> 0.22 ms.
        // GECCO_CRF_GENERAL_VITERBI=wave|chunked forces either for 9 <= L <= 32 (tests, A/B); =split forces the split at k = 1.
        const int L = p.model->L;
        int32_t wave_tmax = -1;  // -1: no wave kernel; 0: every contig; > 0: contigs up to this length
        if (L > 8 && p.n_contigs > 0) {
            const double t_step = L > 16 ? 0.36 : 0.25, t_gene = L >= 28 ? 6.3e-3 : L > 16 ? 4.3e-3 : 1.3e-3;
            wave_tmax = choose_wave_split(p, std::getenv("GECCO_CRF_GENERAL_VITERBI"), L > 12, t_step, t_gene, L > 16 ? 0.07 : 0.04,
                                          L > 16 ? 150.0 : 80.0, p.gen_wave_tmax);
        }
        GenArgs g;
        const bool wave = wave_tmax >= 0, tail_chunked = wave_tmax > 0;
        if ((rc = fill_gen_args(p, d_gene_ptr, d_attr_id, g, !wave || tail_chunked, stream, tail_chunked ? wave_tmax : -1))) return rc;
        g.y = d_y;
        g.score = d_score;
        g.E = nullptr;
        g.smax = nullptr;
        g.wave_tmax = tail_chunked ? wave_tmax : 0;
        if ((rc = check_hip(launch_gen_state(g, stream), "state score launch"))) return rc;
        if (!wave) return check_hip(launch_gen_viterbi(g, stream), "viterbi launch");
        if (!tail_chunked || g.n_chunks <= 0) return check_hip(launch_gen_viterbi_wave(g, stream), "viterbi launch");
        // the long tail (chunked kernels: short in work, long in dependent launches) NEXT TO the waves of the other contigs:
        // forked onto the plan's side stream behind the state scores, joined before anything later on the caller's stream.
        // The two write disjoint genes, contigs and back-pointer regions: a contig's back-pointers, in either kernel's layout, stay
        // inside its own T * L bytes (gl_viterbi_wave: quads of rows 1 .. T - 1 from the first dword boundary).  (A chunkless
        // contig's path score is the waves'.)
        if ((rc = fork_tail(p, stream))) return rc;
        if ((rc = check_hip(launch_gen_viterbi(g, p.side_stream), "viterbi launch"))) return rc;
        if ((rc = check_hip(launch_gen_viterbi_wave(g, stream), "viterbi launch"))) return rc;
        return join_tail(p, stream);
    }
    a.y = d_y;
    a.score = d_score;
    a.csr_gene_ptr = d_gene_ptr;
    a.csr_attr_id = d_attr_id;
    a.csr_wtab01 = p.tables_model->wtab2[1];
    a.csr_n_attrs = p.model->A;
    if (viterbi_delta_ok(a, d_score)) {
        if ((rc = check_hip(launch_seq_state_delta(d_gene_ptr, d_attr_id, p.tables_model->wtab2[1], p.model->A, p.n_genes,
                                                   const_cast<double *>(a.dstate), stream), "state score launch")))
            return rc;
        return check_hip(launch_seq_viterbi_delta(a, stream), "viterbi launch");
    }
    if ((rc = check_hip(launch_seq_state(d_gene_ptr, d_attr_id, p.tables_model->wtab2[1], p.model->A, p.n_genes,
                                         const_cast<double2 *>(a.state), stream), "state score launch")))
        return rc;
    return check_hip(launch_seq_viterbi(a, p.d_contig_ptr, stream), "viterbi launch");
}


int plan_run_decode(Plan &p, const int32_t *d_gene_ptr, const int32_t *d_attr_id, int32_t label, double *d_p_out,
                    int8_t *d_y, double *d_score, hipStream_t stream) {
    p.pipe.pending = false;  // (the workspace of pipelined decode calls is written below)
    // handing the score differences over needs the register-resident 2-label kernel and every gene in slot space
    const bool share = !p.general && p.fast_ok && p.skipped.empty() && p.device >= 0;
    if (!share) {
        int rc = plan_run_windowed(p, d_gene_ptr, d_attr_id, label, d_p_out, stream);
        if (rc) return rc;
        return plan_run_viterbi(p, d_gene_ptr, d_attr_id, d_y, d_score, stream);
    }
    if (label < 0 || label >= p.model->L) {
        set_error("label out of range");
        return GECCO_CRF_EINVAL;
    }
    SeqArgs a;
    int rc = fill_seq_args(p, a, stream);
    if (rc) return rc;
    if (p.n_contigs == 0 || p.n_genes == 0) return GECCO_CRF_OK;
    if (!d_gene_ptr || !d_p_out || !d_y) {
        set_error("null device buffer");
        return GECCO_CRF_EINVAL;
    }
    a.y = d_y;
    a.score = d_score;
    a.csr_gene_ptr = d_gene_ptr;
    a.csr_attr_id = d_attr_id;
    a.csr_wtab01 = p.tables_model->wtab2[1];
    a.csr_n_attrs = p.model->A;
    if (viterbi_delta_ok(a, d_score)) {
        if ((rc = run_windowed_impl(p, d_gene_ptr, d_attr_id, label, d_p_out, nullptr, const_cast<double *>(a.dstate), stream)))
            return rc;
        return check_hip(launch_seq_viterbi_delta(a, stream), "viterbi launch");
    }
    if ((rc = run_windowed_impl(p, d_gene_ptr, d_attr_id, label, d_p_out, const_cast<double2 *>(a.state), nullptr, stream)))
        return rc;
    return check_hip(launch_seq_viterbi(a, p.d_contig_ptr, stream), "viterbi launch");
}

int plan_run_decode_pipelined(Plan *cur, const int32_t *d_gene_ptr, const int32_t *d_attr_id, int32_t label, double *d_p_out,
                              Plan *prev, int8_t *d_prev_y, hipStream_t stream) {
    int rc;
    // ---- the previous batch: labels from the score differences its window tiles left behind (or from its CSR arrays)
    SeqArgs pa{};
    bool prev_delta = false;
    if (prev && prev->n_genes > 0 && prev->n_contigs > 0) {
        if (!d_prev_y) {
            set_error("null device buffer");
            return GECCO_CRF_EINVAL;
        }
        if (prev->pipe.pending && !prev->general) {
            if ((rc = fill_seq_args(*prev, pa, stream))) return rc;
            pa.dstate = reinterpret_cast<const double *>(pa.state) + size_t(prev->pipe.parity) * (size_t(prev->n_genes) + 8);
            pa.y = d_prev_y;
            pa.csr_gene_ptr = prev->pipe.gene_ptr;
            pa.csr_attr_id = prev->pipe.attr_id;
            pa.csr_wtab01 = prev->tables_model->wtab2[1];
            pa.csr_n_attrs = prev->model->A;
            prev_delta = true;
        }
    }
    const bool prev_work = prev && prev->n_genes > 0 && prev->n_contigs > 0;
    if (prev_work && !prev_delta) {
        // no score differences left behind (any-L model, contigs outside slot space, or another call used the workspace since):
        // the state scores are summed again from the CSR arrays -- BEFORE this batch's tiles write into the workspace
        const Plan::Pipe keep = prev->pipe;
        if (!keep.gene_ptr) {
            set_error("pipelined decode: the previous plan has not been scored by a pipelined call (or has been rebuilt since)");
            return GECCO_CRF_EINVAL;
        }
        if ((rc = plan_run_viterbi(*prev, keep.gene_ptr, keep.attr_id, d_prev_y, nullptr, stream))) return rc;
        prev->pipe = keep;
    }
    // ---- this batch: marginals, and score differences for the next call where the kernels hand them over
    bool took = false;
    if (cur) {
        Plan &p = *cur;
        const int parity = p.pipe.parity ^ 1;  // (cur == prev: the Viterbi side reads the other buffer)
        bool delta = false;
        double *d_dstate = nullptr;
        if (!p.general && p.fast_ok && p.skipped.empty() && p.device >= 0 && p.n_genes > 0) {
            SeqArgs ca;
            if (prev_delta && prev == cur) {
                ca = pa;  // (a plan that follows itself: the block has just been filled for the Viterbi side)
            } else if ((rc = fill_seq_args(p, ca, stream))) {
                return rc;
            }
            if (viterbi_delta_ok(ca, nullptr)) {
                delta = true;
                d_dstate = const_cast<double *>(reinterpret_cast<const double *>(ca.state)) + size_t(parity) * (size_t(p.n_genes) + 8);
            }
        }
        PipelinedLaunch fl{};
        fl.seq = &pa;
        fl.took = &took;
        // A/B switch (wrong labels): the tiles keep their score differences to themselves -- what the WRITE half of the hand-over
        // between launches (8 B per gene + its five vector instructions) costs the step (profiles/r06_ab.txt)
        static const bool ab_no_store = [] {
            const char *env = std::getenv("GECCO_CRF_AB_NO_HANDOVER_STORE");
            return env && env[0] == '1';
        }();
        if ((rc = run_windowed_impl(p, d_gene_ptr, d_attr_id, label, d_p_out, nullptr, ab_no_store ? nullptr : d_dstate, stream,
                                    (delta && prev_delta) ? &fl : nullptr)))
            return rc;
        p.pipe.pending = delta;
        p.pipe.parity = delta ? parity : p.pipe.parity;
        p.pipe.gene_ptr = d_gene_ptr;
        p.pipe.attr_id = d_attr_id;
    }
    if (prev_work && prev_delta && !took)
        if ((rc = check_hip(launch_seq_viterbi_delta(pa, stream), "viterbi launch"))) return rc;
    if (prev && prev != cur) prev->pipe.pending = false;
    return GECCO_CRF_OK;
}

int plan_viterbi_stats(Plan &p, int64_t out[4], bool reset) {
    for (int i = 0; i < 4; ++i) out[i] = 0;
    std::lock_guard<std::mutex> lock(p.ws_mutex);
    if (p.device < 0 || !p.d_seq_ws) return GECCO_CRF_OK;  // nothing has been decoded yet
    int rc = use_device(p.device);
    if (rc) return rc;
    if ((rc = check_hip(hipDeviceSynchronize(), "hipDeviceSynchronize"))) return rc;
    uint32_t v[4] = {0, 0, 0, 0};
    char *at = p.d_seq_ws + seq_layout(size_t(p.n_genes)).off_stats;
    if ((rc = check_hip(hipMemcpy(v, at, sizeof(v), hipMemcpyDeviceToHost), "read decoder counters"))) return rc;
    for (int i = 0; i < 4; ++i) out[i] = v[i];
    if (reset) rc = check_hip(hipMemset(at, 0, sizeof(v)), "clear decoder counters");
    return rc;
}

int plan_run_segment(Plan &p, const double *d_p, const uint8_t *d_annotated, const SegParams &params, int32_t *d_seg,
                     int32_t max_seg, int32_t *d_seg_off, int32_t *d_total, hipStream_t stream, double *d_gather, int32_t gather_cap) {
    if (!d_total || max_seg < 0 || (max_seg > 0 && !d_seg)) {
        set_error("plan_run_segment: bad arguments");
        return GECCO_CRF_EINVAL;
    }
    if (params.criterion != 0 && params.criterion != 1) {
        set_error("Unknown cluster filtering criterion");  // refine.py:165
        return GECCO_CRF_EINVAL;
    }
    if (params.criterion == 1 && p.n_genes > 0 && (!params.bio_ptr || !params.bio_id)) {
        set_error("the antismash criterion needs the genes' marker domains");
        return GECCO_CRF_EINVAL;
    }
    // the contig flags of the whole-contig tables when the plan has them; otherwise the segmenter derives them from the
    // contig table on the device (a memset and a launch over the contigs instead of an upload of a byte per gene)
    int rc = GECCO_CRF_OK;
    if (p.n_genes > 0 && (!d_p || !d_annotated)) {
        set_error("null device buffer");
        return GECCO_CRF_EINVAL;
    }
    {
        std::lock_guard<std::mutex> lock(p.ws_mutex);
        if ((rc = grow_ws(p.d_seg_ws, p.seg_ws_cap, segment_workspace_bytes(p.n_genes, p.n_contigs), "hipMalloc segment workspace")))
            return rc;
    }
    return check_hip(launch_segment(d_p, d_annotated, p.seq_ready ? p.d_seq_flags : nullptr, p.d_contig_ptr, p.n_genes, p.n_contigs, params, d_seg, max_seg,
                                    d_seg_off, d_total, p.d_seg_ws, stream, d_gather, gather_cap),
                     "segment launch");
}

}  // namespace gecco
