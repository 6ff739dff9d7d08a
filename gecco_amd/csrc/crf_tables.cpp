// Columnar host side of the path: feature / gene TABLE COLUMNS -> CSR batch
// (row X of 8a without Gene objects), and called clusters -> the rows of clusters.tsv.
//
// What the reference does object by object:
//   * gecco/crf/__init__.py:199-206  genes sorted by (source.id, start) (stable), a gene's domains by
//     start, genes grouped by contig;
//   * gecco/crf/features.py:13-35    one {domain name: True} dict per gene: a repeated name is one
//     feature, order = first occurrence; [EXT] CRFsuite drops names its dictionary does not know;
//   * gecco/model.py:731-760         cluster rows: start / end of the member genes, average_p =
//     statistics.mean (exactly rounded), max_p, sorted protein ids, sorted domain names.
// Here: strings arrive as Arrow-style columns (one byte buffer + int64 offsets per column: the layout
// pandas / polars / pyarrow hold them in), are hashed ONCE (or not at all: rows of one protein are
// normally adjacent and the feature table walks the gene table in order), orders are checked before
// anything is sorted (tables written by GECCO are already in order), and the CSR lands in pinned memory
// ready for the batch driver.
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/gecco_crf.h"
#include "crf_model.hpp"
#include "crf_tables.hpp"

namespace gecco {
namespace {

// GECCO_CRF_TRACE=1: wall time of every phase, to stderr
struct Phase {
    bool on;
    double t;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    Phase() {
        const char *e = std::getenv("GECCO_CRF_TRACE");
        on = e && e[0] == '1';
        t = on ? now() : 0.0;
    }
    void lap(const char *what) {
        if (!on) return;
        const double n = now();
        std::fprintf(stderr, "[gecco_crf] tables %-22s %8.2f ms\n", what, (n - t) * 1e3);
        t = n;
    }
};

struct Str {
    const uint8_t *p;
    uint32_t n;
};
inline Str at(const gecco_crf_strings &c, int64_t i) {
    const int64_t a = c.offsets[i], b = c.offsets[i + 1];
    return Str{c.data + a, uint32_t(b - a)};
}
inline bool same(const Str &a, const Str &b) { return a.n == b.n && (a.n == 0 || std::memcmp(a.p, b.p, a.n) == 0); }
inline bool less(const Str &a, const Str &b) {  // byte order = code point order of UTF-8 = Python's str order
    const int c = std::memcmp(a.p, b.p, std::min(a.n, b.n));
    return c < 0 || (c == 0 && a.n < b.n);
}
inline uint64_t hash_str(const Str &s) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t(s.n) * 0xff51afd7ed558ccdull);
    const uint8_t *p = s.p;
    uint32_t n = s.n;
    while (n >= 8) {
        uint64_t w;
        std::memcpy(&w, p, 8);
        h = (h ^ w) * 0xff51afd7ed558ccdull;
        h ^= h >> 32;
        p += 8;
        n -= 8;
    }
    if (n) {
        uint64_t w = 0;
        std::memcpy(&w, p, n);
        h = (h ^ w) * 0xc4ceb9fe1a85ec53ull;
        h ^= h >> 29;
    }
    h *= 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 32);
}

// string -> dense index in first-appearance order
struct Interner {
    std::vector<Str> keys;
    std::vector<uint64_t> hashes;
    std::vector<uint32_t> slots;  // index + 1, 0 = empty
    uint64_t mask = 0;
    explicit Interner(size_t expect) {
        size_t cap = 16;
        while (cap < expect * 2 + 8) cap <<= 1;
        slots.assign(cap, 0);
        mask = cap - 1;
        keys.reserve(expect);
        hashes.reserve(expect);
    }
    void grow() {
        std::vector<uint32_t> ns(slots.size() * 2, 0);
        const uint64_t nm = ns.size() - 1;
        for (size_t i = 0; i < keys.size(); ++i) {
            uint64_t s = hashes[i] & nm;
            while (ns[s]) s = (s + 1) & nm;
            ns[s] = uint32_t(i + 1);
        }
        slots.swap(ns);
        mask = nm;
    }
    int32_t find(const Str &k) const {
        const uint64_t h = hash_str(k);
        for (uint64_t s = h & mask;; s = (s + 1) & mask) {
            const uint32_t e = slots[s];
            if (!e) return -1;
            if (hashes[e - 1] == h && same(keys[e - 1], k)) return int32_t(e - 1);
        }
    }
    int32_t intern(const Str &k, bool *fresh = nullptr) {
        const uint64_t h = hash_str(k);
        for (uint64_t s = h & mask;; s = (s + 1) & mask) {
            const uint32_t e = slots[s];
            if (!e) {
                if ((keys.size() + 1) * 2 > slots.size()) {
                    grow();
                    return intern(k, fresh);
                }
                keys.push_back(k);
                hashes.push_back(h);
                slots[s] = uint32_t(keys.size());
                if (fresh) *fresh = true;
                return int32_t(keys.size() - 1);
            }
            if (hashes[e - 1] == h && same(keys[e - 1], k)) {
                if (fresh) *fresh = false;
                return int32_t(e - 1);
            }
        }
    }
};

// stable sort of `idx` by 64-bit keys, least significant digit first; skipped when already in order
template <class KeyVec, class IdxVec>
void radix_sort_by_key(KeyVec &key, IdxVec &idx) {
    const size_t n = key.size();
    bool sorted = true;
    uint64_t all_or = 0;
    for (size_t i = 0; i < n; ++i) {
        all_or |= key[i];
        if (i && key[i] < key[i - 1]) sorted = false;
    }
    if (sorted) return;
    KeyVec k2(n);
    IdxVec i2(n);
    for (int shift = 0; shift < 64; shift += 16) {
        if (((all_or >> shift) & 0xffffull) == 0 && (all_or >> shift) == 0) break;
        std::vector<size_t> count(65537, 0);
        for (size_t i = 0; i < n; ++i) ++count[((key[i] >> shift) & 0xffff) + 1];
        for (int b = 0; b < 65536; ++b) count[b + 1] += count[b];
        for (size_t i = 0; i < n; ++i) {
            const size_t d = count[(key[i] >> shift) & 0xffff]++;
            k2[d] = key[i];
            i2[d] = idx[i];
        }
        key.swap(k2);
        idx.swap(i2);
    }
}

// memory the batch driver can copy from asynchronously when a device is there; plain otherwise
bool have_device() {
    int n = 0;
    const bool ok = hipGetDeviceCount(&n) == hipSuccess && n > 0;
    if (!ok) (void)hipGetLastError();
    return ok;
}
void *alloc_staging(size_t bytes, bool pinned) {
    void *p = nullptr;
    if (pinned) {
        if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) throw std::bad_alloc();
        return p;
    }
    p = std::malloc(bytes ? bytes : 1);
    if (!p) throw std::bad_alloc();
    return p;
}

}  // namespace

Packed::~Packed() {
    if (!block) return;
    if (pinned) (void)hipHostFree(block); else std::free(block);
}

namespace {
// ---- a few host threads for the per-row passes (tables of 10^5..10^7 rows; every pass is a plain loop) ----
// GECCO_CRF_HOST_THREADS caps the workers (default: up to 16), GECCO_CRF_HOST_GRAIN the items a worker must
// have to be worth starting (default 16384; the tests lower it to cut tiny tables into many ranges)
int worker_count(int64_t items) {
    int hw = 0;
    if (const char *e = std::getenv("GECCO_CRF_HOST_THREADS")) hw = std::min(std::atoi(e), 64);
    if (hw < 1) {
        const unsigned h = std::thread::hardware_concurrency();
        hw = int(std::min<unsigned>(h ? h : 1, 16));
    }
    int64_t grain = 16384;
    if (const char *e = std::getenv("GECCO_CRF_HOST_GRAIN")) grain = std::max(1, std::atoi(e));
    return int(std::max<int64_t>(1, std::min<int64_t>(hw, items / grain)));
}
// A handful of persistent workers (started on first use): a pass over a table costs a wake-up, not 16 thread
// creations.  One job at a time (calls are serialised by `busy`).
class Pool {
public:
    static Pool &get() {
        static Pool *p = new Pool();  // never destroyed: workers may outlive static destruction order
        return *p;
    }
    template <class Fn>
    void run(int workers, Fn fn) {  // fn(worker) for worker in [0, workers); the caller is worker 0
        if (workers <= 1) {
            fn(0);
            return;
        }
        std::lock_guard<std::mutex> serial(busy_);
        if (pid_ != getpid()) {  // a forked child has none of the parent's threads
            threads_.clear();
            pid_ = getpid();
        }
        ensure(workers - 1);
        std::vector<std::exception_ptr> err(static_cast<size_t>(workers));
        std::function<void(int)> job = [&](int w) {
            try {
                fn(w);
            } catch (...) {
                err[size_t(w)] = std::current_exception();
            }
        };
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &job;
            want_ = workers - 1;
            pending_ = workers - 1;
            ++generation_;
        }
        cv_.notify_all();
        job(0);
        {
            std::unique_lock<std::mutex> lk(m_);
            done_.wait(lk, [&] { return pending_ == 0; });
            job_ = nullptr;
        }
        for (auto &e : err)
            if (e) std::rethrow_exception(e);
    }

private:
    void ensure(int n) {
        while (int(threads_.size()) < n) {
            const int id = int(threads_.size()) + 1;
            threads_.emplace_back([this, id] { loop(id); });
            threads_.back().detach();
        }
    }
    void loop(int id) {
        uint64_t seen = 0;
        for (;;) {
            std::function<void(int)> *job = nullptr;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
                if (id <= want_) job = job_;
            }
            if (!job) continue;
            (*job)(id);
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    pid_t pid_ = getpid();
    std::mutex m_, busy_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> threads_;
    std::function<void(int)> *job_ = nullptr;
    int want_ = 0, pending_ = 0;
    uint64_t generation_ = 0;
};

// fn(begin, end, worker) over [0, n) cut into `workers` contiguous ranges
template <class Fn>
void parallel_ranges(int64_t n, int workers, Fn fn) {
    if (workers <= 1) {
        fn(int64_t(0), n, 0);
        return;
    }
    Pool::get().run(workers, [&](int w) { fn(n * w / workers, n * (w + 1) / workers, w); });
}

// Read-only index of the gene table's ids, built by several threads: ids are dealt to 2^k sub-tables by the
// top bits of their hash, every worker fills its own sub-tables (no two workers touch the same one).
struct GeneIndex {
    static constexpr int kParts = 64;
    std::vector<uint32_t> slots[kParts];  // gene + 1
    uint64_t mask[kParts];
    const UVec<Str> *keys = nullptr;
    const UVec<uint64_t> *hashes = nullptr;
    bool duplicates = false;
    void build(const UVec<Str> &k, const UVec<uint64_t> &h, int workers) {
        keys = &k;
        hashes = &h;
        const int64_t n = int64_t(k.size());
        std::vector<int64_t> counts(size_t(kParts) * size_t(workers), 0);
        parallel_ranges(n, workers, [&](int64_t b, int64_t e, int w) {
            for (int64_t i = b; i < e; ++i) ++counts[size_t(w) * kParts + (h[size_t(i)] >> 58)];
        });
        // ids listed part by part, in gene order inside a part (round 5: every worker used to scan ALL hashes for the parts it
        // owned -- sixteen workers sixteen scans, slower than one): offsets in (part, worker) order, every worker scatters its
        // own range, every part is then inserted from its contiguous list
        std::vector<int64_t> off(size_t(kParts) * size_t(workers) + 1, 0);
        {
            int64_t run = 0;
            for (int p = 0; p < kParts; ++p)
                for (int w = 0; w < workers; ++w) {
                    off[size_t(p) * workers + w] = run;
                    run += counts[size_t(w) * kParts + p];
                }
            off[size_t(kParts) * workers] = run;
        }
        UVec<uint32_t> list(static_cast<size_t>(n));
        parallel_ranges(n, workers, [&](int64_t b, int64_t e, int w) {
            int64_t cur[kParts];
            for (int p = 0; p < kParts; ++p) cur[p] = off[size_t(p) * workers + w];
            for (int64_t i = b; i < e; ++i) list[size_t(cur[h[size_t(i)] >> 58]++)] = uint32_t(i);
        });
        std::vector<char> dup(static_cast<size_t>(kParts), 0);
        const int pw = std::min(workers, kParts);
        parallel_ranges(kParts, pw, [&](int64_t pb, int64_t pe, int) {
            for (int64_t p = pb; p < pe; ++p) {
                const int64_t l0 = off[size_t(p) * workers], l1 = off[size_t(p + 1) * workers];
                size_t cap = 16;
                while (cap < size_t(l1 - l0) * 2 + 8) cap <<= 1;
                std::vector<uint32_t> &sl = slots[p];
                sl.assign(cap, 0);
                mask[p] = cap - 1;
                for (int64_t q = l0; q < l1; ++q) {
                    const uint32_t i = list[size_t(q)];
                    const uint64_t hv = h[i];
                    for (uint64_t s = hv & mask[p];; s = (s + 1) & mask[p]) {
                        const uint32_t cur = sl[s];
                        if (!cur) {
                            sl[s] = i + 1;
                            break;
                        }
                        if (h[cur - 1] == hv && same(k[cur - 1], k[i])) {
                            dup[size_t(p)] = 1;
                            break;
                        }
                    }
                }
            }
        });
        for (char d : dup) duplicates |= d != 0;
    }
    int32_t find(const Str &s, uint64_t hv) const {
        const int p = int(hv >> 58);
        const std::vector<uint32_t> &sl = slots[p];
        for (uint64_t q = hv & mask[p];; q = (q + 1) & mask[p]) {
            const uint32_t cur = sl[q];
            if (!cur) return -1;
            if ((*hashes)[cur - 1] == hv && same((*keys)[cur - 1], s)) return int32_t(cur - 1);
        }
    }
};

}  // namespace

int pack_columns(const Model &m, const gecco_crf_table_columns &t, Packed &out) {
    const int64_t nf = t.n_rows, ng = t.n_genes;
    if (nf < 0 || ng < 0 || nf + ng > int64_t(INT32_MAX) - 8) {
        set_error("pack_columns: table too large (more than 2^31 rows)");
        return GECCO_CRF_EUNSUPPORTED;
    }
    if ((nf && (!t.sequence_id.offsets || !t.protein_id.offsets || !t.domain.offsets || !t.start || !t.domain_start)) ||
        (ng && (!t.gene_sequence_id.offsets || !t.gene_protein_id.offsets || !t.gene_start))) {
        set_error("pack_columns: null column");
        return GECCO_CRF_EINVAL;
    }
    Phase ph;
    const int wg = worker_count(ng), wf = worker_count(nf);
    // ---- genes in first-appearance order: gene-table rows, then proteins only the feature table knows
    UVec<Str> g_key(static_cast<size_t>(ng));  // per gene: its id
    UVec<uint64_t> g_hash(static_cast<size_t>(ng));
    parallel_ranges(ng, wg, [&](int64_t b, int64_t e, int) {
        for (int64_t i = b; i < e; ++i) {
            g_key[size_t(i)] = at(t.gene_protein_id, i);
            g_hash[size_t(i)] = hash_str(g_key[size_t(i)]);
        }
    });
    GeneIndex index;
    index.build(g_key, g_hash, wg);
    ph.lap("gene index");
    std::vector<int32_t> g_sid;    // per gene: contig (first-appearance index)
    std::vector<int64_t> g_start;  // per gene
    std::vector<int64_t> g_row;    // per gene: gene-table row, or -1 - (first feature row)
    g_sid.reserve(size_t(ng) + 16);
    g_start.reserve(size_t(ng) + 16);
    g_row.reserve(size_t(ng) + 16);
    Interner contigs(1024);
    auto contig_of = [&](const gecco_crf_strings &col, int64_t i, const Str &prev, int32_t prev_id) {
        const Str s = at(col, i);
        if (prev_id >= 0 && same(s, prev)) return prev_id;  // rows of one contig are normally adjacent
        return contigs.intern(s);
    };
    out.n_duplicate_gene_ids = 0;
    std::vector<int32_t> dedup;  // gene-table row -> gene, only when ids repeat
    if (!index.duplicates) {
        // contig of every gene row: rows of one contig are normally adjacent, so only the rows where the id
        // CHANGES are interned (found in parallel), the rows in between inherit
        g_sid.resize(size_t(ng));
        g_start.assign(t.gene_start, t.gene_start + ng);
        g_row.resize(size_t(ng));
        std::vector<uint8_t> change(static_cast<size_t>(ng));
        parallel_ranges(ng, wg, [&](int64_t b, int64_t e, int) {
            for (int64_t i = b; i < e; ++i) {
                change[size_t(i)] = i == 0 || !same(at(t.gene_sequence_id, i), at(t.gene_sequence_id, i - 1));
                g_row[size_t(i)] = i;
            }
        });
        std::vector<int64_t> cut;  // rows where a new stretch starts, and the contig of the stretch
        std::vector<int32_t> cut_id;
        for (int64_t i = 0; i < ng; ++i)
            if (change[size_t(i)]) {
                cut.push_back(i);
                cut_id.push_back(contigs.intern(at(t.gene_sequence_id, i)));
            }
        cut.push_back(ng);
        parallel_ranges(int64_t(cut_id.size()), worker_count(ng), [&](int64_t b, int64_t e, int) {
            for (int64_t k = b; k < e; ++k)
                for (int64_t i = cut[size_t(k)]; i < cut[size_t(k) + 1]; ++i) g_sid[size_t(i)] = cut_id[size_t(k)];
        });
    } else {  // repeated ids: the first row fixes the gene's position, the last one its contig and start (like a dict)
        Interner genes(size_t(ng) + 16);
        UVec<Str> keys2;
        UVec<uint64_t> hashes2;
        Str prev{nullptr, 0};
        int32_t prev_id = -1;
        for (int64_t i = 0; i < ng; ++i) {
            bool fresh = false;
            const int32_t g = genes.intern(g_key[size_t(i)], &fresh);
            const int32_t c = contig_of(t.gene_sequence_id, i, prev, prev_id);
            prev = at(t.gene_sequence_id, i);
            prev_id = c;
            if (fresh) {
                g_sid.push_back(c);
                g_start.push_back(t.gene_start[i]);
                g_row.push_back(i);
                keys2.push_back(g_key[size_t(i)]);
                hashes2.push_back(g_hash[size_t(i)]);
            } else {
                ++out.n_duplicate_gene_ids;
                g_sid[size_t(g)] = c;
                g_start[size_t(g)] = t.gene_start[i];
                g_row[size_t(g)] = i;
            }
        }
        g_key.swap(keys2);
        g_hash.swap(hashes2);
        index = GeneIndex();
        index.build(g_key, g_hash, wg);
    }
    const int64_t n_listed = int64_t(g_key.size());
    ph.lap("gene rows");
    // feature rows -> gene.  The feature table normally lists proteins in gene-table order with the rows of a
    // protein adjacent: try the previous row's gene and its successors before looking the id up.  Rows are cut
    // into ranges, one per worker; ids the gene table does not list are resolved afterwards, in row order.
    UVec<int32_t> row_gene_fa(static_cast<size_t>(nf));  // first-appearance gene index of every feature row
    std::vector<std::vector<int64_t>> unlisted(static_cast<size_t>(wf));
    parallel_ranges(nf, wf, [&](int64_t b, int64_t e, int w) {
        int32_t cur = -1;
        for (int64_t i = b; i < e; ++i) {
            const Str pid = at(t.protein_id, i);
            int32_t g = -1;
            if (cur >= 0 && same(g_key[size_t(cur)], pid)) {
                g = cur;
            } else {
                for (int32_t k = cur + 1; k < cur + 6 && k < int32_t(n_listed); ++k)
                    if (k >= 0 && same(g_key[size_t(k)], pid)) {
                        g = k;
                        break;
                    }
                if (g < 0 && n_listed) g = index.find(pid, hash_str(pid));
            }
            if (g < 0) {
                unlisted[size_t(w)].push_back(i);
            } else {
                cur = g;
            }
            row_gene_fa[size_t(i)] = g;
        }
    });
    out.n_unlisted_proteins = 0;
    {
        Interner extra(64);
        Str prev_sid{nullptr, 0};
        int32_t prev_sid_id = -1;
        for (const auto &rows : unlisted)
            for (int64_t i : rows) {  // workers hold consecutive row ranges: this IS row order
                bool fresh = false;
                const int32_t x = extra.intern(at(t.protein_id, i), &fresh);
                if (fresh) {  // the first row of a protein the gene table does not list defines it
                    const int32_t c = contig_of(t.sequence_id, i, prev_sid, prev_sid_id);
                    prev_sid = at(t.sequence_id, i);
                    prev_sid_id = c;
                    g_sid.push_back(c);
                    g_start.push_back(t.start[i]);
                    g_row.push_back(-1 - i);
                    if (ng) ++out.n_unlisted_proteins;
                }
                row_gene_fa[size_t(i)] = int32_t(n_listed) + x;
            }
    }
    const int64_t n = int64_t(g_row.size());
    ph.lap("feature rows -> genes");
    // ---- contigs in str order, genes by (contig, start), ties in first-appearance order (sorted() is stable)
    const size_t nc_all = contigs.keys.size();
    std::vector<int32_t> c_rank(nc_all);
    {
        std::vector<int32_t> by(nc_all);
        for (size_t i = 0; i < nc_all; ++i) by[i] = int32_t(i);
        std::sort(by.begin(), by.end(), [&](int32_t a, int32_t b) { return less(contigs.keys[a], contigs.keys[b]); });
        for (size_t r = 0; r < nc_all; ++r) c_rank[by[r]] = int32_t(r);
    }
    // (round 5: the passes over the genes below are plain loops -- cut over the host threads; the sort itself only runs when
    // the genes do not come in scoring order already, which annotation pipelines see to)
    const int wn = worker_count(n);
    int64_t min_start = 0;
    {
        std::vector<int64_t> part(size_t(std::max(wn, 1)), 0);
        parallel_ranges(n, wn, [&](int64_t b, int64_t e, int w) {
            int64_t m = 0;
            for (int64_t g = b; g < e; ++g) m = std::min(m, g_start[size_t(g)]);
            part[size_t(w)] = m;
        });
        for (int64_t m : part) min_start = std::min(min_start, m);
    }
    UVec<int64_t> perm(static_cast<size_t>(n));
    {
        UVec<uint64_t> key(static_cast<size_t>(n));
        std::vector<char> bad(size_t(std::max(wn, 1)), 0);
        parallel_ranges(n, wn, [&](int64_t b, int64_t e, int w) {
            bool fits_here = true;
            for (int64_t g = b; g < e; ++g) {
                const uint64_t s = uint64_t(g_start[size_t(g)] - min_start);
                if (s >> 40) fits_here = false;
                key[size_t(g)] = (uint64_t(c_rank[size_t(g_sid[size_t(g)])]) << 40) | (s & ((1ull << 40) - 1));
                perm[size_t(g)] = g;
            }
            bad[size_t(w)] = fits_here ? 0 : 1;
        });
        bool fits = true;
        for (char c : bad) fits &= c == 0;
        if (fits && nc_all < (1u << 24)) {
            radix_sort_by_key(key, perm);
        } else {  // coordinates beyond 2^40: comparison sort
            std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) {
                const int32_t ra = c_rank[size_t(g_sid[size_t(a)])], rb = c_rank[size_t(g_sid[size_t(b)])];
                return ra != rb ? ra < rb : g_start[size_t(a)] < g_start[size_t(b)];
            });
        }
    }
    UVec<int32_t> pos_of(static_cast<size_t>(n));  // first-appearance index -> position in scoring order
    out.n_genes = int32_t(n);
    out.gene_row.resize(size_t(n));
    std::vector<std::vector<int32_t>> cuts(size_t(std::max(wn, 1)));  // contig boundaries found by every worker, in order
    parallel_ranges(n, wn, [&](int64_t b, int64_t e, int w) {
        for (int64_t k = b; k < e; ++k) {
            pos_of[size_t(perm[size_t(k)])] = int32_t(k);
            out.gene_row[size_t(k)] = g_row[size_t(perm[size_t(k)])];
            if (k >= 1 && g_sid[size_t(perm[size_t(k)])] != g_sid[size_t(perm[size_t(k - 1)])]) cuts[size_t(w)].push_back(int32_t(k));
        }
    });
    std::vector<int32_t> cptr;
    cptr.push_back(0);
    for (const std::vector<int32_t> &c : cuts) cptr.insert(cptr.end(), c.begin(), c.end());
    if (n) cptr.push_back(int32_t(n));
    out.n_contigs = int32_t(cptr.size()) - 1;
    ph.lap("gene order");
    // ---- feature rows by (gene position, domain_start), stable
    out.row_gene.resize(size_t(nf));
    out.row_order.resize(size_t(nf));
    out.row_ptr.resize(size_t(n) + 1);
    {
        parallel_ranges(nf, wf, [&](int64_t b, int64_t e, int) {
            for (int64_t i = b; i < e; ++i) out.row_gene[size_t(i)] = pos_of[size_t(row_gene_fa[size_t(i)])];
        });
        ph.lap("row genes");
        bool grouped = true;  // rows already come gene by gene, in scoring order: only the runs need sorting
        {
            std::vector<char> broken(size_t(std::max(wf, 1)), 0);
            parallel_ranges(nf, wf, [&](int64_t b, int64_t e, int w) {
                bool ok = true;
                for (int64_t i = std::max<int64_t>(b, 1); i < e; ++i) ok &= out.row_gene[size_t(i)] >= out.row_gene[size_t(i - 1)];
                broken[size_t(w)] = ok ? 0 : 1;
            });
            for (char c : broken) grouped &= c == 0;
        }
        if (grouped && nf > 0) {
            // row_ptr[g] = number of rows of the genes before g = index of the first row whose gene is >= g: the rows where the
            // gene changes give it for the genes that have rows, the others take the next such gene's (round 5: host threads)
            const int wn2 = worker_count(n);
            std::vector<int64_t> first_defined(size_t(std::max(wn2, 1)) + 1, nf);  // per gene range: its first gene's value once filled
            parallel_ranges(n + 1, wn2, [&](int64_t gb, int64_t ge, int) {
                for (int64_t g = gb; g < ge; ++g) out.row_ptr[size_t(g)] = -1;
            });
            parallel_ranges(nf, wf, [&](int64_t b, int64_t e, int) {
                for (int64_t i = b; i < e; ++i)
                    if (i == 0 || out.row_gene[size_t(i)] != out.row_gene[size_t(i - 1)]) out.row_ptr[size_t(out.row_gene[size_t(i)])] = i;
            });
            out.row_ptr[size_t(n)] = nf;
            // fill the genes without rows from behind: every range first reports the value its first gene will get ...
            parallel_ranges(n + 1, wn2, [&](int64_t gb, int64_t ge, int w) {
                int64_t v = -1;
                for (int64_t g = gb; g < ge && v < 0; ++g) v = out.row_ptr[size_t(g)];
                first_defined[size_t(w)] = v;  // (-1: nothing defined in this range)
            });
            for (int w = std::max(wn2, 1) - 1; w >= 0; --w)
                if (first_defined[size_t(w)] < 0) first_defined[size_t(w)] = first_defined[size_t(w) + 1];
            // ... then fills its own genes with the value carried in from the ranges behind it
            parallel_ranges(n + 1, wn2, [&](int64_t gb, int64_t ge, int w) {
                int64_t carry = first_defined[size_t(w) + 1];
                for (int64_t g = ge - 1; g >= gb; --g) {
                    if (out.row_ptr[size_t(g)] < 0) out.row_ptr[size_t(g)] = carry;
                    else carry = out.row_ptr[size_t(g)];
                }
            });
        } else {
            std::fill(out.row_ptr.begin(), out.row_ptr.end(), int64_t(0));
            for (int64_t i = 0; i < nf; ++i) ++out.row_ptr[size_t(out.row_gene[size_t(i)]) + 1];
            for (int64_t g = 0; g < n; ++g) out.row_ptr[size_t(g) + 1] += out.row_ptr[size_t(g)];
        }
        ph.lap("row pointers");
        if (grouped) {
            parallel_ranges(n, worker_count(nf), [&](int64_t gb, int64_t ge, int) {
                for (int64_t g = gb; g < ge; ++g) {
                    const int64_t r0 = out.row_ptr[size_t(g)], r1 = out.row_ptr[size_t(g) + 1];
                    for (int64_t r = r0; r < r1; ++r) {  // insertion sort of a handful of rows, stable
                        const int64_t ds = t.domain_start[r];
                        int64_t q = r;
                        while (q > r0 && t.domain_start[out.row_order[size_t(q - 1)]] > ds) {
                            out.row_order[size_t(q)] = out.row_order[size_t(q - 1)];
                            --q;
                        }
                        out.row_order[size_t(q)] = r;
                    }
                }
            });
        } else {
            int64_t min_ds = 0;
            for (int64_t i = 0; i < nf; ++i) min_ds = std::min(min_ds, t.domain_start[i]);
            UVec<uint64_t> key(static_cast<size_t>(nf));
            bool fits = true;
            for (int64_t i = 0; i < nf; ++i) {
                const uint64_t ds = uint64_t(t.domain_start[i] - min_ds);
                if (ds >> 32) fits = false;
                key[size_t(i)] = (uint64_t(uint32_t(out.row_gene[size_t(i)])) << 32) | (ds & 0xffffffffull);
                out.row_order[size_t(i)] = i;
            }
            if (fits) {
                radix_sort_by_key(key, out.row_order);
            } else {
                std::stable_sort(out.row_order.begin(), out.row_order.end(), [&](int64_t a, int64_t b) {
                    const int32_t pa = out.row_gene[size_t(a)], pb = out.row_gene[size_t(b)];
                    return pa != pb ? pa < pb : t.domain_start[a] < t.domain_start[b];
                });
            }
        }
    }
    ph.lap("row order");
    // ---- domain names -> attribute ids: one dictionary lookup per DISTINCT name.  Accessions ("PF00001") fit
    // a machine word: those go through a small table keyed by the word itself, no memcmp.  Every worker keeps
    // its own code space; codes only have to tell the names of ONE gene apart, so they are made global by
    // adding the worker's base after the pass.
    UVec<int32_t> row_dom(static_cast<size_t>(nf));
    std::vector<int32_t> dom_attr;    // per distinct (worker, name): attribute id or -1
    std::vector<int32_t> dom_marker;  // ... and its index in the caller's marker list or -1 (antismash criterion)
    const bool want_markers = t.n_markers > 0;
    if (t.n_markers > 256) {  // kSegMaxMarkers of the segmenter (crf_device.hpp)
        set_error("at most 256 marker domains");
        return GECCO_CRF_EINVAL;
    }
    std::unordered_map<std::string, int32_t> marker_index;
    for (int64_t k = 0; k < t.n_markers; ++k) {
        const Str d = at(t.markers, k);
        marker_index.emplace(std::string(reinterpret_cast<const char *>(d.p), d.n), int32_t(k));
    }
    {
        std::vector<std::vector<int32_t>> attr_of(static_cast<size_t>(wf)), mark_of(static_cast<size_t>(wf));
        std::vector<std::vector<Str>> name_of(static_cast<size_t>(wf));
        parallel_ranges(nf, wf, [&](int64_t b, int64_t e, int w) {
            Interner doms(4096);
            std::string key;
            std::vector<int32_t> &attrs = attr_of[size_t(w)], &marks = mark_of[size_t(w)];
            auto code_of = [&](const Str &d) {
                bool fresh = false;
                const int32_t code = doms.intern(d, &fresh);
                if (fresh) {
                    key.assign(reinterpret_cast<const char *>(d.p), d.n);
                    auto it = m.attr_index.find(key);
                    attrs.push_back(it == m.attr_index.end() ? -1 : it->second);
                    if (want_markers) {
                        auto mk = marker_index.find(key);
                        marks.push_back(mk == marker_index.end() ? -1 : mk->second);
                    }
                }
                return code;
            };
            constexpr uint32_t kSmall = 1u << 14;
            std::vector<uint64_t> sw(kSmall, 0);
            std::vector<int32_t> sc(kSmall, -1);
            for (int64_t i = b; i < e; ++i) {
                const Str d = at(t.domain, i);
                int32_t code = -1;
                if (d.n >= 1 && d.n <= 7) {
                    uint64_t wd = 0;
                    std::memcpy(&wd, d.p, d.n);
                    wd |= uint64_t(d.n) << 56;
                    uint32_t sl = uint32_t((wd * 0x9E3779B97F4A7C15ull) >> 50);
                    for (int probe = 0; probe < 8; ++probe, sl = (sl + 1) & (kSmall - 1)) {
                        if (sw[sl] == wd) {
                            code = sc[sl];
                            break;
                        }
                        if (sc[sl] < 0) {
                            code = code_of(d);
                            sw[sl] = wd;
                            sc[sl] = code;
                            break;
                        }
                    }
                }
                row_dom[size_t(i)] = code >= 0 ? code : code_of(d);
            }
            name_of[size_t(w)] = doms.keys;
        });
        // one code per distinct NAME across workers (a gene's rows may straddle two ranges): a name the model knows takes its
        // attribute id, the others follow behind the A ids in first-appearance order -- the only names still interned here
        // (round 5: interning every worker's every name again was 2 ms of serial work per 16 workers x 2 659 names)
        const int32_t A = m.A;
        dom_attr.resize(size_t(A));
        for (int32_t a2 = 0; a2 < A; ++a2) dom_attr[size_t(a2)] = a2;
        if (want_markers) dom_marker.assign(size_t(A), -1);
        Interner unknown(64);
        std::vector<std::vector<int32_t>> global(static_cast<size_t>(wf));
        for (int w = 0; w < wf; ++w) {
            global[size_t(w)].resize(name_of[size_t(w)].size());
            for (size_t c = 0; c < name_of[size_t(w)].size(); ++c) {
                const int32_t a2 = attr_of[size_t(w)][c];
                if (a2 >= 0 && a2 < A) {
                    if (want_markers) dom_marker[size_t(a2)] = mark_of[size_t(w)][c];
                    global[size_t(w)][c] = a2;
                    continue;
                }
                bool fresh = false;
                const int32_t g = A + unknown.intern(name_of[size_t(w)][c], &fresh);
                if (fresh) {
                    dom_attr.push_back(-1);
                    if (want_markers) dom_marker.push_back(mark_of[size_t(w)][c]);
                }
                global[size_t(w)][c] = g;
            }
        }
        parallel_ranges(nf, wf, [&](int64_t b, int64_t e, int w) {
                for (int64_t i = b; i < e; ++i) row_dom[size_t(i)] = global[size_t(w)][size_t(row_dom[size_t(i)])];
            });
    }
    ph.lap("domain names");
    // ---- CSR: per gene its distinct names in row order, unknown names dropped
    out.pinned = have_device();
    {   // one block (a pinned allocation costs ~1 ms whatever its size)
        auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
        const size_t o_c = 0, o_g = o_c + up((cptr.size() + 1) * 4), o_a = o_g + up((size_t(n) + 2) * 4),
                     o_n = o_a + up((size_t(nf) + 4) * 4), o_mp = o_n + up(size_t(n) + 8),
                     o_mi = o_mp + (want_markers ? up((size_t(n) + 2) * 4) : 0),
                     bytes = o_mi + (want_markers ? up((size_t(nf) + 4) * 4) : 0);
        out.block = static_cast<char *>(alloc_staging(bytes, out.pinned));
        out.contig_ptr = reinterpret_cast<int32_t *>(out.block + o_c);
        out.gene_ptr = reinterpret_cast<int32_t *>(out.block + o_g);
        out.attr_id = reinterpret_cast<int32_t *>(out.block + o_a);
        out.annotated = reinterpret_cast<uint8_t *>(out.block + o_n);
        if (want_markers) {
            out.marker_ptr = reinterpret_cast<int32_t *>(out.block + o_mp);
            out.marker_id = reinterpret_cast<int32_t *>(out.block + o_mi);
        }
    }
    std::memcpy(out.contig_ptr, cptr.data(), cptr.size() * 4);
    if (cptr.size() == 1) out.contig_ptr[0] = 0;
    // a gene's kept attributes: called twice, to count and to fill
    auto gene_attrs = [&](int64_t g, int32_t *dst) {
        const int64_t r0 = out.row_ptr[size_t(g)], r1 = out.row_ptr[size_t(g) + 1];
        int32_t seen[16];
        int n_seen = 0, kept = 0;
        for (int64_t r = r0; r < r1; ++r) {
            const int32_t code = row_dom[size_t(out.row_order[size_t(r)])];
            bool dup = false;
            if (r1 - r0 <= 16) {
                for (int k = 0; k < n_seen; ++k) dup |= seen[k] == code;
                if (!dup) seen[n_seen++] = code;
            } else {  // a gene with many rows: look the code up among the rows before it
                for (int64_t q = r0; q < r && !dup; ++q) dup = row_dom[size_t(out.row_order[size_t(q)])] == code;
            }
            if (dup) continue;
            const int32_t a = dom_attr[size_t(code)];
            if (a >= 0) {
                if (dst) dst[kept] = a;
                ++kept;
            }
        }
        return kept;
    };
    const int wc = worker_count(nf);
    out.gene_ptr[0] = 0;
    parallel_ranges(n, wc, [&](int64_t gb, int64_t ge, int) {
        for (int64_t g = gb; g < ge; ++g) {
            out.gene_ptr[g + 1] = gene_attrs(g, nullptr);
            out.annotated[g] = out.row_ptr[size_t(g) + 1] > out.row_ptr[size_t(g)] ? 1 : 0;
        }
    });
    for (int64_t g = 0; g < n; ++g) out.gene_ptr[g + 1] += out.gene_ptr[g];
    parallel_ranges(n, wc, [&](int64_t gb, int64_t ge, int) {
        for (int64_t g = gb; g < ge; ++g)
            if (out.gene_ptr[g + 1] > out.gene_ptr[g]) (void)gene_attrs(g, out.attr_id + out.gene_ptr[g]);
    });
    out.nnz = n ? out.gene_ptr[n] : 0;
    ph.lap("csr");
    if (want_markers) {
        // per gene the distinct marker domains among ALL its rows (refine.py:158: names of every domain of the
        // member genes, whether the CRF knows them or not); few rows carry one
        auto gene_markers = [&](int64_t g, int32_t *dst) {
            int32_t kept = 0;
            for (int64_t r = out.row_ptr[size_t(g)]; r < out.row_ptr[size_t(g) + 1]; ++r) {
                const int32_t mk = dom_marker[size_t(row_dom[size_t(out.row_order[size_t(r)])])];
                if (mk < 0) continue;
                bool dup = false;
                for (int64_t q = out.row_ptr[size_t(g)]; q < r && !dup; ++q)
                    dup = dom_marker[size_t(row_dom[size_t(out.row_order[size_t(q)])])] == mk;
                if (dup) continue;
                if (dst) dst[kept] = mk;
                ++kept;
            }
            return kept;
        };
        out.marker_ptr[0] = 0;
        parallel_ranges(n, wc, [&](int64_t gb, int64_t ge, int) {
            for (int64_t g = gb; g < ge; ++g) out.marker_ptr[g + 1] = gene_markers(g, nullptr);
        });
        for (int64_t g = 0; g < n; ++g) out.marker_ptr[g + 1] += out.marker_ptr[g];
        parallel_ranges(n, wc, [&](int64_t gb, int64_t ge, int) {
            for (int64_t g = gb; g < ge; ++g)
                if (out.marker_ptr[g + 1] > out.marker_ptr[g]) (void)gene_markers(g, out.marker_id + out.marker_ptr[g]);
        });
        ph.lap("markers");
    }
    return GECCO_CRF_OK;
}

// ---- output-column helpers of predict_tables (were single-threaded numpy passes: 2 of the table level's 10 ms) -------------
int gather_f64(const double *src, int64_t n_src, const int32_t *idx, int64_t n, double *out) {
    std::atomic<int> bad{0};
    parallel_ranges(n, worker_count(n), [&](int64_t b, int64_t e, int) {
        for (int64_t i = b; i < e; ++i) {
            const int64_t j = idx[i];
            if (j < 0 || j >= n_src) {
                bad.store(1);
                out[i] = std::nan("");
            } else {
                out[i] = src[j];
            }
        }
    });
    if (bad.load()) {
        set_error("gather: index out of range");
        return GECCO_CRF_EINVAL;
    }
    return GECCO_CRF_OK;
}

int order_info(const Packed &pk, const int64_t *gene_start, const int64_t *gene_end, int64_t n_gene_rows, int32_t *rows_in_order,
               int32_t *refiner_order_differs) {
    const int64_t n = pk.n_genes;
    std::atomic<int> out_of_order{0}, differs{0}, bad{0};
    // contig of every range's first gene by binary search, then a walk
    parallel_ranges(n, worker_count(n), [&](int64_t b, int64_t e, int) {
        if (b >= e) return;
        int c = int(std::upper_bound(pk.contig_ptr, pk.contig_ptr + pk.n_contigs + 1, int32_t(b)) - pk.contig_ptr) - 1;
        int ooo = 0, dif = 0;
        for (int64_t g = b; g < e; ++g) {
            while (c + 1 <= pk.n_contigs && pk.contig_ptr[c + 1] <= g) ++c;
            const int64_t r = pk.gene_row[size_t(g)];
            if (r != g) ooo = 1;
            if (r < 0 || r >= n_gene_rows) {
                bad.store(1);
                continue;
            }
            if (g > pk.contig_ptr[c]) {  // the gene before it belongs to the same contig
                const int64_t q = pk.gene_row[size_t(g - 1)];
                if (q >= 0 && q < n_gene_rows && gene_start[r] == gene_start[q] && gene_end[r] < gene_end[q]) dif = 1;
            }
        }
        if (ooo) out_of_order.store(1);
        if (dif) differs.store(1);
    });
    if (bad.load()) {
        set_error("order_info: a gene without a gene-table row");
        return GECCO_CRF_EINVAL;
    }
    *rows_in_order = (n_gene_rows == n && !out_of_order.load()) ? 1 : 0;
    *refiner_order_differs = differs.load();
    return GECCO_CRF_OK;
}

// ---- exactly rounded mean of doubles: what statistics.mean returns -------------------------------------
// Every finite double is an integer multiple of 2^-1074; positive and negative values are accumulated exactly in
// two fixed-point integers with that unit (2^-1074 .. 2^1024 and 64 bits of carry room), their difference is divided
// by the count (long division, remainder kept) and rounded to nearest-even once.  Infinities follow float
// arithmetic (inf, -inf, or NaN when both signs occur), as statistics.mean does.
namespace {
struct ExactSum {
    static constexpr int kLimbs = 34;  // bits 0 .. 2175 in units of 2^-1074: |v| < 2^1024 needs 2098, the rest is carry room
    uint64_t limb[kLimbs];
    ExactSum() { std::memset(limb, 0, sizeof(limb)); }
    void add(double v) {  // v > 0, finite
        int e;
        const double f = std::frexp(v, &e);                  // v = f 2^e, f in [0.5, 1)
        uint64_t mant = uint64_t(std::ldexp(f, 53));       // 53-bit integer
        int64_t pos = int64_t(e) - 53 + 1074;                // bit position of mant's lsb
        if (pos < 0) {                                       // subnormal: low bits are zero anyway
            mant >>= -pos;
            pos = 0;
        }
        const int li = int(pos >> 6), sh = int(pos & 63);
        if (li + 1 >= kLimbs) return;  // unreachable for finite doubles (li <= 32); keeps the writes in bounds regardless
        unsigned __int128 x = (unsigned __int128)mant << sh;
        uint64_t lo = uint64_t(x), hi = uint64_t(x >> 64);
        unsigned __int128 c = (unsigned __int128)limb[li] + lo;
        limb[li] = uint64_t(c);
        c = (c >> 64) + limb[li + 1] + hi;
        limb[li + 1] = uint64_t(c);
        uint64_t carry = uint64_t(c >> 64);
        for (int k = li + 2; carry && k < kLimbs; ++k) {
            c = (unsigned __int128)limb[k] + carry;
            limb[k] = uint64_t(c);
            carry = uint64_t(c >> 64);
        }
    }
    int compare(const ExactSum &o) const {
        for (int k = kLimbs - 1; k >= 0; --k)
            if (limb[k] != o.limb[k]) return limb[k] < o.limb[k] ? -1 : 1;
        return 0;
    }
    void subtract(const ExactSum &o) {  // *this >= o
        uint64_t borrow = 0;
        for (int k = 0; k < kLimbs; ++k) {
            const unsigned __int128 d = (unsigned __int128)limb[k] - o.limb[k] - borrow;
            limb[k] = uint64_t(d);
            borrow = uint64_t(d >> 64) & 1u;
        }
    }
    double mean(uint64_t count) const {
        // Q = floor(S / count), r = S mod count, most significant limb first
        uint64_t q[kLimbs];
        unsigned __int128 rem = 0;
        for (int k = kLimbs - 1; k >= 0; --k) {
            const unsigned __int128 cur = (rem << 64) | limb[k];
            q[k] = uint64_t(cur / count);
            rem = cur % count;
        }
        int top = -1;  // highest set bit of Q
        for (int k = kLimbs - 1; k >= 0 && top < 0; --k)
            if (q[k]) top = k * 64 + 63 - __builtin_clzll(q[k]);
        auto bit = [&](int b) { return b >= 0 ? (q[b >> 6] >> (b & 63)) & 1ull : 0ull; };
        if (top < 53) {  // Q fits the mantissa: unit 2^-1074 is the last place, the remainder decides the rounding
            uint64_t mnt = q[0];
            const unsigned __int128 twice = rem * 2;
            if (twice > count || (twice == count && (mnt & 1))) ++mnt;
            return std::ldexp(double(mnt), -1074);
        }
        const int drop = top - 52;  // bits below the 53 kept
        uint64_t mnt = 0;
        for (int b = top; b >= drop; --b) mnt = (mnt << 1) | bit(b);
        const uint64_t round_bit = bit(drop - 1);
        bool sticky = rem != 0;
        for (int b = drop - 2; b >= 0 && !sticky; --b) sticky = bit(b);
        if (round_bit && (sticky || (mnt & 1))) ++mnt;
        return std::ldexp(double(mnt), drop - 1074);  // overflows to +inf by itself
    }
};
}  // namespace

double exact_mean(const double *v, int64_t n) {
    ExactSum pos, neg;
    uint64_t cnt = 0;
    double inf_sum = 0.0;  // float sum of the infinite values: +inf, -inf, or NaN when both occur
    bool any_inf = false;
    for (int64_t i = 0; i < n; ++i) {
        const double x = v[i];
        if (x != x) continue;
        ++cnt;
        if (std::isinf(x)) {
            inf_sum += x;
            any_inf = true;
        } else if (x > 0.0) {
            pos.add(x);
        } else if (x < 0.0) {
            neg.add(-x);
        }
    }
    if (!cnt) return std::nan("");
    if (any_inf) return inf_sum;
    const int c = pos.compare(neg);
    if (c == 0) return 0.0;
    if (c > 0) {
        pos.subtract(neg);
        return pos.mean(cnt);
    }
    neg.subtract(pos);
    return -neg.mean(cnt);
}

// Rows of clusters.tsv (gecco/model.py:731-760) for the called clusters, columnar.
int cluster_rows(const Packed &pk, const gecco_crf_table_columns &t, const int64_t *gene_end, const int64_t *feat_end,
                 const int32_t *seg, int32_t n_seg, const double *seg_p, const int64_t *seg_off, ClusterRows &out) {
    Phase ph;
    out.start.resize(size_t(n_seg));
    out.end.resize(size_t(n_seg));
    out.average_p.resize(size_t(n_seg));
    out.max_p.resize(size_t(n_seg));
    for (int32_t k = 0; k < n_seg; ++k) {
        const int32_t a = seg[4 * k + 2], b = seg[4 * k + 3];
        if (a < 0 || b < a || b > pk.n_genes) {
            set_error("cluster_rows: segment outside the gene range");
            return GECCO_CRF_EINVAL;
        }
    }
    auto gene_pid = [&](int32_t g) {
        const int64_t r = pk.gene_row[size_t(g)];
        return r >= 0 ? at(t.gene_protein_id, r) : at(t.protein_id, -1 - r);
    };
    auto gene_sid = [&](int32_t g) {
        const int64_t r = pk.gene_row[size_t(g)];
        return r >= 0 ? at(t.gene_sequence_id, r) : at(t.sequence_id, -1 - r);
    };
    // clusters are independent: every worker writes the numeric columns in place and its own piece of the text
    // columns, the pieces are concatenated afterwards
    int64_t member_genes = 0;
    for (int32_t k = 0; k < n_seg; ++k) member_genes += seg[4 * k + 3] - seg[4 * k + 2];
    const int workers = std::min<int>(worker_count(member_genes * 4 + n_seg * 64), std::max(1, n_seg));
    struct Piece {
        StrOut col[4];  // sequence_id, cluster_id, proteins, domains
    };
    std::vector<Piece> pieces(static_cast<size_t>(workers));
    parallel_ranges(n_seg, workers, [&](int64_t kb, int64_t ke, int w) {
        Piece &pc = pieces[size_t(w)];
        for (StrOut &c : pc.col) c.offsets.clear();
        auto put = [](StrOut &c, const Str &s) { c.data.insert(c.data.end(), s.p, s.p + s.n); };
        auto close = [](StrOut &c) { c.offsets.push_back(int64_t(c.data.size())); };
        // sorting member ids / domain names: compare their first 8 bytes as one big-endian word, the rest only on ties
        struct Keyed {
            uint64_t key;
            Str s;
        };
        auto keyed = [](const Str &s) {
            uint64_t wd = 0;
            std::memcpy(&wd, s.p, std::min<uint32_t>(s.n, 8));
            return Keyed{__builtin_bswap64(wd), s};
        };
        std::vector<Keyed> names;
        auto join_sorted = [&](StrOut &col) {
            std::sort(names.begin(), names.end(),
                      [](const Keyed &x, const Keyed &y) { return x.key != y.key ? x.key < y.key : less(x.s, y.s); });
            for (size_t i = 0; i < names.size(); ++i) {
                if (i) col.data.push_back(uint8_t(';'));
                put(col, names[i].s);
            }
            close(col);
        };
        for (int64_t k = kb; k < ke; ++k) {
            const int32_t number = seg[4 * k + 1], a = seg[4 * k + 2], b = seg[4 * k + 3];
            int64_t lo = INT64_MAX, hi = INT64_MIN;
            for (int32_t g = a; g < b; ++g) {
                const int64_t r = pk.gene_row[size_t(g)];
                lo = std::min(lo, r >= 0 ? t.gene_start[r] : t.start[-1 - r]);
                hi = std::max(hi, r >= 0 ? gene_end[r] : feat_end[-1 - r]);
            }
            out.start[size_t(k)] = b > a ? lo : 0;
            out.end[size_t(k)] = b > a ? hi : 0;
            const double *ps = seg_p + seg_off[k];
            const int64_t np = seg_off[k + 1] - seg_off[k];
            out.average_p[size_t(k)] = exact_mean(ps, np);
            double mx = std::nan("");
            for (int64_t i = 0; i < np; ++i)
                if (ps[i] == ps[i] && !(mx >= ps[i])) mx = ps[i];
            out.max_p[size_t(k)] = mx;
            const Str sid = b > a ? gene_sid(a) : Str{nullptr, 0};
            put(pc.col[0], sid);
            close(pc.col[0]);
            put(pc.col[1], sid);
            const std::string tail = "_cluster_" + std::to_string(number);
            pc.col[1].data.insert(pc.col[1].data.end(), tail.begin(), tail.end());
            close(pc.col[1]);
            names.clear();
            for (int32_t g = a; g < b; ++g) names.push_back(keyed(gene_pid(g)));
            join_sorted(pc.col[2]);
            names.clear();
            for (int64_t r = pk.row_ptr[size_t(a)]; r < pk.row_ptr[size_t(b)]; ++r)
                names.push_back(keyed(at(t.domain, pk.row_order[size_t(r)])));
            join_sorted(pc.col[3]);
        }
    });
    StrOut *cols[4] = {&out.sequence_id, &out.cluster_id, &out.proteins, &out.domains};
    for (int c = 0; c < 4; ++c) {
        StrOut &dst = *cols[c];
        size_t bytes = 0;
        for (const Piece &pc : pieces) bytes += pc.col[c].data.size();
        dst.data.clear();
        dst.data.reserve(bytes);
        dst.offsets.assign(1, 0);
        dst.offsets.reserve(size_t(n_seg) + 1);
        for (const Piece &pc : pieces) {
            const int64_t base = int64_t(dst.data.size());
            dst.data.insert(dst.data.end(), pc.col[c].data.begin(), pc.col[c].data.end());
            for (int64_t o : pc.col[c].offsets) dst.offsets.push_back(base + o);
        }
    }
    ph.lap("cluster rows");
    return GECCO_CRF_OK;
}

}  // namespace gecco

// ---- TSV text of a table (gecco/_base.py:133-152 dump rules: tab-separated, NaN as an empty field) ---------
// Floats are written the way Python's repr() writes them (the reference goes through polars / csv with repr
// digits; the golden tables are compared byte for byte): the SHORTEST digit string that round-trips
// (std::to_chars), in positional notation when -4 <= exponent < 16, else d.ddde+XX.
namespace gecco {
namespace {
inline char *put_float_repr(char *p, double v) {
    if (v != v) return p;  // NaN: empty field
    if (std::isinf(v)) {
        const char *s = v > 0 ? "inf" : "-inf";
        const size_t n = std::strlen(s);
        std::memcpy(p, s, n);
        return p + n;
    }
    char sci[40];
    const auto r = std::to_chars(sci, sci + sizeof(sci) - 1, v, std::chars_format::scientific);
    *r.ptr = 0;  // sci = [-]d[.ddd]e[+-]XX
    char *q = sci;
    if (*q == '-') *p++ = *q++;
    char digits[24];
    int nd = 0;
    digits[nd++] = *q++;
    if (*q == '.') {
        ++q;
        while (*q != 'e') digits[nd++] = *q++;
    }
    ++q;  // 'e'
    const int ex = int(std::strtol(q, nullptr, 10));
    const int decpt = ex + 1;  // position of the decimal point relative to the first digit
    if (decpt > 16 || decpt < -3) {  // repr: exponential outside 1e-4 <= |v| < 1e16
        *p++ = digits[0];
        if (nd > 1) {
            *p++ = '.';
            std::memcpy(p, digits + 1, size_t(nd - 1));
            p += nd - 1;
        }
        *p++ = 'e';
        *p++ = ex < 0 ? '-' : '+';
        const int ax = ex < 0 ? -ex : ex;
        if (ax < 10) *p++ = '0';
        const auto r2 = std::to_chars(p, p + 8, ax);
        return r2.ptr;
    }
    if (decpt <= 0) {  // 0.000ddd
        *p++ = '0';
        *p++ = '.';
        for (int k = 0; k < -decpt; ++k) *p++ = '0';
        std::memcpy(p, digits, size_t(nd));
        return p + nd;
    }
    if (decpt >= nd) {  // ddd000.0
        std::memcpy(p, digits, size_t(nd));
        p += nd;
        for (int k = nd; k < decpt; ++k) *p++ = '0';
        *p++ = '.';
        *p++ = '0';
        return p;
    }
    std::memcpy(p, digits, size_t(decpt));
    p += decpt;
    *p++ = '.';
    std::memcpy(p, digits + decpt, size_t(nd - decpt));
    return p + (nd - decpt);
}
}  // namespace

// kinds: 0 = text (data + offsets), 1 = int64, 2 = float64.  Returns a malloc'ed buffer.
int format_tsv(int64_t n_rows, int32_t n_cols, const int32_t *kinds, const void *const *data, const int64_t *const *offsets,
               const char *header, uint8_t **out, int64_t *out_len) {
    const int workers = worker_count(n_rows * n_cols / 4 + 1);
    std::vector<std::vector<char>> piece(static_cast<size_t>(workers));
    parallel_ranges(n_rows, workers, [&](int64_t b, int64_t e, int w) {
        std::vector<char> &buf = piece[size_t(w)];
        size_t text = 0;
        for (int32_t c = 0; c < n_cols; ++c)
            if (kinds[c] == 0) text += size_t(offsets[c][e] - offsets[c][b]);
        buf.resize(text + size_t(e - b) * size_t(n_cols) * 26 + 64);
        char *p = buf.data();
        for (int64_t i = b; i < e; ++i) {
            for (int32_t c = 0; c < n_cols; ++c) {
                if (c) *p++ = '\t';
                if (kinds[c] == 0) {
                    const int64_t a0 = offsets[c][i], a1 = offsets[c][i + 1];
                    std::memcpy(p, static_cast<const uint8_t *>(data[c]) + a0, size_t(a1 - a0));
                    p += a1 - a0;
                } else if (kinds[c] == 1) {
                    p = std::to_chars(p, p + 24, static_cast<const int64_t *>(data[c])[i]).ptr;
                } else {
                    p = put_float_repr(p, static_cast<const double *>(data[c])[i]);
                }
            }
            *p++ = '\n';
        }
        buf.resize(size_t(p - buf.data()));
    });
    const size_t hl = header ? std::strlen(header) : 0;
    size_t total = hl;
    for (const auto &b : piece) total += b.size();
    uint8_t *res = static_cast<uint8_t *>(std::malloc(total ? total : 1));
    if (!res) {
        set_error("out of host memory");
        return GECCO_CRF_ENOMEM;
    }
    size_t at = 0;
    if (hl) std::memcpy(res, header, hl);
    at = hl;
    for (const auto &b : piece) {
        if (!b.empty()) std::memcpy(res + at, b.data(), b.size());
        at += b.size();
    }
    *out = res;
    *out_len = int64_t(total);
    return GECCO_CRF_OK;
}

}  // namespace gecco
