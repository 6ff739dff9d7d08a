// lCRF / FOMC v100 reader.  Layout facts ([EXT] CRFsuite 0.12 crf1d_model.c; verified on
// the embedded model): little-endian throughout;
//   header  48 B : "lCRF" size "FOMC" version num_features num_labels num_attrs
//                  off_features off_labels off_attrs off_labelrefs off_attrrefs
//   FEAT chunk   : "FEAT" size num, then num x {u32 type, u32 src, u32 dst, f64 weight}
//                  type 0 = state (src=attr, dst=label), 1 = transition (src,dst labels)
//   CQDB chunk   : "CQDB" size flag byteorder(0x62445371) bwd_size bwd_offset, 256x(off,num)
//                  hash refs, records {u32 id, u32 ksize, key\0}; bwd[id] = record offset
//                  relative to the chunk start.  Only the id->string direction is read; the
//                  string->id map is rebuilt as an std::unordered_map (no need for CQDB's hash).
//   LFRF / AFRF  : "LFRF"/"AFRF" size num, num x u32 absolute offsets -> {u32 n, u32 fid[n]}
// The dense tables are filled by walking AFRF/LFRF exactly the way the tagger scores
// ([EXT] crf1dt_state_score: state[t][f.dst] += f.weight; crf1dt_transition_score:
// trans[i][f.dst] = f.weight), so duplicated state features would accumulate as they do there.
#include "crf_model.hpp"

#include <cstring>

#include "../../include/gecco_crf.h"

namespace gecco {

static thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }
const char *last_error() { return g_last_error.c_str(); }

namespace {
struct Reader {
    const uint8_t *p;
    size_t n;
    bool ok(size_t off, size_t len) const { return off <= n && len <= n - off; }
    uint32_t u32(size_t off) const {
        uint32_t v;
        std::memcpy(&v, p + off, 4);
        return v;
    }
    double f64(size_t off) const {
        double v;
        std::memcpy(&v, p + off, 8);
        return v;
    }
};

int read_cqdb(const Reader &r, size_t off, uint32_t expect, std::vector<std::string> &names, const char *what) {
    if (!r.ok(off, 24) || std::memcmp(r.p + off, "CQDB", 4) != 0) {
        set_error(std::string("lCRF: bad CQDB chunk for ") + what);
        return GECCO_CRF_EFORMAT;
    }
    uint32_t flag = r.u32(off + 8), byteorder = r.u32(off + 12);
    uint32_t bwd_size = r.u32(off + 16), bwd_offset = r.u32(off + 20);
    if (byteorder != 0x62445371u) {
        set_error("lCRF: CQDB byte-order mark mismatch");
        return GECCO_CRF_EFORMAT;
    }
    if ((flag & 1u) || bwd_size != expect || !r.ok(off + bwd_offset, size_t(bwd_size) * 4)) {
        set_error(std::string("lCRF: CQDB backward array missing or wrong size for ") + what);
        return GECCO_CRF_EFORMAT;
    }
    names.resize(bwd_size);
    for (uint32_t i = 0; i < bwd_size; ++i) {
        size_t rec = off + r.u32(off + bwd_offset + size_t(i) * 4);
        if (!r.ok(rec, 8)) goto bad;
        {
            uint32_t id = r.u32(rec), ksize = r.u32(rec + 4);
            if (id != i || ksize == 0 || !r.ok(rec + 8, ksize) || r.p[rec + 8 + ksize - 1] != 0) goto bad;
            names[i].assign(reinterpret_cast<const char *>(r.p + rec + 8), ksize - 1);
        }
    }
    return GECCO_CRF_OK;
bad:
    set_error(std::string("lCRF: corrupt CQDB record for ") + what);
    return GECCO_CRF_EFORMAT;
}
}  // namespace

int parse_lcrf(const uint8_t *blob, size_t n, Model &m) {
    Reader r{blob, n};
    if (blob == nullptr || n < 48 || std::memcmp(blob, "lCRF", 4) != 0 || std::memcmp(blob + 8, "FOMC", 4) != 0) {
        set_error("not a CRFsuite lCRF/FOMC model");
        return GECCO_CRF_EFORMAT;
    }
    if (r.u32(4) != n) {
        set_error("lCRF: size field does not match the buffer length");
        return GECCO_CRF_EFORMAT;
    }
    if (r.u32(12) != 100) {
        set_error("lCRF: unsupported model version");
        return GECCO_CRF_EFORMAT;
    }
    const uint32_t L = r.u32(20), A = r.u32(24);
    const size_t off_feat = r.u32(28), off_labels = r.u32(32), off_attrs = r.u32(36), off_lref = r.u32(40),
                 off_aref = r.u32(44);
    // every label and every attribute owns a CQDB record of >= 10 bytes: counts a blob of this size cannot
    // hold are corruption, and the dense A x L table must stay allocatable (2^28 weights = 2 GiB)
    if (L == 0 || L > (1u << 20) || A > (1u << 28) || uint64_t(L) * 10 > n || uint64_t(A) * 10 > n ||
        uint64_t(A) * uint64_t(L) > (1ull << 28)) {
        set_error("lCRF: implausible label/attribute counts");
        return GECCO_CRF_EFORMAT;
    }
    if (!r.ok(off_feat, 12) || std::memcmp(blob + off_feat, "FEAT", 4) != 0) {
        set_error("lCRF: FEAT chunk missing");
        return GECCO_CRF_EFORMAT;
    }
    const uint32_t nfeat = r.u32(off_feat + 8);
    if (!r.ok(off_feat + 12, size_t(nfeat) * 20)) {
        set_error("lCRF: FEAT chunk truncated");
        return GECCO_CRF_EFORMAT;
    }
    int rc;
    if ((rc = read_cqdb(r, off_labels, L, m.labels, "labels"))) return rc;
    if ((rc = read_cqdb(r, off_attrs, A, m.attrs, "attributes"))) return rc;

    m.L = int32_t(L);
    m.A = int32_t(A);
    m.n_features = int32_t(nfeat);
    m.state.assign(size_t(A) * L, 0.0);
    m.state_mask.assign(size_t(A) * L, 0);
    m.trans.assign(size_t(L) * L, 0.0);
    m.trans_mask.assign(size_t(L) * L, 0);

    auto feature = [&](uint32_t fid, uint32_t &type, uint32_t &src, uint32_t &dst, double &w) -> bool {
        if (fid >= nfeat) return false;
        size_t o = off_feat + 12 + size_t(fid) * 20;
        type = r.u32(o);
        src = r.u32(o + 4);
        dst = r.u32(o + 8);
        w = r.f64(o + 12);
        return true;
    };
    auto walk = [&](size_t off_ref, const char *magic, uint32_t count, bool is_state) -> int {
        if (!r.ok(off_ref, 12) || std::memcmp(blob + off_ref, magic, 4) != 0 || r.u32(off_ref + 8) < count ||
            !r.ok(off_ref + 12, size_t(count) * 4)) {
            set_error(std::string("lCRF: bad reference chunk ") + magic);
            return GECCO_CRF_EFORMAT;
        }
        for (uint32_t i = 0; i < count; ++i) {
            size_t o = r.u32(off_ref + 12 + size_t(i) * 4);
            if (!r.ok(o, 4)) goto bad;
            {
                uint32_t k = r.u32(o);
                if (!r.ok(o + 4, size_t(k) * 4)) goto bad;
                for (uint32_t j = 0; j < k; ++j) {
                    uint32_t type, src, dst;
                    double w;
                    if (!feature(r.u32(o + 4 + size_t(j) * 4), type, src, dst, w)) goto bad;
                    if (dst >= L || src != i || type != (is_state ? 0u : 1u)) goto bad;
                    if (is_state) {
                        m.state[size_t(i) * L + dst] += w;
                        m.state_mask[size_t(i) * L + dst] = 1;
                    } else {
                        m.trans[size_t(i) * L + dst] = w;
                        m.trans_mask[size_t(i) * L + dst] = 1;
                    }
                }
            }
        }
        return GECCO_CRF_OK;
    bad:
        set_error(std::string("lCRF: corrupt feature reference in ") + magic);
        return GECCO_CRF_EFORMAT;
    };
    if ((rc = walk(off_aref, "AFRF", A, true))) return rc;
    if ((rc = walk(off_lref, "LFRF", L, false))) return rc;

    for (int32_t i = 0; i < m.L; ++i) m.label_index.emplace(m.labels[i], i);
    m.attr_index.reserve(size_t(A) * 2);
    for (int32_t i = 0; i < m.A; ++i) m.attr_index.emplace(m.attrs[i], i);
    return GECCO_CRF_OK;
}

int model_from_tables(const double *state, const double *trans, int32_t A, int32_t L, Model &m) {
    if (!state || !trans || A < 0 || L <= 0) {
        set_error("model_from_tables: bad arguments");
        return GECCO_CRF_EINVAL;
    }
    m.L = L;
    m.A = A;
    m.state.assign(state, state + size_t(A) * L);
    m.trans.assign(trans, trans + size_t(L) * L);
    m.state_mask.resize(size_t(A) * L);
    m.trans_mask.assign(size_t(L) * L, 1);
    m.n_features = L * L;
    for (size_t i = 0; i < m.state.size(); ++i) {
        m.state_mask[i] = m.state[i] != 0.0;
        m.n_features += m.state_mask[i];
    }
    m.labels.resize(L);
    m.attrs.resize(A);
    for (int32_t i = 0; i < L; ++i) {
        m.labels[i] = std::to_string(i);
        m.label_index.emplace(m.labels[i], i);
    }
    m.attr_index.reserve(size_t(A) * 2);
    for (int32_t i = 0; i < A; ++i) {
        m.attrs[i] = "a" + std::to_string(i);
        m.attr_index.emplace(m.attrs[i], i);
    }
    return GECCO_CRF_OK;
}

}  // namespace gecco
