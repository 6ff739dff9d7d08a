// Host-side CRF model: parses the CRFsuite `lCRF` blob that GECCO's model.pkl carries
// (what [EXT] pycrfsuite.Tagger.open does for gecco/crf/__init__.py:99) into dense tables.
#pragma once
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace gecco {

void set_error(const std::string &msg);
const char *last_error();

struct DeviceTables;  // defined in crf_plan.hpp (per-device uploaded copies)
struct Session;       // defined in crf_session.cpp (batch driver)

struct Model {
    int32_t L = 0, A = 0, n_features = 0;
    std::vector<std::string> labels, attrs;
    std::unordered_map<std::string, int32_t> label_index, attr_index;
    std::vector<double> state;        // A x L, 0 where the model has no feature
    std::vector<uint8_t> state_mask;  // A x L
    std::vector<double> trans;        // L x L
    std::vector<uint8_t> trans_mask;  // L x L

    // lazily created per-device copies, owned by the model
    mutable std::mutex dev_mutex;
    mutable std::vector<DeviceTables *> dev_tables;
    // per-device batch drivers behind the one-shot entry points (created on first use, owned by the model)
    mutable std::vector<std::pair<int, Session *>> sessions;
    ~Model();
};

// returns 0 or a GECCO_CRF_E* code (message via set_error)
int parse_lcrf(const uint8_t *blob, size_t n, Model &out);
int model_from_tables(const double *state, const double *trans, int32_t A, int32_t L, Model &out);

}  // namespace gecco
