// ASan/UBSan fuzz driver for the lCRF parser (host-only, no HIP):
//   g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-sanitize-recover=all -Igecco_amd/csrc -include cstring \
//       gecco_amd/csrc/crf_model.cpp tools/fuzz_model.cpp -o /tmp/fuzz && /tmp/fuzz <blob.bin>
// Round 1: 3000 corruptions of the embedded model, 2919 rejected, 81 parsed, no sanitizer report.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "crf_model.hpp"
namespace gecco { Model::~Model() {} }
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); std::vector<uint8_t> blob; uint8_t buf[65536]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) blob.insert(blob.end(), buf, buf + n);
    fclose(f);
    std::mt19937 rng(42); int ok = 0, bad = 0;
    for (int t = 0; t < 3000; ++t) {
        std::vector<uint8_t> b = blob;
        int kind = t % 4;
        if (kind == 0) for (int i = 0; i < 6; ++i) b[rng() % b.size()] ^= 1 + rng() % 255;
        else if (kind == 1) b[rng() % 48] = rng() % 256;
        else if (kind == 2) { size_t cut = 48 + rng() % (b.size() - 48); b.resize(cut); uint32_t s = cut; memcpy(&b[4], &s, 4); }
        else for (int i = 0; i < 16; ++i) b[184288 + rng() % (b.size() - 184288)] = rng() % 256;
        gecco::Model m;
        if (gecco::parse_lcrf(b.data(), b.size(), m) == 0) ++ok; else ++bad;
    }
    printf("ok=%d rejected=%d\n", ok, bad);
    return 0;
}
