// exp_correctly_rounded (Ziv fast path, gecco_amd/csrc/crf_exact_exp.hpp) against exp_correctly_rounded_slow on random and special
// arguments: must agree bit for bit.  Built and run by tests/test_native_cpu.py (g++ -O2 -ffp-contract=off -I gecco_amd/csrc).
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include "crf_exact_exp.hpp"
using namespace gecco::ddx;
static long fallbacks = 0;
int main(int argc, char **argv) {
    long n = argc > 1 ? atol(argv[1]) : 10000000;
    std::mt19937_64 g(12345);
    long bad = 0, tested = 0;
    auto check = [&](double x) {
        const double a = exp_correctly_rounded(x), b = exp_correctly_rounded_slow(x);
        ++tested;
        if (std::memcmp(&a, &b, 8) != 0 && !(a != a && b != b)) {
            if (bad < 10) std::printf("MISMATCH x=%a fast=%a slow=%a\n", x, a, b);
            ++bad;
        }
    };
    std::uniform_real_distribution<double> u1(-40.0, 40.0), u2(-708.0, 709.0), u3(-1.0, 1.0);
    for (long i = 0; i < n; ++i) check(u1(g));
    for (long i = 0; i < n / 4; ++i) check(u2(g));
    for (long i = 0; i < n / 4; ++i) check(u3(g));
    for (long i = 0; i < n / 8; ++i) check(std::ldexp(u3(g), -int(g() % 80)));      // tiny arguments
    for (long i = 0; i < n / 8; ++i) {                                                // near the reduction boundaries
        const long k = long(g() % 130000) - 65000;
        check(double(k) * 0.010830424696249145 * (1.0 + u3(g) * 1e-12) + 0.005415212348124572 * ((i & 1) ? 1 : 0));
    }
    const double sp[] = {0.0, -0.0, 1.0, -1.0, 0x1p-54, -0x1p-54, 0x1p-53, -0x1p-53, 0.6931471805599453, 0.6931471805599452, 709.0, -708.0, 709.78, -745.0, -708.5, 1e-300, -1e-300};
    for (double x : sp) check(x);
    std::printf("tested %ld mismatches %ld\n", tested, bad);
    return bad != 0;
}
