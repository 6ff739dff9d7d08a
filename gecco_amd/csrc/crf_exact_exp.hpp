// exp(x) rounded correctly, for host and device: what the reference-bits mode (crf_exact.hip) puts where CRFsuite calls libm's
// exp ([EXT] crf1dc_exp_state).  glibc's exp is within 0.51 ulp of the true value, i.e. it returns the correctly rounded
// double except when the true value lies within a hundredth of an ulp of a rounding boundary; a correctly rounded exp is
// therefore the libm-independent way to land on the reference's bits.
//
// Double-double arithmetic (error-free transformations; no fast-math, no contraction of the a * b + c the transformations rely on):
//   x = k ln2 + r          ln2 in three pieces of 32 + 32 + 53 bits: the first two products are exact, r is a double-double
//   exp(r) = (exp(r / 8))^8,  exp(r / 8) = sum_{i <= 15} (r / 8)^i / i!   (|r / 8| <= 0.0434: truncation < 1e-34), coefficients
//   as double-doubles; three squarings; the high word of the normalised result is the double nearest to it.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GECCO_HD __host__ __device__
#else
#define GECCO_HD
#endif

namespace gecco {
namespace ddx {

struct dd {
    double hi, lo;
};
#pragma clang fp contract(off)
GECCO_HD inline dd two_sum(double a, double b) {
    const double s = a + b, bb = s - a;
    return dd{s, (a - (s - bb)) + (b - bb)};
}
GECCO_HD inline dd quick_two_sum(double a, double b) {  // |a| >= |b|
    const double s = a + b;
    return dd{s, b - (s - a)};
}
GECCO_HD inline dd two_prod(double a, double b) {
    const double p = a * b;
    return dd{p, ::fma(a, b, -p)};
}
GECCO_HD inline dd add(dd a, dd b) {
    dd s = two_sum(a.hi, b.hi);
    const dd t = two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = quick_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return quick_two_sum(s.hi, s.lo);
}
GECCO_HD inline dd mul(dd a, dd b) {
    dd p = two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return quick_two_sum(p.hi, p.lo);
}

GECCO_HD inline double exp_correctly_rounded(double x) {
    if (!(x == x)) return x;
    if (x > 709.782712893384) return HUGE_VAL;
    if (x < -745.2) return 0.0;
    const double k = ::rint(x * 1.4426950408889634);
    constexpr double L1 = 0x1.62e42ff000000p-1, L2 = -0x1.718432a200000p-35, L3 = 0x1.3c7673007e5edp-69;  // ln2 = L1 + L2 + L3 (+ 1.4e-37)
    dd r = two_sum(x, -k * L1);                   // (k L1, k L2: exact products -- 11 x 32 bits)
    r = add(r, dd{-k * L2, 0.0});
    r = add(r, two_prod(-k, L3));
    r.hi *= 0.125;
    r.lo *= 0.125;
    // 1 / i! as double-doubles, i = 15 .. 0
    constexpr double C[16][2] = {
        {0x1.ae7f3e733b81fp-41, 0x1.1d8656b0ee8cbp-97},  {0x1.93974a8c07c9dp-37, 0x1.05d6f8a2efd1fp-92},
        {0x1.6124613a86d09p-33, 0x1.f28e0cc748ebep-87},  {0x1.1eed8eff8d898p-29, -0x1.2aec959e14c06p-83},
        {0x1.ae64567f544e4p-26, -0x1.c062e06d1f209p-80}, {0x1.27e4fb7789f5cp-22, 0x1.cbbc05b4fa99ap-76},
        {0x1.71de3a556c734p-19, -0x1.c154f8ddc6c00p-73}, {0x1.a01a01a01a01ap-16, 0x1.a01a01a01a01ap-76},
        {0x1.a01a01a01a01ap-13, 0x1.a01a01a01a01ap-73},  {0x1.6c16c16c16c17p-10, -0x1.f49f49f49f49fp-65},
        {0x1.1111111111111p-7, 0x1.1111111111111p-63},   {0x1.5555555555555p-5, 0x1.5555555555555p-59},
        {0x1.5555555555555p-3, 0x1.5555555555555p-57},   {0x1.0000000000000p-1, 0.0},
        {1.0, 0.0},                                      {1.0, 0.0}};
    dd p{C[0][0], C[0][1]};
    for (int i = 1; i < 16; ++i) p = add(mul(p, r), dd{C[i][0], C[i][1]});
    p = mul(p, p);
    p = mul(p, p);
    p = mul(p, p);
    return ::ldexp(p.hi, int(k));
}

}  // namespace ddx
}  // namespace gecco
