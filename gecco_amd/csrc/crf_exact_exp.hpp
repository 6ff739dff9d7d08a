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

GECCO_HD inline double exp_correctly_rounded_slow(double x) {
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

// The same function, fast path first (Ziv's strategy): exp(x) = 2^m T[j] exp(r), x = (64 m + j) ln2/64 + r, |r| <= ln2/128, with
//   * r as a double-double (ln2/64 in pieces of 32 + 32 + 53 bits: the first two products with k < 2^17 are exact),
//   * T[j] = 2^(j/64) as double-doubles (generated with 60-digit decimal arithmetic: error < 2^-105),
//   * exp(r) - 1 = r + r^2/2 (both as double-doubles, the cross term r_hi r_lo included) + r^3/6 + ... + r^8/8! in plain double
//     (|r^3/6| < 2^-25: its rounding is below 2^-78; truncation r^9/9! < 2^-86),
// so the double-double result carries a relative error below 2^-76.  It is accepted -- its high word IS the correctly rounded
// exponential -- when it is more than 2^-70 (relative) away from the midpoint between two doubles, which is the case for all
// but ~1 argument in 2^16; the others, and results outside the normal range, take the routine above.  Every accepted result
// therefore equals the routine above bit for bit (tests/test_native_cpu.py: both against 60-digit decimal arithmetic, and
// against each other on millions of arguments).
GECCO_HD inline double exp_correctly_rounded(double x) {
    if (!(x == x)) return x;
    if (x == 0.0) return 1.0;  // (a gene without domains: the common case)
    if (!(x > -708.0 && x < 709.0)) return exp_correctly_rounded_slow(x);  // (overflow, subnormal results: a second rounding)
    constexpr double T[64][2] = {
        {0x1.0000000000000p+0, 0.0}, {0x1.02c9a3e778061p+0, -0x1.19083535b085dp-56},
        {0x1.059b0d3158574p+0, 0x1.d73e2a475b465p-55}, {0x1.0874518759bc8p+0, 0x1.186be4bb284ffp-57},
        {0x1.0b5586cf9890fp+0, 0x1.8a62e4adc610bp-54}, {0x1.0e3ec32d3d1a2p+0, 0x1.03a1727c57b53p-59},
        {0x1.11301d0125b51p+0, -0x1.6c51039449b3ap-54}, {0x1.1429aaea92de0p+0, -0x1.32fbf9af1369ep-54},
        {0x1.172b83c7d517bp+0, -0x1.19041b9d78a76p-55}, {0x1.1a35beb6fcb75p+0, 0x1.e5b4c7b4968e4p-55},
        {0x1.1d4873168b9aap+0, 0x1.e016e00a2643cp-54}, {0x1.2063b88628cd6p+0, 0x1.dc775814a8495p-55},
        {0x1.2387a6e756238p+0, 0x1.9b07eb6c70573p-54}, {0x1.26b4565e27cddp+0, 0x1.2bd339940e9d9p-55},
        {0x1.29e9df51fdee1p+0, 0x1.612e8afad1255p-55}, {0x1.2d285a6e4030bp+0, 0x1.0024754db41d5p-54},
        {0x1.306fe0a31b715p+0, 0x1.6f46ad23182e4p-55}, {0x1.33c08b26416ffp+0, 0x1.32721843659a6p-54},
        {0x1.371a7373aa9cbp+0, -0x1.63aeabf42eae2p-54}, {0x1.3a7db34e59ff7p+0, -0x1.5e436d661f5e3p-56},
        {0x1.3dea64c123422p+0, 0x1.ada0911f09ebcp-55}, {0x1.4160a21f72e2ap+0, -0x1.ef3691c309278p-58},
        {0x1.44e086061892dp+0, 0x1.89b7a04ef80d0p-59}, {0x1.486a2b5c13cd0p+0, 0x1.3c1a3b69062f0p-56},
        {0x1.4bfdad5362a27p+0, 0x1.d4397afec42e2p-56}, {0x1.4f9b2769d2ca7p+0, -0x1.4b309d25957e3p-54},
        {0x1.5342b569d4f82p+0, -0x1.07abe1db13cadp-55}, {0x1.56f4736b527dap+0, 0x1.9bb2c011d93adp-54},
        {0x1.5ab07dd485429p+0, 0x1.6324c054647adp-54}, {0x1.5e76f15ad2148p+0, 0x1.ba6f93080e65ep-54},
        {0x1.6247eb03a5585p+0, -0x1.383c17e40b497p-54}, {0x1.6623882552225p+0, -0x1.bb60987591c34p-54},
        {0x1.6a09e667f3bcdp+0, -0x1.bdd3413b26456p-54}, {0x1.6dfb23c651a2fp+0, -0x1.bbe3a683c88abp-57},
        {0x1.71f75e8ec5f74p+0, -0x1.16e4786887a99p-55}, {0x1.75feb564267c9p+0, -0x1.0245957316dd3p-54},
        {0x1.7a11473eb0187p+0, -0x1.41577ee04992fp-55}, {0x1.7e2f336cf4e62p+0, 0x1.05d02ba15797ep-56},
        {0x1.82589994cce13p+0, -0x1.d4c1dd41532d8p-54}, {0x1.868d99b4492edp+0, -0x1.fc6f89bd4f6bap-54},
        {0x1.8ace5422aa0dbp+0, 0x1.6e9f156864b27p-54}, {0x1.8f1ae99157736p+0, 0x1.5cc13a2e3976cp-55},
        {0x1.93737b0cdc5e5p+0, -0x1.75fc781b57ebcp-57}, {0x1.97d829fde4e50p+0, -0x1.d185b7c1b85d1p-54},
        {0x1.9c49182a3f090p+0, 0x1.c7c46b071f2bep-56}, {0x1.a0c667b5de565p+0, -0x1.359495d1cd533p-54},
        {0x1.a5503b23e255dp+0, -0x1.d2f6edb8d41e1p-54}, {0x1.a9e6b5579fdbfp+0, 0x1.0fac90ef7fd31p-54},
        {0x1.ae89f995ad3adp+0, 0x1.7a1cd345dcc81p-54}, {0x1.b33a2b84f15fbp+0, -0x1.2805e3084d708p-57},
        {0x1.b7f76f2fb5e47p+0, -0x1.5584f7e54ac3bp-56}, {0x1.bcc1e904bc1d2p+0, 0x1.23dd07a2d9e84p-55},
        {0x1.c199bdd85529cp+0, 0x1.11065895048ddp-55}, {0x1.c67f12e57d14bp+0, 0x1.2884dff483cadp-54},
        {0x1.cb720dcef9069p+0, 0x1.503cbd1e949dbp-56}, {0x1.d072d4a07897cp+0, -0x1.cbc3743797a9cp-54},
        {0x1.d5818dcfba487p+0, 0x1.2ed02d75b3707p-55}, {0x1.da9e603db3285p+0, 0x1.c2300696db532p-54},
        {0x1.dfc97337b9b5fp+0, -0x1.1a5cd4f184b5cp-54}, {0x1.e502ee78b3ff6p+0, 0x1.39e8980a9cc8fp-55},
        {0x1.ea4afa2a490dap+0, -0x1.e9c23179c2893p-54}, {0x1.efa1bee615a27p+0, 0x1.dc7f486a4b6b0p-54},
        {0x1.f50765b6e4540p+0, 0x1.9d3e12dd8a18bp-54}, {0x1.fa7c1819e90d8p+0, 0x1.74853f3a5931ep-55},
    };
    const double kd = ::rint(x * 0x1.71547652b82fep+6);  // 64 / ln2
    constexpr double C1 = 0x1.62e42ff000000p-7, C2 = -0x1.718432a200000p-41, C3 = 0x1.3c7673007e5edp-75;  // ln2 / 64
    dd r = two_sum(x, -kd * C1);
    r = add(r, dd{-kd * C2, 0.0});
    r = add(r, two_prod(-kd, C3));
    const int k = int(kd), j = k & 63, m = (k - j) / 64;
    const double rh = r.hi, rl = r.lo;
    // exp(r) - 1 = q
    double p = 0x1.a01a01a01a01ap-16;                        // 1 / 8!
    p = p * rh + 0x1.a01a01a01a01ap-13;                      // 1 / 7!
    p = p * rh + 0x1.6c16c16c16c17p-10;                      // 1 / 6!
    p = p * rh + 0x1.1111111111111p-7;                       // 1 / 5!
    p = p * rh + 0x1.5555555555555p-5;                       // 1 / 4!
    p = p * rh + 0x1.5555555555555p-3;                       // 1 / 3!
    p = p * (rh * rh * rh);
    const dd h2 = two_prod(rh, rh);                          // r^2 (the halving below is exact)
    const dd s = two_sum(rh, 0.5 * h2.hi);
    const double tail = s.lo + (0.5 * h2.lo + (rl + rh * rl) + p);
    const dd q = quick_two_sum(s.hi, tail);
    // T (1 + q)
    const dd t{T[j][0], T[j][1]};
    const dd res = add(t, mul(t, q));
    // acceptance: |res.lo| stays clear of half an ulp of res.hi (res.hi lies in [0.99, 2.02); at a power of two the doubles
    // below are twice as dense: left to the routine above)
    if (res.hi == 1.0 || res.hi == 2.0) return exp_correctly_rounded_slow(x);
    const double half_ulp = res.hi >= 2.0 ? 0x1p-52 : res.hi >= 1.0 ? 0x1p-53 : 0x1p-54;
    if (half_ulp - ::fabs(res.lo) > 0x1p-69) return ::ldexp(res.hi, m);
    return exp_correctly_rounded_slow(x);
}

}  // namespace ddx
}  // namespace gecco
