// klatt_math.h -- exp() and cos() for the resonator coefficients of MODE_FAST.
//
// Reference src/speechWaveGenerator.cpp:116,118 call libm exp and cos once per resonator on
// every sample of a fade.  These are straight-line fused-multiply-add versions (about 20 and
// 30 instructions instead of the library's ~70 each), accurate to better than 1 ulp over the
// argument ranges a frame can produce; tests/test_host_logic.py measures them against libm
// on the host (this header compiles for both).
#pragma once

#if defined(__HIPCC__)
#define KLATT_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define KLATT_HD inline
#endif

namespace klatt {

// e^r for |r| <= ln2/2: 1 + r + r^2 P(r), Taylor through r^13 (truncation < 4e-18 relative)
KLATT_HD double exp_kernel(double r)
{
    double p = 1.6059043836821613e-10;            // 1/13!
    p = __builtin_fma(p, r, 2.0876756987868100e-09);   // 1/12!
    p = __builtin_fma(p, r, 2.5052108385441720e-08);   // 1/11!
    p = __builtin_fma(p, r, 2.7557319223985890e-07);   // 1/10!
    p = __builtin_fma(p, r, 2.7557319223985893e-06);   // 1/9!
    p = __builtin_fma(p, r, 2.4801587301587302e-05);   // 1/8!
    p = __builtin_fma(p, r, 1.9841269841269841e-04);   // 1/7!
    p = __builtin_fma(p, r, 1.3888888888888889e-03);   // 1/6!
    p = __builtin_fma(p, r, 8.3333333333333332e-03);   // 1/5!
    p = __builtin_fma(p, r, 4.1666666666666664e-02);   // 1/4!
    p = __builtin_fma(p, r, 1.6666666666666666e-01);   // 1/3!
    p = __builtin_fma(p, r, 0.5);
    return __builtin_fma(r * r, p, r) + 1.0;
}

constexpr double kLog2e = 1.4426950408889634074;
constexpr double kTwoOverPi = 0.63661977236758134308;

// e^x, |x| <= 700.  x = k ln2 + r, |r| <= ln2/2 (fdlibm's two-part ln2), scaled by 2^k.
KLATT_HD double fast_exp(double x)
{
    const double k = __builtin_rint(x * kLog2e);
    double r = __builtin_fma(-k, 6.93147180369123816490e-01, x);   // ln2 high part
    r = __builtin_fma(-k, 1.90821492927058770002e-10, r);           // ln2 low part
    return __builtin_ldexp(exp_kernel(r), (int)k);
}

// sin r = r + r z S(z), z = r^2, |r| <= pi/4, Taylor through r^17
KLATT_HD double sin_kernel(double r, double z)
{
    double s = 2.8114572543455206e-15;                 // 1/17!
    s = __builtin_fma(s, z, -7.6471637318198164e-13);  // -1/15!
    s = __builtin_fma(s, z, 1.6059043836821613e-10);   // 1/13!
    s = __builtin_fma(s, z, -2.5052108385441720e-08);  // -1/11!
    s = __builtin_fma(s, z, 2.7557319223985893e-06);   // 1/9!
    s = __builtin_fma(s, z, -1.9841269841269841e-04);  // -1/7!
    s = __builtin_fma(s, z, 8.3333333333333332e-03);   // 1/5!
    s = __builtin_fma(s, z, -1.6666666666666666e-01);  // -1/3!
    return __builtin_fma(r * z, s, r);
}

// cos r = 1 + z (-1/2 + z C(z)), z = r^2, |r| <= pi/4, Taylor through r^16
KLATT_HD double cos_kernel(double z)
{
    double c = 4.7794773323873853e-14;                 // 1/16!
    c = __builtin_fma(c, z, -1.1470745597729725e-11);  // -1/14!
    c = __builtin_fma(c, z, 2.0876756987868100e-09);   // 1/12!
    c = __builtin_fma(c, z, -2.7557319223985890e-07);  // -1/10!
    c = __builtin_fma(c, z, 2.4801587301587302e-05);   // 1/8!
    c = __builtin_fma(c, z, -1.3888888888888889e-03);  // -1/6!
    c = __builtin_fma(c, z, 4.1666666666666664e-02);   // 1/4!
    return __builtin_fma(z, __builtin_fma(z, c, -0.5), 1.0);
}

// cos(t), |t| <= 1e4.  t = n pi/2 + r, |r| <= pi/4 (two-part pi/2 with FMA), then the sine or
// cosine kernel in r by quadrant (truncation < 3e-18).
KLATT_HD double fast_cos(double t)
{
    const double n = __builtin_rint(t * kTwoOverPi);
    double r = __builtin_fma(-n, 1.5707963267948965580e+00, t);
    r = __builtin_fma(-n, 6.1232339957367660359e-17, r);
    const double z = r * r;
    const double sinr = sin_kernel(r, z);
    const double cosr = cos_kernel(z);
    const int q = (int)n & 3;
    const double v = (q & 1) ? sinr : cosr;
    return (q == 1 || q == 2) ? -v : v;
}

// sin(t), |t| <= 1e4: the same reduction as fast_cos.  (Used by klatt_seeds.h, once per fade: the start of MODE_FAST's coefficient
// recurrences; nothing on the MODE_EXACT path calls it.)
KLATT_HD double fast_sin(double t)
{
    const double n = __builtin_rint(t * kTwoOverPi);
    double r = __builtin_fma(-n, 1.5707963267948965580e+00, t);
    r = __builtin_fma(-n, 6.1232339957367660359e-17, r);
    const double z = r * r;
    const double sinr = sin_kernel(r, z);
    const double cosr = cos_kernel(z);
    const int q = (int)n & 3;
    const double v = (q & 1) ? cosr : sinr;
    return (q >= 2) ? -v : v;
}

// The arguments of a formant are small: exp(-pi bw / sr) has k = 0 up to bw = 2433 Hz at 22.05 kHz and
// cos(2 pi f / sr) has n = 0 up to f = 2756 Hz.  With k = 0 and n = 0 the reductions above are the identity
// (r = x - 0, ldexp(e, 0) = e, quadrant 0 = the cosine kernel), so these two return bit for bit what fast_exp
// and fast_cos return; a caller that knows k = n = 0 for all its lanes skips the reductions, the sine kernel
// and the quadrant selects (about 35 of 60 instructions).
// Third formant and up: cos(2 pi (-f) / sr) has n = -1 for f between 2756 and 8268 Hz at 22.05 kHz.  With n = -1 the
// reduction is r = (t + pi/2_hi) + pi/2_lo (fma(-n, c, t) with -n = 1 is one rounded addition), the quadrant is 3 and
// the result is +sin_kernel(r): cos_quadrant_m1 returns bit for bit what fast_cos returns there, without the cosine
// kernel and the selects.
KLATT_HD bool cos_is_quadrant_m1(double t) { return __builtin_rint(t * kTwoOverPi) == -1.0; }
KLATT_HD double cos_quadrant_m1(double t)
{
    double r = __builtin_fma(1.0, 1.5707963267948965580e+00, t);
    r = __builtin_fma(1.0, 6.1232339957367660359e-17, r);
    return sin_kernel(r, r * r);
}
KLATT_HD bool exp_is_unreduced(double x) { return __builtin_rint(x * kLog2e) == 0.0; }
KLATT_HD bool cos_is_unreduced(double t) { return __builtin_rint(t * kTwoOverPi) == 0.0; }
KLATT_HD double exp_unreduced(double x) { return exp_kernel(x); }
KLATT_HD double cos_unreduced(double t) { return cos_kernel(t * t); }
}  // namespace klatt
