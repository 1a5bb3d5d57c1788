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

// A Horner step p * x + C with a literal C.  On the device the constant is a SCALAR operand of one v_fma_f64: left to the compiler
// the step becomes `v_mov_b64 tmp, C; v_fmac_f64 tmp, p, x` -- the two-address form wants the addend in the destination, and C,
// kept in a VGPR pair for the whole kernel, must survive: two issue slots per step and two dozen constants' worth of VGPRs in
// every kernel that evaluates coefficients (klatt_direct.h has the count: 130 of ~230 instructions per sample and stage).  Same
// operation, same operands: the same bits.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(KLATT_NO_SCALAR_CONSTANTS)
__device__ __forceinline__ double horner(double p, double x, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(p), "v"(x), "s"(c));
    return d;
}
// c * x + a with the literal as the multiplicand (the first step of a polynomial, the reductions' k * ln2 terms)
__device__ __forceinline__ double horner0(double c, double x, double a)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "s"(c), "v"(x), "v"(a));
    return d;
}
// x + c and x * c with a literal c
__device__ __forceinline__ double add_const(double x, double c)
{
    double d;
    asm("v_add_f64 %0, %1, %2" : "=v"(d) : "v"(x), "s"(c));
    return d;
}
__device__ __forceinline__ double mul_const(double x, double c)
{
    double d;
    asm("v_mul_f64 %0, %1, %2" : "=v"(d) : "v"(x), "s"(c));
    return d;
}
#else
KLATT_HD double horner(double p, double x, double c) { return __builtin_fma(p, x, c); }
KLATT_HD double horner0(double c, double x, double a) { return __builtin_fma(c, x, a); }
KLATT_HD double add_const(double x, double c) { return x + c; }
KLATT_HD double mul_const(double x, double c) { return x * c; }
#endif

// e^r for |r| <= ln2/2: 1 + r + r^2 P(r), Taylor through r^13 (truncation < 4e-18 relative)
KLATT_HD double exp_kernel(double r)
{
    double p = horner0(1.6059043836821613e-10, r, 2.0876756987868100e-09);   // (1/13!) r + 1/12!
    p = horner(p, r, 2.5052108385441720e-08);   // 1/11!
    p = horner(p, r, 2.7557319223985890e-07);   // 1/10!
    p = horner(p, r, 2.7557319223985893e-06);   // 1/9!
    p = horner(p, r, 2.4801587301587302e-05);   // 1/8!
    p = horner(p, r, 1.9841269841269841e-04);   // 1/7!
    p = horner(p, r, 1.3888888888888889e-03);   // 1/6!
    p = horner(p, r, 8.3333333333333332e-03);   // 1/5!
    p = horner(p, r, 4.1666666666666664e-02);   // 1/4!
    p = horner(p, r, 1.6666666666666666e-01);   // 1/3!
    p = __builtin_fma(p, r, 0.5);
    return __builtin_fma(r * r, p, r) + 1.0;
}

constexpr double kLog2e = 1.4426950408889634074;
constexpr double kTwoOverPi = 0.63661977236758134308;

// e^x, |x| <= 700.  x = k ln2 + r, |r| <= ln2/2 (fdlibm's two-part ln2), scaled by 2^k.
KLATT_HD double fast_exp(double x)
{
    const double k = __builtin_rint(mul_const(x, kLog2e));
    double r = horner0(-6.93147180369123816490e-01, k, x);   // -k ln2_hi + x (the sign moved to the constant: the same product)
    r = horner0(-1.90821492927058770002e-10, k, r);           // ln2 low part
    return __builtin_ldexp(exp_kernel(r), (int)k);
}

// sin r = r + r z S(z), z = r^2, |r| <= pi/4, Taylor through r^17
KLATT_HD double sin_kernel(double r, double z)
{
    double s = horner0(2.8114572543455206e-15, z, -7.6471637318198164e-13);  // (1/17!) z - 1/15!
    s = horner(s, z, 1.6059043836821613e-10);   // 1/13!
    s = horner(s, z, -2.5052108385441720e-08);  // -1/11!
    s = horner(s, z, 2.7557319223985893e-06);   // 1/9!
    s = horner(s, z, -1.9841269841269841e-04);  // -1/7!
    s = horner(s, z, 8.3333333333333332e-03);   // 1/5!
    s = horner(s, z, -1.6666666666666666e-01);  // -1/3!
    return __builtin_fma(r * z, s, r);
}

// cos r = 1 + z (-1/2 + z C(z)), z = r^2, |r| <= pi/4, Taylor through r^16
KLATT_HD double cos_kernel(double z)
{
    double c = horner0(4.7794773323873853e-14, z, -1.1470745597729725e-11);  // (1/16!) z - 1/14!
    c = horner(c, z, 2.0876756987868100e-09);   // 1/12!
    c = horner(c, z, -2.7557319223985890e-07);  // -1/10!
    c = horner(c, z, 2.4801587301587302e-05);   // 1/8!
    c = horner(c, z, -1.3888888888888889e-03);  // -1/6!
    c = horner(c, z, 4.1666666666666664e-02);   // 1/4!
    return __builtin_fma(z, __builtin_fma(z, c, -0.5), 1.0);
}

// cos(t), |t| <= 1e4.  t = n pi/2 + r, |r| <= pi/4 (two-part pi/2 with FMA), then the sine or
// cosine kernel in r by quadrant (truncation < 3e-18).
KLATT_HD double fast_cos(double t)
{
    const double n = __builtin_rint(mul_const(t, kTwoOverPi));
    double r = horner0(-1.5707963267948965580e+00, n, t);
    r = horner0(-6.1232339957367660359e-17, n, r);
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
    const double n = __builtin_rint(mul_const(t, kTwoOverPi));
    double r = horner0(-1.5707963267948965580e+00, n, t);
    r = horner0(-6.1232339957367660359e-17, n, r);
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
    double r = add_const(t, 1.5707963267948965580e+00);      // fma(1.0, pi/2_hi, t): one rounded addition
    r = add_const(r, 6.1232339957367660359e-17);
    return sin_kernel(r, r * r);
}
KLATT_HD bool exp_is_unreduced(double x) { return __builtin_rint(x * kLog2e) == 0.0; }
KLATT_HD bool cos_is_unreduced(double t) { return __builtin_rint(t * kTwoOverPi) == 0.0; }
KLATT_HD double exp_unreduced(double x) { return exp_kernel(x); }
KLATT_HD double cos_unreduced(double t) { return cos_kernel(t * t); }
}  // namespace klatt
