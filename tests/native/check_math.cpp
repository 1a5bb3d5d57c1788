// Host check of the kernel's arithmetic helpers (compiled and run by tests/test_host_logic.py):
//  * div_by(): correctly rounded division from a rounded reciprocal, against the '/' operator;
//  * noise_uniform()'s two-operation division by 2^31-1, against '/' for every 31-bit integer;
//  * fast_exp()/fast_cos(): ulp error against libm over the ranges a frame can produce.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

#include "../../nvspeechplayer_amd/csrc/klatt_math.h"

static double div_by(double x, double b, double y)
{
    double q = x * y;
    double r = std::fma(-b, q, x);
    return std::fma(r, y, q);
}

static double ulp_diff(double a, double b)
{
    if (a == b) return 0;
    int64_t ia, ib;
    memcpy(&ia, &a, 8); memcpy(&ib, &b, 8);
    if ((ia < 0) != (ib < 0)) return 1e9;
    return (double)(ia > ib ? ia - ib : ib - ia);
}

int main()
{
    std::mt19937_64 rng(12345);
    long long divBad = 0, divN = 0;
    const double bs[] = {22050.0, 16000.0, 44100.0, 48000.0, 8000.0, 11025.0, 2147483647.0, 3.0, 7.0, 441.0, 1102.0};
    for (double b : bs) {
        const double y = 1.0 / b;
        std::uniform_real_distribution<double> mant(1.0, 2.0);
        std::uniform_int_distribution<int> ex(-40, 40);
        for (int i = 0; i < 2000000; ++i) {
            double x = std::ldexp(mant(rng), ex(rng));
            if (i & 1) x = -x;
            divN++;
            if (div_by(x, b, y) != x / b) divBad++;
        }
        for (uint32_t r = 0; r < 2000000; ++r) {   // integers, as in rand()/RAND_MAX and counter/fadeSamples
            double x = (double)(r * 1073u + 17u);
            divN++;
            if (div_by(x, b, y) != x / b) divBad++;
        }
    }
    // counter / fadeSamples exhaustively for small fades, sampled for large ones
    for (uint32_t F = 1; F <= 3000; ++F) {
        const double y = 1.0 / (double)F;
        for (uint32_t c = 0; c <= F; ++c) {
            divN++;
            if (div_by((double)c, (double)F, y) != (double)c / (double)F) divBad++;
        }
    }
    for (int i = 0; i < 3000000; ++i) {
        uint32_t F = (uint32_t)(rng() % 4000000000ull) + 1u;
        uint32_t c = (uint32_t)(rng() % F);
        divN++;
        if (div_by((double)c, (double)F, 1.0 / (double)F) != (double)c / (double)F) divBad++;
    }

    // noise_uniform(): r / (2^31 - 1) as fma(r, yh, r * yl), exhaustively over every value noise31() can return
    long long noiseBad = 0;
    for (uint64_t r = 0; r < 2147483648ull; ++r) {
        const double x = (double)r;
        if (std::fma(x, 0x1.00000002p-31, x * 0x1p-93) != x / 2147483647.0) noiseBad++;
    }
    divN += 2147483648ll;
    divBad += noiseBad;

    double expMax = 0, cosMax = 0, expSum = 0, cosSum = 0;
    long long n = 0;
    std::uniform_real_distribution<double> ux(-3.0, 0.5), ut(-7.0, 7.0);
    for (int i = 0; i < 4000000; ++i) {
        double x = ux(rng), t = ut(rng);
        double e = ulp_diff(klatt::fast_exp(x), std::exp(x));
        double c = ulp_diff(klatt::fast_cos(t), std::cos(t));
        if (e > expMax) expMax = e;
        if (c > cosMax) cosMax = c;
        expSum += e; cosSum += c; n++;
    }
    // wide ranges
    double expWide = 0, cosWide = 0;
    std::uniform_real_distribution<double> wx(-600.0, 600.0), wt(-9000.0, 9000.0);
    for (int i = 0; i < 1000000; ++i) {
        double x = wx(rng), t = wt(rng);
        double e = ulp_diff(klatt::fast_exp(x), std::exp(x));
        double c = ulp_diff(klatt::fast_cos(t), std::cos(t));
        // near a zero of cos the ulp measure blows up for any finite-precision reduction; use absolute error there
        if (std::fabs(std::cos(t)) < 1e-3) c = std::fabs(klatt::fast_cos(t) - std::cos(t)) / 1.1e-19 > 4 ? c : 0;
        if (e > expWide) expWide = e;
        if (c > cosWide) cosWide = c;
    }
    // the unreduced short path must return the same bits as the full functions wherever it is allowed
    long long unredN = 0, unredBad = 0;
    std::uniform_real_distribution<double> sx(-0.36, 0.36), st(-0.80, 0.80);
    for (int i = 0; i < 4000000; ++i) {
        double x = sx(rng), t = st(rng);
        if (i < 64) { x = (i & 1 ? -1 : 1) * 0.34657359027997264 * (1.0 - (i >> 1) * 1e-16); t = (i & 1 ? -1 : 1) * 0.78539816339744828 * (1.0 - (i >> 1) * 1e-16); }
        if (klatt::exp_is_unreduced(x)) { unredN++; double a = klatt::exp_unreduced(x), b = klatt::fast_exp(x); if (memcmp(&a, &b, 8)) unredBad++; }
        if (klatt::cos_is_unreduced(t)) { unredN++; double a = klatt::cos_unreduced(t), b = klatt::fast_cos(t); if (memcmp(&a, &b, 8)) unredBad++; }
    }
    // ... and so must the n = -1 quadrant path of cos (F3 and up)
    std::uniform_real_distribution<double> qt(-2.40, -0.75);
    for (int i = 0; i < 4000000; ++i) {
        double t = qt(rng);
        if (i < 64) t = -((i & 1) ? 0.78539816339744839 : 2.3561944901923448) * (1.0 + ((i & 2) ? 1 : -1) * (i >> 2) * 1e-16);
        if (klatt::cos_is_quadrant_m1(t)) { unredN++; double a = klatt::cos_quadrant_m1(t), b = klatt::fast_cos(t); if (memcmp(&a, &b, 8)) unredBad++; }
    }
    printf("{\"unreduced_checked\": %lld, \"unreduced_bad\": %lld, ", unredN, unredBad);
    printf("\"div_checked\": %lld, \"div_bad\": %lld, \"exp_max_ulp\": %.0f, \"cos_max_ulp\": %.0f, "
           "\"exp_mean_ulp\": %.4f, \"cos_mean_ulp\": %.4f, \"exp_wide_max_ulp\": %.0f, \"cos_wide_max_ulp\": %.0f, "
           "\"exp0\": %.17g, \"cos0\": %.17g}\n",
           divN, divBad, expMax, cosMax, expSum / n, cosSum / n, expWide, cosWide, klatt::fast_exp(0.0), klatt::fast_cos(0.0));
    return 0;
}
