/*
 * phip_fmath.h -- deterministic single-precision elementary functions.
 *
 * The reference calls libm (sincosf, expf, logf, acosf, atan2f, tanf, powf; see
 * include/mitsuba/core/math.h:185-237, src/bsdfs/microfacet.h:421-459,573-644,
 * src/libcore/warp.cpp:81-102).  glibc's and ROCm OCML's versions of these differ by a few
 * ulp, which is enough to flip `sample.x <= F` / Russian-roulette branches between a CPU and
 * a GPU run.  These restatements use only IEEE-754 basic operations (+ - * / sqrt, floor,
 * int<->float conversion, bit casts), evaluated in a fixed order, so that -- compiled with
 * -ffp-contract=off on both sides -- host and gfx950 device code return bit-identical
 * results.  Accuracy: <= ~2 ulp on the argument ranges the path uses (tests/test_fmath.py
 * checks them against libm).  Polynomial forms follow the classic Cephes single-precision
 * algorithms (Moshier), restated here.
 *
 * Usable from plain C++ (g++) and from HIP device code.
 */
#ifndef PHIP_FMATH_H
#define PHIP_FMATH_H

#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#define PM_FN __host__ __device__ static inline
#else
#define PM_FN static inline
#endif

PM_FN float pm_from_bits(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f; memcpy(&f, &u, 4); return f;
#endif
}
PM_FN uint32_t pm_to_bits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u; memcpy(&u, &f, 4); return u;
#endif
}

#define PM_PI        3.14159265358979323846f
#define PM_PI_2      1.57079632679489661923f
#define PM_PI_4      0.78539816339744830962f
#define PM_INV_PI    0.31830988618379067154f

/* x * 2^n for |n| small enough that one or two exact power-of-two multiplies suffice */
PM_FN float pm_ldexpf(float x, int n) {
    if (n > 127) {
        x *= pm_from_bits(0x7f000000u); /* 2^127 */
        n -= 127;
        if (n > 127) n = 127;
    } else if (n < -126) {
        x *= pm_from_bits(0x00800000u); /* 2^-126 */
        n += 126;
        if (n < -126) n = -126;
    }
    return x * pm_from_bits((uint32_t)(n + 127) << 23);
}

/* mantissa in [0.5,1) and exponent, for finite positive normal or subnormal x */
PM_FN float pm_frexpf(float x, int *e) {
    uint32_t u = pm_to_bits(x);
    int ex = (int)((u >> 23) & 0xff);
    if (ex == 0) { /* subnormal: renormalise */
        x *= 8388608.0f; /* 2^23 */
        u = pm_to_bits(x);
        ex = (int)((u >> 23) & 0xff) - 23;
    }
    *e = ex - 126;
    u = (u & 0x807fffffu) | 0x3f000000u;
    return pm_from_bits(u);
}

#if defined(PHIP_FMATH_NATIVE) && defined(__HIP_DEVICE_COMPILE__)
/* ======================================================================================
 *  MEASUREMENT ONLY (round 6, VERDICT r5 item 5: what does exactness cost?): the device's own transcendentals -- v_exp_f32 / v_log_f32 / v_sin_f32 / v_cos_f32 behind
 *  the __expf / __logf / __sinf / __cosf intrinsics (~1-2 ulp on a reduced range), ocml's acosf / atanf / atan2f / tanf -- instead of the correctly rounded double-
 *  precision evaluations below.  Built by tools/build_variant.sh with -DPHIP_FMATH_NATIVE; never in the product: results are no longer the oracle's bits.
 * ====================================================================================== */
PM_FN void pm_sincosf(float xx, float *s, float *c) { *s = __sinf(xx); *c = __cosf(xx); }
PM_FN float pm_expf(float x) { return __expf(x); }
PM_FN float pm_logf(float xx) { return __logf(xx); }
PM_FN float pm_powf(float x, float y) { return x == 0.0f ? (y > 0.0f ? 0.0f : 1.0f) : __expf(y * __logf(x)); }
PM_FN float pm_acosf(float x) { return acosf(x); }
PM_FN float pm_atanf(float xx) { return atanf(xx); }
PM_FN float pm_atan2f(float y, float x) { return atan2f(y, x); }
PM_FN float pm_tanf(float xx) { return tanf(xx); }
#elif !defined(PHIP_FMATH_CEPHES)
/* ======================================================================================
 *  Default implementation: every function is evaluated in DOUBLE precision from IEEE basic operations (argument
 *  reduction with split constants, Taylor / artanh series of 1/n! and 1/n coefficients taken far enough that the
 *  truncation error is < 1e-17) and rounded to float once.  The result is the correctly rounded float value except when
 *  the exact value lies within ~1e-9 ulp of a rounding boundary -- i.e. what `(float) ::exp((double) x)` (the reference's
 *  math::fastexp / fastlog on Linux x86-64, core/math.h:175-199) returns, and what glibc's sinf / cosf / expf / logf /
 *  acosf / atan2f return in all but ~0.1-1 % of the calls (their error bounds are 0.50-0.56 ulp).  Host and device evaluate the
 *  same IEEE double operations in the same order (-ffp-contract=off): bit-identical results, as before.
 * ====================================================================================== */
PM_FN double pmd_from_bits(uint64_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __longlong_as_double((long long) u);
#else
    double f; memcpy(&f, &u, 8); return f;
#endif
}
PM_FN uint64_t pmd_to_bits(double f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint64_t) __double_as_longlong(f);
#else
    uint64_t u; memcpy(&u, &f, 8); return u;
#endif
}

#define PMD_PI      3.14159265358979311600e+00
#define PMD_PI_2    1.57079632679489655800e+00
#define PMD_PI_4    7.85398163397448278999e-01

/* sin(x), cos(x) for a double argument of float magnitude (|x| < 2^31 * pi/2; beyond that the quadrant is still right
   modulo the precision of x itself) */
PM_FN void pmd_sincos(double x, double *s, double *c) {
    const double INV_PIO2 = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632673412561417e+00;      /* 33 significant bits: k * PIO2_HI is exact for |k| < 2^20 */
    const double PIO2_LO = 6.07710050650619224932e-11;
    const double PIO2_LO2 = 3.52155986518361559378e-27;
    double kd = floor(x * INV_PIO2 + 0.5);
    double r = fma(-kd, PIO2_LO2, fma(-kd, PIO2_LO, fma(-kd, PIO2_HI, x)));
    double q4 = kd - 4.0 * floor(kd * 0.25);                  /* kd mod 4, exact */
    int q = (int) q4;
    double z = r * r;
    double ps = fma(r * z, fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, 2.81145725434552059811e-15, -7.64716373181981640551e-13), 1.60590438368216133409e-10), -2.50521083854417202239e-08), 2.75573192239858925110e-06), -1.98412698412698412526e-04), 8.33333333333333321769e-03), -1.66666666666666657415e-01), r);
    double pc = fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, 4.77947733238738525345e-14, -1.14707455977297245073e-11), 2.08767569878681001866e-09), -2.75573192239858882758e-07), 2.48015873015873015658e-05), -1.38888888888888894189e-03), 4.16666666666666643537e-02), -5.00000000000000000000e-01), 1.0);
    if (q == 0) { *s = ps; *c = pc; }
    else if (q == 1) { *s = pc; *c = -ps; }
    else if (q == 2) { *s = -ps; *c = -pc; }
    else { *s = -pc; *c = ps; }
}

PM_FN double pmd_exp(double x) {                               /* |x| < 700 */
    const double INV_LN2 = 1.44269504088896338700e+00;
    const double LN2_HI = 6.93147180369123816490e-01;          /* 32 significant bits */
    const double LN2_LO = 1.90821492927058770002e-10;
    double kd = floor(x * INV_LN2 + 0.5);
    double r = fma(-kd, LN2_LO, fma(-kd, LN2_HI, x));
    double p = fma(r * r, fma(r, fma(r, fma(r, fma(r, fma(r, fma(r, fma(r, fma(r, fma(r, fma(r, fma(r, fma(r, 1.14707455977297245073e-11, 1.60590438368216133409e-10), 2.08767569878681001866e-09), 2.50521083854417202239e-08), 2.75573192239858882758e-07), 2.75573192239858925110e-06), 2.48015873015873015658e-05), 1.98412698412698412526e-04), 1.38888888888888894189e-03), 8.33333333333333321769e-03), 4.16666666666666643537e-02), 1.66666666666666657415e-01), 5.00000000000000000000e-01), 1.0 + r);
    int k = (int) kd;
    return p * pmd_from_bits((uint64_t) (k + 1023) << 52);     /* 2^k, k in [-1010, 1010] */
}

PM_FN double pmd_log(double x) {                               /* finite x > 0, normal double */
    const double LN2 = 6.93147180559945286227e-01;
    uint64_t u = pmd_to_bits(x);
    int e = (int) ((u >> 52) & 0x7ff) - 1023;
    double m = pmd_from_bits((u & 0x000fffffffffffffull) | 0x3ff0000000000000ull);     /* [1, 2) */
    if (m > 1.41421356237309514547e+00) { m *= 0.5; e += 1; }
    double t = (m - 1.0) / (m + 1.0), z = t * t;               /* log m = 2 artanh t, |t| <= 0.1716 */
    double lm = t * fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, 8.69565217391304323691e-02, 9.52380952380952328085e-02), 1.05263157894736836262e-01), 1.17647058823529410132e-01), 1.33333333333333331483e-01), 1.53846153846153854694e-01), 1.81818181818181823228e-01), 2.22222222222222209886e-01), 2.85714285714285698425e-01), 4.00000000000000022204e-01), 6.66666666666666629659e-01), 2.00000000000000000000e+00);
    return fma((double) e, LN2, lm);
}

PM_FN double pmd_atan(double xx) {
    double x = fabs(xx), y;
    if (x > 2.41421356237309492343e+00) { y = PMD_PI_2; x = -(1.0 / x); }             /* tan(3 pi / 8) */
    else if (x > 4.14213562373095034329e-01) { y = PMD_PI_4; x = (x - 1.0) / (x + 1.0); }   /* tan(pi / 8) */
    else y = 0.0;
    double z = x * x;                                         /* |x| <= 0.4143: 22 terms of the alternating series */
    double p = fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, -2.32558139534883717703e-02, 2.43902439024390252365e-02), -2.56410256410256401360e-02), 2.70270270270270285273e-02), -2.85714285714285705364e-02), 3.03030303030303038714e-02), -3.22580645161290313627e-02), 3.44827586206896546939e-02), -3.70370370370370349811e-02), 4.00000000000000008327e-02), -4.34782608695652161845e-02), 4.76190476190476164042e-02), -5.26315789473684181310e-02), 5.88235294117647050660e-02), -6.66666666666666657415e-02), 7.69230769230769273470e-02), -9.09090909090909116141e-02), 1.11111111111111104943e-01), -1.42857142857142849213e-01), 2.00000000000000011102e-01), -3.33333333333333314830e-01), 1.00000000000000000000e+00);
    y = fma(x, p, y);
    return xx < 0.0 ? -y : y;
}

PM_FN double pmd_atan2(double y, double x) {                   /* finite, not both zero */
    if (x == 0.0) return y > 0.0 ? PMD_PI_2 : -PMD_PI_2;
    double z = pmd_atan(y / x);
    if (x < 0.0) z += (y < 0.0) ? -PMD_PI : PMD_PI;
    return z;
}

PM_FN void pm_sincosf(float xx, float *s, float *c) {
    double sd, cd;
    pmd_sincos((double) xx, &sd, &cd);
    *s = (float) sd; *c = (float) cd;
}

PM_FN float pm_expf(float x) {
    if (x != x) return x;
    if (x > 89.0f) return pm_from_bits(0x7f800000u);
    if (x < -104.0f) return 0.0f;
    return (float) pmd_exp((double) x);                        /* the cast rounds into the subnormal range / to infinity */
}

PM_FN float pm_logf(float xx) {
    if (xx != xx) return xx;
    if (xx < 0.0f) return pm_from_bits(0x7fc00000u);
    if (xx == 0.0f) return pm_from_bits(0xff800000u);
    if (pm_to_bits(xx) == 0x7f800000u) return xx;
    return (float) pmd_log((double) xx);                       /* float subnormals are normal doubles */
}

/* x > 0 only (the path calls it with a base in (0,1]) */
PM_FN float pm_powf(float x, float y) {
    if (x == 0.0f) return y > 0.0f ? 0.0f : 1.0f;
    double t = (double) y * pmd_log((double) x);
    if (t > 89.0) return pm_from_bits(0x7f800000u);
    if (t < -104.0) return 0.0f;
    return (float) pmd_exp(t);
}

PM_FN float pm_acosf(float x) {
    if (x < -1.0f || x > 1.0f || x != x) return pm_from_bits(0x7fc00000u);
    double xd = (double) x;
    if (xd == 1.0) return 0.0f;
    return (float) pmd_atan2(sqrt((1.0 - xd) * (1.0 + xd)), xd);     /* (1 - x) and (1 + x) are exact in double */
}

PM_FN float pm_atanf(float xx) {
    if (xx != xx) return xx;
    return (float) pmd_atan((double) xx);
}

PM_FN float pm_atan2f(float y, float x) {
    if (x != x || y != y) return pm_from_bits(0x7fc00000u);
    if (x == 0.0f && y == 0.0f) return 0.0f;
    if (y == 0.0f) return x > 0.0f ? 0.0f : PM_PI;
    return (float) pmd_atan2((double) y, (double) x);
}

PM_FN float pm_tanf(float xx) {
    double sd, cd;
    pmd_sincos((double) xx, &sd, &cd);
    return (float) (sd / cd);
}

#else /* PHIP_FMATH_CEPHES: the round-1 implementation, single-precision Cephes-style polynomials, <= ~4 ulp (kept for A/B) */

/* sin and cos, |x| <= 8192 (larger arguments lose accuracy but stay finite) */
PM_FN void pm_sincosf(float xx, float *s, float *c) {
    const float DP1 = 0.78515625f;
    const float DP2 = 2.4187564849853515625e-4f;
    const float DP3 = 3.77489497744594108e-8f;
    const float FOPI = 1.27323954473516f; /* 4/pi */
    float x = fabsf(xx);
    int sign_s = xx < 0.0f ? -1 : 1;
    int sign_c = 1;

    int j = (int)(FOPI * x);
    float y = (float) j;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    if (j > 3) { sign_s = -sign_s; sign_c = -sign_c; j -= 4; }
    if (j > 1) sign_c = -sign_c;

    x = ((x - y * DP1) - y * DP2) - y * DP3;
    float z = x * x;

    float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * x + x;
    float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z
                + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;

    float rs, rc;
    if (j == 1 || j == 2) { rs = pc; rc = ps; } else { rs = ps; rc = pc; }
    *s = sign_s < 0 ? -rs : rs;
    *c = sign_c < 0 ? -rc : rc;
}

PM_FN float pm_expf(float x) {
    if (x > 88.72283905206835f) return pm_from_bits(0x7f800000u);
    if (x < -103.278929903431851103f) return 0.0f;
    if (x != x) return x;
    const float LOG2EF = 1.44269504088896341f;
    const float C1 = 0.693359375f;
    const float C2 = -2.12194440e-4f;
    float z = floorf(LOG2EF * x + 0.5f);
    x -= z * C1;
    x -= z * C2;
    int n = (int) z;
    z = x * x;
    float p = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x
                 + 4.1665795894e-2f) * x + 1.6666665459e-1f) * x + 5.0000001201e-1f) * z + x + 1.0f;
    return pm_ldexpf(p, n);
}

PM_FN float pm_logf(float xx) {
    if (xx != xx) return xx;
    if (xx < 0.0f) return pm_from_bits(0x7fc00000u);
    if (xx == 0.0f) return pm_from_bits(0xff800000u);
    if (pm_to_bits(xx) == 0x7f800000u) return xx;
    const float SQRTHF = 0.707106781186547524f;
    int e;
    float x = pm_frexpf(xx, &e);
    if (x < SQRTHF) { e -= 1; x = x + x - 1.0f; } else { x = x - 1.0f; }
    float z = x * x;
    float y = ((((((((7.0376836292e-2f * x - 1.1514610310e-1f) * x + 1.1676998740e-1f) * x
                    - 1.2420140846e-1f) * x + 1.4249322787e-1f) * x - 1.6668057665e-1f) * x
                 + 2.0000714765e-1f) * x - 2.4999993993e-1f) * x + 3.3333331174e-1f) * x * z;
    float fe = (float) e;
    if (e != 0) y += -2.12194440e-4f * fe;
    y += -0.5f * z;
    z = x + y;
    if (e != 0) z += 0.693359375f * fe;
    return z;
}

/* x > 0 only (the path calls it with a base in (0,1]) */
PM_FN float pm_powf(float x, float y) {
    if (x == 0.0f) return y > 0.0f ? 0.0f : 1.0f;
    return pm_expf(y * pm_logf(x));
}

PM_FN float pm_asinf_core(float a) { /* 0 <= a <= 0.5 */
    float z = a * a;
    return ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z
             + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * a + a;
}

PM_FN float pm_acosf(float x) {
    if (x < -1.0f || x > 1.0f || x != x) return pm_from_bits(0x7fc00000u);
    if (x < -0.5f)
        return PM_PI - 2.0f * pm_asinf_core(sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f)
        return 2.0f * pm_asinf_core(sqrtf(0.5f * (1.0f - x)));
    float a = fabsf(x);
    float r = pm_asinf_core(a);
    return PM_PI_2 - (x < 0.0f ? -r : r);
}

PM_FN float pm_atanf(float xx) {
    float x = fabsf(xx), y;
    if (x > 2.414213562373095f) { y = PM_PI_2; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = PM_PI_4; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    y += (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z
          - 3.33329491539e-1f) * z * x + x;
    return xx < 0.0f ? -y : y;
}

PM_FN float pm_atan2f(float y, float x) {
    if (x != x || y != y) return pm_from_bits(0x7fc00000u);
    if (x == 0.0f) {
        if (y == 0.0f) return 0.0f;
        return y > 0.0f ? PM_PI_2 : -PM_PI_2;
    }
    if (y == 0.0f)
        return x > 0.0f ? 0.0f : PM_PI;
    float z = pm_atanf(y / x);
    if (x < 0.0f)
        z += (y < 0.0f) ? -PM_PI : PM_PI;
    return z;
}

PM_FN float pm_tanf(float xx) {
    const float DP1 = 0.78515625f;
    const float DP2 = 2.4187564849853515625e-4f;
    const float DP3 = 3.77489497744594108e-8f;
    const float FOPI = 1.27323954473516f;
    float x = fabsf(xx);
    int j = (int)(FOPI * x);
    float y = (float) j;
    if (j & 1) { j += 1; y += 1.0f; }
    float z = ((x - y * DP1) - y * DP2) - y * DP3;
    float zz = z * z;
    if (zz > 1.0e-8f) {
        y = (((((9.38540185543e-3f * zz + 3.11992232697e-3f) * zz + 2.44301354525e-2f) * zz
               + 5.34112807005e-2f) * zz + 1.33387994085e-1f) * zz + 3.33331568548e-1f) * zz * z + z;
    } else {
        y = z;
    }
    if (j & 2) y = -1.0f / y;
    return xx < 0.0f ? -y : y;
}

#endif /* PHIP_FMATH_CEPHES */

#endif /* PHIP_FMATH_H */
