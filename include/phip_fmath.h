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

#endif /* PHIP_FMATH_H */
