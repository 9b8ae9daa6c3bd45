/* ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * libcrm.so: the float transcendentals the reference's path calls (sincosf / sinf / cosf / expf / logf / acosf / atan2f / atanf /
 * tanf / powf, and exp / log with float-valued arguments: math::fastexp / fastlog, core/math.h:175-199), answered by
 * include/phip_fmath.h -- the correctly rounded functions the GPU kernels and the oracle's parity build compute with -- instead
 * of glibc's (within 1 ulp, not always correctly rounded).  LD_PRELOADed into a process that runs the REFERENCE (oracle/_ref),
 * it removes the one remaining difference between Mitsuba 0.6 on the CPU and path_hip on the GPU: the experiment behind
 * DESIGN.md's statement that the full-size image differences (C3: 8.9e-4 relative L2) are libm rounding and nothing else.
 * Never linked into the product.
 *
 *     LD_PRELOAD=oracle/_build/libcrm.so python tools/fullsize_vs_reference.py out.json
 */
#include <dlfcn.h>
#include <cmath>
#include "../../include/phip_fmath.h"

extern "C" {

void sincosf(float x, float *s, float *c) { pm_sincosf(x, s, c); }
float sinf(float x) { float s, c; pm_sincosf(x, &s, &c); return s; }
float cosf(float x) { float s, c; pm_sincosf(x, &s, &c); return c; }
float expf(float x) { return pm_expf(x); }
float logf(float x) { return pm_logf(x); }
float acosf(float x) { return pm_acosf(x); }
float atan2f(float y, float x) { return pm_atan2f(y, x); }
float atanf(float x) { return pm_atanf(x); }
float tanf(float x) { return pm_tanf(x); }
float powf(float x, float y) { return pm_powf(x, y); }

/* math::fastexp(x) = (float) exp((double) x): a float-valued argument gets the float routine's (exactly representable) answer,
   anything else glibc's own double routine */
double exp(double x) {
    const float f = (float) x;
    if ((double) f == x) return (double) pm_expf(f);
    static double (*real)(double) = (double (*)(double)) dlsym(RTLD_NEXT, "exp");
    return real(x);
}
double log(double x) {
    const float f = (float) x;
    if ((double) f == x) return (double) pm_logf(f);
    static double (*real)(double) = (double (*)(double)) dlsym(RTLD_NEXT, "log");
    return real(x);
}

} // extern "C"
