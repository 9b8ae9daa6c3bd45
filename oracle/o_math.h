/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * o_math.h: vector / spectrum / frame types and leaf math, restating (file:line under
 * /root/reference):
 *   include/mitsuba/core/vector.h:535-626   (v / f multiplies by the reciprocal; normalize)
 *   include/mitsuba/core/spectrum.h:365-583 (RGB Spectrum ops; `/ f` multiplies by 1/f)
 *   include/mitsuba/core/frame.h:30-140     (Frame, toLocal/toWorld, cosTheta ...)
 *   src/libcore/util.cpp:592-608            (coordinateSystem, computeShadingFrame)
 *   src/libcore/util.cpp:651-681,739-761    (fresnelDielectricExt, fresnelConductorExact)
 *   src/libcore/math.cpp:25-72              (erfinv, erf)
 *   src/libcore/warp.cpp:43-52,76-102       (cosine hemisphere, uniform triangle, concentric disk)
 *   include/mitsuba/core/constants.h:24-32  (Epsilon, ShadowEpsilon)
 *
 * All arithmetic is float32 in the reference's operation order; build with -ffp-contract=off.
 * Transcendentals go through om::sincos/exp/log/... which map to phip_fmath.h (default, the
 * parity mode shared bit-for-bit with the GPU) or to libm when ORACLE_LIBM is defined (to
 * measure how much the substitution matters).
 */
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <limits>
#include "../include/phip_fmath.h"

namespace orc {

typedef float Float;

#define ORC_EPSILON        1e-4f
#define ORC_SHADOW_EPSILON 1e-3f
#define ORC_DELTA_EPSILON  1e-3f
#define ORC_PI             3.14159265358979323846f
#define ORC_INV_PI         0.31830988618379067154f
#define ORC_INV_FOURPI     0.07957747154594766788f   /* constants.h:66 */
#define ORC_INV_TWOPI      0.15915494309189533577f   /* constants.h:65 */
#define ORC_RCPOVERFLOW    2.93873587705571876e-39f  /* constants.h:53,58: 2^-128 */

namespace om {
#if defined(ORACLE_LIBM)
inline void sincos(float x, float *s, float *c) { ::sincosf(x, s, c); }
inline float exp(float x) { return ::expf(x); }
inline float log(float x) { return ::logf(x); }
/* math::fastexp / math::fastlog on Linux x86_64 (core/math.h:175-199): the double-precision routine, rounded to float */
inline float fastexp(float x) { return (float) ::exp((double) x); }
inline float fastlog(float x) { return (float) ::log((double) x); }
inline float acos(float x) { return ::acosf(x); }
inline float atan2(float y, float x) { return ::atan2f(y, x); }
inline float tan(float x) { return ::tanf(x); }
inline float atan(float x) { return ::atanf(x); }
inline float pow(float x, float y) { return ::powf(x, y); }
#else
inline void sincos(float x, float *s, float *c) { pm_sincosf(x, s, c); }
inline float exp(float x) { return pm_expf(x); }
inline float log(float x) { return pm_logf(x); }
inline float fastexp(float x) { return pm_expf(x); }
inline float fastlog(float x) { return pm_logf(x); }
inline float acos(float x) { return pm_acosf(x); }
inline float atan2(float y, float x) { return pm_atan2f(y, x); }
inline float tan(float x) { return pm_tanf(x); }
inline float atan(float x) { return pm_atanf(x); }
inline float pow(float x, float y) { return pm_powf(x, y); }
#endif
inline float safe_sqrt(float v) { return std::sqrt(std::max(0.0f, v)); }
inline float signum(float v) { return copysignf(1.0f, v); }
}

struct Vec3 {
    Float x, y, z;
    Vec3() {}
    explicit Vec3(Float v) : x(v), y(v), z(v) {}
    Vec3(Float x, Float y, Float z) : x(x), y(y), z(z) {}
    Float operator[](int i) const { return (&x)[i]; }
    Float &operator[](int i) { return (&x)[i]; }
    Vec3 operator+(const Vec3 &v) const { return Vec3(x + v.x, y + v.y, z + v.z); }
    Vec3 operator-(const Vec3 &v) const { return Vec3(x - v.x, y - v.y, z - v.z); }
    Vec3 &operator+=(const Vec3 &v) { x += v.x; y += v.y; z += v.z; return *this; }
    Vec3 &operator-=(const Vec3 &v) { x -= v.x; y -= v.y; z -= v.z; return *this; }
    Vec3 operator*(Float f) const { return Vec3(x * f, y * f, z * f); }
    Vec3 &operator*=(Float f) { x *= f; y *= f; z *= f; return *this; }
    Vec3 operator-() const { return Vec3(-x, -y, -z); }
    /* vector.h:535-541: division multiplies by the reciprocal */
    Vec3 operator/(Float f) const { Float r = 1.0f / f; return Vec3(x * r, y * r, z * r); }
    Vec3 &operator/=(Float f) { Float r = 1.0f / f; x *= r; y *= r; z *= r; return *this; }
    Float lengthSquared() const { return x * x + y * y + z * z; }
    Float length() const { return std::sqrt(lengthSquared()); }
    bool isZero() const { return x == 0 && y == 0 && z == 0; }
};
inline Vec3 operator*(Float f, const Vec3 &v) { return v * f; }
inline Float dot(const Vec3 &a, const Vec3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Float absDot(const Vec3 &a, const Vec3 &b) { return std::abs(dot(a, b)); }
inline Vec3 cross(const Vec3 &a, const Vec3 &b) {
    return Vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline Vec3 normalize(const Vec3 &v) { return v / v.length(); }

struct Vec2 {
    Float x, y;
    Vec2() {}
    Vec2(Float x, Float y) : x(x), y(y) {}
};

/* RGB spectrum, spectrum.h (SPECTRUM_SAMPLES == 3) */
struct Spectrum {
    Float s[3];
    Spectrum() {}
    explicit Spectrum(Float v) { s[0] = s[1] = s[2] = v; }
    Spectrum(Float r, Float g, Float b) { s[0] = r; s[1] = g; s[2] = b; }
    explicit Spectrum(const float *p) { s[0] = p[0]; s[1] = p[1]; s[2] = p[2]; }
    Float operator[](int i) const { return s[i]; }
    Float &operator[](int i) { return s[i]; }
    Spectrum operator+(const Spectrum &o) const { return Spectrum(s[0] + o.s[0], s[1] + o.s[1], s[2] + o.s[2]); }
    Spectrum operator-(const Spectrum &o) const { return Spectrum(s[0] - o.s[0], s[1] - o.s[1], s[2] - o.s[2]); }
    Spectrum &operator+=(const Spectrum &o) { s[0] += o.s[0]; s[1] += o.s[1]; s[2] += o.s[2]; return *this; }
    Spectrum operator*(const Spectrum &o) const { return Spectrum(s[0] * o.s[0], s[1] * o.s[1], s[2] * o.s[2]); }
    Spectrum &operator*=(const Spectrum &o) { s[0] *= o.s[0]; s[1] *= o.s[1]; s[2] *= o.s[2]; return *this; }
    Spectrum operator*(Float f) const { return Spectrum(s[0] * f, s[1] * f, s[2] * f); }
    Spectrum &operator*=(Float f) { s[0] *= f; s[1] *= f; s[2] *= f; return *this; }
    Spectrum operator/(const Spectrum &o) const { return Spectrum(s[0] / o.s[0], s[1] / o.s[1], s[2] / o.s[2]); }
    /* spectrum.h:415-425,447-456: scalar division multiplies by the reciprocal */
    Spectrum operator/(Float f) const { Float r = 1.0f / f; return Spectrum(s[0] * r, s[1] * r, s[2] * r); }
    Spectrum &operator/=(Float f) { Float r = 1.0f / f; s[0] *= r; s[1] *= r; s[2] *= r; return *this; }
    bool isZero() const { return s[0] == 0.0f && s[1] == 0.0f && s[2] == 0.0f; }
    Float max() const { Float r = s[0]; for (int i = 1; i < 3; i++) r = std::max(r, s[i]); return r; }
    /* spectrum.h:481-486 */
    Float average() const { Float r = 0.0f; for (int i = 0; i < 3; i++) r += s[i]; return r * (1.0f / 3); }
    Spectrum safe_sqrt() const { return Spectrum(om::safe_sqrt(s[0]), om::safe_sqrt(s[1]), om::safe_sqrt(s[2])); }
};
inline Spectrum operator*(Float f, const Spectrum &s) { return s * f; }

/* util.cpp:592-601 */
inline void coordinateSystem(const Vec3 &a, Vec3 &b, Vec3 &c) {
    if (std::abs(a.x) > std::abs(a.y)) {
        Float invLen = 1.0f / std::sqrt(a.x * a.x + a.z * a.z);
        c = Vec3(a.z * invLen, 0.0f, -a.x * invLen);
    } else {
        Float invLen = 1.0f / std::sqrt(a.y * a.y + a.z * a.z);
        c = Vec3(0.0f, a.z * invLen, -a.y * invLen);
    }
    b = cross(c, a);
}

/* frame.h */
struct Frame {
    Vec3 s, t, n;
    Frame() {}
    explicit Frame(const Vec3 &n_) : n(n_) { coordinateSystem(n, s, t); }
    Vec3 toLocal(const Vec3 &v) const { return Vec3(dot(v, s), dot(v, t), dot(v, n)); }
    Vec3 toWorld(const Vec3 &v) const { return s * v.x + t * v.y + n * v.z; }
    static Float cosTheta(const Vec3 &v) { return v.z; }
    static Float cosTheta2(const Vec3 &v) { return v.z * v.z; }
    static Float sinTheta2(const Vec3 &v) { return 1.0f - v.z * v.z; }
    static Float tanTheta(const Vec3 &v) {
        Float temp = 1 - v.z * v.z;
        if (temp <= 0.0f) return 0.0f;
        return std::sqrt(temp) / v.z;
    }
};

/* util.cpp:603-608 */
inline void computeShadingFrame(const Vec3 &n, const Vec3 &dpdu, Frame &frame) {
    frame.n = n;
    frame.s = normalize(dpdu - frame.n * dot(frame.n, dpdu));
    frame.t = cross(frame.n, frame.s);
}

/* ---------------- warp.cpp ---------------- */
inline Vec2 squareToUniformDiskConcentric(const Vec2 &sample) { /* warp.cpp:81-102 */
    Float r1 = 2.0f * sample.x - 1.0f;
    Float r2 = 2.0f * sample.y - 1.0f;
    Float phi, r;
    if (r1 == 0 && r2 == 0) {
        r = phi = 0;
    } else if (r1 * r1 > r2 * r2) {
        r = r1;
        phi = (ORC_PI / 4.0f) * (r2 / r1);
    } else {
        r = r2;
        phi = (ORC_PI / 2.0f) - (r1 / r2) * (ORC_PI / 4.0f);
    }
    Float cosPhi, sinPhi;
    om::sincos(phi, &sinPhi, &cosPhi);
    return Vec2(r * cosPhi, r * sinPhi);
}

inline Vec3 squareToCosineHemisphere(const Vec2 &sample) { /* warp.cpp:43-52 */
    Vec2 p = squareToUniformDiskConcentric(sample);
    Float z = om::safe_sqrt(1.0f - p.x * p.x - p.y * p.y);
    if (z == 0) z = 1e-10f;
    return Vec3(p.x, p.y, z);
}

inline Float squareToCosineHemispherePdf(const Vec3 &d) { return ORC_INV_PI * Frame::cosTheta(d); } /* warp.h:55-56 */

inline Vec2 squareToUniformTriangle(const Vec2 &sample) { /* warp.cpp:76-79 */
    Float a = om::safe_sqrt(1.0f - sample.x);
    return Vec2(1 - a, a * sample.y);
}

inline Vec3 squareToUniformSphere(const Vec2 &sample) { /* warp.cpp:27-34 (test_kd workload) */
    Float z = 1.0f - 2.0f * sample.y;
    Float r = om::safe_sqrt(1.0f - z * z);
    Float sinPhi, cosPhi;
    om::sincos(2.0f * ORC_PI * sample.x, &sinPhi, &cosPhi);
    return Vec3(r * cosPhi, r * sinPhi, z);
}

inline Float squareToUniformSpherePdf() { return ORC_INV_FOURPI; } /* warp.h:43 */

/* util.cpp:447-485 */
inline bool solveQuadratic(Float a, Float b, Float c, Float &x0, Float &x1) {
    if (a == 0) {
        if (b != 0) {
            x0 = x1 = -c / b;
            return true;
        }
        return false;
    }
    Float discrim = b * b - 4.0f * a * c;
    if (discrim < 0)
        return false;
    Float temp, sqrtDiscrim = std::sqrt(discrim);
    if (b < 0)
        temp = -0.5f * (b - sqrtDiscrim);
    else
        temp = -0.5f * (b + sqrtDiscrim);
    x0 = temp / a;
    x1 = c / temp;
    if (x0 > x1)
        std::swap(x0, x1);
    return true;
}

/* bsphere.h:30-95 */
struct BSphere {
    Vec3 center; Float radius;
    BSphere() : center(0.0f), radius(0.0f) {}
    BSphere(const Vec3 &c, Float r) : center(c), radius(r) {}
    bool rayIntersect(const Vec3 &ro, const Vec3 &rd, Float &nearHit, Float &farHit) const {
        Vec3 o = ro - center;
        Float A = rd.lengthSquared();
        Float B = 2 * dot(o, rd);
        Float C = o.lengthSquared() - radius * radius;
        return solveQuadratic(A, B, C, nearHit, farHit);
    }
};

/* ---------------- math.cpp:25-72 ---------------- */
inline Float mts_erfinv(Float x) {
    Float w = -om::fastlog(((Float) 1 - x) * ((Float) 1 + x));
    Float p;
    if (w < (Float) 5) {
        w = w - (Float) 2.5;
        p = (Float) 2.81022636e-08;
        p = (Float) 3.43273939e-07 + p * w;
        p = (Float) -3.5233877e-06 + p * w;
        p = (Float) -4.39150654e-06 + p * w;
        p = (Float) 0.00021858087 + p * w;
        p = (Float) -0.00125372503 + p * w;
        p = (Float) -0.00417768164 + p * w;
        p = (Float) 0.246640727 + p * w;
        p = (Float) 1.50140941 + p * w;
    } else {
        w = std::sqrt(w) - (Float) 3;
        p = (Float) -0.000200214257;
        p = (Float) 0.000100950558 + p * w;
        p = (Float) 0.00134934322 + p * w;
        p = (Float) -0.00367342844 + p * w;
        p = (Float) 0.00573950773 + p * w;
        p = (Float) -0.0076224613 + p * w;
        p = (Float) 0.00943887047 + p * w;
        p = (Float) 1.00167406 + p * w;
        p = (Float) 2.83297682 + p * w;
    }
    return p * x;
}

inline Float mts_erf(Float x) {
    Float a1 = (Float) 0.254829592;
    Float a2 = (Float) -0.284496736;
    Float a3 = (Float) 1.421413741;
    Float a4 = (Float) -1.453152027;
    Float a5 = (Float) 1.061405429;
    Float p = (Float) 0.3275911;
    Float sign = om::signum(x);
    x = std::abs(x);
    Float t = (Float) 1.0 / ((Float) 1.0 + p * x);
    Float y = (Float) 1.0 - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * om::fastexp(-x * x);
    return sign * y;
}

inline float hypot2(float a, float b) { /* math.cpp:74-86 */
    float r;
    if (std::abs(a) > std::abs(b)) {
        r = b / a;
        r = std::abs(a) * std::sqrt(1.0f + r * r);
    } else if (b != 0.0f) {
        r = a / b;
        r = std::abs(b) * std::sqrt(1.0f + r * r);
    } else {
        r = 0.0f;
    }
    return r;
}

/* ---------------- Fresnel, util.cpp ---------------- */
inline Float fresnelDielectricExt(Float cosThetaI_, Float &cosThetaT_, Float eta) { /* util.cpp:651-681 */
    if (eta == 1) {
        cosThetaT_ = -cosThetaI_;
        return 0.0f;
    }
    Float scale = (cosThetaI_ > 0) ? 1 / eta : eta,
          cosThetaTSqr = 1 - (1 - cosThetaI_ * cosThetaI_) * (scale * scale);
    if (cosThetaTSqr <= 0.0f) {
        cosThetaT_ = 0.0f;
        return 1.0f;
    }
    Float cosThetaI = std::abs(cosThetaI_);
    Float cosThetaT = std::sqrt(cosThetaTSqr);
    Float Rs = (cosThetaI - eta * cosThetaT) / (cosThetaI + eta * cosThetaT);
    Float Rp = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    cosThetaT_ = (cosThetaI_ > 0) ? -cosThetaT : cosThetaT;
    return 0.5f * (Rs * Rs + Rp * Rp);
}

inline Spectrum fresnelConductorExact(Float cosThetaI, const Spectrum &eta, const Spectrum &k) { /* util.cpp:739-761 */
    Float cosThetaI2 = cosThetaI * cosThetaI,
          sinThetaI2 = 1 - cosThetaI2,
          sinThetaI4 = sinThetaI2 * sinThetaI2;
    Spectrum temp1 = eta * eta - k * k - Spectrum(sinThetaI2),
             a2pb2 = (temp1 * temp1 + k * k * eta * eta * 4).safe_sqrt(),
             a = ((a2pb2 + temp1) * 0.5f).safe_sqrt();
    Spectrum term1 = a2pb2 + Spectrum(cosThetaI2),
             term2 = a * (2 * cosThetaI);
    Spectrum Rs2 = (term1 - term2) / (term1 + term2);
    Spectrum term3 = a2pb2 * cosThetaI2 + Spectrum(sinThetaI4),
             term4 = term2 * sinThetaI2;
    Spectrum Rp2 = Rs2 * (term3 - term4) / (term3 + term4);
    return 0.5f * (Rp2 + Rs2);
}

/* util.cpp:527-539; RCPOVERFLOW = constants.h:46 (single precision) */
inline bool solveLinearSystem2x2(const Float a[2][2], const Float b[2], Float x[2]) {
    Float det = a[0][0] * a[1][1] - a[0][1] * a[1][0];
    if (std::abs(det) <= ORC_RCPOVERFLOW)
        return false;
    Float inverse = (Float) 1.0f / det;
    x[0] = (a[1][1] * b[0] - a[0][1] * b[1]) * inverse;
    x[1] = (a[0][0] * b[1] - a[1][0] * b[0]) * inverse;
    return true;
}

/* ---------------- 4x4 matrix (matrix.h / matrix.inl) ---------------- */
struct Mat4 {
    Float m[4][4];
    static Mat4 identity() { Mat4 r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = i == j ? 1.0f : 0.0f; return r; }
    Mat4 operator*(const Mat4 &o) const { /* matrix.h:744-757 */
        Mat4 r;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                Float sum = 0;
                for (int k = 0; k < 4; ++k) sum += m[i][k] * o.m[k][j];
                r.m[i][j] = sum;
            }
        return r;
    }
    bool invert(Mat4 &target) const { /* matrix.inl:138-193 */
        const int N = 4;
        int indxc[N], indxr[N], ipiv[N];
        memset(ipiv, 0, sizeof(ipiv));
        memcpy(target.m, m, sizeof(m));
        for (int i = 0; i < N; i++) {
            int irow = -1, icol = -1;
            Float big = 0;
            for (int j = 0; j < N; j++) {
                if (ipiv[j] != 1) {
                    for (int k = 0; k < N; k++) {
                        if (ipiv[k] == 0) {
                            if (std::abs(target.m[j][k]) >= big) { big = std::abs(target.m[j][k]); irow = j; icol = k; }
                        } else if (ipiv[k] > 1) return false;
                    }
                }
            }
            ++ipiv[icol];
            if (irow != icol) for (int k = 0; k < N; ++k) std::swap(target.m[irow][k], target.m[icol][k]);
            indxr[i] = irow; indxc[i] = icol;
            if (target.m[icol][icol] == 0) return false;
            Float pivinv = 1.f / target.m[icol][icol];
            target.m[icol][icol] = 1.f;
            for (int j = 0; j < N; j++) target.m[icol][j] *= pivinv;
            for (int j = 0; j < N; j++) {
                if (j != icol) {
                    Float save = target.m[j][icol];
                    target.m[j][icol] = 0;
                    for (int k = 0; k < N; k++) target.m[j][k] -= target.m[icol][k] * save;
                }
            }
        }
        for (int j = N - 1; j >= 0; j--)
            if (indxr[j] != indxc[j])
                for (int k = 0; k < N; k++) std::swap(target.m[k][indxr[j]], target.m[k][indxc[j]]);
        return true;
    }
};


} // namespace orc
