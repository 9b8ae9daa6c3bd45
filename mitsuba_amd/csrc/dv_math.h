/*
 * dv_math.h -- float3 / RGB helpers and leaf math for the path_hip kernels.
 *
 * Every function is __host__ __device__: the kernels use the device instantiation; the
 * phip_debug_host_* entry points run the SAME source on the host so that CPU-only tests can
 * compare the product's shading arithmetic with the oracle bit for bit.
 *
 * Operation order follows the reference (so results match the CPU `path` integrator up to the
 * transcendental substitution of include/phip_fmath.h); the file must be compiled with
 * -ffp-contract=off.  Reference semantics restated here (file:line under /root/reference):
 *   include/mitsuba/core/vector.h:535-626   `v / f` multiplies by the reciprocal
 *   include/mitsuba/core/spectrum.h:415-456 same for Spectrum / scalar
 *   src/libcore/util.cpp:592-608            coordinateSystem, computeShadingFrame
 *   src/libcore/util.cpp:651-681,739-761    fresnelDielectricExt, fresnelConductorExact
 *   src/libcore/math.cpp:25-72              erfinv, erf
 *   src/libcore/warp.cpp:43-52,76-102       cosine hemisphere, uniform triangle, concentric disk
 */
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/phip_fmath.h"

#define DV __host__ __device__ __forceinline__
/* PHIP_EXPERIMENTS (tools/build_variant.sh -DPHIP_EXPERIMENTS=1): the alternatives that were measured against the product -- earlier kernel generations,
   algorithm-selecting environment variables -- are compiled in; the shipped library carries one algorithm per job and reads no environment beyond the
   documented knobs */
#ifndef PHIP_EXPERIMENTS
#define PHIP_EXPERIMENTS 0
#endif
/* the Sobol' row loops of sobolseq.h (one table read per index bit): the device code of the product draws through the byte tables only; the host keeps both
   forms (tests/test_host_parity.py holds them against each other) */
#if defined(__HIP_DEVICE_COMPILE__) && !PHIP_EXPERIMENTS
#define SOBOL_ROW_LOOPS 0
#else
#define SOBOL_ROW_LOOPS 1
#endif

/* float -> int the way the reference's hardware does it (x86 cvttss2si: NaN and out-of-range values give INT_MIN, the
   "integer indefinite" -- e.g. floorToInt(NaN) < 0 sends MIPMap::eval down its bilinear branch, mipmap.h:657-662); AMD's
   v_cvt_i32_f32 saturates and maps NaN to 0, and in C the conversion is undefined: spelled out so that the device takes the
   reference's branch (found by the fuzz test: a camera ray through the pole of an envmap has NaN differentials) */
DV int f2i(float x) { return (x >= -2147483648.0f && x < 2147483648.0f) ? (int) x : (int) 0x80000000; }


#define PT_EPSILON        1e-4f
#define PT_SHADOW_EPSILON 1e-3f
#define PT_PI             3.14159265358979323846f
#define PT_INV_PI         0.31830988618379067154f
#define PT_INV_FOURPI     0.07957747154594766788f
#define PT_INV_TWOPI      0.15915494309189533577f

namespace pt {

/* std::max / std::min semantics (NOT fmaxf: NaN behaviour must match the CPU restatement) */
DV float smax(float a, float b) { return (a < b) ? b : a; }
DV float smin(float a, float b) { return (b < a) ? b : a; }
DV float safe_sqrt(float v) { return sqrtf(smax(0.0f, v)); }

struct V3 {
    float x, y, z;
    DV V3() {}
    DV explicit V3(float v) : x(v), y(v), z(v) {}
    DV V3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    DV V3 operator+(const V3 &v) const { return V3(x + v.x, y + v.y, z + v.z); }
    DV V3 operator-(const V3 &v) const { return V3(x - v.x, y - v.y, z - v.z); }
    DV V3 operator*(float f) const { return V3(x * f, y * f, z * f); }
    DV V3 operator*(const V3 &v) const { return V3(x * v.x, y * v.y, z * v.z); }   /* Spectrum * Spectrum */
    DV V3 operator-() const { return V3(-x, -y, -z); }
    DV V3 operator/(float f) const { float r = 1.0f / f; return V3(x * r, y * r, z * r); }
    DV V3 div(const V3 &v) const { return V3(x / v.x, y / v.y, z / v.z); }          /* Spectrum / Spectrum */
    DV float lengthSquared() const { return x * x + y * y + z * z; }
    DV float length() const { return sqrtf(lengthSquared()); }
    DV bool isZero() const { return x == 0.0f && y == 0.0f && z == 0.0f; }
    DV float maxc() const { float r = x; r = smax(r, y); r = smax(r, z); return r; }
    DV float average() const { float r = 0.0f; r += x; r += y; r += z; return r * (1.0f / 3); }
    DV float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
DV V3 operator*(float f, const V3 &v) { return v * f; }
DV float dot(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DV float absDot(const V3 &a, const V3 &b) { return fabsf(dot(a, b)); }
DV V3 cross(const V3 &a, const V3 &b) { return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DV V3 normalize(const V3 &v) { return v / v.length(); }
DV V3 sqrt3(const V3 &v) { return V3(safe_sqrt(v.x), safe_sqrt(v.y), safe_sqrt(v.z)); }

struct V2 { float x, y; DV V2() {} DV V2(float x_, float y_) : x(x_), y(y_) {} };

struct Frame {
    V3 s, t, n;
    DV V3 toLocal(const V3 &v) const { return V3(dot(v, s), dot(v, t), dot(v, n)); }
    DV V3 toWorld(const V3 &v) const { return s * v.x + t * v.y + n * v.z; }
};
DV float cosTheta(const V3 &v) { return v.z; }
DV float tanTheta(const V3 &v) {
    float temp = 1 - v.z * v.z;
    if (temp <= 0.0f) return 0.0f;
    return sqrtf(temp) / v.z;
}

/* util.cpp:592-601 */
DV void coordinateSystem(const V3 &a, V3 &b, V3 &c) {
    if (fabsf(a.x) > fabsf(a.y)) {
        float invLen = 1.0f / sqrtf(a.x * a.x + a.z * a.z);
        c = V3(a.z * invLen, 0.0f, -a.x * invLen);
    } else {
        float invLen = 1.0f / sqrtf(a.y * a.y + a.z * a.z);
        c = V3(0.0f, a.z * invLen, -a.y * invLen);
    }
    b = cross(c, a);
}

/* ---- warp.cpp ---- */
DV V2 squareToUniformDiskConcentric(const V2 &sample) {
    float r1 = 2.0f * sample.x - 1.0f;
    float r2 = 2.0f * sample.y - 1.0f;
    float phi, r;
    if (r1 == 0 && r2 == 0) {
        r = phi = 0;
    } else if (r1 * r1 > r2 * r2) {
        r = r1;
        phi = (PT_PI / 4.0f) * (r2 / r1);
    } else {
        r = r2;
        phi = (PT_PI / 2.0f) - (r1 / r2) * (PT_PI / 4.0f);
    }
    float cosPhi, sinPhi;
    pm_sincosf(phi, &sinPhi, &cosPhi);
    return V2(r * cosPhi, r * sinPhi);
}
DV V3 squareToCosineHemisphere(const V2 &sample) {
    V2 p = squareToUniformDiskConcentric(sample);
    float z = safe_sqrt(1.0f - p.x * p.x - p.y * p.y);
    if (z == 0) z = 1e-10f;
    return V3(p.x, p.y, z);
}
DV V3 squareToUniformSphere(const V2 &sample) {   /* warp.cpp:27-34 */
    float z = 1.0f - 2.0f * sample.y;
    float r = safe_sqrt(1.0f - z * z);
    float sinPhi, cosPhi;
    pm_sincosf(2.0f * PT_PI * sample.x, &sinPhi, &cosPhi);
    return V3(r * cosPhi, r * sinPhi, z);
}
/* util.cpp:447-485 */
DV bool solveQuadratic(float a, float b, float c, float &x0, float &x1) {
    if (a == 0) {
        if (b != 0) { x0 = x1 = -c / b; return true; }
        return false;
    }
    float discrim = b * b - 4.0f * a * c;
    if (discrim < 0) return false;
    float temp, sqrtDiscrim = sqrtf(discrim);
    if (b < 0) temp = -0.5f * (b - sqrtDiscrim);
    else temp = -0.5f * (b + sqrtDiscrim);
    x0 = temp / a;
    x1 = c / temp;
    if (x0 > x1) { float t = x0; x0 = x1; x1 = t; }
    return true;
}
DV V2 squareToUniformTriangle(const V2 &sample) {
    float a = safe_sqrt(1.0f - sample.x);
    return V2(1 - a, a * sample.y);
}

/* ---- math.cpp:25-72 ---- */
DV float mts_erfinv(float x) {
    float w = -pm_logf((1.0f - x) * (1.0f + x));
    float p;
    if (w < 5.0f) {
        w = w - 2.5f;
        p = 2.81022636e-08f;
        p = 3.43273939e-07f + p * w;
        p = -3.5233877e-06f + p * w;
        p = -4.39150654e-06f + p * w;
        p = 0.00021858087f + p * w;
        p = -0.00125372503f + p * w;
        p = -0.00417768164f + p * w;
        p = 0.246640727f + p * w;
        p = 1.50140941f + p * w;
    } else {
        w = sqrtf(w) - 3.0f;
        p = -0.000200214257f;
        p = 0.000100950558f + p * w;
        p = 0.00134934322f + p * w;
        p = -0.00367342844f + p * w;
        p = 0.00573950773f + p * w;
        p = -0.0076224613f + p * w;
        p = 0.00943887047f + p * w;
        p = 1.00167406f + p * w;
        p = 2.83297682f + p * w;
    }
    return p * x;
}
DV float mts_erf(float x) {
    const float a1 = 0.254829592f, a2 = -0.284496736f, a3 = 1.421413741f, a4 = -1.453152027f, a5 = 1.061405429f, p = 0.3275911f;
    float sign = copysignf(1.0f, x);
    x = fabsf(x);
    float t = 1.0f / (1.0f + p * x);
    float y = 1.0f - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * pm_expf(-x * x);
    return sign * y;
}
DV float hypot2(float a, float b) {
    float r;
    if (fabsf(a) > fabsf(b)) { r = b / a; r = fabsf(a) * sqrtf(1.0f + r * r); }
    else if (b != 0.0f) { r = a / b; r = fabsf(b) * sqrtf(1.0f + r * r); }
    else r = 0.0f;
    return r;
}

/* ---- util.cpp:651-681 ---- */
DV float fresnelDielectricExt(float cosThetaI_, float &cosThetaT_, float eta) {
    if (eta == 1) { cosThetaT_ = -cosThetaI_; return 0.0f; }
    float scale = (cosThetaI_ > 0) ? 1 / eta : eta,
          cosThetaTSqr = 1 - (1 - cosThetaI_ * cosThetaI_) * (scale * scale);
    if (cosThetaTSqr <= 0.0f) { cosThetaT_ = 0.0f; return 1.0f; }
    float cosThetaI = fabsf(cosThetaI_);
    float cosThetaT = sqrtf(cosThetaTSqr);
    float Rs = (cosThetaI - eta * cosThetaT) / (cosThetaI + eta * cosThetaT);
    float Rp = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    cosThetaT_ = (cosThetaI_ > 0) ? -cosThetaT : cosThetaT;
    return 0.5f * (Rs * Rs + Rp * Rp);
}

/* ---- util.cpp:739-761 (Spectrum version) ---- */
DV V3 fresnelConductorExact(float cosThetaI, const V3 &eta, const V3 &k) {
    float cosThetaI2 = cosThetaI * cosThetaI,
          sinThetaI2 = 1 - cosThetaI2,
          sinThetaI4 = sinThetaI2 * sinThetaI2;
    V3 temp1 = eta * eta - k * k - V3(sinThetaI2),
       a2pb2 = sqrt3(temp1 * temp1 + k * k * eta * eta * 4),
       a = sqrt3((a2pb2 + temp1) * 0.5f);
    V3 term1 = a2pb2 + V3(cosThetaI2),
       term2 = a * (2 * cosThetaI);
    V3 Rs2 = (term1 - term2).div(term1 + term2);
    V3 term3 = a2pb2 * cosThetaI2 + V3(sinThetaI4),
       term4 = term2 * sinThetaI2;
    V3 Rp2 = (Rs2 * (term3 - term4)).div(term3 + term4);
    return 0.5f * (Rp2 + Rs2);
}

/* ---- the counter-based parity stream (phip_sampler_kind CTR), see include/phip.h ----
 * pcg4d (Jarzynski & Olano, JCGT 9(3) 2020) keyed by (pixel, sample, block, seed); words become
 * floats exactly like Random::nextFloat (src/libcore/random.cpp:632-641). */
struct U4 { uint32_t x, y, z, w; };
DV U4 pcg4d(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    U4 v;
    v.x = a * 1664525u + 1013904223u; v.y = b * 1664525u + 1013904223u;
    v.z = c * 1664525u + 1013904223u; v.w = d * 1664525u + 1013904223u;
    v.x += v.y * v.w; v.y += v.z * v.x; v.z += v.x * v.y; v.w += v.y * v.z;
    v.x ^= v.x >> 16; v.y ^= v.y >> 16; v.z ^= v.z >> 16; v.w ^= v.w >> 16;
    v.x += v.y * v.w; v.y += v.z * v.x; v.z += v.x * v.y; v.w += v.y * v.z;
    return v;
}
DV float u32ToFloat(uint32_t u) { return pm_from_bits((u >> 9) | 0x3f800000u) - 1.0f; }

/* ---- PHIP_SAMPLER_LD: the construction of src/samplers/ldsampler.cpp (per pixel and dimension a randomly scrambled (0,2)-sequence
 * in a random order, ldsampler.cpp:151-186) with the scrambles and the order drawn from the counter-based generator instead of the
 * worker's sequential Random, so that every (pixel, sample, dimension) is addressable.  Point construction: include/mitsuba/core/qmc.h. */
#define LD_DIMENSIONS 4u               /* ldsampler's `dimension` default: that many 1D and 2D requests per sample are stratified (ldsampler.cpp:79) */
DV float radicalInverse2Single(uint32_t n, uint32_t scramble) {      /* qmc.h:43-59: van der Corput, 24 bits */
    n = (n << 16) | (n >> 16);
    n = ((n & 0x00ff00ffu) << 8) | ((n & 0xff00ff00u) >> 8);
    n = ((n & 0x0f0f0f0fu) << 4) | ((n & 0xf0f0f0f0u) >> 4);
    n = ((n & 0x33333333u) << 2) | ((n & 0xccccccccu) >> 2);
    n = ((n & 0x55555555u) << 1) | ((n & 0xaaaaaaaau) >> 1);
    n = (n >> 8) ^ (scramble & 0x00ffffffu);
    return (float) n / 16777216.0f;
}
DV float sobol2Single(uint32_t n, uint32_t scramble) {               /* qmc.h:82-87 (may round to 1.0f, as in the reference) */
    for (uint32_t v = 1u << 31; n != 0; n >>= 1, v ^= v >> 1)
        if (n & 1u) scramble ^= v;
    return (float) scramble / 4294967296.0f;
}
/* a keyed pseudo-random permutation of [0, mask] (mask = 2^m - 1): the "random order" (Random::shuffle, ldsampler.cpp:163,186) */
DV uint32_t ldPermute(uint32_t i, uint32_t mask, uint32_t key) {
    /* A. Kensler, "Correlated Multi-Jittered Sampling", Pixar TR 13-01, listing `permute` for a power-of-two domain (no cycle walking):
       multiplications by odd constants and xor-shifts, all confined to the low bits */
    i ^= key;                i *= 0xe170893du;
    i ^= key >> 16;
    i ^= (i & mask) >> 4;
    i ^= key >> 8;           i *= 0x0929eb3fu;
    i ^= key >> 23;
    i ^= (i & mask) >> 1;    i *= 1u | key >> 27;
                             i *= 0x6935fa69u;
    i ^= (i & mask) >> 11;   i *= 0x74dcb303u;
    i ^= (i & mask) >> 2;    i *= 0x9e501cc3u;
    i ^= (i & mask) >> 2;    i *= 0xc860a3dfu;
    i &= mask;
    i ^= i >> 5;
    return (i + key) & mask;
}
/* the same over a domain of any size (`direct`: a sample array of sampleCount * N entries is ONE scrambled sequence in a random order,
   ldsampler.cpp:193-197): the permutation of the next power of two, walked until it lands inside (Kensler, ibid.) */
DV uint32_t ldPermuteAny(uint32_t i, uint32_t l, uint32_t key) {
    uint32_t w = l - 1u;
    w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16;
    do {
        i ^= key;             i *= 0xe170893du;
        i ^= key >> 16;
        i ^= (i & w) >> 4;
        i ^= key >> 8;        i *= 0x0929eb3fu;
        i ^= key >> 23;
        i ^= (i & w) >> 1;    i *= 1u | key >> 27;
                              i *= 0x6935fa69u;
        i ^= (i & w) >> 11;   i *= 0x74dcb303u;
        i ^= (i & w) >> 2;    i *= 0x9e501cc3u;
        i ^= (i & w) >> 2;    i *= 0xc860a3dfu;
        i &= w;
        i ^= i >> 5;
    } while (i >= l);
    return (i + key) % l;
}
/* entry `idx` of the 2D sample array `a` (in request order) of `pixel`, `total` = sampleCount * entries per sample */
DV void ldArrayPoint(uint32_t pixel, uint32_t a, uint32_t idx, uint32_t total, uint32_t seed, float &x, float &y) {
    const U4 h = pcg4d(pixel, 0x100u + a, 0x4c44u, seed);
    const uint32_t i = ldPermuteAny(idx, total, h.x);
    x = radicalInverse2Single(i, h.y);
    y = sobol2Single(i, h.z);
}
/* the request `dim` (2D: 2 * q, 1D: 2 * j + 1) of sample `k` of `pixel`; mask = sampleCount - 1 (a power of two, ldsampler.cpp:83-87) */
DV void ldPoint(uint32_t pixel, uint32_t k, uint32_t dim, uint32_t seed, uint32_t mask, float &x, float &y) {
    const U4 h = pcg4d(pixel, dim, 0x4c44u /* 'LD' */, seed);
    const uint32_t i = ldPermute(k & mask, mask, h.x);
    x = radicalInverse2Single(i, h.y);
    y = sobol2Single(i, h.z);
}

/* ---- PHIP_SAMPLER_SOBOL: the reference's `sobol` sampler as it stands (src/samplers/sobol.cpp, sobolseq.h), SINGLE_PRECISION.  The
 *      direction numbers are DATA handed through the ABI (phip_render_params.sobol_*): nothing of sobolseq.cpp is in this repository. ---- */
struct SobolTab {
    const uint32_t *matrices;           /* sobol::Matrices::matrices32: dims x 52 */
    const uint64_t *vdc, *vdcInv;       /* rows [m - 1] of vdc_sobol_matrices / vdc_sobol_matrices_inv */
    uint32_t dims, logRes, scramble;    /* m_scramble truncated to 32 bits (sobolseq.h:86: sampleSingle(index, dimension, (uint32_t) scramble)) */
    float resolution;                   /* 2^logRes */
    /* byte tables (device only; built by phip.hip, NULL on the host side of the CPU tests): entry [dimension][b][v] = the XOR of the rows 8 b + j over
       the set bits j of v -- the contribution of byte b of the index.  A number is then 4 look-ups for a 28-bit index (C2: 2 x 10 bits of pixel, 8 of
       sample) instead of 28 row reads and 28 selects; XOR is associative, so it is the same number.  vdcBt / vdcInvBt: the same for the two
       enumeration rows (64-bit entries). */
    const uint32_t *matBt;              /* dims x SOBOL_BT_BYTES x 256 */
    const uint64_t *vdcBt, *vdcInvBt;   /* 4 x 256 (the frame has 32 bits), 7 x 256 (a pixel code has at most 52 bits) */
};
#define SOBOL_BT_BYTES 8u               /* every byte of a 64-bit index (rows 52 .. 63 of a dimension are, as in sobolseq.h, the first rows of the next one) */
DV uint32_t byteLength64(uint64_t v) { return v ? (71u - (uint32_t) __builtin_clzll((unsigned long long) v)) >> 3 : 0u; }
/* The loops of sobolseq.h are `for (; bits; bits >>= 1, ++c) if (bits & 1) acc ^= table[c]`: a table read behind a branch behind a shift, one
   memory round trip per bit -- ~28 dependent round trips per number drawn (C2 with <sampler type="sobol"/>: 158.8 ms per frame in k_mega, 173 ms in
   the film pass that re-derives the pixel jitter).  XOR does not care about order and a zero changes nothing, so every row up to the highest set
   bit is read unconditionally, eight reads in flight, and masked by its bit: the same number. */
DV uint32_t bitLength32(uint32_t v) { return v ? 32u - (uint32_t) __builtin_clz(v) : 0u; }
DV uint32_t bitLength64(uint64_t v) { return v ? 64u - (uint32_t) __builtin_clzll((unsigned long long) v) : 0u; }
/* index of the frame-th point of the sequence that falls into pixel (px, py): sobol::look_up, sobolseq.h:94-130 */
DV uint64_t sobolLookUp(const SobolTab &T, uint32_t frame, uint32_t px, uint32_t py) {
    const uint32_t m = T.logRes, m2 = m << 1;
    uint64_t index = (uint64_t) frame << m2;
    uint64_t delta = 0;
    if (!SOBOL_ROW_LOOPS || T.vdcBt) {
        const uint64_t *t = T.vdcBt;
#pragma unroll
        for (uint32_t b = 0; b < 4u; ++b) delta ^= t[b * 256u + ((frame >> (8u * b)) & 255u)];      /* (entry [b][0] = 0) */
        const uint64_t scr = (uint64_t) (T.scramble >> (32u - m));
        const uint64_t bb = ((((uint64_t) px ^ scr) << m) | ((uint64_t) py ^ scr)) ^ delta;
        const uint32_t nb = byteLength64(bb);
        const uint64_t *ti = T.vdcInvBt;
        for (uint32_t b = 0; b < nb; ++b) index ^= ti[b * 256u + (uint32_t) ((bb >> (8u * b)) & 255ull)];
        return index;
    }
    const uint32_t nf = bitLength32(frame);
#pragma unroll 8
    for (uint32_t c = 0; c < nf; ++c) { const uint64_t v = T.vdc[c]; delta ^= ((frame >> c) & 1u) ? v : 0ull; }
    const uint64_t scramble = (uint64_t) (T.scramble >> (32u - m));
    const uint64_t b = ((((uint64_t) px ^ scramble) << m) | ((uint64_t) py ^ scramble)) ^ delta;
    const uint32_t nb = bitLength64(b);
#pragma unroll 8
    for (uint32_t c = 0; c < nb; ++c) { const uint64_t v = T.vdcInv[c]; index ^= ((b >> c) & 1ull) ? v : 0ull; }
    return index;
}
/* SobolSampler::setSampleIndex, sobol.cpp:207-219 */
DV uint64_t sobolSampleIndex(const SobolTab &T, uint32_t sampleIndex, uint32_t px, uint32_t py) {
    return T.logRes > 1u ? sobolLookUp(T, sampleIndex, px, py) : (uint64_t) sampleIndex;
}
/* sobol::sampleSingle, sobolseq.h:42-58 */
DV float sobolSample(const SobolTab &T, uint64_t index, uint32_t dimension) {
    uint32_t result = T.scramble;
    if (!SOBOL_ROW_LOOPS || T.matBt) {
        const uint32_t *t = T.matBt + (size_t) dimension * (SOBOL_BT_BYTES * 256u);
        const uint32_t nb = byteLength64(index);
        for (uint32_t b = 0; b < nb; ++b) result ^= t[b * 256u + (uint32_t) ((index >> (8u * b)) & 255ull)];
        const float v = (float) result * (1.0f / 4294967296.0f);
        return 0.99999994f < v ? 0.99999994f : v;
    }
    const uint32_t *row = T.matrices + dimension * 52u;
    const uint32_t n = bitLength64(index);
#pragma unroll 8
    for (uint32_t i = 0; i < n; ++i) { const uint32_t v = row[i]; result ^= ((index >> i) & 1ull) ? v : 0u; }
    const float v = (float) result * (1.0f / 4294967296.0f);
    return 0.99999994f < v ? 0.99999994f : v;            /* std::min(result * 2^-32, ONE_MINUS_EPS_FLT) */
}
/* two consecutive dimensions / two pairs of them of ONE point in one pass over the index bits: the reads of all rows are in flight together
   (a vertex draws its emitter and its BSDF sample from the same point: four chains of round trips become one) */
DV void sobolSample2(const SobolTab &T, uint64_t index, uint32_t dimension, float &a, float &b) {
    uint32_t r0 = T.scramble, r1 = T.scramble;
    if (!SOBOL_ROW_LOOPS || T.matBt) {
        const uint32_t *t = T.matBt + (size_t) dimension * (SOBOL_BT_BYTES * 256u);
        const uint32_t nb = byteLength64(index);
        for (uint32_t i = 0; i < nb; ++i) {
            const uint32_t e = i * 256u + (uint32_t) ((index >> (8u * i)) & 255ull);
            r0 ^= t[e]; r1 ^= t[SOBOL_BT_BYTES * 256u + e];
        }
        const float f0 = (float) r0 * (1.0f / 4294967296.0f), f1 = (float) r1 * (1.0f / 4294967296.0f);
        a = 0.99999994f < f0 ? 0.99999994f : f0; b = 0.99999994f < f1 ? 0.99999994f : f1;
        return;
    }
    const uint32_t *row = T.matrices + dimension * 52u;
    const uint32_t n = bitLength64(index);
#pragma unroll 8
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t v0 = row[i], v1 = row[52u + i];
        const bool bit = ((index >> i) & 1ull) != 0;
        r0 ^= bit ? v0 : 0u; r1 ^= bit ? v1 : 0u;
    }
    const float f0 = (float) r0 * (1.0f / 4294967296.0f), f1 = (float) r1 * (1.0f / 4294967296.0f);
    a = 0.99999994f < f0 ? 0.99999994f : f0; b = 0.99999994f < f1 ? 0.99999994f : f1;
}
DV void sobolSample2x2(const SobolTab &T, uint64_t index, uint32_t dimA, uint32_t dimB, float out[4]) {
    uint32_t r0 = T.scramble, r1 = T.scramble, r2 = T.scramble, r3 = T.scramble;
    if (!SOBOL_ROW_LOOPS || T.matBt) {
        const uint32_t *ta = T.matBt + (size_t) dimA * (SOBOL_BT_BYTES * 256u), *tb = T.matBt + (size_t) dimB * (SOBOL_BT_BYTES * 256u);
        const uint32_t nb = byteLength64(index);
        for (uint32_t i = 0; i < nb; ++i) {
            const uint32_t e = i * 256u + (uint32_t) ((index >> (8u * i)) & 255ull);
            r0 ^= ta[e]; r1 ^= ta[SOBOL_BT_BYTES * 256u + e]; r2 ^= tb[e]; r3 ^= tb[SOBOL_BT_BYTES * 256u + e];
        }
        const uint32_t r[4] = { r0, r1, r2, r3 };
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float f = (float) r[k] * (1.0f / 4294967296.0f); out[k] = 0.99999994f < f ? 0.99999994f : f; }
        return;
    }
    const uint32_t *rowA = T.matrices + dimA * 52u, *rowB = T.matrices + dimB * 52u;
    const uint32_t n = bitLength64(index);
#pragma unroll 4                                                 /* (16 reads in flight; 32 put the QMC build of k_mega 100 B into scratch) */
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t v0 = rowA[i], v1 = rowA[52u + i], v2 = rowB[i], v3 = rowB[52u + i];
        const bool bit = ((index >> i) & 1ull) != 0;
        r0 ^= bit ? v0 : 0u; r1 ^= bit ? v1 : 0u; r2 ^= bit ? v2 : 0u; r3 ^= bit ? v3 : 0u;
    }
    const uint32_t r[4] = { r0, r1, r2, r3 };
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float f = (float) r[k] * (1.0f / 4294967296.0f); out[k] = 0.99999994f < f ? 0.99999994f : f; }
}
/* the camera sample: SobolSampler::next2D at dimension 0 (sobol.cpp:244-247) -> the jitter renderBlock adds to the pixel (integrator.cpp:171) */
DV void sobolCameraSample(const SobolTab &T, uint32_t sampleIndex, uint32_t px, uint32_t py, float &jx, float &jy) {
    const uint64_t idx = sobolSampleIndex(T, sampleIndex, px, py);
    if (idx != (uint64_t) sampleIndex) {
        float a, b; sobolSample2(T, idx, 0u, a, b);
        jx = a * T.resolution - (float) (int) px;
        jy = b * T.resolution - (float) (int) py;
    } else sobolSample2(T, idx, 0u, jx, jy);
}

/* ---- PHIP_SAMPLER_HALTON / _HAMMERSLEY: the reference's radical-inverse samplers as they stand (src/samplers/halton.cpp, hammersley.cpp).  The
 *      primes and the digit permutations (PermutationStorage, src/samplers/faure.cpp) are DATA handed through the ABI (phip_render_params.qmc_*);
 *      the partition of the sequence over the pixels (setFilmResolution) is computed by the host (phip.hip: setupRadicalInverse). ---- */
#define RINV_MAX_RESOLUTION 128u        /* halton.cpp:29, hammersley.cpp:29: pixel positions enter modulo this */
struct RinvTab {
    const uint32_t *primes;             /* primeTable[0 .. dims) */
    const uint16_t *perm;               /* the permutations of those bases, concatenated; NULL: no scrambling */
    const uint32_t *permOffset;         /* start of the permutation of dimension d */
    uint32_t dims;
    uint32_t hammersley;                /* 0: halton, 1: hammersley */
    uint32_t stride;                    /* halton: 2^expX 3^expY (<= 128 * 243); hammersley: resY */
    uint32_t powX, powY, expX, expY;    /* halton: m_primePowers, m_primeExponents; hammersley: resX, resY, -, log2 resY */
    uint32_t multInvX, multInvY;        /* halton: m_multInverse */
    uint32_t invPerm2, invPerm3;        /* inverse permutations of bases 2 and 3, two bits per digit (identity without scrambling) */
    uint32_t sampleCount;               /* hammersley: m_sampleCount (of the whole render) */
    float factor;                       /* hammersley: m_factor */
    /* Multi-digit tables (round 5; NULL: the digit-by-digit loops).  The radical inverse is a chain of DEPENDENT steps, one per digit of the index
       (23 for base 2 on C2), each a division by a run-time prime; the Faure permutation is per base, so the permuted value of a chunk of k digits is
       ONE look-up (base 2: 10 digits per chunk, 3: 6, 5: 4, 7: 3, 11..31: 2, larger: 1), and the division by the chunk size b^k a multiply-high with a
       correction.  Built by the host for the first RINV_TAB_DIMS dimensions (phip.hip: buildRinvTables):
         dimInfo[8 d ..]  base, B = base^k, k | first chunk entry << 8, floor(2^32 / B) + 1, bits(1 / base), bits(radical * perm[0] / (1 - radical)), -, -
         chunk[..]        per value c of a chunk: V_full (the k digits of c, least significant first, as value = value * base + perm[digit]) |
                          V_sig << 10 (the same over the SIGNIFICANT digits of c only: the top chunk of an index) | number of significant digits << 20
         fac[34 d + m]    radical^m as the loop computes it (m sequential float multiplications), pw[11 d + n] = base^n */
    const uint32_t *dimInfo, *chunk, *pw; const float *fac; uint32_t tabDims;
};
#define RINV_TAB_DIMS 128u
#define RINV_FAC_STRIDE 34u
#define RINV_PW_STRIDE 11u
/* inverseScrambledRadicalInverse, halton.cpp:198-211 / hammersley.cpp:165-178 */
DV uint32_t rinvInverse(uint32_t base, uint32_t inverse, uint32_t digits, uint32_t invPerm) {
    uint32_t index = 0;
    for (; digits; --digits) {
        const uint32_t digit = (invPerm >> (2u * (inverse % base))) & 3u;
        inverse /= base;
        index = index * base + digit;
    }
    return index;
}
/* the index of sample k of pixel (px, py): generate() + next*D, halton.cpp:272-296,352-372 / hammersley.cpp:206-222,252-268 */
DV uint64_t rinvSampleIndex(const RinvTab &T, uint32_t k, uint32_t px, uint32_t py) {
    const uint32_t x = px % RINV_MAX_RESOLUTION, y = py % RINV_MAX_RESOLUTION;
    uint32_t offset = 0;
    if (T.stride > 1u) {
        if (T.hammersley)
            offset = x * T.powY * T.sampleCount + rinvInverse(2u, y, T.expY, T.invPerm2);
        else {
            /* (each term is below 128 * 243 * 243 < 2^23: the 64-bit arithmetic of the reference fits 32 bits) */
            offset = rinvInverse(2u, x, T.expX, T.invPerm2) * (T.stride / T.powX) * T.multInvX
                   + rinvInverse(3u, y, T.expY, T.invPerm3) * (T.stride / T.powY) * T.multInvY;
            offset %= T.stride;
        }
    }
    return (uint64_t) offset + (uint64_t) T.stride * (uint64_t) k;
}
/* radicalInverseFast / scrambledRadicalInverseFast, qmc.cpp:141-166,169-1198,1201-2230 (the macros RINV / SCRAMBLED_RINV: integer digits, one
   float factor): bit for bit */
/* floor(n / B) and the remainder by M = floor(2^32 / B) + 1: n M / 2^32 = n / B + n d / 2^32 with 0 < d <= 1, so the multiply-high is the quotient or one
   more; one more shows as a remainder that wrapped below zero (>= 2^32 - B >= B in unsigned arithmetic, whether or not q B itself overflowed) */
DV void rinvDivMod(uint32_t n, uint32_t B, uint32_t M, uint32_t &q, uint32_t &c) {
    q = (uint32_t) (((uint64_t) n * (uint64_t) M) >> 32);
    c = n - q * B;
    if (c >= B) { q -= 1u; c += B; }
}
/* (Measured and not kept, profiles/r05_gpu_call_g_*: a straight-line form that derives up to four chunks and the top chunk's digit count by integer
   arithmetic alone and requests all look-ups together -- two memory round trips per number instead of one per chunk: k_mega with `halton` 1.600 x the counter
   stream's time against 1.585 x for this loop; the extra divisions cost what the parallel loads save.) */
DV float rinvRadicalInverseTab(const RinvTab &T, uint32_t baseIndex, uint32_t i32) {
    const uint32_t *di = T.dimInfo + 8u * baseIndex;
    const uint32_t B = di[1], k = di[2] & 0xffu, M = di[3];
    const float tail = pm_from_bits(di[5]);                 /* radical * perm[0] / (1 - radical) (0 without scrambling) */
    const uint32_t *tab = T.chunk + (di[2] >> 8);
    uint64_t value = 0;
    uint32_t digits = 0;
    while (i32 >= B) {                                      /* a full chunk: k digits, leading zeros included */
        uint32_t q, c; rinvDivMod(i32, B, M, q, c);
        value = value * B + (uint64_t) (tab[c] & 0x3ffu);
        digits += k; i32 = q;
    }
    const uint32_t e = tab[i32], n = e >> 20;               /* the top chunk: its significant digits only (the loops stop when nothing is left) */
    value = value * T.pw[RINV_PW_STRIDE * baseIndex + n] + (uint64_t) ((e >> 10) & 0x3ffu);
    digits += n;
    const float factor = T.fac[RINV_FAC_STRIDE * baseIndex + digits];
    const float inverse = T.perm ? factor * ((float) value + tail) : (float) value * factor;
    return 0.99999994f < inverse ? 0.99999994f : inverse;
}
DV float rinvRadicalInverse(const RinvTab &T, uint32_t baseIndex, uint64_t index) {
    if (T.dimInfo && baseIndex < T.tabDims && !(index >> 32)) return rinvRadicalInverseTab(T, baseIndex, (uint32_t) index);
    const uint32_t base = T.primes[baseIndex];
    const float radical = 1.0f / (float) (int) base;
    const uint16_t *perm = T.perm ? T.perm + T.permOffset[baseIndex] : nullptr;
    uint64_t value = 0;
    float factor = 1.0f;
    if (index >> 32) {                                       /* (only with more than 2^32 / stride samples per pixel) */
        while (index >> 32) {
            const uint64_t next = index / base, digit = index - next * base;
            value = value * base + (perm ? (uint64_t) perm[digit] : digit);
            factor *= radical;
            index = next;
        }
    }
    uint32_t i32 = (uint32_t) index;
    while (i32) {
        const uint32_t next = i32 / base, digit = i32 - next * base;
        value = value * base + (perm ? (uint64_t) perm[digit] : (uint64_t) digit);
        factor *= radical;
        i32 = next;
    }
    float inverse;
    if (perm) inverse = factor * ((float) value + radical * (float) (int) perm[0] / (1.0f - radical));
    else inverse = (float) value * factor;
    return 0.99999994f < inverse ? 0.99999994f : inverse;    /* std::min(inverse, ONE_MINUS_EPS) */
}
/* Two consecutive dimensions of ONE point (a 2D request) in one interleaved pass: the chunk loops of the two bases advance together, so their look-ups -- two
   descriptors, then two table entries per level, then the two factors -- are in flight at the same time instead of one behind the other (the radical inverse
   is a chain of dependent L1 / L2 round trips at four waves per SIMD: latency, not bandwidth).  The same arithmetic per base as rinvRadicalInverseTab. */
DV void rinvRadicalInverseTab2(const RinvTab &T, uint32_t b0, uint32_t i32, float &r0, float &r1) {
    const uint32_t *di0 = T.dimInfo + 8u * b0, *di1 = di0 + 8u;
    const uint32_t B0 = di0[1], k0 = di0[2] & 0xffu, M0 = di0[3], B1 = di1[1], k1 = di1[2] & 0xffu, M1 = di1[3];
    const float tail0 = pm_from_bits(di0[5]), tail1 = pm_from_bits(di1[5]);
    const uint32_t *tab0 = T.chunk + (di0[2] >> 8), *tab1 = T.chunk + (di1[2] >> 8);
    uint64_t v0 = 0, v1 = 0;
    uint32_t dg0 = 0, dg1 = 0, i0 = i32, i1 = i32;
    while (i0 >= B0 || i1 >= B1) {                          /* a full chunk of whichever base still has one (the other keeps its state: selects, no branch) */
        const bool a0 = i0 >= B0, a1 = i1 >= B1;
        uint32_t q0, c0, q1, c1;
        rinvDivMod(i0, B0, M0, q0, c0); rinvDivMod(i1, B1, M1, q1, c1);
        const uint32_t e0 = tab0[a0 ? c0 : 0u], e1 = tab1[a1 ? c1 : 0u];
        v0 = a0 ? v0 * B0 + (uint64_t) (e0 & 0x3ffu) : v0; dg0 += a0 ? k0 : 0u; i0 = a0 ? q0 : i0;
        v1 = a1 ? v1 * B1 + (uint64_t) (e1 & 0x3ffu) : v1; dg1 += a1 ? k1 : 0u; i1 = a1 ? q1 : i1;
    }
    const uint32_t e0 = tab0[i0], e1 = tab1[i1], n0 = e0 >> 20, n1 = e1 >> 20;
    const uint32_t p0 = T.pw[RINV_PW_STRIDE * b0 + n0], p1 = T.pw[RINV_PW_STRIDE * (b0 + 1u) + n1];
    const float f0 = T.fac[RINV_FAC_STRIDE * b0 + dg0 + n0], f1 = T.fac[RINV_FAC_STRIDE * (b0 + 1u) + dg1 + n1];
    v0 = v0 * p0 + (uint64_t) ((e0 >> 10) & 0x3ffu); v1 = v1 * p1 + (uint64_t) ((e1 >> 10) & 0x3ffu);
    const float x0 = T.perm ? f0 * ((float) v0 + tail0) : (float) v0 * f0, x1 = T.perm ? f1 * ((float) v1 + tail1) : (float) v1 * f1;
    r0 = 0.99999994f < x0 ? 0.99999994f : x0; r1 = 0.99999994f < x1 ? 0.99999994f : x1;
}
/* dimension `dimension` of point `index`: nextFloat, halton.cpp:343-350 / hammersley.cpp:235-243 */
DV float rinvSample(const RinvTab &T, uint64_t index, uint32_t dimension) {
    if (T.hammersley) {
        if (dimension == 0u) return (float) index * T.factor;
        return rinvRadicalInverse(T, dimension - 1u, index);
    }
    return rinvRadicalInverse(T, dimension, index);
}
/* dimensions `dimension`, `dimension` + 1 of point `index` (a 2D request behind the camera sample: dimension >= 2) */
DV void rinvSample2(const RinvTab &T, uint64_t index, uint32_t dimension, float &a, float &b) {
    const uint32_t b0 = dimension - T.hammersley;           /* hammersley: its dimension d > 0 is the radical inverse in prime d - 1 */
    if (T.dimInfo && dimension >= 1u && b0 + 1u < T.tabDims && !(index >> 32)) { rinvRadicalInverseTab2(T, b0, (uint32_t) index, a, b); return; }
    a = rinvSample(T, index, dimension); b = rinvSample(T, index, dimension + 1u);
}
/* the camera sample: next2D at dimension 0, halton.cpp:375-378 / hammersley.cpp:271-274 */
DV void rinvCameraSample(const RinvTab &T, uint32_t sampleIndex, uint32_t px, uint32_t py, float &jx, float &jy) {
    const uint64_t idx = rinvSampleIndex(T, sampleIndex, px, py);
    jx = rinvSample(T, idx, 0u) * (float) (int) T.powX - (float) (int) (px % RINV_MAX_RESOLUTION);
    jy = rinvSample(T, idx, 1u) * (float) (int) T.powY - (float) (int) (py % RINV_MAX_RESOLUTION);
}

/* ---- PHIP_SAMPLER_STRATIFIED: the construction of `stratified` (src/samplers/stratified.cpp:147-200) made addressable: the cell a sample
 *      visits in dimension `dim` (2D request q: 2 q, 1D request j: 2 j + 1) is a keyed permutation of its index (stratified.cpp:147-158 shuffles with
 *      the worker's Random), the jitter inside the cell is the counter stream's number for that request. ---- */
#define ST_DIMENSIONS 4u                /* `dimension` default: that many 2D and 1D requests per sample are stratified (stratified.cpp:79) */
DV uint32_t stCell(uint32_t pixel, uint32_t k, uint32_t dim, uint32_t seed, uint32_t count) {
    const U4 h = pcg4d(pixel, dim, 0x5354u /* 'ST' */, seed);
    return ldPermuteAny(k % count, count, h.x);
}
DV void stPoint2D(uint32_t pixel, uint32_t k, uint32_t q, uint32_t seed, uint32_t res, float u1, float u2, float &x, float &y) {
    const uint32_t c = stCell(pixel, k, 2u * q, seed, res * res);
    const float inv = 1.0f / (float) (int) res;
    x = ((float) (int) (c % res) + u1) * inv; y = ((float) (int) (c / res) + u2) * inv;       /* stratified.cpp:181-189 */
}
/* `direct` with more than one sample of a kind: the requested 2D array a (direct.cpp:139-146) is a Latin hypercube over ALL its sampleCount * count entries
   (stratified.cpp:160-164 -> latinHypercube, src/libcore/qmc.cpp: entry i of a dimension starts as (i + xi) / N, then each dimension is shuffled on its own): entry
   e = k * count + i holds the stratum a keyed permutation of e gives it, per dimension, jittered by the counter stream's numbers for that entry */
DV void stArrayPoint(uint32_t pixel, uint32_t a, uint32_t e, uint32_t total, uint32_t seed, float u1, float u2, float &x, float &y) {
    const U4 h = pcg4d(pixel, 0x200u + a, 0x5354u /* 'ST' */, seed);
    const float delta = 1.0f / (float) (size_t) total;
    x = ((float) (int) ldPermuteAny(e, total, h.x) + u1) * delta;
    y = ((float) (int) ldPermuteAny(e, total, h.y) + u2) * delta;
}
DV float stPoint1D(uint32_t pixel, uint32_t k, uint32_t j, uint32_t seed, uint32_t res, float u) {
    const uint32_t c = stCell(pixel, k, 2u * j + 1u, seed, res * res);
    return ((float) (int) c + u) * (1.0f / (float) (size_t) (res * res));                      /* stratified.cpp:170-173 */
}

} // namespace pt
