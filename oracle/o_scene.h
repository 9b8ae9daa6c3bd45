/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see o_math.h).
 *
 * o_scene.h: scene container, intersection record, emitter sampling, sample streams.  Restates
 * (file:line under /root/reference):
 *   include/mitsuba/core/pmf.h:33-200            DiscreteDistribution (append/normalize/sample/sampleReuse)
 *   src/librender/scene.cpp:375-381,828-852,949-952  emitter PDF, sampleEmitterDirect, pdfEmitterDirect
 *   include/mitsuba/render/scene.h:848-850       pdfEmitterDiscrete
 *   src/emitters/area.cpp:104-109,158-182        AreaLight::eval / sampleDirect / pdfDirect
 *   src/librender/shape.cpp:102-126              Shape::sampleDirect / pdfDirect
 *   src/librender/trimesh.cpp:358-360,388-423    pdfPosition, prepareSamplingTable, samplePosition
 *   src/libcore/triangle.cpp:24-67               Triangle::sample / surfaceArea
 *   include/mitsuba/render/skdtree.h:343-428     fillIntersectionRecord<true>
 *   include/mitsuba/render/records.inl:160-178   DirectSamplingRecord(its) / setQuery
 *   src/samplers/independent.cpp:71-103, src/librender/renderjob.cpp:58-69   per-worker SFMT streams
 */
#pragma once
#include <functional>
#include <cstdio>
#include "o_kdtree.h"
#include "o_sfmt.h"
#include "o_envmap.h"
#include "../include/phip.h"
#include <string>
#include <stdexcept>

namespace orc {

/* pmf.h */
struct DiscreteDistribution {
    std::vector<Float> cdf;
    Float sum = 0, normalization = 0;
    bool normalized = false;
    DiscreteDistribution() { cdf.push_back(0.0f); }
    void append(Float v) { cdf.push_back(cdf[cdf.size() - 1] + v); }
    size_t size() const { return cdf.size() - 1; }
    Float operator[](size_t e) const { return cdf[e + 1] - cdf[e]; }
    Float normalize() {
        sum = cdf[cdf.size() - 1];
        if (sum > 0) {
            normalization = 1.0f / sum;
            for (size_t i = 1; i < cdf.size(); ++i) cdf[i] *= normalization;
            cdf[cdf.size() - 1] = 1.0f;
            normalized = true;
        } else {
            normalization = 0.0f;
        }
        return sum;
    }
    size_t sample(Float sampleValue) const {
        std::vector<Float>::const_iterator entry = std::lower_bound(cdf.begin(), cdf.end(), sampleValue);
        size_t index = std::min(cdf.size() - 2, (size_t) std::max((ptrdiff_t) 0, entry - cdf.begin() - 1));
        /* pmf.h:131-134 evaluates `operator[](index) == 0 && index < size-1`, which reads one entry
           past the end when a zero-probability tail is reached; the bounds check comes first here */
        while (index < cdf.size() - 2 && operator[](index) == 0)
            ++index;
        return index;
    }
    size_t sample(Float sampleValue, Float &pdf) const { size_t i = sample(sampleValue); pdf = operator[](i); return i; }
    size_t sampleReuse(Float &sampleValue) const {
        size_t index = sample(sampleValue);
        sampleValue = (sampleValue - cdf[index]) / (cdf[index + 1] - cdf[index]);
        return index;
    }
    size_t sampleReuse(Float &sampleValue, Float &pdf) const {
        size_t index = sample(sampleValue, pdf);
        sampleValue = (sampleValue - cdf[index]) / (cdf[index + 1] - cdf[index]);
        return index;
    }
};

/* BSDF type bits used by the path (bsdf.h:224-283) */
enum { EDiffuseReflection = 0x8, EGlossyReflection = 0x10, EDeltaReflection = 0x100, EDeltaTransmission = 0x400,
       ESmoothMask = 0x8 | 0x10 | 0x20 | 0x40 | 0x4, EDeltaMask = 0x100 | 0x400 | 0x1,
       EFrontSide = 0x20000, EBackSide = 0x40000 };
/* (numeric values are local to the oracle; only the predicates matter) */

struct Material {
    phip_material m;
    /* derived at configure time */
    bool smooth;          /* getType() & ESmooth                           */
    bool transOrBack;     /* getType() & (ETransmission | EBackSide)       */
    Float alphaU, alphaV; /* roughconductor, after .average() and clamp    */
};

struct Shape {
    phip_shape s;
    DiscreteDistribution areaDistr;
    Float surfaceArea = 0, invSurfaceArea = 0;
};

struct Intersection {
    Float t;
    Vec3 p;
    Frame geoFrame, shFrame;
    Vec2 uv;
    Vec3 dpdu, dpdv;
    bool hasUVPartials = false;
    Float dudx = 0, dudy = 0, dvdx = 0, dvdy = 0;
    Vec3 wi;
    int shape;            /* -1 = invalid */
    uint32_t primIndex;   /* global triangle id */
    bool isValid() const { return shape >= 0; }
    Vec3 toWorld(const Vec3 &v) const { return shFrame.toWorld(v); }
    Vec3 toLocal(const Vec3 &v) const { return shFrame.toLocal(v); }
};

enum EMeasure { EInvalidMeasure = 0, ESolidAngle = 1, ELength = 2, EArea = 3, EDiscrete = 4 };

struct DirectSamplingRecord {
    Vec3 p, n; Vec2 uv; Float pdf; int measure; int emitter;
    Vec3 ref, refN, d; Float dist;
};

/* Sample source: the reference consumes one sequential stream per worker (`sfmt` mode); the
 * parity stream (`ctr`) is addressed by (pixel, sample, dimension block) so CPU and GPU see the
 * same numbers regardless of scheduling.  Block layout (shared with the HIP kernels; HISTORY.md 3.5):
 *   block 0              : (jitter.x, jitter.y, -, -)
 *   block 1 + 2*(k>>1)   : pair k & 1 = the k-th 2D request after the jitter (= (emitter.xy, bsdf.xy) of depth k/2+1 without dielectrics)
 *   block 2 + 2*(d-1)    : (rr, -, -, -)
 * word -> float exactly like Random::nextFloat (random.cpp:632-641): (u >> 9 | 0x3f800000) - 1.
 */
inline void pcg4d(uint32_t v[4]) {
    /* Jarzynski & Olano, "Hash Functions for GPU Rendering", JCGT 9(3) 2020, pcg4d */
    for (int i = 0; i < 4; ++i) v[i] = v[i] * 1664525u + 1013904223u;
    v[0] += v[1] * v[3]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1]; v[3] += v[1] * v[2];
    for (int i = 0; i < 4; ++i) v[i] ^= v[i] >> 16;
    v[0] += v[1] * v[3]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1]; v[3] += v[1] * v[2];
}
inline float u32ToFloat(uint32_t u) {
    uint32_t b = (u >> 9) | 0x3f800000u; float f; memcpy(&f, &b, 4); return f - 1.0f;
}

/* ---- PHIP_SAMPLER_LD (HISTORY.md 3.5): ldsampler.cpp's construction on the counter-based generator.  Points: core/qmc.h:43-59,82-87 ---- */
inline float radicalInverse2Single(uint32_t n, uint32_t scramble) {
    n = __builtin_bswap32(n);
    n = ((n & 0x0f0f0f0f) << 4) | ((n & 0xf0f0f0f0) >> 4);
    n = ((n & 0x33333333) << 2) | ((n & 0xcccccccc) >> 2);
    n = ((n & 0x55555555) << 1) | ((n & 0xaaaaaaaa) >> 1);
    n = (n >> (32 - 24)) ^ (scramble & ~-(1 << 24));
    return (float) n / (float) (1U << 24);
}
inline float sobol2Single(uint32_t n, uint32_t scramble) {
    for (uint32_t v = 1U << 31; n != 0; n >>= 1, v ^= v >> 1)
        if (n & 1) scramble ^= v;
    return (float) scramble / (float) (1ULL << 32);
}
/* the keyed bijection of [0, mask] that stands in for Random::shuffle (every step is invertible modulo mask + 1) */
inline uint32_t ldPermute(uint32_t i, uint32_t mask, uint32_t key) {
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
inline uint32_t ldPermuteAny(uint32_t i, uint32_t l, uint32_t key) {     /* ... for a domain of any size: cycle walking (Kensler, ibid.) */
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
static const uint32_t LD_DIMENSIONS = 4;     /* ldsampler.cpp:79 */

/* ---- PHIP_SAMPLER_SOBOL: the reference's `sobol` sampler (src/samplers/sobol.cpp), SINGLE_PRECISION.  The direction numbers are the
 *      plugin's own tables, handed in as data (phip_render_params.sobol_*). ---- */
struct SobolTables {
    const uint32_t *matrices = nullptr;              /* sobol::Matrices::matrices32 (sobolseq.h:31) */
    const uint64_t *vdc = nullptr, *vdcInv = nullptr; /* rows [m - 1] of vdc_sobol_matrices / _inv (sobolseq.h:33-34) */
    uint32_t dims = 0, logRes = 0; uint64_t scramble = 0;
    float resolution = 1.0f;
    uint64_t lookUp(uint32_t frame, uint32_t px, uint32_t py) const {          /* sobol::look_up, sobolseq.h:94-130 */
        const uint32_t m = logRes, m2 = m << 1;
        uint64_t index = uint64_t(frame) << m2;
        uint64_t delta = 0;
        for (uint32_t c = 0; frame; frame >>= 1, ++c)
            if (frame & 1) delta ^= vdc[c];
        const uint64_t scr = (scramble & 0xFFFFFFFFull) >> (32 - m);
        uint64_t b = (((uint64_t) (px ^ scr) << m) | (py ^ scr)) ^ delta;
        for (uint32_t c = 0; b; b >>= 1, ++c)
            if (b & 1) index ^= vdcInv[c];
        return index;
    }
    uint64_t sampleIndex(uint32_t sample, uint32_t px, uint32_t py) const {    /* SobolSampler::setSampleIndex, sobol.cpp:207-219 */
        return logRes > 1 ? lookUp(sample, px, py) : (uint64_t) sample;
    }
    float sample(uint64_t index, uint32_t dimension) const {                   /* sobol::sampleSingle, sobolseq.h:42-58 */
        uint32_t result = (uint32_t) scramble;
        for (uint32_t i = dimension * 52; index; index >>= 1, ++i)
            if (index & 1) result ^= matrices[i];
        return std::min(result * (1.0f / (1ULL << 32)), 0.999999940395355225f /* ONE_MINUS_EPS_FLT */);
    }
};
/* ---- PHIP_SAMPLER_HALTON / _HAMMERSLEY: the reference's radical-inverse samplers (src/samplers/halton.cpp, hammersley.cpp), SINGLE_PRECISION.
 *      Primes and digit permutations are the reference's own tables, handed in as data (phip_render_params.qmc_*). ---- */
struct RinvTables {
    const uint32_t *primes = nullptr; const uint16_t *perm = nullptr; uint32_t dims = 0;
    std::vector<size_t> permOffset;                  /* start of the permutation of dimension d: primes[0] + ... + primes[d - 1] */
    bool hammersley = false;
    uint64_t stride = 1, multInverse[2] = { 0, 0 };  /* m_stride, m_multInverse */
    int primePowers[2] = { 1, 1 }, primeExponents[2] = { 0, 0 };   /* halton; hammersley: m_resolution, (-, m_logHeight) */
    size_t sampleCount = 1; float factor = 1.0f;     /* hammersley: m_sampleCount, m_factor */
    static const int MAX_RESOLUTION = 128;           /* halton.cpp:29, hammersley.cpp:29 */
    /* getPermutation(d) / getInversePermutation(d) of PermutationStorage (faure.h:44-52; the inverse by inversion, faure.cpp) */
    const uint16_t *permutation(uint32_t d) const { return perm ? perm + permOffset[d] : nullptr; }
    uint64_t inverseDigit(uint32_t d, uint64_t digit) const {
        if (!perm) return digit;
        const uint16_t *pm = permutation(d);
        for (uint32_t i = 0; i < primes[d]; ++i) if (pm[i] == digit) return i;
        return digit;
    }
    static void extendedGCD(uint64_t a, uint64_t b, int64_t &x, int64_t &y) {           /* halton.cpp:219-229 */
        if (b == 0) { x = 1; y = 0; return; }
        int64_t d = (int64_t) (a / b), x_, y_;
        extendedGCD(b, a % b, x_, y_);
        x = y_; y = x_ - d * y_;
    }
    static uint64_t multiplicativeInverse(int64_t a, int64_t n) {                       /* halton.cpp:236-241; math::modulo: the non-negative remainder */
        int64_t x, y; extendedGCD((uint64_t) a, (uint64_t) n, x, y);
        int64_t r = x % n; return (uint64_t) (r < 0 ? r + n : r);
    }
    void setFilmResolution(int resX, int resY, size_t sampleCount_) {                   /* blocked = true: halton.cpp:244-266, hammersley.cpp:181-196 */
        const int res[2] = { resX, resY };
        sampleCount = sampleCount_;
        if (!hammersley) {
            stride = 1;
            for (int i = 0; i < 2; ++i) {
                int prime = (int) primes[i], value = 1, exp = 0;
                while (value < std::min(res[i], MAX_RESOLUTION)) { value *= prime; ++exp; }
                primePowers[i] = value; primeExponents[i] = exp; stride *= (uint64_t) value;
            }
            multInverse[0] = multiplicativeInverse(primePowers[1], primePowers[0]);
            multInverse[1] = multiplicativeInverse(primePowers[0], primePowers[1]);
        } else {
            for (int i = 0; i < 2; ++i) { uint32_t v = 1; while (v < (uint32_t) res[i]) v <<= 1; primePowers[i] = (int) std::min((uint32_t) MAX_RESOLUTION, v); }
            primeExponents[0] = 0; primeExponents[1] = 0; while ((1 << primeExponents[1]) < primePowers[1]) ++primeExponents[1];
            factor = (float) 1.0f / (sampleCount * (size_t) primePowers[0] * (size_t) primePowers[1]);
            stride = (uint64_t) primePowers[1];
        }
    }
    uint64_t inverseScrambledRadicalInverse(uint32_t d, uint64_t inverse, uint64_t digits) const {     /* halton.cpp:198-211 */
        const uint64_t base = primes[d];
        uint64_t index = 0;
        while (digits) {
            uint64_t digit = inverseDigit(d, inverse % base);
            inverse /= base;
            index = index * base + digit;
            --digits;
        }
        return index;
    }
    uint64_t sampleIndex(uint32_t sample, uint32_t px, uint32_t py) const {             /* generate() + next*D: halton.cpp:272-296,372 / hammersley.cpp:206-222,268 */
        const uint64_t pos[2] = { px % (uint32_t) MAX_RESOLUTION, py % (uint32_t) MAX_RESOLUTION };
        uint64_t offset = 0;
        if (stride > 1) {
            if (hammersley)
                offset = pos[0] * (uint64_t) primePowers[1] * sampleCount + inverseScrambledRadicalInverse(0, pos[1], (uint64_t) primeExponents[1]);
            else {
                for (int i = 0; i < 2; ++i)
                    offset += inverseScrambledRadicalInverse((uint32_t) i, pos[i], (uint64_t) primeExponents[i]) * (stride / (uint64_t) primePowers[i]) * multInverse[i];
                offset %= stride;
            }
        }
        return offset + stride * (uint64_t) sample;
    }
    float radicalInverse(uint32_t baseIndex, uint64_t index) const {                    /* RINV / SCRAMBLED_RINV, qmc.cpp:141-166 */
        const int base = (int) primes[baseIndex];
        const uint16_t *pm = permutation(baseIndex);
        const float radical = (float) 1 / (float) base;
        uint64_t value = 0;
        float factor_ = 1.0f;
        while (index) {
            uint64_t next = index / (uint64_t) base;
            uint64_t digit = index - next * (uint64_t) base;
            value = value * (uint64_t) base + (pm ? (uint64_t) pm[digit] : digit);
            factor_ *= radical;
            index = next;
        }
        float inverse = pm ? factor_ * ((float) value + radical * pm[0] / (1 - radical)) : (float) value * factor_;
        return std::min(inverse, 0.999999940395355225f /* ONE_MINUS_EPS_FLT */);
    }
    float sample(uint64_t index, uint32_t dimension) const {                            /* nextFloat, halton.cpp:343-350 / hammersley.cpp:235-243 */
        if (hammersley) return dimension == 0 ? index * factor : radicalInverse(dimension - 1, index);
        return radicalInverse(dimension, index);
    }
    uint32_t dimensions() const { return dims + (hammersley ? 1u : 0u); }
};
static const uint32_t ST_DIMENSIONS = 4;     /* stratified.cpp:79 */

struct SampleSource {
    /* ctr mode */
    bool ctr = true;
    uint32_t pixel = 0, sample = 0, seed = 0;
    /* ld mode (on top of ctr): the first LD_DIMENSIONS 2D requests (the pixel jitter is request 0) and 1D requests of a sample */
    bool ld = false; uint32_t ldMask = 0; int rrDepth = 5;
    Vec2 ldPoint(uint32_t dim) const {
        uint32_t h[4] = { pixel, dim, 0x4c44u, seed };
        pcg4d(h);
        const uint32_t i = ldPermute(sample & ldMask, ldMask, h[0]);
        return Vec2(radicalInverse2Single(i, h[1]), sobol2Single(i, h[2]));
    }
    /* sobol mode (the reference's stream as it stands) / stratified mode (its construction on the counter stream, as ld): PHIP_SAMPLER_SOBOL / _STRATIFIED */
    int qmc = 0;                             /* 0: off, 1: sobol, 2: stratified, 3: halton / hammersley (sequence samplers: 1 and 3) */
    const RinvTables *rinv = nullptr;
    bool sequence() const { return qmc == 1 || qmc == 3; }
    uint64_t seqIndex() const { return qmc == 1 ? sobol->sampleIndex(sample, px, py) : rinv->sampleIndex(sample, px, py); }
    float seqSample(uint64_t idx, uint32_t dim) const { return qmc == 1 ? sobol->sample(idx, dim) : rinv->sample(idx, dim); }
    uint32_t seqDims() const { return qmc == 1 ? sobol->dims : rinv->dimensions(); }
    const SobolTables *sobol = nullptr; uint32_t px = 0, py = 0;             /* pixel position SobolSampler::generate received */
    uint32_t stRes = 1;
    uint32_t stCell(uint32_t dim) const {    /* the cell sample `sample` visits in dimension dim: a keyed permutation (stratified.cpp:147-158 shuffles) */
        uint32_t h[4] = { pixel, dim, 0x5354u, seed };
        pcg4d(h);
        return ldPermuteAny(sample % (stRes * stRes), stRes * stRes, h[0]);
    }
    Vec2 stPoint2D(uint32_t q, Vec2 u) const {           /* StratifiedSampler::next2D, stratified.cpp:177-189 */
        const uint32_t c = stCell(2 * q);
        const int x = (int) (c % stRes), y = (int) (c / stRes);
        const float inv = 1 / (Float) (int) stRes;
        return Vec2((x + u.x) * inv, (y + u.y) * inv);
    }
    Float stPoint1D(uint32_t j, Float u) const {         /* StratifiedSampler::next1D, stratified.cpp:167-175 */
        const int c = (int) stCell(2 * j + 1);
        return (c + u) * (1 / (Float) (size_t) (stRes * stRes));
    }
    Vec2 sobol2D(uint32_t dim, Vec2 fallback) const {
        if (dim + 1 >= seqDims()) return fallback;       /* (the reference stops with an error here, sobol.cpp:243-245) */
        const uint64_t idx = seqIndex();
        return Vec2(seqSample(idx, dim), seqSample(idx, dim + 1));
    }
    /* sfmt mode */
    SFMT *rng = nullptr;

    void block(uint32_t blk, float out[4]) const {
        uint32_t v[4] = { pixel, sample, blk, seed };
        pcg4d(v);
        for (int i = 0; i < 4; ++i) out[i] = u32ToFloat(v[i]);
    }
    /* ctr mode, `path`: the stream is defined by CALL ORDER, like a Sampler is consumed (HISTORY.md 3.5): the k-th 2D request after
       the pixel jitter (k = 0, 1, ...) is pair k & 1 (.xy / .zw) of block 1 + 2 (k >> 1).  A vertex with a smooth BSDF makes two
       requests (emitter sample, BSDF sample), a vertex without (dielectric) one: k = 2 (depth - 1) - ns at the start of vertex
       `depth`, ns = non-smooth vertices so far (modulo 64: six bits of device state). */
    uint32_t ns = 0;
    Vec2 ctrPair(uint32_t k) const { float f[4]; block(1 + 2 * (k >> 1), f); return (k & 1u) ? Vec2(f[2], f[3]) : Vec2(f[0], f[1]); }
    /* the k-th 2D request after the camera sample, made at vertex `depth`.  sobol: dimensions are consumed in call order -- two per 2D request
       (the camera sample took 0 and 1), one per 1D request; before the requests of vertex `depth` there were max(0, depth - rrDepth)
       Russian-roulette requests (one behind every vertex from rrDepth on, path.cpp:276-283) */
    Vec2 pair(uint32_t k, int depth) const {
        if (ld && k + 1 < LD_DIMENSIONS) return ldPoint(2 * (k + 1));
        /* ... and SobolSampler::next2D skips dimension 4 (sobol.cpp:241-242: `m_dimension + 1 >= m_arrayStartDim && m_dimension < m_arrayEndDim` with both = 5
           when no sample array is requested): the third 2D request of a sample starts there (rrDepth >= 2: no 1D request comes earlier), so it
           and everything behind it is shifted by one */
        if (sequence()) return sobol2D(2 * (1 + k) + (uint32_t) std::max(0, depth - rrDepth) + (k >= 1 ? 1u : 0u), ctrPair(k));      /* (halton.cpp:364-366 and hammersley.cpp:257-259 are the same test) */
        if (qmc == 2 && k + 1 < ST_DIMENSIONS) return stPoint2D(k + 1, ctrPair(k));
        return ctrPair(k);
    }
    Vec2 cameraSample() {
        if (!ctr) { Float a = rng->nextFloat(); Float b = rng->nextFloat(); return Vec2(a, b); }
        ns = 0;
        if (ld) return ldPoint(0);
        if (qmc == 1) {                                      /* SobolSampler::next2D at dimension 0, sobol.cpp:246-252 */
            const uint64_t idx = sobol->sampleIndex(sample, px, py);
            if (idx != (uint64_t) sample)
                return Vec2(sobol->sample(idx, 0) * sobol->resolution - (int) px, sobol->sample(idx, 1) * sobol->resolution - (int) py);
            return Vec2(sobol->sample(idx, 0), sobol->sample(idx, 1));
        }
        if (qmc == 3) {                                      /* next2D at dimension 0, halton.cpp:375-378 / hammersley.cpp:271-274 */
            const uint64_t idx = rinv->sampleIndex(sample, px, py);
            const float v1 = rinv->sample(idx, 0) * rinv->primePowers[0] - (int) (px % (uint32_t) RinvTables::MAX_RESOLUTION);
            const float v2 = rinv->sample(idx, 1) * rinv->primePowers[1] - (int) (py % (uint32_t) RinvTables::MAX_RESOLUTION);
            return Vec2(v1, v2);
        }
        float f[4]; block(0, f);
        if (qmc == 2) return stPoint2D(0, Vec2(f[0], f[1]));
        return Vec2(f[0], f[1]);
    }
    Vec2 emitterSample(int depth) {                      /* (only requested at vertices with a smooth BSDF, path.cpp:174-176) */
        if (!ctr) { Float a = rng->nextFloat(); Float b = rng->nextFloat(); return Vec2(a, b); }
        return pair(2 * (uint32_t) (depth - 1) - ns, depth);
    }
    Vec2 bsdfSample(int depth, bool smooth) {
        if (!ctr) { Float a = rng->nextFloat(); Float b = rng->nextFloat(); return Vec2(a, b); }
        const uint32_t k = 2 * (uint32_t) (depth - 1) - ns + (smooth ? 1u : 0u);
        if (!smooth) ns = (ns + 1u) & 63u;
        return pair(k, depth);
    }
    Float rrSample(int depth) {
        if (!ctr) return rng->nextFloat();
        if (ld && (uint32_t) (depth - rrDepth) < LD_DIMENSIONS) return ldPoint(2 * (uint32_t) (depth - rrDepth) + 1).x;   /* the (depth - rrDepth)-th 1D request */
        float f[4]; block(2 + 2 * (uint32_t) (depth - 1), f);
        if (sequence()) {                                    /* SobolSampler::next1D: after 1 + (2 depth - ns) 2D requests and depth - rrDepth 1D requests */
            const uint32_t dim = 2 * (1 + 2 * (uint32_t) depth - ns) + (uint32_t) (depth - rrDepth) + 1u;      /* (+ 1: the skipped dimension 4, see pair()) */
            return dim < seqDims() ? seqSample(seqIndex(), dim) : f[0];
        }
        if (qmc == 2 && (uint32_t) (depth - rrDepth) < ST_DIMENSIONS) return stPoint1D((uint32_t) (depth - rrDepth), f[0]);
        return f[0];
    }

    /* ---- `direct`: sample arrays (which = 0: emitter samples, 1: BSDF samples) ----
       ctr stream: block 1 + i holds (emitter sample i, BSDF sample i).
       sfmt stream: like `independent` -- arrays with more than one entry are drawn for all samples of the pixel by
       Sampler::generate (independent.cpp:82-93) and handed out per sample by next2DArray (sampler.cpp:82-92); a
       single sample is drawn on the spot by nextSample2D, even when it ends up unused (direct.cpp:212-216, 251-255). */
    std::vector<Vec2> arrays[2];
    Vec2 single[2];
    void generateDirectArrays(size_t sampleCount, size_t emitterSamples, size_t bsdfSamples) {
        dirEmitterSamples = emitterSamples;
        if (qmc == 3 && rinv->hammersley && (emitterSamples > 1 || bsdfSamples > 1))
            throw std::runtime_error("request2DArray(): Not supported for the Hammersley QMC sequence!");       /* hammersley.cpp:293-300 */
        if (ctr) return;
        const size_t n[2] = { emitterSamples, bsdfSamples };
        for (int w = 0; w < 2; ++w) {
            arrays[w].clear();
            if (n[w] > 1)
                for (size_t j = 0; j < sampleCount * n[w]; ++j) {
                    /* independent.cpp:88-90 builds `Point2(m_random->nextFloat(), m_random->nextFloat())`: the order of the two
                       calls is unspecified in C++; GCC (the reference's Linux compiler) evaluates the arguments right to left, so
                       y receives the first number -- verified against the reference compiled with GCC (oracle/_ref) */
                    Float y = rng->nextFloat(); Float x = rng->nextFloat();
                    arrays[w].push_back(Vec2(x, y));
                }
        }
    }
    void beginDirectArray(int which, size_t count) {
        if (!ctr && count <= 1) { Float a = rng->nextFloat(); Float b = rng->nextFloat(); single[which] = Vec2(a, b); }
    }
    size_t dirEmitterSamples = 1;      /* (set by generateDirectArrays: which request / array the BSDF samples are depends on it) */
    Vec2 directSample(int which, size_t i, size_t count) const {
        if (!ctr) return count > 1 ? arrays[which][(size_t) sample * count + i] : single[which];
        if (sequence()) {
            /* sobol.cpp:170-196 / halton.cpp:274-328 (generate) and :238-257 / :352-384 (next2D).  More than one sample of a kind: requested array a
               (direct.cpp:139-146: the emitter array first) owns dimensions 5 + 2 a and 6 + 2 a; its element sample * count + i is the point of
               "sample j of this pixel", j = sample * count + i -- sobol::look_up(j, pixel), m_offset + j * m_stride.  A single sample: the next 2D
               request -- dimensions (2, 3) for the emitter sample (the camera sample took 0 and 1); for the BSDF sample (2, 3) when the emitter
               samples were an array, else m_dimension is 4 and next2D() moves on to the end of the arrays' range, 5 with no array requested: (5, 6) */
            const uint32_t E = (uint32_t) dirEmitterSamples;
            float f[4]; block(1 + (uint32_t) i, f);
            const Vec2 fallback = which == 0 ? Vec2(f[0], f[1]) : Vec2(f[2], f[3]);
            if (count > 1) {
                const uint32_t dim = 5u + 2u * (which ? (E > 1 ? 1u : 0u) : 0u);
                const uint32_t j = sample * (uint32_t) count + (uint32_t) i;
                const uint64_t idx = qmc == 1 ? sobol->lookUp(j, px, py) : rinv->sampleIndex(j, px, py);
                if (dim + 1 >= seqDims()) return fallback;
                return Vec2(seqSample(idx, dim), seqSample(idx, dim + 1));
            }
            return sobol2D(which == 0 ? 2u : (E > 1 ? 2u : 5u), fallback);
        }
        if (ld) {
            /* more than one sample of a kind: a requested 2D array (direct.cpp:139-146, the emitter array first) = ONE scrambled
               sequence of sampleCount * count points in a random order (ldsampler.cpp:193-197); a single one: the sample's next 2D request */
            const uint32_t E = (uint32_t) dirEmitterSamples;
            if (count > 1) {
                const uint32_t a = which ? (E > 1 ? 1u : 0u) : 0u;
                uint32_t h[4] = { pixel, 0x100u + a, 0x4c44u, seed };
                pcg4d(h);
                const uint32_t p = ldPermuteAny((sample & ldMask) * (uint32_t) count + (uint32_t) i, (ldMask + 1u) * (uint32_t) count, h[0]);
                return Vec2(radicalInverse2Single(p, h[1]), sobol2Single(p, h[2]));
            }
            return ldPoint(2u * (which ? (E > 1 ? 1u : 2u) : 1u));
        }
        float f[4]; block(1 + (uint32_t) i, f);
        const Vec2 u = which == 0 ? Vec2(f[0], f[1]) : Vec2(f[2], f[3]);
        if (qmc == 2) {
            /* `stratified` with `direct` (round 5): an array of more than one sample per kind is ONE Latin hypercube over its sampleCount * count entries
               (stratified.cpp:160-164 -> latinHypercube, qmc.cpp: (i + xi) / N per dimension, each dimension shuffled on its own) -- entry e = sample * count + i
               gets, per dimension, the stratum a keyed permutation assigns it; a single sample is the sample's next 2D request (stratified.cpp:177-189) */
            const uint32_t E = (uint32_t) dirEmitterSamples, n = stRes * stRes;
            if (count > 1) {
                const uint32_t a = which ? (E > 1 ? 1u : 0u) : 0u, total = n * (uint32_t) count, e = (sample % n) * (uint32_t) count + (uint32_t) i;
                uint32_t h[4] = { pixel, 0x200u + a, 0x5354u, seed };
                pcg4d(h);
                const float delta = 1.0f / (Float) (size_t) total;
                return Vec2(((Float) (int) ldPermuteAny(e, total, h[0]) + u.x) * delta, ((Float) (int) ldPermuteAny(e, total, h[1]) + u.y) * delta);
            }
            return stPoint2D(which ? (E > 1 ? 1u : 2u) : 1u, u);
        }
        return u;
    }
};

struct PathCounters {
    uint64_t closestRays = 0, shadowRays = 0, pathVertices = 0, samples = 0, invalidSamples = 0;
    uint32_t smoothMask = 0;     /* of the sample being evaluated: bit d-1 = the BSDF at path vertex d has a smooth component (d <= 32) */
    bool verbose = false;        /* oracle_path_sample: print every vertex of the path to stderr */
    TraversalCounters closest, shadow;
    void add(const PathCounters &o) {
        closestRays += o.closestRays; shadowRays += o.shadowRays; pathVertices += o.pathVertices;
        samples += o.samples; invalidSamples += o.invalidSamples;
        closest.nodeVisits += o.closest.nodeVisits; closest.triTests += o.closest.triTests; closest.leafVisits += o.closest.leafVisits;
        shadow.nodeVisits += o.shadow.nodeVisits; shadow.triTests += o.shadow.triTests; shadow.leafVisits += o.shadow.leafVisits;
    }
};

/* `bitmap` texture: Texture2D::eval (texture.cpp:112-121) over BitmapTexture::eval (bitmap.cpp:431-454,486-499) */
struct Texture {
    MipMap mip;
    Vec2 uvScale, uvOffset;
    Spectrum eval(const Intersection &its) const {
        Vec2 uv(its.uv.x * uvScale.x + uvOffset.x, its.uv.y * uvScale.y + uvOffset.y);
        if (its.hasUVPartials)
            return mip.eval(uv, Vec2(its.dudx * uvScale.x, its.dvdx * uvScale.y), Vec2(its.dudy * uvScale.x, its.dvdy * uvScale.y));
        return mip.filterType != PHIP_FILTER_NEAREST ? mip.evalBilinear(0, uv) : mip.evalBox(0, uv);
    }
};

class Scene {
public:
    std::vector<float> positions, normals, texcoords;
    std::vector<Vec3> tanU, tanV;        /* TriMesh::m_tangents per triangle (shapes with texcoords) */
    std::vector<Texture> textures;
    std::vector<uint32_t> indices;
    std::vector<uint32_t> triShape, triPrim;
    std::vector<Shape> shapes;
    std::vector<Material> materials;
    std::vector<phip_emitter> emitters;
    DiscreteDistribution emitterPDF;
    int envEmitter = -1;                 /* Scene::m_environmentEmitter (index into emitters) */
    BSphere envSphere;                   /* ConstantBackgroundEmitter / EnvironmentMap::m_sceneBSphere */
    EnvMap envmap;                       /* valid() iff the environment emitter is an `envmap` */
    KDTree kdtree;
    phip_camera camera;
    phip_film film;
    bool haveNormals = false, haveTexcoords = false;

    void load(const phip_scene_desc &d) {
        if (d.abi_version != PHIP_ABI_VERSION) throw std::runtime_error("oracle: ABI version mismatch");
        positions.assign(d.positions, d.positions + 3 * (size_t) d.n_vertices);
        haveNormals = d.normals != nullptr;
        if (haveNormals) normals.assign(d.normals, d.normals + 3 * (size_t) d.n_vertices);
        indices.assign(d.indices, d.indices + 3 * (size_t) d.n_triangles);
        haveTexcoords = d.texcoords != nullptr;
        if (haveTexcoords) texcoords.assign(d.texcoords, d.texcoords + 2 * (size_t) d.n_vertices);

        camera = d.camera; film = d.film;
textures.resize(d.n_textures);
        for (uint32_t i = 0; i < d.n_textures; ++i) {
            const phip_texture &t = d.textures[i];
            Texture &o = textures[i];
            o.mip.bcu = t.wrap_u; o.mip.bcv = t.wrap_v; o.mip.filterType = t.filter_type; o.mip.maxAnisotropy = t.max_anisotropy;
            if (t.wrap_u > PHIP_WRAP_ONE || t.wrap_v > PHIP_WRAP_ONE || t.filter_type > PHIP_FILTER_EWA) throw std::runtime_error("oracle: bad texture wrap mode / filter type");
            o.mip.load(t.width, t.height, t.n_levels > 1 ? t.n_levels : 1, t.levels);
            o.uvScale = Vec2(t.uv_scale[0], t.uv_scale[1]); o.uvOffset = Vec2(t.uv_offset[0], t.uv_offset[1]);
        }
        materials.resize(d.n_materials);
        for (uint32_t i = 0; i < d.n_materials; ++i) materials[i].m = d.materials[i];
        for (uint32_t i = 0; i < d.n_materials; ++i) configureMaterial(i);
        emitters.assign(d.emitters, d.emitters + d.n_emitters);
        shapes.resize(d.n_shapes);
        triShape.resize(d.n_triangles); triPrim.resize(d.n_triangles);
        uint32_t expect = 0;
        for (uint32_t i = 0; i < d.n_shapes; ++i) {
            shapes[i].s = d.shapes[i];
            const phip_shape &s = d.shapes[i];
            if (s.first_triangle != expect) throw std::runtime_error("oracle: shape triangle ranges must tile the index array");
            if (s.material >= d.n_materials) throw std::runtime_error("oracle: bad material id");
            if (s.emitter >= (int32_t) d.n_emitters) throw std::runtime_error("oracle: bad emitter id");
            if (s.emitter >= 0 && d.emitters[s.emitter].type != PHIP_EMITTER_AREA) throw std::runtime_error("oracle: a shape can only carry an area emitter");
            expect += s.n_triangles;
            for (uint32_t j = 0; j < s.n_triangles; ++j) { triShape[s.first_triangle + j] = i; triPrim[s.first_triangle + j] = j; }
        }
        if (expect != d.n_triangles) throw std::runtime_error("oracle: shape triangle ranges do not cover the index array");
        /* TriMesh::computeUVTangents, trimesh.cpp:683-735 (always run for meshes with texture coordinates, :383-385) */
        tanU.assign(d.n_triangles, Vec3(0.0f)); tanV.assign(d.n_triangles, Vec3(0.0f));
        for (uint32_t i = 0; i < d.n_shapes; ++i) {
            if (!shapes[i].s.has_texcoords) continue;
            if (!haveTexcoords) throw std::runtime_error("oracle: shape has_texcoords but texcoords is NULL");
            for (uint32_t j = 0; j < shapes[i].s.n_triangles; ++j) {
                const uint32_t t = shapes[i].s.first_triangle + j;
                const Vec3 v0 = P(t, 0), v1 = P(t, 1), v2 = P(t, 2);
                const Vec2 uv0 = UV(t, 0), uv1 = UV(t, 1), uv2 = UV(t, 2);
                Vec3 dP1 = v1 - v0, dP2 = v2 - v0;
                Vec2 dUV1(uv1.x - uv0.x, uv1.y - uv0.y), dUV2(uv2.x - uv0.x, uv2.y - uv0.y);
                Vec3 n = cross(dP1, dP2);
                Float length = n.length();
                if (length == 0) continue;
                Float determinant = dUV1.x * dUV2.y - dUV1.y * dUV2.x;
                if (determinant == 0) {
                    coordinateSystem(n / length, tanU[t], tanV[t]);
                } else {
                    Float invDet = 1.0f / determinant;
                    tanU[t] = (dP1 * dUV2.y - dP2 * dUV1.y) * invDet;
                    tanV[t] = (dP1 * (-dUV2.x) + dP2 * dUV1.x) * invDet;
                }
            }
        }
        for (uint32_t i = 0; i < d.n_materials; ++i)
            for (uint32_t t : { materials[i].m.reflectance_texture, materials[i].m.alpha_u_texture, materials[i].m.alpha_v_texture, materials[i].m.transmittance_texture })
                if (t > d.n_textures) throw std::runtime_error("oracle: bad texture id");
        /* TriMesh::computeUVTangents, trimesh.cpp:683-693: an anisotropic BSDF (roughconductor.cpp:196-200,230-231: the clamped
           alphaU != alphaV; twosided.cpp:96-100 inherits the flag) needs texture coordinates for its tangent frame */
        for (uint32_t i = 0; i < d.n_shapes; ++i) {
            std::function<bool(uint32_t, int)> aniso = [&](uint32_t m, int depth) -> bool {
                if (m >= d.n_materials || depth > 2) return false;
                const phip_material &M = d.materials[m];
                if (M.type == PHIP_BSDF_ROUGHCONDUCTOR)          /* m_alphaU != m_alphaV as objects, roughconductor.cpp:228-229 */
                    return (M.alpha_u_texture | M.alpha_v_texture) ? M.alpha_u_texture != M.alpha_v_texture : std::max(M.alpha_u, 1e-4f) != std::max(M.alpha_v, 1e-4f);
                if (M.type == PHIP_BSDF_TWOSIDED) return aniso(M.nested[0], depth + 1) || aniso(M.nested[1], depth + 1);
                return false;
            };
            if (!shapes[i].s.has_texcoords && aniso(shapes[i].s.material, 0))
                throw std::runtime_error("computeUVTangents(): texture coordinates are required to generate tangent vectors (anisotropic BSDF on a shape without them)");
        }
        /* area sampling tables, trimesh.cpp:388-404 (built lazily in the reference) */
        for (uint32_t i = 0; i < d.n_shapes; ++i) {
            Shape &sh = shapes[i];
            if (sh.s.emitter < 0) continue;
            for (uint32_t j = 0; j < sh.s.n_triangles; ++j) {
                uint32_t t = sh.s.first_triangle + j;
                Vec3 p0 = P(t, 0), p1 = P(t, 1), p2 = P(t, 2);
                Vec3 sideA = p1 - p0, sideB = p2 - p0;
                sh.areaDistr.append(0.5f * cross(sideA, sideB).length());   /* triangle.cpp:61-67 */
            }
            sh.surfaceArea = sh.areaDistr.normalize();
            sh.invSurfaceArea = 1.0f / sh.surfaceArea;
        }
        /* scene.cpp:375-381 */
        for (uint32_t i = 0; i < d.n_emitters; ++i) emitterPDF.append(emitters[i].sampling_weight);
        if (d.n_emitters > 0) emitterPDF.normalize();
        kdtree.build(positions.data(), indices.data(), d.n_triangles, triShape.data(), triPrim.data());
        for (uint32_t i = 0; i < d.n_emitters; ++i) {
            if (emitters[i].type == PHIP_EMITTER_AREA) continue;
            if (emitters[i].type != PHIP_EMITTER_CONSTANT && emitters[i].type != PHIP_EMITTER_ENVMAP) throw std::runtime_error("oracle: unknown emitter type");
            if (emitters[i].type == PHIP_EMITTER_ENVMAP) envmap.load(d.envmap);
            if (envEmitter >= 0) throw std::runtime_error("The scene may only contain one environment emitter");   /* scene.cpp:510-513 */
            envEmitter = (int) i;
        }
        if (envEmitter >= 0) {
            /* Scene::initializeBidirectional (scene.cpp:384-413): the scene box seen by createShape() is the
               kd-tree's (enlarged) box expanded by the sensor's translation bounds (track.cpp:79-83: the image of
               the origin); constant.cpp:67-72 = envmap.cpp:330-334: bounding sphere of that box (aabb.cpp:44-47), radius x 1.5 */
            AABB aabb = kdtree.aabb;
            const float *m = camera.to_world;
            Vec3 sp(m[3], m[7], m[11]);
            const Float w = m[15];
            if (w != 1.0f) sp = sp / w;              /* transform.h: Transform::operator()(Point) */
            aabb.expandBy(sp);
            const Vec3 center = (aabb.max + aabb.min) * (Float) 0.5;
            BSphere bs(center, (center - aabb.max).length());
            bs.radius = std::max(ORC_EPSILON, bs.radius * 1.5f);
            envSphere = bs;
        }
    }

    Vec3 P(uint32_t tri, int c) const { const float *p = &positions[3 * (size_t) indices[3 * (size_t) tri + c]]; return Vec3(p[0], p[1], p[2]); }
    Vec2 UV(uint32_t tri, int c) const { const float *p = &texcoords[2 * (size_t) indices[3 * (size_t) tri + c]]; return Vec2(p[0], p[1]); }
    Vec3 Nrm(uint32_t tri, int c) const { const float *p = &normals[3 * (size_t) indices[3 * (size_t) tri + c]]; return Vec3(p[0], p[1], p[2]); }

    /* a `bitmap` texture on specularReflectance: ensureEnergyConservation (dielectric.cpp:159-160, roughconductor.cpp:236) */
    void specularTexture(const Material &M) const {
        if (M.m.reflectance_texture == 0) return;
        if (M.m.reflectance_texture > textures.size()) throw std::runtime_error("oracle: bad texture id");
        Float mx = 0;
        for (const Spectrum &t : textures[M.m.reflectance_texture - 1].mip.levels[0]) mx = std::max(mx, t.max());
        if (mx > 1.0f) throw std::runtime_error("specularReflectance texture > 1 (ensureEnergyConservation)");
    }

    void configureMaterial(uint32_t i) {
        Material &M = materials[i];
        switch (M.m.type) {
            case PHIP_BSDF_DIFFUSE: {
                /* diffuse.cpp:93-103: clamp to <=1 (ensureEnergyConservation), component only if max > 0 */
                Float mx = std::max(M.m.reflectance[0], std::max(M.m.reflectance[1], M.m.reflectance[2]));
                if (M.m.reflectance_texture != 0) {        /* m_reflectance->getMaximum().max(), level 0 of the bitmap */
                    if (M.m.reflectance_texture > textures.size()) throw std::runtime_error("oracle: bad texture id");
                    mx = 0;
                    for (const Spectrum &t : textures[M.m.reflectance_texture - 1].mip.levels[0]) mx = std::max(mx, t.max());
                    if (mx > 1.0f) throw std::runtime_error("diffuse reflectance texture > 1 (ensureEnergyConservation, diffuse.cpp:95)");
                }
                M.smooth = mx > 0; M.transOrBack = false;
            } break;
            case PHIP_BSDF_DIELECTRIC: {
                M.smooth = false; M.transOrBack = true; specularTexture(M);
                if (M.m.transmittance_texture != 0) {      /* ensureEnergyConservation(specularTransmittance), dielectric.cpp:207-208 */
                    Float mx = 0;
                    for (const Spectrum &t : textures[M.m.transmittance_texture - 1].mip.levels[0]) mx = std::max(mx, t.max());
                    if (mx > 1.0f) throw std::runtime_error("specularTransmittance texture > 1 (ensureEnergyConservation)");
                }
            } break;
            case PHIP_BSDF_ROUGHCONDUCTOR: {
                M.smooth = true; M.transOrBack = false; specularTexture(M);
                /* roughconductor.cpp:275-280: alpha = texture.eval().average(); microfacet.h:113-114 clamp */
                M.alphaU = std::max(Spectrum(M.m.alpha_u).average(), (Float) 1e-4f);
                M.alphaV = std::max(Spectrum(M.m.alpha_v).average(), (Float) 1e-4f);
                if (M.m.distribution > PHIP_MF_GGX) throw std::runtime_error("oracle: unsupported microfacet distribution");
            } break;
            case PHIP_BSDF_TWOSIDED: {
                if (M.m.nested[0] >= materials.size() || M.m.nested[1] >= materials.size()) throw std::runtime_error("oracle: bad nested material");
                if (M.m.nested[0] >= i || M.m.nested[1] >= i) throw std::runtime_error("oracle: nested materials must precede the twosided adapter");
                const Material &a = materials[M.m.nested[0]], &b = materials[M.m.nested[1]];
                if (a.m.type == PHIP_BSDF_DIELECTRIC || b.m.type == PHIP_BSDF_DIELECTRIC || a.m.type == PHIP_BSDF_TWOSIDED || b.m.type == PHIP_BSDF_TWOSIDED)
                    throw std::runtime_error("oracle: twosided can only nest one-sided reflection models (twosided.cpp:104-106)");
                M.smooth = a.smooth || b.smooth; M.transOrBack = true;   /* twosided.cpp:96-100: EBackSide set */
            } break;
            default: throw std::runtime_error("oracle: unknown material type");
        }
    }

    /* skdtree.cpp:112-142 + skdtree.h:343-428 */
    bool rayIntersect(const Ray &ray, Intersection &its, PathCounters *pc) const {
        Float t, u, v; uint32_t prim;
        its.shape = -1; its.t = std::numeric_limits<Float>::infinity();
        if (pc) pc->closestRays++;
        if (!kdtree.rayIntersect(ray, t, u, v, prim, pc ? &pc->closest : nullptr))
            return false;
        its.t = t;
        fillIntersectionRecord(ray, prim, u, v, its);
        return true;
    }

    bool rayIntersectShadow(const Ray &ray, PathCounters *pc) const {
        if (pc) pc->shadowRays++;
        return kdtree.rayIntersectShadow(ray, pc ? &pc->shadow : nullptr);
    }

    void fillIntersectionRecord(const Ray &ray, uint32_t prim, Float cu, Float cv, Intersection &its) const {
        const uint32_t shapeIdx = triShape[prim];
        const Shape &sh = shapes[shapeIdx];
        const Vec3 b(1 - cu - cv, cu, cv);
        const Vec3 p0 = P(prim, 0), p1 = P(prim, 1), p2 = P(prim, 2);
        its.p = p0 * b.x + p1 * b.y + p2 * b.z;
        Vec3 side1(p1 - p0), side2(p2 - p0);
        Vec3 faceNormal(cross(side1, side2));
        Float length = faceNormal.length();
        if (!faceNormal.isZero())
            faceNormal /= length;
        if (sh.s.has_texcoords) {           /* vertexTangents, skdtree.h:373-377 */
            its.dpdu = tanU[prim];
            its.dpdv = tanV[prim];
        } else {
            its.dpdu = side1;
            its.dpdv = side2;
        }
        if (sh.s.has_normals && haveNormals) {
            const Vec3 n0 = Nrm(prim, 0), n1 = Nrm(prim, 1), n2 = Nrm(prim, 2);
            its.shFrame.n = normalize(n0 * b.x + n1 * b.y + n2 * b.z);
            if (dot(faceNormal, its.shFrame.n) < 0)
                faceNormal = -faceNormal;
        } else {
            its.shFrame.n = faceNormal;
        }
        its.geoFrame = Frame(faceNormal);
        if (sh.s.has_texcoords) {           /* skdtree.h:397-404 */
            const Vec2 t0 = UV(prim, 0), t1 = UV(prim, 1), t2 = UV(prim, 2);
            its.uv = Vec2(t0.x * b.x + t1.x * b.y + t2.x * b.z, t0.y * b.x + t1.y * b.y + t2.y * b.z);
        } else {
            its.uv = Vec2(b.y, b.z);
        }
        its.hasUVPartials = false;
        its.shape = (int) shapeIdx;
        its.primIndex = prim;
        computeShadingFrame(its.shFrame.n, its.dpdu, its.shFrame);
        its.wi = its.toLocal(-ray.d);
    }

    const Material &bsdfOf(const Intersection &its) const { return materials[shapes[its.shape].s.material]; }

    /* BSDF::usesRayDifferentials: a bitmap texture somewhere below (diffuse.cpp, twosided.cpp) */
    bool usesRayDifferentials(const Material &M) const {
        if (M.m.type == PHIP_BSDF_TWOSIDED)
            return usesRayDifferentials(materials[M.m.nested[0]]) || usesRayDifferentials(materials[M.m.nested[1]]);
        return (M.m.reflectance_texture | M.m.alpha_u_texture | M.m.alpha_v_texture | M.m.transmittance_texture) != 0;   /* diffuse.cpp, roughconductor.cpp:244-248, dielectric.cpp:196-198 */
    }

    /* Intersection::computePartials, intersection.cpp:5-76 (rxOrigin = ryOrigin = the ray origin for a pinhole camera) */
    static void computePartials(Intersection &its, const Vec3 &rayO, const Vec3 &rxDirection, const Vec3 &ryDirection) {
        Float A[2][2], Bx[2], By[2], x[2];
        int axes[2];
        if (its.hasUVPartials) return;
        its.hasUVPartials = true;
        if (its.dpdu.isZero() && its.dpdv.isZero()) { its.dudx = its.dvdx = its.dudy = its.dvdy = 0.0f; return; }
        const Vec3 &gn = its.geoFrame.n;
        const Float pp = dot(gn, its.p), pox = dot(gn, rayO), poy = dot(gn, rayO),
                    prx = dot(gn, rxDirection), pry = dot(gn, ryDirection);
        if (prx == 0 || pry == 0) { its.dudx = its.dvdx = its.dudy = its.dvdy = 0.0f; return; }
        const Float tx = (pp - pox) / prx, ty = (pp - poy) / pry;
        Float absX = std::abs(gn.x), absY = std::abs(gn.y), absZ = std::abs(gn.z);
        if (absX > absY && absX > absZ) { axes[0] = 1; axes[1] = 2; }
        else if (absY > absZ) { axes[0] = 0; axes[1] = 2; }
        else { axes[0] = 0; axes[1] = 1; }
        A[0][0] = its.dpdu[axes[0]]; A[0][1] = its.dpdv[axes[0]];
        A[1][0] = its.dpdu[axes[1]]; A[1][1] = its.dpdv[axes[1]];
        Vec3 px = rayO + rxDirection * tx, py = rayO + ryDirection * ty;
        Bx[0] = px[axes[0]] - its.p[axes[0]]; Bx[1] = px[axes[1]] - its.p[axes[1]];
        By[0] = py[axes[0]] - its.p[axes[0]]; By[1] = py[axes[1]] - its.p[axes[1]];
        if (solveLinearSystem2x2(A, Bx, x)) { its.dudx = x[0]; its.dvdx = x[1]; }
        else { its.dudx = 1; its.dvdx = 0; }
        if (solveLinearSystem2x2(A, By, x)) { its.dudy = x[0]; its.dvdy = x[1]; }
        else { its.dudy = 0; its.dudy = 1; }      /* (sic, intersection.cpp:74) */
    }
    bool isEmitter(const Intersection &its) const { return shapes[its.shape].s.emitter >= 0; }

    /* area.cpp:104-109 */
    Spectrum Le(const Intersection &its, const Vec3 &d) const {
        const phip_emitter &e = emitters[shapes[its.shape].s.emitter];
        if (dot(its.shFrame.n, d) <= 0) return Spectrum(0.0f);
        return Spectrum(e.radiance);
    }

    /* records.inl:160-164 */
    void initDirectRecord(DirectSamplingRecord &dRec, const Intersection &refIts) const {
        dRec.ref = refIts.p; dRec.refN = Vec3(0.0f);
        if (!bsdfOf(refIts).transOrBack)
            dRec.refN = refIts.shFrame.n;
        dRec.emitter = -1; dRec.pdf = 0; dRec.measure = EInvalidMeasure;
    }
    /* records.inl:170-178 */
    void setQuery(DirectSamplingRecord &dRec, const Ray &ray, const Intersection &its) const {
        dRec.p = its.p; dRec.n = its.shFrame.n; dRec.measure = ESolidAngle; dRec.uv = its.uv;
        dRec.emitter = shapes[its.shape].s.emitter; dRec.d = ray.d; dRec.dist = its.t;
    }

    /* trimesh.cpp:412-423 + triangle.cpp:24-59 */
    void samplePosition(const Shape &sh, DirectSamplingRecord &dRec, const Vec2 &_sample) const {
        Vec2 sample(_sample);
        size_t index = sh.areaDistr.sampleReuse(sample.y);
        uint32_t t = sh.s.first_triangle + (uint32_t) index;
        const Vec3 p0 = P(t, 0), p1 = P(t, 1), p2 = P(t, 2);
        Vec2 bary = squareToUniformTriangle(sample);
        Vec3 sideA = p1 - p0, sideB = p2 - p0;
        dRec.p = p0 + (sideA * bary.x) + (sideB * bary.y);
        if (sh.s.has_normals && haveNormals) {
            const Vec3 n0 = Nrm(t, 0), n1 = Nrm(t, 1), n2 = Nrm(t, 2);
            dRec.n = normalize(n0 * (1.0f - bary.x - bary.y) + n1 * bary.x + n2 * bary.y);
        } else {
            dRec.n = normalize(cross(sideA, sideB));
        }
        dRec.uv = bary;
        dRec.pdf = sh.invSurfaceArea;
        dRec.measure = EArea;
    }

    /* shape.cpp:102-115 */
    void shapeSampleDirect(const Shape &sh, DirectSamplingRecord &dRec, const Vec2 &sample) const {
        samplePosition(sh, dRec, sample);
        dRec.d = dRec.p - dRec.ref;
        Float distSquared = dRec.d.lengthSquared();
        dRec.dist = std::sqrt(distSquared);
        dRec.d /= dRec.dist;
        Float dp = absDot(dRec.d, dRec.n);
        dRec.pdf *= dp != 0 ? (distSquared / dp) : 0.0f;
        dRec.measure = ESolidAngle;
    }

    /* scene.h:910-913 + constant.cpp:254-256 / envmap.cpp:380-409 (ray without differentials) */
    Spectrum evalEnvironment(const Ray &ray) const {
        if (envEmitter < 0) return Spectrum(0.0f);
        if (envmap.valid()) return envmap.evalEnvironment(ray.d);
        return Spectrum(emitters[envEmitter].radiance);
    }
    /* the same for a ray with differentials (camera rays): the envmap filters the lookup, envmap.cpp:395-407 */
    Spectrum evalEnvironment(const Ray &ray, const Vec3 &rx, const Vec3 &ry) const {
        if (envEmitter >= 0 && envmap.valid() && envmap.nLevels > 1) return envmap.evalEnvironment(ray.d, rx, ry);
        return evalEnvironment(ray);
    }

    /* constant.cpp:258-273 = envmap.cpp:354-370; false = "internal error" (the path is terminated, path.cpp:242-243) */
    bool fillDirectSamplingRecord(DirectSamplingRecord &dRec, const Ray &ray) const {
        Float nearT, farT;
        if (!envSphere.rayIntersect(ray.o, ray.d, nearT, farT) || nearT > 0 || farT < 0)
            return false;
        dRec.p = ray.o + ray.d * farT;
        dRec.n = normalize(envSphere.center - dRec.p);
        dRec.measure = ESolidAngle;
        dRec.emitter = envEmitter;
        dRec.d = ray.d;
        dRec.dist = farT;
        return true;
    }

    /* constant.cpp:184-225 */
    Spectrum constantSampleDirect(const phip_emitter &em, DirectSamplingRecord &dRec, const Vec2 &sample) const {
        Vec3 d;
        Float pdf;
        if (!dRec.refN.isZero()) {
            d = squareToCosineHemisphere(sample);
            pdf = squareToCosineHemispherePdf(d);
            d = Frame(dRec.refN).toWorld(d);
        } else {
            d = squareToUniformSphere(sample);
            pdf = squareToUniformSpherePdf();
        }
        Float nearT, farT;
        dRec.pdf = 0.0f;
        if (!envSphere.rayIntersect(dRec.ref, d, nearT, farT))
            return Spectrum(0.0f);
        if (!(nearT < 0 && farT > 0))
            return Spectrum(0.0f);
        dRec.p = dRec.ref + d * farT;
        dRec.n = normalize(envSphere.center - dRec.p);
        dRec.measure = ESolidAngle;
        dRec.d = d;
        dRec.dist = farT;
        dRec.pdf = pdf;
        if (!dRec.refN.isZero() && dot(dRec.d, dRec.refN) <= 0)
            return Spectrum(0.0f);
        return Spectrum(em.radiance) / pdf;
    }

    /* constant.cpp:227-243 */
    Float constantPdfDirect(const DirectSamplingRecord &dRec) const {
        Float pdfSA;
        if (!dRec.refN.isZero())
            pdfSA = ORC_INV_PI * std::max((Float) 0.0f, dot(dRec.d, dRec.refN));
        else
            pdfSA = squareToUniformSpherePdf();
        if (dRec.measure == ESolidAngle)
            return pdfSA;
        else if (dRec.measure == EArea)
            return pdfSA * absDot(dRec.d, dRec.n) / (dRec.dist * dRec.dist);
        else
            return 0.0f;
    }

    /* envmap.cpp:516-542 */
    Spectrum envmapSampleDirect(DirectSamplingRecord &dRec, const Vec2 &sample) const {
        Spectrum value; Vec3 d; Float pdf;
        envmap.internalSampleDirection(sample, d, value, pdf);
        const Vec3 rd = EnvMap::xformVec(envmap.toWorld, d);
        Float nearT, farT;
        if (value.isZero() || pdf == 0 || !envSphere.rayIntersect(dRec.ref, rd, nearT, farT) || nearT >= 0 || farT <= 0) {
            dRec.pdf = 0.0f;
            return Spectrum(0.0f);
        }
        dRec.pdf = pdf;
        dRec.p = dRec.ref + rd * farT;
        dRec.n = normalize(envSphere.center - dRec.p);
        dRec.dist = farT;
        dRec.d = rd;
        dRec.measure = ESolidAngle;
        return value / pdf;
    }

    /* envmap.cpp:545-556 */
    Float envmapPdfDirect(const DirectSamplingRecord &dRec) const {
        Float pdfSA = envmap.internalPdfDirection(EnvMap::xformVec(envmap.toLocal, dRec.d));
        if (dRec.measure == ESolidAngle)
            return pdfSA;
        else if (dRec.measure == EArea)
            return pdfSA * absDot(dRec.d, dRec.n) / (dRec.dist * dRec.dist);
        else
            return 0.0f;
    }

    /* area.cpp:158-173 */
    Spectrum areaSampleDirect(const phip_emitter &em, DirectSamplingRecord &dRec, const Vec2 &sample) const {
        shapeSampleDirect(shapes[em.shape], dRec, sample);
        if (dot(dRec.d, dRec.refN) >= 0 && dot(dRec.d, dRec.n) < 0 && dRec.pdf != 0) {
            return Spectrum(em.radiance) / dRec.pdf;
        } else {
            dRec.pdf = 0.0f;
            return Spectrum(0.0f);
        }
    }

    /* scene.cpp:828-852 (visibility test included) */
    Spectrum sampleEmitterDirect(DirectSamplingRecord &dRec, const Vec2 &_sample, PathCounters *pc) const {
        Vec2 sample(_sample);
        if (emitters.empty()) { dRec.pdf = 0; return Spectrum(0.0f); }
        Float emPdf;
        size_t index = emitterPDF.sampleReuse(sample.x, emPdf);
        const phip_emitter &em = emitters[index];
        Spectrum value = em.type == PHIP_EMITTER_CONSTANT ? constantSampleDirect(em, dRec, sample)
                       : em.type == PHIP_EMITTER_ENVMAP ? envmapSampleDirect(dRec, sample)
                                                        : areaSampleDirect(em, dRec, sample);
        if (pc && pc->verbose)
            fprintf(stderr, "    NEE: emitter %zu pdf %.9g value (%.9g %.9g %.9g) point (%.9g %.9g %.9g) d (%.9g %.9g %.9g) dist %.9g\n", index, dRec.pdf, value[0], value[1], value[2],
                    dRec.p.x, dRec.p.y, dRec.p.z, dRec.d.x, dRec.d.y, dRec.d.z, dRec.dist);
        if (dRec.pdf != 0) {
            Ray ray(dRec.ref, dRec.d, ORC_EPSILON, dRec.dist * (1 - ORC_SHADOW_EPSILON));
            const bool occ = rayIntersectShadow(ray, pc);
            if (pc && pc->verbose) fprintf(stderr, "    NEE: shadow ray %s\n", occ ? "OCCLUDED" : "free");
            if (occ)
                return Spectrum(0.0f);
            dRec.emitter = (int) index;
            dRec.pdf *= emPdf;
            value /= emPdf;
            return value;
        }
        return Spectrum(0.0f);
    }

    /* scene.cpp:949-952, scene.h:848-850, area.cpp:175-182, shape.cpp:117-126 */
    Float pdfEmitterDirect(const DirectSamplingRecord &dRec) const {
        const phip_emitter &em = emitters[dRec.emitter];
        Float pdf;
        if (em.type == PHIP_EMITTER_CONSTANT) {
            pdf = constantPdfDirect(dRec);
        } else if (em.type == PHIP_EMITTER_ENVMAP) {
            pdf = envmapPdfDirect(dRec);
        } else if (dot(dRec.d, dRec.refN) >= 0 && dot(dRec.d, dRec.n) < 0) {
            Float pdfPos = shapes[em.shape].invSurfaceArea;
            if (dRec.measure == ESolidAngle)
                pdf = pdfPos * (dRec.dist * dRec.dist) / absDot(dRec.d, dRec.n);
            else if (dRec.measure == EArea)
                pdf = pdfPos;
            else
                pdf = 0.0f;
        } else {
            pdf = 0.0f;
        }
        return pdf * (em.sampling_weight * emitterPDF.normalization);
    }
};

} // namespace orc
