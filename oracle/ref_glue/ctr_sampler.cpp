/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * ctr_sampler.cpp: a Sampler plugin FOR THE REFERENCE ("ctr", oracle/_ref/plugins/ctr.so) that hands the reference's own
 * integrators the counter-based parity stream of HISTORY.md 3.5 -- pcg4d(pixel, sampleIndex, block, seed) -- so that the
 * reference's `path` / `direct` and the GPU consume THE SAME random numbers and their images can be compared directly
 * (tests/test_gpu_dropin.py).  The stream is defined by CALL ORDER, which is all a sampler sees:
 *   `path`:   2D request 0 of a sample = the pixel jitter (block 0 .xy, integrator.cpp:171); 2D request 1 + k = pair k & 1 (.xy / .zw)
 *             of block 1 + 2 (k >> 1) -- a vertex with a smooth BSDF makes two requests (emitter sample, path.cpp:176; BSDF sample,
 *             :209), one without (dielectric) makes one, and the device / the oracle count exactly like that (k_shade.h,
 *             o_scene.h); 1D request j (Russian roulette, path.cpp:283: one per vertex from depth rrDepth on) = .x of block
 *             2 + 2 (rrDepth - 1 + j).
 *   `direct`: the sample counts tell which 2D calls are single samples (direct.cpp:212-216,251-255); arrays as in generate().
 * Nothing about the scene is needed (round 1 needed the oracle's per-sample smooth-vertex masks for scenes with dielectrics).
 * ld = true (`path` only): PHIP_SAMPLER_LD -- the first four 2D and the first four 1D requests of a sample are points of scrambled
 * (0,2)-sequences in a keyed order (include/phip.h, HISTORY.md 3.5), made with the reference's own qmc.h functions.
 */
#include <mitsuba/render/sampler.h>
#include <mitsuba/render/scene.h>
#include <mitsuba/core/qmc.h>     /* PHIP_SAMPLER_LD: the reference's own radicalInverse2Single / sobol2Single make the points */

MTS_NAMESPACE_BEGIN

class CtrSampler : public Sampler {
public:
    CtrSampler(const Properties &props) : Sampler(props) {
        m_sampleCount = props.getSize("sampleCount", 4);
        m_seed = (uint32_t) props.getInteger("seed", 0);
        m_width = props.getInteger("cropWidth", 0);
        m_direct = props.getString("mode", "path") == "direct";
        m_emitterSamples = props.getSize("emitterSamples", 1);
        m_bsdfSamples = props.getSize("bsdfSamples", 1);
        m_rrFirst = (uint32_t) props.getInteger("rrDepth", 5);
        m_ld = props.getBoolean("ld", false);
        m_ldMask = (uint32_t) props.getSize("sampleTotal", m_sampleCount) - 1u;
        /* PHIP_SAMPLER_STRATIFIED: the construction of stratified.cpp:147-200 on the counter stream (include/phip.h); `path` only */
        m_stratified = props.getBoolean("stratified", false);
        m_stRes = 1; while ((size_t) m_stRes * m_stRes < props.getSize("sampleTotal", m_sampleCount)) ++m_stRes;
        m_pixel = 0; m_call2D = 0; m_call1D = 0;
    }
    CtrSampler(Stream *stream, InstanceManager *manager) : Sampler(stream, manager) { Log(EError, "ctr sampler: not serializable"); }

    ref<Sampler> clone() {
        ref<CtrSampler> s = new CtrSampler(getProperties());
        s->m_sampleCount = m_sampleCount; s->m_seed = m_seed; s->m_width = m_width; s->m_direct = m_direct;
        s->m_emitterSamples = m_emitterSamples; s->m_bsdfSamples = m_bsdfSamples; s->m_rrFirst = m_rrFirst;
        s->m_ld = m_ld; s->m_ldMask = m_ldMask; s->m_stratified = m_stratified; s->m_stRes = m_stRes;
        for (size_t i = 0; i < m_req1D.size(); ++i) s->request1DArray(m_req1D[i]);
        for (size_t i = 0; i < m_req2D.size(); ++i) s->request2DArray(m_req2D[i]);
        return s.get();
    }

    void setFilmResolution(const Vector2i &res, bool blocked) { if (m_width == 0) m_width = res.x; }

    void generate(const Point2i &offset) {
        m_pixel = (uint32_t) (offset.y * m_width + offset.x);           /* crop-relative pixel, row-major: the parity stream's key */
        /* sample arrays of `direct`: array a of sample j, entry i = block 1 + i of that sample; the emitter array (if
           requested, direct.cpp:140-146) comes first and takes .xy, the BSDF array .zw */
        for (size_t a = 0; a < m_req2D.size() && m_ld; ++a)           /* ldsampler.cpp:193-197: one scrambled sequence per array, in a random order */
            for (size_t e = 0; e < m_sampleCount * m_req2D[a]; ++e)
                m_sampleArrays2D[a][e] = ldArrayPoint((uint32_t) a, (uint32_t) e, (uint32_t) ((m_ldMask + 1u) * m_req2D[a]));
        for (size_t a = 0; a < m_req2D.size() && !m_ld; ++a) {
            const bool emitter = (a == 0 && m_emitterSamples > 1);
            for (size_t j = 0; j < m_sampleCount; ++j)
                for (size_t i = 0; i < m_req2D[a]; ++i) {
                    float f[4]; block((uint32_t) j, 1 + (uint32_t) i, f);
                    const Point2 u = emitter ? Point2(f[0], f[1]) : Point2(f[2], f[3]);
                    /* PHIP_SAMPLER_STRATIFIED (round 5): the array is one Latin hypercube over its sampleCount * count entries (stratified.cpp:160-164) */
                    m_sampleArrays2D[a][j * m_req2D[a] + i] = m_stratified ? stArrayPoint((uint32_t) a, (uint32_t) ((j % ((size_t) m_stRes * m_stRes)) * m_req2D[a] + i), (uint32_t) ((size_t) m_stRes * m_stRes * m_req2D[a]), u) : u;
                }
        }
        m_sampleIndex = 0; m_dimension1DArray = m_dimension2DArray = 0;
        beginSample();
    }
    void advance() { Sampler::advance(); beginSample(); }
    void setSampleIndex(size_t i) { Sampler::setSampleIndex(i); beginSample(); }

    Point2 next2D() {
        float f[4];
        const uint32_t call = m_call2D++;
        if (m_ld && call < 4) return ldPoint(2 * call);                                             /* ldsampler.cpp:218-224 */
        if (call == 0) {                                                                            /* integrator.cpp:171 */
            block((uint32_t) m_sampleIndex, 0, f);
            return m_stratified ? stPoint2D(0, f[0], f[1]) : Point2(f[0], f[1]);
        }
        if (m_direct) {
            /* direct.cpp:212-216: a single emitter sample (also drawn when emitterSamples == 0); :251-255 the same for the BSDF */
            const bool emitterCall = (m_emitterSamples <= 1) && call == 1;
            block((uint32_t) m_sampleIndex, 1, f);
            const Point2 u = emitterCall ? Point2(f[0], f[1]) : Point2(f[2], f[3]);
            return (m_stratified && call < 4) ? stPoint2D(call, u.x, u.y) : u;                       /* stratified.cpp:177-189: the sample's request number `call` */
        }
        /* `path`: 2D request 1 + k of the sample */
        const uint32_t k = call - 1;
        block((uint32_t) m_sampleIndex, 1 + 2 * (k >> 1), f);
        const Point2 u = (k & 1u) ? Point2(f[2], f[3]) : Point2(f[0], f[1]);
        return (m_stratified && call < 4) ? stPoint2D(call, u.x, u.y) : u;                          /* stratified.cpp:177-195: the first `dimension` = 4 requests */
    }
    Float next1D() {                                                                                /* path.cpp:283, Russian roulette */
        const uint32_t call = m_call1D++;
        if (m_ld && !m_direct && call < 4) return ldPoint(2 * call + 1).x;                          /* ldsampler.cpp:212-216 */
        float f[4]; block((uint32_t) m_sampleIndex, 2 + 2 * (m_rrFirst - 1 + call), f);
        if (m_stratified && !m_direct && call < 4)                                                  /* stratified.cpp:167-175 */
            return ((int) stCell(2 * call + 1) + f[0]) * (1 / (Float) (size_t) (m_stRes * m_stRes));
        return f[0];
    }

    std::string toString() const { return "CtrSampler[]"; }
    MTS_DECLARE_CLASS()
private:
    void beginSample() { m_call2D = 0; m_call1D = 0; }
    /* the cell sample m_sampleIndex visits in dimension dim (2D request q: 2 q, 1D request j: 2 j + 1): a keyed permutation of the sample index */
    uint32_t stCell(uint32_t dim) const {
        uint32_t v[4] = { m_pixel, dim, 0x5354u, m_seed };
        for (int i = 0; i < 4; ++i) v[i] = v[i] * 1664525u + 1013904223u;
        v[0] += v[1] * v[3]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1]; v[3] += v[1] * v[2];
        for (int i = 0; i < 4; ++i) v[i] ^= v[i] >> 16;
        v[0] += v[1] * v[3]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1]; v[3] += v[1] * v[2];
        const uint32_t n = m_stRes * m_stRes;
        return permuteAny((uint32_t) m_sampleIndex % n, n, v[0]);
    }
    Point2 stArrayPoint(uint32_t a, uint32_t e, uint32_t total, const Point2 &u) const {
        uint32_t v[4] = { m_pixel, 0x200u + a, 0x5354u, m_seed };
        for (int i = 0; i < 4; ++i) v[i] = v[i] * 1664525u + 1013904223u;
        v[0] += v[1] * v[3]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1]; v[3] += v[1] * v[2];
        for (int i = 0; i < 4; ++i) v[i] ^= v[i] >> 16;
        v[0] += v[1] * v[3]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1]; v[3] += v[1] * v[2];
        const Float delta = 1 / (Float) (size_t) total;
        return Point2(((Float) (int) permuteAny(e, total, v[0]) + u.x) * delta, ((Float) (int) permuteAny(e, total, v[1]) + u.y) * delta);
    }
    Point2 stPoint2D(uint32_t q, Float u1, Float u2) const {
        const uint32_t c = stCell(2 * q);
        const int x = (int) (c % m_stRes), y = (int) (c / m_stRes);
        const Float inv = 1 / (Float) (int) m_stRes;
        return Point2((x + u1) * inv, (y + u2) * inv);
    }
    static float toFloat(uint32_t u) { uint32_t b = (u >> 9) | 0x3f800000u; float f; memcpy(&f, &b, 4); return f - 1.0f; }   /* random.cpp:632-641 */
    void block(uint32_t sample, uint32_t blk, float out[4]) const {
        /* pcg4d (Jarzynski & Olano, JCGT 9(3) 2020) over (pixel, sample, block, seed) */
        uint32_t v[4] = { m_pixel, sample, blk, m_seed };
        for (int i = 0; i < 4; ++i) v[i] = v[i] * 1664525u + 1013904223u;
        v[0] += v[1] * v[3]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1]; v[3] += v[1] * v[2];
        for (int i = 0; i < 4; ++i) v[i] ^= v[i] >> 16;
        v[0] += v[1] * v[3]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1]; v[3] += v[1] * v[2];
        for (int i = 0; i < 4; ++i) out[i] = toFloat(v[i]);
    }
    /* the keyed order (stands in for Random::shuffle, ldsampler.cpp:163,186) and the scrambles: words of pcg4d(pixel, dim, 'LD', seed) */
    static uint32_t permute(uint32_t i, uint32_t mask, uint32_t key) {
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
    static uint32_t permuteAny(uint32_t i, uint32_t l, uint32_t key) {      /* the same for a domain of any size: cycle walking */
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
    Point2 ldArrayPoint(uint32_t a, uint32_t idx, uint32_t total) const {
        uint32_t v[4] = { m_pixel, 0x100u + a, 0x4c44u, m_seed };
        for (int i = 0; i < 4; ++i) v[i] = v[i] * 1664525u + 1013904223u;
        v[0] += v[1] * v[3]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1]; v[3] += v[1] * v[2];
        for (int i = 0; i < 4; ++i) v[i] ^= v[i] >> 16;
        v[0] += v[1] * v[3]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1]; v[3] += v[1] * v[2];
        const uint32_t i = permuteAny(idx, total, v[0]);
        return Point2(radicalInverse2Single(i, v[1]), sobol2Single(i, v[2]));
    }
    Point2 ldPoint(uint32_t dim) const {
        uint32_t v[4] = { m_pixel, dim, 0x4c44u, m_seed };
        for (int i = 0; i < 4; ++i) v[i] = v[i] * 1664525u + 1013904223u;
        v[0] += v[1] * v[3]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1]; v[3] += v[1] * v[2];
        for (int i = 0; i < 4; ++i) v[i] ^= v[i] >> 16;
        v[0] += v[1] * v[3]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1]; v[3] += v[1] * v[2];
        const uint32_t i = permute((uint32_t) m_sampleIndex & m_ldMask, m_ldMask, v[0]);
        return Point2(radicalInverse2Single(i, v[1]), sobol2Single(i, v[2]));
    }
    bool m_ld; uint32_t m_ldMask;
    bool m_stratified; uint32_t m_stRes;
    uint32_t m_seed, m_pixel, m_call2D, m_call1D, m_rrFirst;
    int m_width;
    bool m_direct;
    size_t m_emitterSamples, m_bsdfSamples;
};

MTS_IMPLEMENT_CLASS_S(CtrSampler, false, Sampler)
MTS_EXPORT_PLUGIN(CtrSampler, "Counter-based parity sampler (test infrastructure)");
MTS_NAMESPACE_END
