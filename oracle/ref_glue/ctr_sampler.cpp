/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * ctr_sampler.cpp: a Sampler plugin FOR THE REFERENCE ("ctr", oracle/_ref/plugins/ctr.so) that hands the reference's own
 * integrators the counter-based parity stream of DESIGN.md 3.5 -- pcg4d(pixel, sampleIndex, block, seed) -- so that the
 * reference's `path` / `direct` and the GPU consume THE SAME random numbers and their images can be compared directly
 * (tests/test_gpu_dropin.py).  A sampler only sees next1D / next2D calls, so the block a call belongs to is inferred from
 * the call order of MIPathTracer::Li (path.cpp:119-300): pixel jitter, then per path vertex [emitter 2D] [BSDF 2D]
 * [Russian-roulette 1D].  The emitter sample is skipped by the integrator for non-smooth BSDFs (path.cpp:174), which a sampler
 * cannot see.  Without further information the plugin assumes that every BSDF has a smooth component (diffuse,
 * roughconductor, two-sided wrappers of those -- BASELINE configs C2 and C3).  For scenes with dielectrics the caller passes
 * `smoothMasks` (property of type data): per (pixel, sample) a 32-bit word whose bit d-1 says whether the BSDF at path
 * vertex d is smooth, recorded by the oracle on the same stream (OracleScene.smooth_masks).  A wrong mask cannot make a
 * wrong render look right: the reference would then consume other numbers than the oracle and the comparison fails.
 * `mode` = "path" or "direct"; for `direct` the sample counts tell which 2D calls are single samples (direct.cpp:212-216,251-255).
 */
#include <mitsuba/render/sampler.h>
#include <mitsuba/render/scene.h>

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
        m_masks = props.hasProperty("smoothMasks") ? (const uint32_t *) props.getData("smoothMasks").ptr : NULL;
        m_pixel = 0; m_call2D = 0; m_depth = 1; m_rrDepth = 1; m_phase = 0;
    }
    CtrSampler(Stream *stream, InstanceManager *manager) : Sampler(stream, manager) { Log(EError, "ctr sampler: not serializable"); }

    ref<Sampler> clone() {
        ref<CtrSampler> s = new CtrSampler(getProperties());
        s->m_sampleCount = m_sampleCount; s->m_seed = m_seed; s->m_width = m_width; s->m_direct = m_direct;
        s->m_emitterSamples = m_emitterSamples; s->m_bsdfSamples = m_bsdfSamples; s->m_masks = m_masks;
        for (size_t i = 0; i < m_req1D.size(); ++i) s->request1DArray(m_req1D[i]);
        for (size_t i = 0; i < m_req2D.size(); ++i) s->request2DArray(m_req2D[i]);
        return s.get();
    }

    void setFilmResolution(const Vector2i &res, bool blocked) { if (m_width == 0) m_width = res.x; }

    void generate(const Point2i &offset) {
        m_pixel = (uint32_t) (offset.y * m_width + offset.x);           /* crop-relative pixel, row-major: the parity stream's key */
        /* sample arrays of `direct`: array a of sample j, entry i = block 1 + i of that sample; the emitter array (if
           requested, direct.cpp:140-146) comes first and takes .xy, the BSDF array .zw */
        for (size_t a = 0; a < m_req2D.size(); ++a) {
            const bool emitter = (a == 0 && m_emitterSamples > 1);
            for (size_t j = 0; j < m_sampleCount; ++j)
                for (size_t i = 0; i < m_req2D[a]; ++i) {
                    float f[4]; block((uint32_t) j, 1 + (uint32_t) i, f);
                    m_sampleArrays2D[a][j * m_req2D[a] + i] = emitter ? Point2(f[0], f[1]) : Point2(f[2], f[3]);
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
        if (call == 0) { block((uint32_t) m_sampleIndex, 0, f); return Point2(f[0], f[1]); }        /* integrator.cpp:171 */
        if (m_direct) {
            /* direct.cpp:212-216: a single emitter sample (also drawn when emitterSamples == 0); :251-255 the same for the BSDF */
            const bool emitterCall = (m_emitterSamples <= 1) && call == 1;
            block((uint32_t) m_sampleIndex, 1, f);
            return emitterCall ? Point2(f[0], f[1]) : Point2(f[2], f[3]);
        }
        /* path.cpp:176 (emitter sample of vertex m_depth, only for smooth BSDFs), :209 (BSDF sample of the same vertex) */
        bool emitterCall;
        if (m_phase == 1) { emitterCall = false; }                                  /* the BSDF sample after an emitter sample */
        else {
            const bool smooth = !m_masks || m_depth > 32 ||
                ((m_masks[(size_t) m_pixel * m_sampleCount + m_sampleIndex] >> (m_depth - 1)) & 1u);
            emitterCall = smooth;
        }
        block((uint32_t) m_sampleIndex, 1 + 2 * (m_depth - 1), f);
        if (emitterCall) { m_phase = 1; return Point2(f[0], f[1]); }
        m_rrDepth = m_depth; ++m_depth; m_phase = 0;
        return Point2(f[2], f[3]);
    }
    Float next1D() {                                                                                /* path.cpp:283, Russian roulette */
        float f[4]; block((uint32_t) m_sampleIndex, 2 + 2 * (m_rrDepth - 1), f);
        return f[0];
    }

    std::string toString() const { return "CtrSampler[]"; }
    MTS_DECLARE_CLASS()
private:
    void beginSample() { m_call2D = 0; m_depth = 1; m_rrDepth = 1; m_phase = 0; }
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
    uint32_t m_seed, m_pixel, m_call2D, m_depth, m_rrDepth, m_phase;
    const uint32_t *m_masks;
    int m_width;
    bool m_direct;
    size_t m_emitterSamples, m_bsdfSamples;
};

MTS_IMPLEMENT_CLASS_S(CtrSampler, false, Sampler)
MTS_EXPORT_PLUGIN(CtrSampler, "Counter-based parity sampler (test infrastructure)");
MTS_NAMESPACE_END
