/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see o_math.h).
 *
 * o_direct.h: MIDirectIntegrator::Li restated from src/integrators/direct/direct.cpp:149-312 (top-level query:
 * rRec.depth == 1, no adaptive query, rRec.type == ERadiance), with the control flow, the random-number
 * consumption order and the operation order of the reference.  Subsurface scattering is outside the scope.
 */
#pragma once
#include "o_path.h"

namespace orc {

struct DirectParams {
    size_t emitterSamples = 1, bsdfSamples = 1;      /* direct.cpp:96-101 (shadingSamples sets both) */
    bool strictNormals = false, hideEmitters = false;
    Float weightBSDF, weightLum, fracBSDF, fracLum;
    void configure() {                               /* direct.cpp:130-138 */
        if (emitterSamples + bsdfSamples == 0) throw std::runtime_error("direct: emitterSamples + bsdfSamples must be > 0");   /* Assert, direct.cpp:107 */
        size_t sum = emitterSamples + bsdfSamples;
        weightBSDF = 1 / (Float) bsdfSamples;
        weightLum = 1 / (Float) emitterSamples;
        fracBSDF = bsdfSamples / (Float) sum;
        fracLum = emitterSamples / (Float) sum;
    }
};

inline Spectrum directLi(const Scene &scene, const DirectParams &dp, const Ray &r, SampleSource &smp,
                         Float &alpha, PathCounters *pc, const Vec3 &rxDirection, const Vec3 &ryDirection) {
    BSDF bsdfs(scene);
    Intersection its;
    Ray ray(r);
    Spectrum Li(0.0f);
    if (pc) { pc->samples++; pc->pathVertices++; }

    /* direct.cpp:157-165 */
    bool hit = scene.rayIntersect(ray, its, pc);
    alpha = hit ? 1.0f : 0.0f;                       /* records.inl:117-144 */
    if (!hit) {
        if (!dp.hideEmitters)
            return scene.evalEnvironment(ray, rxDirection, ryDirection);
        return Spectrum(0.0f);
    }

    /* direct.cpp:168-169 */
    if (scene.isEmitter(its) && !dp.hideEmitters)
        Li += scene.Le(its, -ray.d);

    /* its.getBSDF(ray), records.inl:69-75 */
    const Material &bsdf = scene.bsdfOf(its);
    if (scene.usesRayDifferentials(bsdf))
        Scene::computePartials(its, ray.o, rxDirection, ryDirection);
    bsdfs.its = &its;

    /* direct.cpp:177-190 */
    if (dp.strictNormals && dot(ray.d, its.geoFrame.n) * Frame::cosTheta(its.wi) >= 0)
        return Li;

    /* ---- emitter sampling, direct.cpp:195-247 ---- */
    const size_t numDirectSamples = dp.emitterSamples, numBSDFSamples = dp.bsdfSamples;
    const Float fracLum = dp.fracLum, fracBSDF = dp.fracBSDF, weightLum = dp.weightLum, weightBSDF = dp.weightBSDF;

    smp.beginDirectArray(0, numDirectSamples);       /* next2DArray / nextSample2D, direct.cpp:212-216 */

    DirectSamplingRecord dRec;
    scene.initDirectRecord(dRec, its);
    if (bsdf.smooth) {
        for (size_t i = 0; i < numDirectSamples; ++i) {
            Spectrum value = scene.sampleEmitterDirect(dRec, smp.directSample(0, i, numDirectSamples), pc);
            if (!value.isZero()) {
                const Vec3 wo = its.toLocal(dRec.d);
                const Spectrum bsdfVal = bsdfs.eval(bsdf, its.wi, wo);
                if (!bsdfVal.isZero() && (!dp.strictNormals || dot(its.geoFrame.n, dRec.d) * Frame::cosTheta(wo) > 0)) {
                    /* every emitter of this scope is on a surface (area lights; the environment's bounding sphere) */
                    Float bsdfPdf = bsdfs.pdf(bsdf, its.wi, wo);
                    const Float weight = miWeight(dRec.pdf * fracLum, bsdfPdf * fracBSDF) * weightLum;
                    Li += value * bsdfVal * weight;
                }
            }
        }
    }

    /* ---- BSDF sampling, direct.cpp:249-307 ---- */
    smp.beginDirectArray(1, numBSDFSamples);         /* direct.cpp:251-255 */

    Intersection bsdfIts;
    for (size_t i = 0; i < numBSDFSamples; ++i) {
        Float bsdfPdf = 0;
        BSDFSamplingRecord bRec;
        bRec.wi = its.wi; bRec.eta = 1.0f; bRec.sampledDelta = false;
        Spectrum bsdfVal = bsdfs.sample(bsdf, bRec, bsdfPdf, smp.directSample(1, i, numBSDFSamples));
        if (bsdfVal.isZero())
            continue;

        const Vec3 wo = its.toWorld(bRec.wo);
        Float woDotGeoN = dot(its.geoFrame.n, wo);
        if (dp.strictNormals && woDotGeoN * Frame::cosTheta(bRec.wo) <= 0)
            continue;

        Ray bsdfRay(its.p, wo);
        Spectrum value;
        if (scene.rayIntersect(bsdfRay, bsdfIts, pc)) {
            if (!scene.isEmitter(bsdfIts))
                continue;
            value = scene.Le(bsdfIts, -bsdfRay.d);
            scene.setQuery(dRec, bsdfRay, bsdfIts);
        } else {
            /* no BSDF of this scope samples a null (ENull) interaction, so hideEmitters does not apply here */
            if (scene.envEmitter < 0)
                continue;
            value = scene.evalEnvironment(bsdfRay);
            if (!scene.fillDirectSamplingRecord(dRec, bsdfRay))
                continue;
        }

        const Float lumPdf = (!bRec.sampledDelta) ? scene.pdfEmitterDirect(dRec) : 0;
        const Float weight = miWeight(bsdfPdf * fracBSDF, lumPdf * fracLum) * weightBSDF;
        Li += value * bsdfVal * weight;
    }
    return Li;
}

} // namespace orc
