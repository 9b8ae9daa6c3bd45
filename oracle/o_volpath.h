/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see o_math.h).
 *
 * o_volpath.h: SimpleVolumetricPathTracer::Li restated from src/integrators/path/volpath_simple.cpp:88-318 for a scene WITHOUT participating media
 * (rRec.medium == NULL throughout: the branch :116-164 and the transmittance factors drop out; no BSDF of the path's scope is ENull, so nullChain
 * stays true and `scattered` is set by every bounce).  Control flow, random-number consumption order and operation order of the reference.
 *
 * One deviation, stated: the reference's emitter sample is sampleAttenuatedEmitterDirect (scene.cpp:876-898) -> evalTransmittance (scene.cpp:619-679), whose
 * visibility ray is a CLOSEST-hit query: its adaptive epsilon is Epsilon * max(|o.x|, |o.y|, |o.z|, Epsilon) (skdtree.cpp:124) where the shadow-ray query of
 * `path` has Epsilon * max(|o.x|, |o.y|, |o.z|) (skdtree.cpp:215).  The two differ only for a ray origin within 1e-4 of the world origin on all three axes;
 * oracle and device use the shadow-ray form for both integrators.  Everything else of the two calls is the same arithmetic: the same direction
 * ((p2 - p1) / |p2 - p1| = dRec.d), the same interval, value * (transmittance 1 / emPdf) = value / emPdf (Spectrum divides by multiplying with the reciprocal).
 */
#pragma once
#include "o_path.h"

namespace orc {

inline Spectrum volpathSimpleLi(const Scene &scene, const IntegratorParams &ip, const Ray &r, SampleSource &smp,
                                Float &alpha, PathCounters *pc, const Vec3 *rxDirection = nullptr, const Vec3 *ryDirection = nullptr) {
    BSDF bsdfs(scene);
    Intersection its;
    Ray ray(r);
    Spectrum Li(0.0f);
    bool scattered = false;
    Float eta = 1.0f;
    int depth = 1;
    /* rRec.type: ERadiance at the camera (integrator.cpp:166), then what :239-261 grants */
    bool emitted = true, directSurface = true;

    scene.rayIntersect(ray, its, pc);
    alpha = its.isValid() ? 1.0f : 0.0f;
    ray.mint = ORC_EPSILON;
    Spectrum throughput(1.0f);

    if (ip.maxDepth == 1)                /* :103-104: rRec.type &= EEmittedRadiance */
        directSurface = false;

    while (depth <= ip.maxDepth || ip.maxDepth < 0) {
        if (!its.isValid()) {
            /* :172-183 */
            if (emitted && (!ip.hideEmitters || scattered))
                Li += throughput * ((rxDirection && ryDirection && depth == 1) ? scene.evalEnvironment(ray, *rxDirection, *ryDirection)
                                                                                : scene.evalEnvironment(ray));
            break;
        }

        const Material &bsdf = scene.bsdfOf(its);
        if (depth == 1 && rxDirection && ryDirection && scene.usesRayDifferentials(bsdf))
            Scene::computePartials(its, ray.o, *rxDirection, *ryDirection);
        bsdfs.its = &its;
        if (pc && bsdf.smooth && directSurface && depth <= 32) pc->smoothMask |= 1u << (depth - 1);

        /* :186-188 */
        if (scene.isEmitter(its) && emitted && (!ip.hideEmitters || scattered))
            Li += throughput * scene.Le(its, -ray.d);

        /* :194-198 */
        const Float wiDotGeoN = -dot(its.geoFrame.n, ray.d), wiDotShN = Frame::cosTheta(its.wi);
        if (ip.strictNormals && wiDotGeoN * wiDotShN < 0)
            break;

        /* ---- direct illumination sampling, :207-227 (no multiple importance sampling) ---- */
        const bool requestsEmitterSample = directSurface && bsdf.smooth;
        if (requestsEmitterSample) {
            DirectSamplingRecord dRec;
            scene.initDirectRecord(dRec, its);
            Spectrum value = scene.sampleEmitterDirect(dRec, smp.emitterSample(depth), pc);
            if (!value.isZero()) {
                const Vec3 wo = its.toLocal(dRec.d);
                const Float woDotGeoN = dot(its.geoFrame.n, dRec.d);
                if (!ip.strictNormals || woDotGeoN * Frame::cosTheta(wo) > 0)
                    Li += throughput * value * bsdfs.eval(bsdf, its.wi, wo);
            }
        }

        /* ---- BSDF sampling, :233-237 ---- */
        Float bsdfPdf = 0;
        BSDFSamplingRecord bRec;
        bRec.wi = its.wi; bRec.eta = 1.0f; bRec.sampledDelta = false;
        Spectrum bsdfVal = bsdfs.sample(bsdf, bRec, bsdfPdf, smp.bsdfSample(depth, requestsEmitterSample));
        if (bsdfVal.isZero())
            break;

        /* :239-261 */
        bool nextEmitted = false, nextSurface = false;
        if ((depth + 1 < ip.maxDepth || ip.maxDepth < 0) && directSurface /* (type & EIndirectSurfaceRadiance: the two surface bits travel together) */)
            nextSurface = true;                              /* ERadianceNoEmission: direct + indirect surface radiance */
        if ((depth < ip.maxDepth || ip.maxDepth < 0) && directSurface && bRec.sampledDelta)
            nextEmitted = true;
        if (!nextSurface && !nextEmitted)
            break;
        emitted = nextEmitted; directSurface = nextSurface;

        /* :263-267 */
        const Vec3 wo = its.toWorld(bRec.wo);
        Float woDotGeoN = dot(its.geoFrame.n, wo);
        if (woDotGeoN * Frame::cosTheta(bRec.wo) <= 0 && ip.strictNormals)
            break;

        throughput *= bsdfVal;
        eta *= bRec.eta;

        ray = Ray(its.p, wo);
        scene.rayIntersect(ray, its, pc);
        scattered |= true;

        /* :282-292 (before the loop looks at the new hit) */
        if (depth++ >= ip.rrDepth) {
            Float q = std::min(throughput.max() * eta * eta, (Float) 0.95f);
            if (smp.rrSample(depth - 1) >= q)
                break;
            throughput /= q;
        }
    }
    if (pc) { pc->pathVertices += (uint64_t) depth; pc->samples++; }
    return Li;
}

} // namespace orc
