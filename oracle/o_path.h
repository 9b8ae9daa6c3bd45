/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see o_math.h).
 *
 * o_path.h: MIPathTracer::Li restated from src/integrators/path/path.cpp:119-300 with the
 * control flow, random-number consumption order and operation order of the reference.
 * The `constant` environment emitter is supported (path.cpp:136-143, 233-248); subsurface and media are
 * outside the path's scope.
 */
#pragma once
#include <cstdio>
#include "o_bsdf.h"

namespace orc {

struct IntegratorParams {
    int maxDepth = -1, rrDepth = 5;
    bool strictNormals = false, hideEmitters = false;
};

inline Float miWeight(Float pdfA, Float pdfB) { /* path.cpp:296-300 */
    pdfA *= pdfA;
    pdfB *= pdfB;
    return pdfA / (pdfA + pdfB);
}

/* Returns Li; alpha as set by RadianceQueryRecord::rayIntersect (records.inl:117-144). */
inline Spectrum pathLi(const Scene &scene, const IntegratorParams &ip, const Ray &r, SampleSource &smp,
                       Float &alpha, PathCounters *pc, const Vec3 *rxDirection = nullptr, const Vec3 *ryDirection = nullptr) {
    BSDF bsdfs(scene);
    Intersection its;
    Ray ray(r);
    Spectrum Li(0.0f);
    bool scattered = false;
    int depth = 1;                       /* integrator.h:218-224 */
    bool emittedRadiance = true;         /* rRec.type & EEmittedRadiance (ERadiance initially) */

    scene.rayIntersect(ray, its, pc);
    alpha = its.isValid() ? 1.0f : 0.0f;
    ray.mint = ORC_EPSILON;

    Spectrum throughput(1.0f);
    Float eta = 1.0f;

    while (depth <= ip.maxDepth || ip.maxDepth < 0) {
        if (!its.isValid()) {
            /* path.cpp:136-143 */
            /* only the camera ray can get here with the emission flag set, and it is the only ray with differentials */
            if (emittedRadiance && (!ip.hideEmitters || scattered))
                Li += throughput * ((rxDirection && ryDirection && depth == 1) ? scene.evalEnvironment(ray, *rxDirection, *ryDirection)
                                                                                : scene.evalEnvironment(ray));
            break;
        }

        const Material &bsdf = scene.bsdfOf(its);
        /* Intersection::getBSDF(ray), records.inl:69-75: UV partials for BSDFs that filter textures -- only the camera
           ray carries differentials (path.cpp:229 continues with a plain Ray) */
        if (depth == 1 && rxDirection && ryDirection && scene.usesRayDifferentials(bsdf))
            Scene::computePartials(its, ray.o, *rxDirection, *ryDirection);
        bsdfs.its = &its;
        if (pc && bsdf.smooth && depth <= 32) pc->smoothMask |= 1u << (depth - 1);
        if (pc && pc->verbose)
            fprintf(stderr, "  vertex %d: prim %u t %.9g (%08x) p (%.9g %.9g %.9g) Li (%.9g %.9g %.9g) thr (%.9g %.9g %.9g) shadow rays so far %llu\n", depth, its.primIndex, its.t,
                    (unsigned) *(const uint32_t *) &its.t, its.p.x, its.p.y, its.p.z, Li[0], Li[1], Li[2], throughput[0], throughput[1], throughput[2], (unsigned long long) pc->shadowRays);

        if (scene.isEmitter(its) && emittedRadiance && (!ip.hideEmitters || scattered))
            Li += throughput * scene.Le(its, -ray.d);

        if ((depth >= ip.maxDepth && ip.maxDepth > 0)
            || (ip.strictNormals && dot(ray.d, its.geoFrame.n) * Frame::cosTheta(its.wi) >= 0))
            break;

        /* ---- direct illumination sampling, path.cpp:172-200 ---- */
        DirectSamplingRecord dRec;
        scene.initDirectRecord(dRec, its);

        if (bsdf.smooth) {
            Spectrum value = scene.sampleEmitterDirect(dRec, smp.emitterSample(depth), pc);
            if (!value.isZero()) {
                const Vec3 wo = its.toLocal(dRec.d);
                const Spectrum bsdfVal = bsdfs.eval(bsdf, its.wi, wo);
                if (!bsdfVal.isZero() && (!ip.strictNormals || dot(its.geoFrame.n, dRec.d) * Frame::cosTheta(wo) > 0)) {
                    /* area emitters are on a surface and sampled w.r.t. solid angle */
                    Float bsdfPdf = (dRec.measure == ESolidAngle) ? bsdfs.pdf(bsdf, its.wi, wo) : 0;
                    Float weight = miWeight(dRec.pdf, bsdfPdf);
                    Li += throughput * value * bsdfVal * weight;
                }
            }
        }

        /* ---- BSDF sampling, path.cpp:207-226 ---- */
        Float bsdfPdf = 0;
        BSDFSamplingRecord bRec;
        bRec.wi = its.wi; bRec.eta = 1.0f; bRec.sampledDelta = false;
        Spectrum bsdfWeight = bsdfs.sample(bsdf, bRec, bsdfPdf, smp.bsdfSample(depth, bsdf.smooth));
        if (bsdfWeight.isZero())
            break;

        scattered |= true;               /* sampledType != ENull always on this path */

        const Vec3 wo = its.toWorld(bRec.wo);
        Float woDotGeoN = dot(its.geoFrame.n, wo);
        if (ip.strictNormals && woDotGeoN * Frame::cosTheta(bRec.wo) <= 0)
            break;

        bool hitEmitter = false;
        Spectrum value;

        ray = Ray(its.p, wo);
        if (scene.rayIntersect(ray, its, pc)) {
            if (scene.isEmitter(its)) {
                value = scene.Le(its, -ray.d);
                scene.setQuery(dRec, ray, its);
                hitEmitter = true;
            }
        } else {
            /* path.cpp:233-248 */
            if (scene.envEmitter >= 0) {
                if (ip.hideEmitters && !scattered)
                    break;
                value = scene.evalEnvironment(ray);
                if (!scene.fillDirectSamplingRecord(dRec, ray))
                    break;
                hitEmitter = true;
            } else {
                break;
            }
        }

        throughput *= bsdfWeight;
        eta *= bRec.eta;

        if (hitEmitter) {
            const Float lumPdf = (!bRec.sampledDelta) ? scene.pdfEmitterDirect(dRec) : 0;
            Li += throughput * value * miWeight(bsdfPdf, lumPdf);
        }

        /* ---- indirect illumination, path.cpp:270-286 ---- */
        if (!its.isValid())
            break;
        emittedRadiance = false;         /* rRec.type = ERadianceNoEmission */

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
