/*
 * k_shade_direct.h -- k_shade_direct: MIDirectIntegrator::Li (src/integrators/direct/direct.cpp:149-312) on the wavefront
 * Included by phip_shade.hip after k_shade.h, whose LDS staging and epilogue it shares.
 *
 * The camera vertex stays in its slot for emitterSamples + bsdfSamples "rounds" (launches); state.depth - 1 = round.
 *   round r < E:                    emitter sample r           -> one shadow-queue entry (as in k_shade)
 *   round r = bs0 + i (i < B):      BSDF sample i              -> the slot's closest-hit query
 *   round r = bs0 + i + 1:          resolves BSDF sample i     (emitter hit / environment miss, MIS against the emitter density)
 * with bs0 = max(E, 1) - 1: the last emitter sample and the first BSDF sample share a round.  Radiance is therefore added in
 * the reference's order -- Le, emitter samples 0..E-1, BSDF samples 0..B-1 -- which float addition needs for a bit-identical
 * sum; the default shadingSamples = 1 costs two rounds per camera sample.  Rounds without a BSDF ray set F_NOTRACE.
 * Per slot it keeps: camHit (the camera ray's hit record; its direction is recomputed from the sample's counter stream),
 * thr = (bsdfVal, bsdfPdf) of the BSDF sample in flight, F_PREV_DELTA, F_SCATTERED (= a BSDF ray is in flight).
 */
/* One round of MIDirectIntegrator::Li on the register image of a camera sample -- the statement both device paths compile (round 6): the wavefront kernel
   k_shade_direct below (state streamed through the pool, radiance accumulated in L[id]) and the fused kernel (k_mega<.., DIRECT>: a lane owns the camera sample, state and
   accumulator live in registers), so that both produce the same bits.
     in:   v.hit / v.rayD = the record and direction of the ray that was traced since the last round (the camera ray in the first round, BSDF sample i - 1 afterwards),
           v.thr = (bsdfVal, bsdfPdf) of that BSDF sample, v.state = (round + 1) | flags, camHit = the camera ray's hit record (written in the first round);
     out:  returns true when the sample is complete (vertices = 1); newRay: v.rayO / v.rayD / v.thr hold the next BSDF sample's ray (the caller clips / stores it);
           pushShadow + sh: the round's emitter sample; v.state is updated when the sample goes on (F_NOTRACE: no closest-hit query this round). */
template <int MM, int FEAT, typename LAcc>
__device__ __forceinline__ bool directVertex(const DevScene &S, const EmitterTab &T, const DevMaterial *materials, const RenderConst &rc,
                                             PathVertex &v, float4 &camHit, const LAcc &acc, bool &newRay, bool &pushShadow, ShadowEntry &sh, uint32_t &vertices) {
    constexpr bool ENV = (FEAT & 1) != 0, TEX = (FEAT & 2) != 0, QMC = (FEAT & 8) != 0;
    const float4 hit = v.hit, rd = v.rayD, thr4 = v.thr;
    const uint4 info = make_uint4(v.id, v.pixel, v.k, v.state);
    newRay = false; pushShadow = false;
    float4 sh0 = make_float4(0, 0, 0, 0), sh1 = sh0, sh2 = sh0;
    {
        const int round = (int) (info.w & DEPTH_MASK) - 1;
        uint32_t flags = info.w & ~DEPTH_MASK;
        const uint32_t id = info.x;
        const int E = rc.emitterSamples, B = rc.bsdfSamples;
        const int bs0 = (E > 1 ? E : 1) - 1;
        const int lastIssue = B > 0 ? bs0 + B - 1 : E - 1;
        const bool first = (flags & F_FIRST) != 0;
        bool terminate = false, haveAdd = false;
        float4 l = make_float4(0, 0, 0, 0);

        /* the camera ray of this sample: direction, differentials (integrator.cpp:171-181) */
        const uint32_t px = info.y % (uint32_t) S.film.width, py = info.y / (uint32_t) S.film.width;
        const V2 hc = filmJitter<QMC>(rc, id, info.y, info.z, (uint32_t) S.film.width);      /* (sequence samplers: read back, the regeneration stored it) */
        const float sx = (float) px + hc.x, sy = (float) py + hc.y;
        V3 camD;
        if (first) {
            camD = V3(rd.x, rd.y, rd.z);
            camHit = hit;
        } else {
            V3 o; float mint, maxt;
            cameraRay(S.cam, sx, sy, o, camD, mint, maxt);
        }
        const uint32_t prim = pm_to_bits(camHit.w);

        if (prim == PHIP_NO_HIT) {                               /* direct.cpp:157-165 (only in round 0) */
            terminate = true;
            haveAdd = true;                                      /* (0,0,0, alpha 0) is written: the sample buffer needs no clear */
            if (ENV && S.envEmitter >= 0 && !rc.hideEmitters) {
                const float *em = emitterRecord(T, (uint32_t) S.envEmitter);
                V3 bg = (pm_to_bits(em[EM_TYPE]) == PHIP_EMITTER_ENVMAP) ? envmapEval(S.env, camD) : rgb(em + EM_RADIANCE);
                if (rc.envFiltered) {
                    V3 rx, ry;
                    cameraRayDifferentials(S.cam, sx, sy, rx, ry);
                    rx = camD + (rx - camD) * rc.diffScaleFactor;
                    ry = camD + (ry - camD) * rc.diffScaleFactor;
                    bg = envmapEvalDiff(S.env, camD, rx, ry);
                }
                l.x = bg.x; l.y = bg.y; l.z = bg.z;                /* alpha stays 0 */
            }
        } else {
            Isect its;
            fillIntersection(S, camD, prim, camHit.y, camHit.z, camHit.x, its);
            if (first) {
                l.w = 1.0f;                                      /* alpha, records.inl:117-144 */
                haveAdd = true;
                if (its.emitter >= 0 && !rc.hideEmitters) {       /* direct.cpp:168-169 */
                    const float *em = emitterRecord(T, (uint32_t) its.emitter);
                    const V3 le = (dot(its.sh.n, -camD) <= 0) ? V3(0.0f) : rgb(em + EM_RADIANCE);
                    l.x += le.x; l.y += le.y; l.z += le.z;
                }
                if (rc.strictNormals && dot(camD, its.geoN) * cosTheta(its.wi) >= 0)     /* direct.cpp:177-190 */
                    terminate = true;
                flags &= ~F_FIRST;
            }
            V3 shD(0.0f), shC(0.0f); float shMaxt = 0;
            if (!terminate) {
                const V3 refN = (its.flags & TS_TRANS_OR_BACK) ? V3(0.0f) : its.sh.n;     /* DirectSamplingRecord(its), records.inl:146-153 */
                BsdfCtx bctx = bsdfResolve(materials, its);
                if (TEX && bctx.textured) {
                    /* its.getBSDF(ray): every query at the camera vertex sees the UV partials of the camera-ray differentials */
                    float dudx = 0, dudy = 0, dvdx = 0, dvdy = 0;
                    V3 rx, ry;
                    cameraRayDifferentials(S.cam, sx, sy, rx, ry);
                    rx = camD + (rx - camD) * rc.diffScaleFactor;
                    ry = camD + (ry - camD) * rc.diffScaleFactor;
                    const float *cw = S.cam.c2w;
                    computePartials(its, V3(cw[3], cw[7], cw[11]), rx, ry, dudx, dudy, dvdx, dvdy);
                    bsdfTextures(S, bctx, its.uv, true, dudx, dudy, dvdx, dvdy);
                }

                /* ---- the BSDF sample traced since the last round, direct.cpp:273-306 ---- */
                if (flags & F_SCATTERED) {
                    const uint32_t prim2 = pm_to_bits(hit.w);
                    const V3 d2(rd.x, rd.y, rd.z);
                    V3 value(0.0f); bool have = false; float lumPdf = 0;
                    if (prim2 != PHIP_NO_HIT) {
                        Isect its2;
                        fillIntersection(S, d2, prim2, hit.y, hit.z, hit.x, its2);
                        if (its2.emitter >= 0) {
                            const float *em = emitterRecord(T, (uint32_t) its2.emitter);
                            value = (dot(its2.sh.n, -d2) <= 0) ? V3(0.0f) : rgb(em + EM_RADIANCE);
                            /* dRec.setQuery(bsdfRay, bsdfIts), records.inl:170-178 */
                            if (!(flags & F_PREV_DELTA))
                                lumPdf = pdfEmitterDirectDot<ENV>(S, T, (uint32_t) its2.emitter, d2, dot(d2, refN), refN.isZero(), dot(d2, its2.sh.n), its2.t);
                            have = true;
                        }
                    } else if (ENV && S.envEmitter >= 0) {
                        /* no BSDF of this scope samples a null interaction: the hideEmitters clause of direct.cpp:289 never holds */
                        const float *em = emitterRecord(T, (uint32_t) S.envEmitter);
                        value = (pm_to_bits(em[EM_TYPE]) == PHIP_EMITTER_ENVMAP) ? envmapEval(S.env, d2) : rgb(em + EM_RADIANCE);
                        if (envFillDirectRecord(S, its.p, d2)) {
                            if (!(flags & F_PREV_DELTA))
                                lumPdf = pdfEmitterDirectDot<ENV>(S, T, (uint32_t) S.envEmitter, d2, dot(d2, refN), refN.isZero(), 0.0f, 0.0f);
                            have = true;
                        }
                    }
                    if (have) {
                        const float weight = miWeight(thr4.w * rc.fracBSDF, lumPdf * rc.fracLum) * rc.weightBSDF;
                        const V3 c = value * V3(thr4.x, thr4.y, thr4.z) * weight;
                        l = acc.load(id);
                        l.x += c.x; l.y += c.y; l.z += c.z;
                        haveAdd = true;
                    }
                    flags &= ~F_SCATTERED;
                }

                /* ---- emitter sample `round`, direct.cpp:218-247 ---- */
                if (round < E && (its.flags & TS_MF_SMOOTH)) {
                    DirectRec dRec;
                    dRec.ref = its.p; dRec.refN = refN; dRec.pdf = 0; dRec.emitter = -1;
                    const V3 value = sampleEmitterDirect<ENV>(S, T, dRec, streamDirectSample<QMC>(rc, info.y, info.z, 0, (uint32_t) round, (uint32_t) S.film.width));
                    if (dRec.pdf != 0 && !value.isZero()) {
                        const V3 wo = its.sh.toLocal(dRec.d);
                        float bPdf;
                        const V3 bsdfVal = bsdfEvalPdf<MM>(bctx, wo, bPdf);
                        if (!bsdfVal.isZero() && (!rc.strictNormals || dot(its.geoN, dRec.d) * cosTheta(wo) > 0)) {
                            /* every emitter of this scope isOnSurface() */
                            const float weight = miWeight(dRec.pdf * rc.fracLum, bPdf * rc.fracBSDF) * rc.weightLum;
                            shC = value * bsdfVal * weight;
                            shD = dRec.d; shMaxt = dRec.dist * (1 - PT_SHADOW_EPSILON);
                            pushShadow = true;
                        }
                    }
                }

                /* ---- BSDF sample round - bs0, direct.cpp:257-271 ---- */
                bool issued = false;
                const int i = round - bs0;
                if (i >= 0 && i < B) {
                    BSDFSample bs;
                    const V3 bsdfVal = bsdfSample<MM>(bctx, streamDirectSample<QMC>(rc, info.y, info.z, 1, (uint32_t) i, (uint32_t) S.film.width), bs);
                    if (!bsdfVal.isZero()) {
                        const V3 wo = its.sh.toWorld(bs.wo);
                        const float woDotGeoN = dot(its.geoN, wo);
                        if (!(rc.strictNormals && woDotGeoN * cosTheta(bs.wo) <= 0)) {
                            v.rayO = make_float4(its.p.x, its.p.y, its.p.z, PT_EPSILON); v.rayD = make_float4(wo.x, wo.y, wo.z, INFINITY);
                            v.thr = make_float4(bsdfVal.x, bsdfVal.y, bsdfVal.z, bs.pdf);
                            newRay = true;
                            flags = bs.delta ? (flags | F_PREV_DELTA) : (flags & ~F_PREV_DELTA);
                            issued = true;
                        }
                    }
                }
                flags = issued ? ((flags & ~F_NOTRACE) | F_SCATTERED) : (flags | F_NOTRACE);
                if (!issued && round >= lastIssue)
                    terminate = true;
            }
            if (pushShadow) {
                sh0 = make_float4(its.p.x, its.p.y, its.p.z, shMaxt);
                sh1 = make_float4(shD.x, shD.y, shD.z, 0.0f);
                sh2 = make_float4(shC.x, shC.y, shC.z, pm_from_bits(id));
            }
        }
        if (haveAdd) acc.store(id, l);
        sh.e0 = sh0; sh.e1 = sh1; sh.e2 = sh2;
        if (terminate) { vertices = 1; return true; }
        v.state = flags | (uint32_t) (round + 2);
        return false;
    }
}

template <int MM, int FEAT> __global__ __launch_bounds__(BLOCK, SHADE_WAVES) void k_shade_direct(DevScene S, PathPool P, RenderConst rc, float4 *L) {
    constexpr bool QMC = (FEAT & 8) != 0;
    __shared__ uint32_t waveCnt[BLOCK / 64];
    __shared__ __align__(16) float ldsEm[EMITTER_LDS_FLOATS];
    __shared__ DevMaterial ldsMat[MATERIAL_LDS_MAX];
    if (rc.draining && P.blockDead[blockIdx.x]) return;         /* (block-uniform; see k_shade) */
    /* slot state and the LDS tables in ONE round trip (see k_shade) */
    const uint32_t slot = blockIdx.x * BLOCK + threadIdx.x;
    const bool inRange = slot < P.capacity;
    const uint32_t lslot = inRange ? slot : 0u;
    uint4 info = P.info[lslot];
    info.w = P.state[lslot];
    PathVertex v;
    v.hit = P.hit[lslot];
    v.rayD = P.rayD[lslot];
    v.thr = P.thr[lslot];
    float4 camHit = P.camHit[lslot];
    const ShadeTables tab = stageShadeTables(S, ldsEm, ldsMat);
    v.hit.w = pm_from_bits(hitPrim(pm_to_bits(v.hit.w)));       /* (class bits of k_rays_w: k_pool.h) */
    if (!inRange) info = make_uint4(0, 0, 0, 0);
    __syncthreads();                                            /* LDS tables are complete */
    const bool alive = inRange && (info.w & F_ALIVE);
    bool needNew = inRange && !alive && !(info.w & F_DEAD);
    unsigned long long vertices = 0, done = 0;
    bool pushShadow = false, newRay = false;
    ShadowEntry sh; sh.e0 = sh.e1 = sh.e2 = make_float4(0, 0, 0, 0);

    if (alive) {
        v.id = info.x; v.pixel = info.y; v.k = info.z; v.state = info.w;
        v.rayO = make_float4(0, 0, 0, 0); v.mis = make_float2(0, 0);
        const bool first = (info.w & F_FIRST) != 0;
        uint32_t nv = 0;
        const LGlobal acc{ L, P, slot };
        if (directVertex<MM, FEAT>(S, tab.T, tab.materials, rc, v, camHit, acc, newRay, pushShadow, sh, nv)) {
            vertices = nv; done = 1;
            needNew = true;
        } else {
            info.w = v.state;
            P.state[slot] = info.w;
        }
        if (first && pm_to_bits(camHit.w) != PHIP_NO_HIT) P.camHit[slot] = camHit;
        if (newRay) {
            float4 ro = v.rayO, rdn = v.rayD;
            if (S.preclip) preclipRay(S, ro, rdn);
            P.rayO[slot] = ro; P.rayD[slot] = rdn; P.thr[slot] = v.thr;
        }
    }
    shadeEpilogue<QMC>(S, P, rc, waveCnt, slot, inRange, info, alive, needNew, pushShadow, sh.e0, sh.e1, sh.e2, vertices, done);
}
