/*
 * k_clip.h -- the scene-box clip + adaptive epsilon of Scene::rayIntersect (src/librender/skdtree.cpp:112-142 closest hit, :207-226 shadow
 * rays) and the reciprocal direction of the slab tests.  Included by every translation unit through phip_common.h: the ray kernels clip
 * the rays they fetch (BVH4 path, k_mega, phip_trace); on scenes that use the wide tree the SHADING kernels clip the rays they make
 * (DevScene::preclip), where every lane of the wave has one, and k_rays_w fetches (o, mint' | d, maxt') ready to traverse.
 */
#pragma once

/* Reciprocal direction for the slab tests.  A zero (or denormal) component must not become +-inf: the slab form
 * fma(plane, rcp, -o * rcp) would then evaluate inf - inf = NaN for EVERY box, fminf/fmaxf drop the NaN and the node
 * is rejected -- an axis-aligned ray missed the whole tree (the reference handles d == 0 explicitly, aabb.h / skdtree.cpp:
 * the ray is inside the slab iff min <= o <= max).  A finite +-2^90 keeps the arithmetic meaningful: inside the slab the
 * two plane distances are -huge / +huge (no constraint), outside both have the same sign and |t| >= 2^90 * distance
 * exceeds every finite maxt; boxes are padded (bvh.h), so the rounding of o * rcp cannot flip a decision. */
DV float slabRcp(float d) {
    return fabsf(d) < 8.0779357e-28f /* 2^-90 */ ? copysignf(1.2379400e27f /* 2^90 */, d) : 1.0f / d;
}
/* ... from a reciprocal that is already there (the scene-box clip divides by the same components) */
DV float slabRcpFrom(float d, float rcp) {
    return fabsf(d) < 8.0779357e-28f ? copysignf(1.2379400e27f, d) : rcp;
}

/* scene-box clip + adaptive epsilon, src/librender/skdtree.cpp:112-142 (closest) / :207-226 (shadow).  Also hands out the
   reciprocal direction for the slab tests: the clip divides by the same three components (an IEEE division is ~12 instructions;
   a ray used to pay six of them). */
/* one axis of the clip (skdtree.cpp:118-133): false = the ray misses the slab */
__device__ __forceinline__ bool clipAxis(float origin, float dir, float rcp, float minVal, float maxVal, float &nearT, float &farT) {
    if (dir == 0) return !(origin < minVal || origin > maxVal);
    float t1 = (minVal - origin) * rcp;
    float t2 = (maxVal - origin) * rcp;
    if (t1 > t2) { float tmp = t1; t1 = t2; t2 = tmp; }
    nearT = smax(t1, nearT);
    farT = smin(t2, farT);
    return nearT <= farT;
}
template <bool SHADOW>
__device__ __forceinline__ bool clipToScene(const DevScene &S, const V3 &o, const V3 &d, float rayMint, float rayMaxt,
                                            float &mint, float &maxt, V3 &slab) {
    float nearT = -INFINITY, farT = INFINITY;
    /* (scalars, not float[3] arrays walked by an unrolled loop: the arrays left a 36-byte private segment behind in every kernel that clips --
       allocated per wave at launch although no instruction touched it) */
    const float rx = 1.0f / d.x, ry = 1.0f / d.y, rz = 1.0f / d.z;       /* (inf for a zero component: not used by the clip then) */
    slab = V3(slabRcpFrom(d.x, rx), slabRcpFrom(d.y, ry), slabRcpFrom(d.z, rz));
    if (!clipAxis(o.x, d.x, rx, S.sceneMin[0], S.sceneMax[0], nearT, farT)) return false;
    if (!clipAxis(o.y, d.y, ry, S.sceneMin[1], S.sceneMax[1], nearT, farT)) return false;
    if (!clipAxis(o.z, d.z, rz, S.sceneMin[2], S.sceneMax[2], nearT, farT)) return false;
    mint = nearT; maxt = farT;
    float rayMinT = rayMint;
    if (rayMinT == PT_EPSILON) {
        float m = smax(smax(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
        if (!SHADOW) m = smax(m, PT_EPSILON);
        rayMinT *= m;
    }
    if (rayMinT > mint) mint = rayMinT;
    if (rayMaxt < maxt) maxt = rayMaxt;
    return maxt > mint;
}

/* The same statement without control flow (k_mega): the early returns of the loop above are divergent branches -- twenty exec-mask
   manipulations around three short blocks -- for a test that fails for no ray of a closed scene.  Every axis is evaluated, the verdicts
   are and-ed: nearT only grows and farT only shrinks along the axes, so "nearT <= farT after every axis" and the early return say the same;
   the arithmetic on a ray that passes is the loop's, operation for operation. */
template <bool SHADOW>
__device__ __forceinline__ bool clipToSceneSel(const DevScene &S, const V3 &o, const V3 &d, float rayMint, float rayMaxt,
                                               float &mint, float &maxt, V3 &slab) {
    float nearT = -INFINITY, farT = INFINITY;
    const float oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
    const float rr[3] = { 1.0f / d.x, 1.0f / d.y, 1.0f / d.z };
    slab = V3(slabRcpFrom(d.x, rr[0]), slabRcpFrom(d.y, rr[1]), slabRcpFrom(d.z, rr[2]));
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float origin = oo[i], minVal = S.sceneMin[i], maxVal = S.sceneMax[i];
        const bool zero = dd[i] == 0;
        const float a = (minVal - origin) * rr[i], b = (maxVal - origin) * rr[i];
        const bool sw = a > b;
        const float t1 = sw ? b : a, t2 = sw ? a : b;
        const float n2 = smax(t1, nearT), f2 = smin(t2, farT);
        nearT = zero ? nearT : n2; farT = zero ? farT : f2;
        ok = ok & (zero ? !((origin < minVal) | (origin > maxVal)) : (nearT <= farT));
    }
    mint = nearT; maxt = farT;
    float rayMinT = rayMint;
    {
        float m = smax(smax(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
        if (!SHADOW) m = smax(m, PT_EPSILON);
        rayMinT = (rayMinT == PT_EPSILON) ? rayMinT * m : rayMinT;
    }
    mint = (rayMinT > mint) ? rayMinT : mint;
    maxt = (rayMaxt < maxt) ? rayMaxt : maxt;
    return ok & (maxt > mint);
}

/* the same with the kind of the ray as a per-lane flag (the kernels that trace closest-hit and any-hit rays in one loop) */
__device__ __forceinline__ bool clipToSceneRT(const DevScene &S, const V3 &o, const V3 &d, float rayMint, float rayMaxt,
                                              float &mint, float &maxt, bool shadow, V3 &slab) {
    float nearT = -INFINITY, farT = INFINITY;
    const float oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
    const float rr[3] = { 1.0f / d.x, 1.0f / d.y, 1.0f / d.z };
    slab = V3(slabRcpFrom(d.x, rr[0]), slabRcpFrom(d.y, rr[1]), slabRcpFrom(d.z, rr[2]));
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float origin = oo[i], minVal = S.sceneMin[i], maxVal = S.sceneMax[i];
        if (dd[i] == 0) {
            if (origin < minVal || origin > maxVal) return false;
        } else {
            const float rcp = rr[i];
            float t1 = (minVal - origin) * rcp;
            float t2 = (maxVal - origin) * rcp;
            if (t1 > t2) { float tmp = t1; t1 = t2; t2 = tmp; }
            nearT = smax(t1, nearT);
            farT = smin(t2, farT);
            if (!(nearT <= farT)) return false;
        }
    }
    mint = nearT; maxt = farT;
    float rayMinT = rayMint;
    if (rayMinT == PT_EPSILON) {
        float m = smax(smax(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
        if (!shadow) m = smax(m, PT_EPSILON);               /* skdtree.cpp:124 vs :215 */
        rayMinT *= m;
    }
    if (rayMinT > mint) mint = rayMinT;
    if (rayMaxt < maxt) maxt = rayMaxt;
    return maxt > mint;
}

/* ---- DevScene::preclip: rays enter the pool already clipped.  A closest-hit ray is (o, mint' | d, maxt'), a shadow-queue entry
 *      (o, maxt' | d, mint' | contribution, bits(sample id)); a ray that misses the scene box or has an empty interval is stored with
 *      maxt' < mint' (the ray kernel reports a miss / an unoccluded entry without traversing).  The clip is the reference's
 *      arithmetic (IEEE divisions), so mint' / maxt' are the values the ray kernels used to compute themselves. ---- */
__device__ __forceinline__ void preclipRay(const DevScene &S, float4 &ro, float4 &rd) {
    float mint, maxt; V3 slab;
    const bool ok = clipToScene<false>(S, V3(ro.x, ro.y, ro.z), V3(rd.x, rd.y, rd.z), ro.w, rd.w, mint, maxt, slab);
    ro.w = ok ? mint : 0.0f; rd.w = ok ? maxt : -1.0f;
}
__device__ __forceinline__ void preclipShadow(const DevScene &S, float4 &e0, float4 &e1) {
    float mint, maxt; V3 slab;
    const bool ok = clipToScene<true>(S, V3(e0.x, e0.y, e0.z), V3(e1.x, e1.y, e1.z), PT_EPSILON, e0.w, mint, maxt, slab);
    e0.w = ok ? maxt : -1.0f; e1.w = ok ? mint : 0.0f;
}
