/*
 * dv_scene.h -- device-resident scene layout and the per-vertex shading functions of path_hip.
 *
 * HBM layout (all arrays are immutable after phip_scene_create):
 *   nodes      float4[8*nNodes]    128-byte BVH4 nodes, SoA over the four children (bvh.h)
 *   tris       float4[3*nTriRefs]  48-byte Wald triangle records in leaf order
 *   triShade   float4[6*nTriangles] 96-byte shading record per GLOBAL triangle id (vertices, leaf
 *                                  BSDF ids, emitter id, flat-shading frame or vertex normals)
 *   materials  DevMaterial[]       80 bytes each; staged in LDS by k_shade when there are few
 *   emitterTab float[]             selection CDF + emitter records + area CDFs (EmitterTab), LDS-staged when small
 *
 * Functions restate (file:line under /root/reference) -- same arithmetic as oracle/, written
 * independently for the device:
 *   include/mitsuba/render/skdtree.h:343-428     fillIntersectionRecord<true>
 *   include/mitsuba/render/triaccel.h:96-158     TriAccel::rayIntersect
 *   include/mitsuba/core/pmf.h:124-188           DiscreteDistribution::sample / sampleReuse
 *   src/librender/scene.cpp:828-852,949-952      sampleEmitterDirect / pdfEmitterDirect
 *   src/emitters/area.cpp:104-109,158-182        AreaLight eval / sampleDirect / pdfDirect
 *   src/librender/shape.cpp:102-126              Shape::sampleDirect / pdfDirect
 *   src/librender/trimesh.cpp:412-423, src/libcore/triangle.cpp:24-59   area sampling
 *   src/bsdfs/{diffuse,dielectric,roughconductor,twosided}.cpp, microfacet.h   the BSDFs
 *   src/sensors/perspective.cpp:271-297          sampleRayDifferential
 */
#pragma once
#include "dv_math.h"
#include "../../include/phip.h"

namespace pt {

struct DevShape {
    uint32_t material; int32_t emitter; uint32_t hasNormals; uint32_t firstTri;
    uint32_t nTris; uint32_t cdfOffset; float invSurfaceArea; uint32_t pad;
};

enum { MF_SMOOTH = 1, MF_TRANS_OR_BACK = 2 };
struct DevMaterial {
    uint32_t type, nested0, nested1, flags;
    float refl[3]; float alphaU;
    float trans[3]; float alphaV;
    float eta[3]; uint32_t distribution;
    float k[3]; uint32_t sampleVisible;
    uint32_t reflTexture;                /* 0 or 1 + id of the bitmap texture of `reflectance` (DIFFUSE) / `specularReflectance` */
    uint32_t alphaUTexture, alphaVTexture;   /* ROUGHCONDUCTOR: 0 or 1 + id of the bitmap texture of alpha / alphaU / alphaV */
    uint32_t transTexture;               /* DIELECTRIC: 0 or 1 + id of the bitmap texture of specularTransmittance */
};

struct DevEmitter { float radiance[3]; float samplingWeight; uint32_t shape; uint32_t pad[3]; };

struct DevCamera {
    float s2c[16];      /* sampleToCamera */
    float c2w[12];      /* camera-to-world, top 3 rows */
    float nearClip, farClip, invResX, invResY;
    float dx[3], dy[3]; /* position differentials on the near plane, perspective.cpp:159-163 */
};

struct DevFilm {
    int width, height;          /* crop window size (pixel coordinates are crop-relative) */
    int blockSize, border;
    float radius, scaleFactor;
    float table[PHIP_FILTER_RESOLUTION + 1];
};

/* `envmap` emitter (src/emitters/envmap.cpp), illumination side: MIP level 0 as float4 texels, the marginal /
   conditional CDFs over luminance * sin(theta) built by the host like EnvironmentMap::configure (envmap.cpp:262-328) */
/* A MIP pyramid (levels as the reference built them, consecutive in a float4 texel array) with its lookup parameters
   and the EWA weight table: the envmap's and every bitmap texture's descriptor (global memory) */
struct DevMipLevels {
    int32_t nLevels;                         /* 1: no pyramid */
    int32_t lw[PHIP_MIP_MAX_LEVELS], lh[PHIP_MIP_MAX_LEVELS];
    uint32_t offset[PHIP_MIP_MAX_LEVELS];    /* first texel of the level (relative to the texel array passed along) */
    uint32_t bcu, bcv, filterType;           /* phip_wrap_mode x 2, phip_filter_type */
    float maxAnisotropy;
    float uvScale[2], uvOffset[2];           /* Texture2D (bitmap textures only) */
    float weightLut[64];                     /* mipmap.h:296-301 */
};

struct DevEnvMap {
    const float4 *texels;                    /* level 0: w * h, rgb + pad; further levels follow */
    const DevMipLevels *levels;              /* read only by camera rays that miss the scene */
    const float *cdfRows, *cdfCols, *rowWeights;
    int32_t w, h;                            /* w == 0: the environment emitter (if any) is not an envmap */
    float scale, normalization, pixelSizeX, pixelSizeY;
    float toWorld[9], toLocal[9];            /* 3x3 parts, row-major (Transform::operator()(Vector), transform.h:175-183) */
};

struct DevScene {
    const float4 *nodes; const float4 *tris; const float4 *triShade;
    uint32_t flatMode;                                      /* layout of flatLeaves: 1 = (min, ref)(max, 0) per leaf; 2 = packed planes + record masks (traverseFlat2) */
    const float4 *flatLeaves; uint32_t nFlatLeaves;         /* k_mega: the leaves of a tree of <= FLAT_LEAVES_MAX leaves as a flat table (k_traverse.h: traverseFlat); 0: walk the BVH4 */
    const uint4 *wnodes; uint32_t wideNodeCache;             /* the compressed 8-wide tree (k_wide_node.h) and -- wtris -- the Wald records in ITS leaf order.  Scenes past the packed leaf
                                                                table (more than 64 records) hold nothing else: tris == wtris, nodes unused; the LDS-resident scenes keep the BVH4-ordered
                                                                records (tris) and the leaf table for k_mega / k_shade_trace beside the wide tree the ray kernels walk (round 6) */
    const float4 *wtris;
    const DevMaterial *materials; uint32_t nMaterials;
    const float *emitterTab; uint32_t emitterTabSize;       /* EmitterTab layout, floats */
    uint32_t nEmitters; float emitterNormalization;
    const float4 *texTexels; const DevMipLevels *textures;   /* bitmap textures: all pyramids in one texel array + one descriptor each */
    uint32_t triShadeStride;             /* float4s per shading record: 6, or 9 when a mesh has texture coordinates */
    int32_t envEmitter; float envCenter[3]; float envRadius;   /* environment emitter (or -1) and its m_sceneBSphere */
    DevEnvMap env;
    int32_t rootRef; uint32_t nTriangles;
    uint32_t stackDepth, nodeCache, triCache;   /* LDS staging plan of the traversal kernels */
    uint32_t preclip;                    /* the shading kernels clip the rays they make against the scene box (k_clip.h); k_rays_w expects it */
    uint32_t shadeSort;                  /* the Wald records carry shade classes and the scene has more than one: k_shade deals slots to lanes by class */
    float sceneMin[3], sceneMax[3];
    DevCamera cam; DevFilm film;
};

/* Shading record of one triangle: 96 B = 6 x float4, indexed by the global triangle id
 * (Intersection::primIndex).  Everything k_shade needs about a hit in ONE dependent fetch instead of
 * the chain index -> vertices -> shape -> material -> nested material:
 *   r0 = (p0, frontLeaf)  r1 = (p1, backLeaf)  r2 = (p2, emitter)
 *   flat shading:   r3 = (face normal, flags)  r4 = (frame.s, -)  r5 = (frame.t, -)   -- constant per triangle,
 *                   computed on the host by the very functions below (same IEEE operations, same bits)
 *   vertex normals: r3 = (n0, flags)           r4 = (n1, -)       r5 = (n2, -)
 * frontLeaf/backLeaf = the one-sided model seen from either side (the twosided adapter's nested ids, or
 * the material itself twice). */
enum { TS_VERTEX_NORMALS = 1, TS_TWOSIDED = 2, TS_MF_SMOOTH = 4, TS_TRANS_OR_BACK = 8, TS_TEXCOORDS = 16 };
#define TRISHADE_FLOAT4S 6
/* (Round 5 measured "material heads": two more float4s per record carrying the leaf BSDF's parameters, so that record and material arrive in ONE round
   trip instead of two.  k_shade got SLOWER -- atrium 63.5 -> 66.4 ms per frame, glass room 119.6 -> 129.7, profiles/r05_gpu_call_a_*: the kernel is bound
   by the number of vector-memory instructions and lines it moves (texture-data path 83 % busy), not by the length of its dependency chain; removed.) */
/* meshes with texture coordinates append three float4s: r6 = (uv0, uv1), r7 = (uv2, dpdu.xy), r8 = (dpdu.z, dpdv) with
   dpdu / dpdv = TriMesh::computeUVTangents (trimesh.cpp:683-735; they replace side1 / side2 in the shading frame,
   skdtree.h:373-380) */
#define TRISHADE_FLOAT4S_UV 9

struct Isect {
    V3 p; Frame sh; V3 geoN; V3 wi; float t; uint32_t prim;
    uint32_t front, back, flags; int32_t emitter;
    V2 uv; V3 dpdu, dpdv;               /* filled for triangles with texture coordinates (TS_TEXCOORDS) */
};

DV V3 ld3(const float4 *a, uint32_t i) { float4 v = a[i]; return V3(v.x, v.y, v.z); }
DV V3 xyz(const float4 &v) { return V3(v.x, v.y, v.z); }

/* face normal of skdtree.h:367-373 (left unnormalised when it is zero) */
DV V3 triFaceNormal(const V3 &side1, const V3 &side2) {
    V3 faceNormal(cross(side1, side2));
    float length = faceNormal.length();
    if (!faceNormal.isZero())
        faceNormal = faceNormal / length;
    return faceNormal;
}
/* computeShadingFrame(n, dpdu = side1), util.cpp:603-608 */
DV void triShadingFrame(const V3 &shN, const V3 &side1, Frame &f) {
    f.n = shN;
    f.s = normalize(side1 - shN * dot(shN, side1));
    f.t = cross(shN, f.s);
}

/* skdtree.h:343-428 with BarycentricPos = true, no UV tangents, no texcoords */
DV void fillIntersection(const DevScene &S, const V3 &rayD, uint32_t prim, float cu, float cv, float t, Isect &its) {
    const float4 *r = S.triShade + (size_t) S.triShadeStride * prim;
    const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4], r5 = r[5];
    const V3 b(1 - cu - cv, cu, cv);
    const V3 p0 = xyz(r0), p1 = xyz(r1), p2 = xyz(r2);
    its.p = p0 * b.x + p1 * b.y + p2 * b.z;
    its.front = pm_to_bits(r0.w); its.back = pm_to_bits(r1.w); its.emitter = (int32_t) pm_to_bits(r2.w);
    its.flags = pm_to_bits(r3.w);
    V3 dpdu = p1 - p0;                                   /* side1, skdtree.h:378-379 */
    if (its.flags & TS_TEXCOORDS) {                      /* vertexTangents + texture coordinates, skdtree.h:373-377,397-404 */
        const float4 r6 = r[6], r7 = r[7], r8 = r[8];
        its.uv = V2(r6.x * b.x + r6.z * b.y + r7.x * b.z, r6.y * b.x + r6.w * b.y + r7.y * b.z);
        its.dpdu = V3(r7.z, r7.w, r8.x); its.dpdv = V3(r8.y, r8.z, r8.w);
        dpdu = its.dpdu;
    }
    if (its.flags & TS_VERTEX_NORMALS) {
        const V3 side1(p1 - p0), side2(p2 - p0);
        V3 faceNormal = triFaceNormal(side1, side2);
        const V3 shN = normalize(xyz(r3) * b.x + xyz(r4) * b.y + xyz(r5) * b.z);
        if (dot(faceNormal, shN) < 0)
            faceNormal = -faceNormal;
        its.geoN = faceNormal;
        triShadingFrame(shN, dpdu, its.sh);
    } else {
        its.geoN = xyz(r3);
        its.sh.n = its.geoN; its.sh.s = xyz(r4); its.sh.t = xyz(r5);
    }
    its.wi = its.sh.toLocal(-rayD);
    its.t = t; its.prim = prim;
}

/* Intersection::computePartials, intersection.cpp:5-76 (rxOrigin = ryOrigin = rayO: pinhole camera) */
DV bool solveLinearSystem2x2(const float a[2][2], const float b[2], float x[2]) {   /* util.cpp:527-539 */
    const float det = a[0][0] * a[1][1] - a[0][1] * a[1][0];
    if (fabsf(det) <= 2.93873587705571876e-39f) return false;      /* RCPOVERFLOW */
    const float inverse = 1.0f / det;
    x[0] = (a[1][1] * b[0] - a[0][1] * b[1]) * inverse;
    x[1] = (a[0][0] * b[1] - a[1][0] * b[0]) * inverse;
    return true;
}
DV void computePartials(const Isect &its, const V3 &rayO, const V3 &rxDirection, const V3 &ryDirection,
                        float &dudx, float &dudy, float &dvdx, float &dvdy) {
    dudx = dvdx = dudy = dvdy = 0.0f;
    if (its.dpdu.isZero() && its.dpdv.isZero()) return;
    const V3 &gn = its.geoN;
    const float pp = dot(gn, its.p), pox = dot(gn, rayO), poy = dot(gn, rayO), prx = dot(gn, rxDirection), pry = dot(gn, ryDirection);
    if (prx == 0 || pry == 0) return;
    const float tx = (pp - pox) / prx, ty = (pp - poy) / pry;
    const float absX = fabsf(gn.x), absY = fabsf(gn.y), absZ = fabsf(gn.z);
    int a0, a1;
    if (absX > absY && absX > absZ) { a0 = 1; a1 = 2; }
    else if (absY > absZ) { a0 = 0; a1 = 2; }
    else { a0 = 0; a1 = 1; }
    float A[2][2], Bx[2], By[2], x[2];
    A[0][0] = its.dpdu[a0]; A[0][1] = its.dpdv[a0];
    A[1][0] = its.dpdu[a1]; A[1][1] = its.dpdv[a1];
    const V3 px = rayO + rxDirection * tx, py = rayO + ryDirection * ty;
    Bx[0] = px[a0] - its.p[a0]; Bx[1] = px[a1] - its.p[a1];
    By[0] = py[a0] - its.p[a0]; By[1] = py[a1] - its.p[a1];
    if (solveLinearSystem2x2(A, Bx, x)) { dudx = x[0]; dvdx = x[1]; } else { dudx = 1; dvdx = 0; }
    if (solveLinearSystem2x2(A, By, x)) { dudy = x[0]; dvdy = x[1]; } else { dudy = 1; }     /* (sic: intersection.cpp:74 assigns dudy twice) */
}

/* triaccel.h:96-158 on a 48-byte record */
DV bool waldIntersect(const float4 &a, const float4 &b, const float4 &c, const V3 &o, const V3 &d,
                      float mint, float maxt, float &u, float &v, float &t) {
    const uint32_t k = pm_to_bits(a.x);
    float o_u, o_v, o_k, d_u, d_v, d_k;
    if (k == 0) { o_u = o.y; o_v = o.z; o_k = o.x; d_u = d.y; d_v = d.z; d_k = d.x; }
    else if (k == 1) { o_u = o.z; o_v = o.x; o_k = o.y; d_u = d.z; d_v = d.x; d_k = d.y; }
    else if (k == 2) { o_u = o.x; o_v = o.y; o_k = o.z; d_u = d.x; d_v = d.y; d_k = d.z; }
    else return false;
    const float n_u = a.y, n_v = a.z, n_d = a.w;
    t = (n_d - o_u * n_u - o_v * n_v - o_k) / (d_u * n_u + d_v * n_v + d_k);
    if (t < mint || t > maxt)
        return false;
    const float hu = o_u + t * d_u - b.x;
    const float hv = o_v + t * d_v - b.y;
    u = hv * b.z + hu * b.w;
    v = hu * c.x + hv * c.y;
    return u >= 0 && v >= 0 && u + v <= 1.0f;
}

/* Closest-hit bookkeeping.  Several triangles can report EXACTLY the same distance (a ray through a shared edge, coincident
   surfaces), and the Wald test accepts t == maxt (triaccel.h:140: `t > maxt` rejects), so which of them a structure reports is
   whichever it tests last -- in the reference that is a property of its kd-tree's leaf order.  To make the answer independent
   of the structure (BVH4 / 8-wide tree, traversal order, spatial splits) the HIGHEST triangle index wins a tie: what a sweep
   over all triangles in index order returns (oracle: set_bruteforce; tests/test_gpu_parity.py, C2 at full size). */
DV bool winsTie(float tt, uint32_t prim, float bestT, uint32_t bestPrim) { return !(tt == bestT && prim < bestPrim); }

/* std::lower_bound over cdf[0..n] + DiscreteDistribution::sample, pmf.h:124-136 */
DV uint32_t cdfSample(const float *cdf, uint32_t nEntries, float sampleValue) {
    /* cdf has nEntries+1 values */
    if (nEntries - 1u < 3u) {                /* 1 .. 3 entries */
        /* Few entries (one or two emitters, an emitter of two triangles -- every area light that is a rectangle): the binary search is a
           chain of DEPENDENT reads (LDS or memory round trips), two chains per NEE sample.  The array is non-decreasing, so
           lower_bound = the number of elements < value: all (at most four) elements are read at once and counted -- the same index.
           (The kernel's tables continue behind the CDF, so element nEntries + 1 .. 3 is readable; it is not counted.) */
        const float c0 = cdf[0], c1 = cdf[1], c2 = nEntries >= 2u ? cdf[2] : 0.0f, c3 = nEntries >= 3u ? cdf[3] : 0.0f;
        const uint32_t lo = (c0 < sampleValue ? 1u : 0u) + (c1 < sampleValue ? 1u : 0u)
                          + ((nEntries >= 2u && c2 < sampleValue) ? 1u : 0u) + ((nEntries >= 3u && c3 < sampleValue) ? 1u : 0u);
        uint32_t index = lo ? lo - 1u : 0u;
        if (index > nEntries - 1u) index = nEntries - 1u;
        /* skip entries of zero probability, as below (cdf[index + 1] - cdf[index] == 0), on the values already read */
        if (index == 0u && 1u < nEntries && c1 - c0 == 0) index = 1u;
        if (index == 1u && 2u < nEntries && c2 - c1 == 0) index = 2u;
        return index;
    }
    uint32_t lo = 0, len = nEntries + 1;
    while (len > 0) {                       /* lower_bound: first element not < value */
        uint32_t half = len >> 1, mid = lo + half;
        if (cdf[mid] < sampleValue) { lo = mid + 1; len = len - half - 1; } else len = half;
    }
    long idx = (long) lo - 1;
    if (idx < 0) idx = 0;
    uint32_t index = (uint32_t) idx;
    if (index > nEntries - 1) index = nEntries - 1;
    while (index < nEntries - 1 && cdf[index + 1] - cdf[index] == 0)
        ++index;
    return index;
}

struct DirectRec {
    V3 p, n, d, ref, refN; float dist, pdf; int emitter; int solidAngle;
};

/* Emitter table: everything direct-illumination sampling looks up, packed into ONE float array so that
 * k_shade can stage it in LDS when it is small (the lookups are a chain of dependent loads:
 * selection CDF -> emitter -> area CDF of its mesh -> triangle):
 *   t[0 .. n]                     emitter selection CDF (pmf.h layout, n + 1 entries)
 *   t[n + 1 + 12 e .. + 11]       emitter e: radiance rgb, samplingWeight, firstTri, nTris, cdfOffset (index into t), 1 / surface area,
 *                                 recOffset (index into t of a copy of its triangles' shading records, 16-byte aligned; 0 = none),
 *                                 type (PHIP_EMITTER_*; the constant environment emitter has no triangles)
 *   t[cdfOffset .. + nTris]       area CDF of the emitter's mesh (trimesh.cpp:388-404)
 *   t[recOffset .. ]              shading records of the emitter's triangles, when all emitters together have few */
struct EmitterTab { const float *t; uint32_t n; float normalization; };
enum { EM_RADIANCE = 0, EM_WEIGHT = 3, EM_FIRST_TRI = 4, EM_N_TRIS = 5, EM_CDF = 6, EM_INV_AREA = 7, EM_REC = 8, EM_TYPE = 9, EM_STRIDE = 12 };
DV const float *emitterRecord(const EmitterTab &T, uint32_t e) { return T.t + (T.n + 1) + EM_STRIDE * e; }

/* trimesh.cpp:412-423 + triangle.cpp:24-59 + shape.cpp:102-115 */
DV void shapeSampleDirect(const DevScene &S, const EmitterTab &T, const float *em, DirectRec &dRec, V2 sample) {
    const float *cdf = T.t + pm_to_bits(em[EM_CDF]);
    uint32_t index = cdfSample(cdf, pm_to_bits(em[EM_N_TRIS]), sample.y);
    sample.y = (sample.y - cdf[index]) / (cdf[index + 1] - cdf[index]);
    const uint32_t recOffset = pm_to_bits(em[EM_REC]);
    /* (one load sequence through a selected pointer: two typed sequences -- ds_read for the copy inside the table, global loads for the scene's
       record -- measured slower in k_mega, 77.2 -> 78.0 ms per C2 frame, and no faster in k_shade: round 3, profiles/r03_gpu_call_n.log) */
    const float4 *r = recOffset ? (const float4 *) (T.t + recOffset) + (size_t) TRISHADE_FLOAT4S * index
                                : S.triShade + (size_t) S.triShadeStride * (pm_to_bits(em[EM_FIRST_TRI]) + index);
    const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
    const V3 p0 = xyz(r0), p1 = xyz(r1), p2 = xyz(r2);
    V2 bary = squareToUniformTriangle(sample);
    V3 sideA = p1 - p0, sideB = p2 - p0;
    dRec.p = p0 + (sideA * bary.x) + (sideB * bary.y);
    if (pm_to_bits(r3.w) & TS_VERTEX_NORMALS) {
        const V3 n0 = xyz(r3), n1 = xyz(r[4]), n2 = xyz(r[5]);
        dRec.n = normalize(n0 * (1.0f - bary.x - bary.y) + n1 * bary.x + n2 * bary.y);
    } else {
        /* normalize(cross(sideA, sideB)) == the stored face normal, operation for operation (a triangle
           of zero area is never selected: its CDF step is empty) */
        dRec.n = xyz(r3);
    }
    dRec.pdf = em[EM_INV_AREA];
    dRec.d = dRec.p - dRec.ref;
    float distSquared = dRec.d.lengthSquared();
    dRec.dist = sqrtf(distSquared);
    dRec.d = dRec.d / dRec.dist;
    float dp = absDot(dRec.d, dRec.n);
    dRec.pdf *= dp != 0 ? (distSquared / dp) : 0.0f;
    dRec.solidAngle = 1;
}

/* bsphere.h:88-95 */
DV bool envSphereIntersect(const DevScene &S, const V3 &ro, const V3 &rd, float &nearT, float &farT) {
    const V3 o = ro - V3(S.envCenter[0], S.envCenter[1], S.envCenter[2]);
    const float A = rd.lengthSquared();
    const float B = 2 * dot(o, rd);
    const float C = o.lengthSquared() - S.envRadius * S.envRadius;
    return solveQuadratic(A, B, C, nearT, farT);
}

/* ConstantBackgroundEmitter::sampleDirect, constant.cpp:184-225 */
DV V3 constantSampleDirect(const DevScene &S, const float *em, DirectRec &dRec, const V2 &sample) {
    V3 d; float pdf;
    if (!dRec.refN.isZero()) {
        d = squareToCosineHemisphere(sample);
        pdf = PT_INV_PI * cosTheta(d);
        Frame f; f.n = dRec.refN; coordinateSystem(f.n, f.s, f.t);
        d = f.toWorld(d);
    } else {
        d = squareToUniformSphere(sample);
        pdf = PT_INV_FOURPI;
    }
    float nearT, farT;
    dRec.pdf = 0.0f;
    if (!envSphereIntersect(S, dRec.ref, d, nearT, farT)) return V3(0.0f);
    if (!(nearT < 0 && farT > 0)) return V3(0.0f);
    dRec.p = dRec.ref + d * farT;
    dRec.n = normalize(V3(S.envCenter[0], S.envCenter[1], S.envCenter[2]) - dRec.p);
    dRec.d = d; dRec.dist = farT; dRec.pdf = pdf; dRec.solidAngle = 1;
    if (!dRec.refN.isZero() && dot(dRec.d, dRec.refN) <= 0)
        return V3(0.0f);                 /* roundoff moved the sample to the back side: pdf stays, value is zero */
    return V3(em[EM_RADIANCE], em[EM_RADIANCE + 1], em[EM_RADIANCE + 2]) / pdf;
}

/* ConstantBackgroundEmitter::pdfDirect, constant.cpp:227-243 (solid-angle measure); dDotRefN = dot(d, refN) */
DV float constantPdfDirect(float dDotRefN, bool refNZero) {
    if (!refNZero)
        return PT_INV_PI * smax(0.0f, dDotRefN);
    return PT_INV_FOURPI;
}

/* ConstantBackgroundEmitter::fillDirectSamplingRecord, constant.cpp:258-273: only its verdict matters here
   (pdfDirect in the solid-angle measure reads d and refN) */
DV bool envFillDirectRecord(const DevScene &S, const V3 &ro, const V3 &rd) {
    float nearT, farT;
    return !(!envSphereIntersect(S, ro, rd, nearT, farT) || nearT > 0 || farT < 0);
}

/* ---------------- envmap.cpp ---------------- */
DV float rgbLuminance(const V3 &s) { return s.x * 0.212671f + s.y * 0.715160f + s.z * 0.072169f; }   /* spectrum.h:724-727 */
DV float intervalToTent(float sample) {   /* warp.cpp:143-155 */
    float sign;
    if (sample < 0.5f) { sign = 1; sample *= 2; }
    else { sign = -1; sample = 2 * (sample - 0.5f); }
    return sign * (1 - sqrtf(sample));
}
DV V3 xform3(const float *m, const V3 &v) {
    return V3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z);
}
/* MIPMap::evalTexel, mipmap.h:503-571, with bcu = ERepeat and bcv = EClamp (envmap.cpp:176-178) */
DV V3 envTexel(const DevEnvMap &E, int x, int y) {
    if (x < 0 || x >= E.w) { int r = x % E.w; x = (r < 0) ? r + E.w : r; }
    if (y < 0 || y >= E.h) y = y < 0 ? 0 : E.h - 1;
    const float4 t = E.texels[(size_t) y * E.w + x];
    return V3(t.x, t.y, t.z);
}
DV V2 envDirToUV(const V3 &v) {
    return V2(pm_atan2f(v.x, -v.z) * PT_INV_TWOPI, pm_acosf(smin(1.0f, smax(-1.0f, v.y))) * PT_INV_PI);
}
/* EnvironmentMap::evalEnvironment for a ray without differentials (envmap.cpp:380-394,408-409) = MIPMap::evalBilinear
   on level 0 (mipmap.h:575-596) */
DV V3 envmapEval(const DevEnvMap &E, const V3 &rayD) {
    const V2 uv = envDirToUV(xform3(E.toLocal, rayD));
    if (!(isfinite(uv.x) && isfinite(uv.y))) return V3(0.0f);
    const float u = uv.x * E.w - 0.5f, v = uv.y * E.h - 0.5f;
    const int xPos = f2i(floorf(u)), yPos = f2i(floorf(v));
    const float dx1 = u - xPos, dx2 = 1.0f - dx1, dy1 = v - yPos, dy2 = 1.0f - dy1;
    const V3 value = envTexel(E, xPos, yPos) * dx2 * dy2 + envTexel(E, xPos, yPos + 1) * dx2 * dy1
                   + envTexel(E, xPos + 1, yPos) * dx1 * dy2 + envTexel(E, xPos + 1, yPos + 1) * dx1 * dy1;
    return value * E.scale;
}
/* ---- MIP pyramid lookups (include/mitsuba/render/mipmap.h), shared by the envmap and the bitmap textures ---- */
DV int mipModulo(int a, int b) { const int r = a % b; return (r < 0) ? r + b : r; }   /* math.h:67-70 */
/* one coordinate under a boundary condition (mipmap.h:506-565); false: the texel is the constant `outside` */
DV bool mipWrap(int &x, int size, uint32_t bc, float &outside) {
    if (x < 0 || x >= size) {
        if (bc == PHIP_WRAP_REPEAT) x = mipModulo(x, size);
        else if (bc == PHIP_WRAP_CLAMP) x = x < 0 ? 0 : size - 1;
        else if (bc == PHIP_WRAP_MIRROR) { x = mipModulo(x, 2 * size); if (x >= size) x = 2 * size - x - 1; }
        else { outside = bc == PHIP_WRAP_ONE ? 1.0f : 0.0f; return false; }
    }
    return true;
}
DV V3 mipTexel(const float4 *texels, const DevMipLevels &Lv, int level, int x, int y) {   /* mipmap.h:503-571 */
    const int sw = Lv.lw[level], sh = Lv.lh[level];
    float outside = 0.0f;
    if (!mipWrap(x, sw, Lv.bcu, outside)) return V3(outside);
    if (!mipWrap(y, sh, Lv.bcv, outside)) return V3(outside);
    const float4 t = texels[(size_t) Lv.offset[level] + (size_t) y * sw + x];
    return V3(t.x, t.y, t.z);
}
DV V3 mipBox(const float4 *texels, const DevMipLevels &Lv, int level, const V2 &uv) {   /* mipmap.h:566-569 */
    return mipTexel(texels, Lv, level, f2i(floorf(uv.x * Lv.lw[level])), f2i(floorf(uv.y * Lv.lh[level])));
}
DV V3 mipBilinear(const float4 *texels, const DevMipLevels &Lv, int level, const V2 &uv) {   /* mipmap.h:575-596 */
    if (!(isfinite(uv.x) && isfinite(uv.y))) return V3(0.0f);
    if (level >= Lv.nLevels) return mipBox(texels, Lv, Lv.nLevels - 1, uv);
    const float u = uv.x * Lv.lw[level] - 0.5f, v = uv.y * Lv.lh[level] - 0.5f;
    const int xPos = f2i(floorf(u)), yPos = f2i(floorf(v));
    const float dx1 = u - xPos, dx2 = 1.0f - dx1, dy1 = v - yPos, dy2 = 1.0f - dy1;
    return mipTexel(texels, Lv, level, xPos, yPos) * dx2 * dy2 + mipTexel(texels, Lv, level, xPos, yPos + 1) * dx2 * dy1
         + mipTexel(texels, Lv, level, xPos + 1, yPos) * dx1 * dy2 + mipTexel(texels, Lv, level, xPos + 1, yPos + 1) * dx1 * dy1;
}
DV float mtsLog2(float value) {   /* math.cpp:103-106 */
    const float invLn2 = 1.0f / pm_logf(2.0f);
    return pm_logf(value) * invLn2;
}
DV V3 mipEWA(const float4 *texels, const DevMipLevels &Lv, int level, const V2 &uv, float A, float B, float C) {   /* mipmap.h:780-833 */
    if (!isfinite(A + B + C + uv.x + uv.y)) return V3(0.0f);
    if (level >= Lv.nLevels) return mipBox(texels, Lv, Lv.nLevels - 1, uv);
    const float u = uv.x * Lv.lw[level] - 0.5f;
    const float v = uv.y * Lv.lh[level] - 0.5f;
    const float ratioX = (float) Lv.lw[level] / (float) Lv.lw[0], ratioY = (float) Lv.lh[level] / (float) Lv.lh[0];
    A /= ratioX * ratioX;
    B /= ratioX * ratioY;
    C /= ratioY * ratioY;
    const float invDet = 1.0f / (-B * B + 4.0f * A * C),
                deltaU = 2.0f * sqrtf(C * invDet),
                deltaV = 2.0f * sqrtf(A * invDet);
    const int u0 = f2i(ceilf(u - deltaU)), u1 = f2i(floorf(u + deltaU));
    const int v0 = f2i(ceilf(v - deltaV)), v1 = f2i(floorf(v + deltaV));
    /* (level selection bounds the footprint to ~2 x maxAnisotropy texels per axis; anything far beyond that is garbage input,
       e.g. a NaN that slipped through: do not loop over it) */
    if ((long) u1 - u0 > 4096 || (long) v1 - v0 > 4096) return mipBilinear(texels, Lv, level, uv);
    const float As = A * 64, Bs = B * 64, Cs = C * 64;
    V3 result(0.0f);
    float denominator = 0.0f;
    const float ddq = 2 * As, uu0 = (float) u0 - u;
    for (int vt = v0; vt <= v1; ++vt) {
        const float vv = (float) vt - v;
        float q = As * uu0 * uu0 + (Bs * uu0 + Cs * vv) * vv;
        float dq = As * (2 * uu0 + 1) + Bs * vv;
        for (int ut = u0; ut <= u1; ++ut) {
            if (q < 64.0f) {
                const uint32_t qi = (uint32_t) (long long) q;     /* x86: (uint32_t) of a negative float wraps, it does not saturate to 0 */
                if (qi < 64) {
                    const float weight = Lv.weightLut[qi];
                    result = result + mipTexel(texels, Lv, level, ut, vt) * weight;
                    denominator += weight;
                }
            }
            q += dq;
            dq += ddq;
        }
    }
    if (denominator == 0) return mipBilinear(texels, Lv, level, uv);
    return result / denominator;
}
DV V3 mipEval(const float4 *texels, const DevMipLevels &Lv, const V2 &uv, const V2 &d0, const V2 &d1) {   /* MIPMap::eval, mipmap.h:629-712 */
    if (Lv.filterType == PHIP_FILTER_NEAREST) return mipBox(texels, Lv, 0, uv);
    if (Lv.filterType == PHIP_FILTER_BILINEAR) return mipBilinear(texels, Lv, 0, uv);
    const float maxAnisotropy = Lv.maxAnisotropy;
    const float du0 = d0.x * Lv.lw[0], dv0 = d0.y * Lv.lh[0], du1 = d1.x * Lv.lw[0], dv1 = d1.y * Lv.lh[0];
    float A = dv0 * dv0 + dv1 * dv1,
          B = -2.0f * (du0 * dv0 + du1 * dv1),
          C = du0 * du0 + du1 * du1,
          F = A * C - B * B * 0.25f;
    const float root = hypot2(A - C, B),
                Aprime = 0.5f * (A + C - root),
                Cprime = 0.5f * (A + C + root),
                majorRadius = Aprime != 0 ? sqrtf(F / Aprime) : 0;
    float minorRadius = Cprime != 0 ? sqrtf(F / Cprime) : 0;
    if (Lv.filterType == PHIP_FILTER_TRILINEAR || !(minorRadius > 0) || !(majorRadius > 0) || F < 0) {
        const float level = mtsLog2(smax(majorRadius, PT_EPSILON));
        const int ilevel = f2i(floorf(level));
        if (ilevel < 0) return mipBilinear(texels, Lv, 0, uv);
        const float a = level - ilevel;
        return mipBilinear(texels, Lv, ilevel, uv) * (1.0f - a) + mipBilinear(texels, Lv, ilevel + 1, uv) * a;
    }
    if (minorRadius * maxAnisotropy < majorRadius) {
        minorRadius = majorRadius / maxAnisotropy;
        const float theta = 0.5f * pm_atanf(B / (A - C));
        float sinTheta, cosTheta;
        pm_sincosf(theta, &sinTheta, &cosTheta);
        const float a2 = majorRadius * majorRadius, b2 = minorRadius * minorRadius,
                    sinTheta2 = sinTheta * sinTheta, cosTheta2 = cosTheta * cosTheta, sin2Theta = 2 * sinTheta * cosTheta;
        A = a2 * cosTheta2 + b2 * sinTheta2;
        B = (a2 - b2) * sin2Theta;
        C = a2 * sinTheta2 + b2 * cosTheta2;
        F = a2 * b2;
    }
    const float scl = 1.0f / F;
    A *= scl; B *= scl; C *= scl;
    const float level = smax(0.0f, mtsLog2(minorRadius));
    const int ilevel = f2i(level);
    const float a = level - ilevel;
    if (majorRadius < 1 || !(A > 0 && C > 0))
        return mipBilinear(texels, Lv, ilevel, uv);
    return mipEWA(texels, Lv, ilevel, uv, A, B, C) * (1.0f - a) + mipEWA(texels, Lv, ilevel + 1, uv, A, B, C) * a;
}
/* EnvironmentMap::evalEnvironment for a ray WITH differentials (envmap.cpp:380-409) */
DV V3 envmapEvalDiff(const DevEnvMap &E, const V3 &rayD, const V3 &rxDirection, const V3 &ryDirection) {
    const DevMipLevels &Lv = *E.levels;
    const V3 v = xform3(E.toLocal, rayD);
    const V2 uv = envDirToUV(v);
    const V3 dvdx = xform3(E.toLocal, rxDirection) - v, dvdy = xform3(E.toLocal, ryDirection) - v;
    const float t1 = PT_INV_TWOPI / (v.x * v.x + v.z * v.z),
                t2 = -PT_INV_PI / smax(safe_sqrt(1.0f - v.y * v.y), PT_EPSILON);
    const V2 dudx(t1 * (dvdx.z * v.x - dvdx.x * v.z), t2 * dvdx.y), dudy(t1 * (dvdy.z * v.x - dvdy.x * v.z), t2 * dvdy.y);
    return mipEval(E.texels, Lv, uv, dudx, dudy) * E.scale;
}

/* EnvironmentMap::sampleReuse, envmap.cpp:657-662 (std::lower_bound over size + 1 entries) */
DV uint32_t envSampleReuse(const float *cdf, uint32_t size, float &sample) {
    uint32_t lo = 0, len = size + 1;
    while (len > 0) {
        const uint32_t half = len >> 1, mid = lo + half;
        if (cdf[mid] < sample) { lo = mid + 1; len = len - half - 1; } else len = half;
    }
    long idx = (long) lo - 1;
    if (idx < 0) idx = 0;
    uint32_t index = (uint32_t) idx;
    if (index > size - 1) index = size - 1;
    sample = (sample - cdf[index]) / (cdf[index + 1] - cdf[index]);
    return index;
}
DV int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
/* bilinear patch shared by internalSampleDirection / internalPdfDirection (envmap.cpp:577-590, 616-631) */
DV float envPatch(const DevEnvMap &E, float px, float py, V3 &value) {
    const int xPos = f2i(floorf(px)), yPos = f2i(floorf(py));
    const float dx1 = px - xPos, dx2 = 1.0f - dx1, dy1 = py - yPos, dy2 = 1.0f - dy1;
    const V3 value1 = envTexel(E, xPos, yPos) * dx2 * dy2 + envTexel(E, xPos + 1, yPos) * dx1 * dy2;
    const V3 value2 = envTexel(E, xPos, yPos + 1) * dx2 * dy1 + envTexel(E, xPos + 1, yPos + 1) * dx1 * dy1;
    value = value1 + value2;
    return (rgbLuminance(value1) * E.rowWeights[clampi(yPos, 0, E.h - 1)] +
            rgbLuminance(value2) * E.rowWeights[clampi(yPos + 1, 0, E.h - 1)]) * E.normalization;
}
/* envmap.cpp:567-600 */
DV void envmapSampleDirection(const DevEnvMap &E, V2 sample, V3 &d, V3 &value, float &pdf) {
    const uint32_t row = envSampleReuse(E.cdfRows, (uint32_t) E.h, sample.y);
    const uint32_t col = envSampleReuse(E.cdfCols + (size_t) row * (E.w + 1), (uint32_t) E.w, sample.x);
    const float px = (float) col + intervalToTent(sample.x), py = (float) row + intervalToTent(sample.y);
    pdf = envPatch(E, px, py, value);
    value = value * E.scale;
    float sinPhi, cosPhi, sinTheta, cosTheta;
    pm_sincosf(E.pixelSizeX * (px + 0.5f), &sinPhi, &cosPhi);
    pm_sincosf(E.pixelSizeY * (py + 0.5f), &sinTheta, &cosTheta);
    d = V3(sinPhi * sinTheta, cosTheta, -cosPhi * sinTheta);
    pdf /= smax(fabsf(sinTheta), PT_EPSILON);
}
/* envmap.cpp:603-632 */
DV float envmapPdfDirection(const DevEnvMap &E, const V3 &d) {
    const V2 uv = envDirToUV(d);
    if (!(isfinite(uv.x) && isfinite(uv.y))) return 0.0f;
    V3 value;
    const float p = envPatch(E, uv.x * E.w - 0.5f, uv.y * E.h - 0.5f, value);
    const float sinTheta = safe_sqrt(1 - d.y * d.y);
    return p / smax(fabsf(sinTheta), PT_EPSILON);
}
/* EnvironmentMap::sampleDirect, envmap.cpp:516-542 */
DV V3 envmapSampleDirect(const DevScene &S, DirectRec &dRec, const V2 &sample) {
    V3 value, d; float pdf;
    envmapSampleDirection(S.env, sample, d, value, pdf);
    const V3 rd = xform3(S.env.toWorld, d);
    float nearT, farT;
    if (value.isZero() || pdf == 0 || !envSphereIntersect(S, dRec.ref, rd, nearT, farT) || nearT >= 0 || farT <= 0) {
        dRec.pdf = 0.0f;
        return V3(0.0f);
    }
    dRec.pdf = pdf;
    dRec.p = dRec.ref + rd * farT;
    dRec.n = normalize(V3(S.envCenter[0], S.envCenter[1], S.envCenter[2]) - dRec.p);
    dRec.dist = farT; dRec.d = rd; dRec.solidAngle = 1;
    return value / pdf;
}

/* scene.cpp:828-852 without the visibility test (the shadow ray is traced by the wavefront),
   area.cpp:158-173.  Returns value (radiance/pdf/emPdf); dRec.pdf == 0 means "no sample". */
/* ENV = false compiles the environment-emitter branches out (scenes without one: most of them) */
template <bool ENV> DV V3 sampleEmitterDirect(const DevScene &S, const EmitterTab &T, DirectRec &dRec, V2 sample) {
    if (T.n == 0) { dRec.pdf = 0; return V3(0.0f); }
    uint32_t index = cdfSample(T.t, T.n, sample.x);
    float emPdf = T.t[index + 1] - T.t[index];
    sample.x = (sample.x - T.t[index]) / (T.t[index + 1] - T.t[index]);
    const float *em = emitterRecord(T, index);
    V3 value;
    const uint32_t type = ENV ? pm_to_bits(em[EM_TYPE]) : (uint32_t) PHIP_EMITTER_AREA;
    if (ENV && type == PHIP_EMITTER_CONSTANT) {
        value = constantSampleDirect(S, em, dRec, sample);
        if (dRec.pdf == 0) return V3(0.0f);
    } else if (ENV && type == PHIP_EMITTER_ENVMAP) {
        value = envmapSampleDirect(S, dRec, sample);
        if (dRec.pdf == 0) return V3(0.0f);
    } else {
        shapeSampleDirect(S, T, em, dRec, sample);
        if (dot(dRec.d, dRec.refN) >= 0 && dot(dRec.d, dRec.n) < 0 && dRec.pdf != 0) {
            value = V3(em[EM_RADIANCE], em[EM_RADIANCE + 1], em[EM_RADIANCE + 2]) / dRec.pdf;
        } else {
            dRec.pdf = 0.0f;
            return V3(0.0f);
        }
    }
    dRec.emitter = (int) index;
    dRec.pdf *= emPdf;
    value = value / emPdf;
    return value;
}

/* scene.cpp:949-952, scene.h:848-850, area.cpp:175-182, shape.cpp:117-126 (solid-angle measure).
   The reference point enters only through dot(d, refN) and refN.isZero(): the wavefront stores those two
   (8 bytes with the BSDF pdf) instead of the normal when it spawns the ray. */
template <bool ENV> DV float pdfEmitterDirectDot(const DevScene &S, const EmitterTab &T, uint32_t emitter, const V3 &d, float dDotRefN, bool refNZero, float dDotN, float dist) {
    const float *em = emitterRecord(T, emitter);
    const uint32_t type = ENV ? pm_to_bits(em[EM_TYPE]) : (uint32_t) PHIP_EMITTER_AREA;
    float pdf;
    if (ENV && type == PHIP_EMITTER_CONSTANT) {
        pdf = constantPdfDirect(dDotRefN, refNZero);
    } else if (ENV && type == PHIP_EMITTER_ENVMAP) {
        pdf = envmapPdfDirection(S.env, xform3(S.env.toLocal, d));        /* envmap.cpp:545-549, solid-angle measure */
    } else if (dDotRefN >= 0 && dDotN < 0) {
        float pdfPos = em[EM_INV_AREA];
        pdf = pdfPos * (dist * dist) / fabsf(dDotN);
    } else {
        pdf = 0.0f;
    }
    return pdf * (em[EM_WEIGHT] * T.normalization);
}
DV float pdfEmitterDirect(const DevScene &S, const EmitterTab &T, const DirectRec &dRec) {
    return pdfEmitterDirectDot<true>(S, T, (uint32_t) dRec.emitter, dRec.d, dot(dRec.d, dRec.refN), dRec.refN.isZero(), dot(dRec.d, dRec.n), dRec.dist);
}

/* `bitmap` texture: Texture2D::eval (texture.cpp:112-121) over BitmapTexture::eval (bitmap.cpp:431-454,486-499);
   id = texture id; partials = the vertex has UV partials (first path vertex: camera-ray differentials) */
DV V3 textureEval(const DevScene &S, uint32_t id, const V2 &itsUV, bool partials, float dudx, float dudy, float dvdx, float dvdy) {
    const DevMipLevels &Lv = S.textures[id];
    const V2 uv(itsUV.x * Lv.uvScale[0] + Lv.uvOffset[0], itsUV.y * Lv.uvScale[1] + Lv.uvOffset[1]);
    if (partials)
        return mipEval(S.texTexels, Lv, uv, V2(dudx * Lv.uvScale[0], dvdx * Lv.uvScale[1]), V2(dudy * Lv.uvScale[0], dvdy * Lv.uvScale[1]));
    return Lv.filterType != PHIP_FILTER_NEAREST ? mipBilinear(S.texTexels, Lv, 0, uv) : mipBox(S.texTexels, Lv, 0, uv);
}

/* ======================================================================================
 *  BSDFs
 * ====================================================================================== */
struct MF {   /* MicrofacetDistribution, microfacet.h */
    int type; float alphaU, alphaV; bool visible;
    /* (aU, aV): the material's constants or, with a texture on alpha / alphaU / alphaV, texture->eval(its).average() at this vertex */
    DV MF(const DevMaterial &M, float aU, float aV) : type((int) M.distribution), alphaU(aU), alphaV(aV), visible(M.sampleVisible != 0) {
        alphaU = smax(alphaU, 1e-4f); alphaV = smax(alphaV, 1e-4f);
    }
    DV bool isIsotropic() const { return alphaU == alphaV; }
    DV float eval(const V3 &m) const {
        if (cosTheta(m) <= 0) return 0.0f;
        float cosTheta2 = m.z * m.z;
        float beckmannExponent = ((m.x * m.x) / (alphaU * alphaU) + (m.y * m.y) / (alphaV * alphaV)) / cosTheta2;
        float result;
        if (type == PHIP_MF_BECKMANN) {
            result = pm_expf(-beckmannExponent) / (PT_PI * alphaU * alphaV * cosTheta2 * cosTheta2);
        } else {
            float root = (1.0f + beckmannExponent) * cosTheta2;
            result = 1.0f / (PT_PI * alphaU * alphaV * root * root);
        }
        if (result * cosTheta(m) < 1e-20f) result = 0;
        return result;
    }
    DV float projectRoughness(const V3 &v) const {
        float invSinTheta2 = 1 / (1.0f - v.z * v.z);
        if (isIsotropic() || invSinTheta2 <= 0) return alphaU;
        float cosPhi2 = v.x * v.x * invSinTheta2;
        float sinPhi2 = v.y * v.y * invSinTheta2;
        return sqrtf(cosPhi2 * alphaU * alphaU + sinPhi2 * alphaV * alphaV);
    }
    DV float smithG1(const V3 &v, const V3 &m) const {
        if (dot(v, m) * cosTheta(v) <= 0) return 0.0f;
        float tanT = fabsf(tanTheta(v));
        if (tanT == 0.0f) return 1.0f;
        float alpha = projectRoughness(v);
        if (type == PHIP_MF_BECKMANN) {
            float a = 1.0f / (alpha * tanT);
            if (a >= 1.6f) return 1.0f;
            float aSqr = a * a;
            return (3.535f * a + 2.181f * aSqr) / (1.0f + 2.276f * a + 2.577f * aSqr);
        } else {
            float root = alpha * tanT;
            return 2.0f / (1.0f + hypot2(1.0f, root));
        }
    }
    DV float G(const V3 &wi, const V3 &wo, const V3 &m) const { return smithG1(wi, m) * smithG1(wo, m); }
    DV float pdfVisible(const V3 &wi, const V3 &m) const {
        if (cosTheta(wi) == 0) return 0.0f;
        return smithG1(wi, m) * absDot(wi, m) * eval(m) / fabsf(cosTheta(wi));
    }
    DV float pdf(const V3 &wi, const V3 &m) const { return visible ? pdfVisible(wi, m) : eval(m) * cosTheta(m); }

    DV V3 sampleAll(const V2 &sample, float &pdf) const {
        float cosThetaM = 0.0f, sinPhiM, cosPhiM, alphaSqr;
        if (isIsotropic()) {
            pm_sincosf((2.0f * PT_PI) * sample.y, &sinPhiM, &cosPhiM);
            alphaSqr = alphaU * alphaU;
        } else {
            float phiM = pm_atanf(alphaV / alphaU * pm_tanf(PT_PI + 2 * PT_PI * sample.y)) + PT_PI * floorf(2 * sample.y + 0.5f);
            pm_sincosf(phiM, &sinPhiM, &cosPhiM);
            float cosSc = cosPhiM / alphaU, sinSc = sinPhiM / alphaV;
            alphaSqr = 1.0f / (cosSc * cosSc + sinSc * sinSc);
        }
        if (type == PHIP_MF_BECKMANN) {
            float tanThetaMSqr = alphaSqr * -pm_logf(1.0f - sample.x);
            cosThetaM = 1.0f / sqrtf(1.0f + tanThetaMSqr);
            pdf = (1.0f - sample.x) / (PT_PI * alphaU * alphaV * cosThetaM * cosThetaM * cosThetaM);
        } else {
            float tanThetaMSqr = alphaSqr * sample.x / (1.0f - sample.x);
            cosThetaM = 1.0f / sqrtf(1.0f + tanThetaMSqr);
            float temp = 1 + tanThetaMSqr / alphaSqr;
            pdf = PT_INV_PI / (alphaU * alphaV * cosThetaM * cosThetaM * cosThetaM * temp * temp);
        }
        if (pdf < 1e-20f) pdf = 0;
        float sinThetaM = sqrtf(smax(0.0f, 1 - cosThetaM * cosThetaM));
        return V3(sinThetaM * cosPhiM, sinThetaM * sinPhiM, cosThetaM);
    }

    DV V2 sampleVisible11(float thetaI, V2 sample) const {
        const float SQRT_PI_INV = 1 / sqrtf(PT_PI);
        V2 slope;
        if (type == PHIP_MF_BECKMANN) {
            if (thetaI < 1e-4f) {
                float sinPhi, cosPhi;
                float r = sqrtf(-pm_logf(1.0f - sample.x));
                pm_sincosf(2 * PT_PI * sample.y, &sinPhi, &cosPhi);
                return V2(r * cosPhi, r * sinPhi);
            }
            float tanThetaI = pm_tanf(thetaI);
            float cotThetaI = 1 / tanThetaI;
            float a = -1, c = mts_erf(cotThetaI);
            float sample_x = smax(sample.x, 1e-6f);
            float fit = 1 + thetaI * (-0.876f + thetaI * (0.4265f - 0.0594f * thetaI));
            float b = c - (1 + c) * pm_powf(1 - sample_x, fit);
            float normalization = 1 / (1 + c + SQRT_PI_INV * tanThetaI * pm_expf(-cotThetaI * cotThetaI));
            int it = 0;
            while (++it < 10) {
                if (!(b >= a && b <= c))
                    b = 0.5f * (a + c);
                float invErf = mts_erfinv(b);
                float value = normalization * (1 + b + SQRT_PI_INV * tanThetaI * pm_expf(-invErf * invErf)) - sample_x;
                float derivative = normalization * (1 - invErf * tanThetaI);
                if (fabsf(value) < 1e-5f)
                    break;
                if (value > 0) c = b; else a = b;
                b -= value / derivative;
            }
            slope.x = mts_erfinv(b);
            slope.y = mts_erfinv(2.0f * smax(sample.y, 1e-6f) - 1.0f);
        } else {
            if (thetaI < 1e-4f) {
                float sinPhi, cosPhi;
                float r = safe_sqrt(sample.x / (1 - sample.x));
                pm_sincosf(2 * PT_PI * sample.y, &sinPhi, &cosPhi);
                return V2(r * cosPhi, r * sinPhi);
            }
            float tanThetaI = pm_tanf(thetaI);
            float a = 1 / tanThetaI;
            float G1 = 2.0f / (1.0f + safe_sqrt(1.0f + 1.0f / (a * a)));
            float A = 2.0f * sample.x / G1 - 1.0f;
            if (fabsf(A) == 1)
                A -= copysignf(1.0f, A) * PT_EPSILON;
            float tmp = 1.0f / (A * A - 1.0f);
            float B = tanThetaI;
            float D = safe_sqrt(B * B * tmp * tmp - (A * A - B * B) * tmp);
            float slope_x_1 = B * tmp - D;
            float slope_x_2 = B * tmp + D;
            slope.x = (A < 0.0f || slope_x_2 > 1.0f / tanThetaI) ? slope_x_1 : slope_x_2;
            float Sg;
            if (sample.y > 0.5f) { Sg = 1.0f; sample.y = 2.0f * (sample.y - 0.5f); }
            else { Sg = -1.0f; sample.y = 2.0f * (0.5f - sample.y); }
            float z = (sample.y * (sample.y * (sample.y * (-0.365728915865723f) + 0.790235037209296f) -
                        0.424965825137544f) + 0.000152998850436920f) /
                      (sample.y * (sample.y * (sample.y * (sample.y * 0.169507819808272f - 0.397203533833404f) -
                        0.232500544458471f) + 1.0f) - 0.539825872510702f);
            slope.y = Sg * z * sqrtf(1.0f + slope.x * slope.x);
        }
        return slope;
    }

    DV V3 sampleVisible(const V3 &_wi, const V2 &sample) const {
        V3 wi = normalize(V3(alphaU * _wi.x, alphaV * _wi.y, _wi.z));
        float theta = 0, phi = 0;
        if (wi.z < 0.99999f) {
            theta = pm_acosf(wi.z);
            phi = pm_atan2f(wi.y, wi.x);
        }
        float sinPhi, cosPhi;
        pm_sincosf(phi, &sinPhi, &cosPhi);
        V2 slope = sampleVisible11(theta, sample);
        slope = V2(cosPhi * slope.x - sinPhi * slope.y, sinPhi * slope.x + cosPhi * slope.y);
        slope.x *= alphaU;
        slope.y *= alphaV;
        float normalization = 1.0f / sqrtf(slope.x * slope.x + slope.y * slope.y + 1.0f);
        return V3(-slope.x * normalization, -slope.y * normalization, normalization);
    }
    DV V3 sample(const V3 &wi, const V2 &smp, float &pdf) const {
        V3 m;
        if (visible) { m = sampleVisible(wi, smp); pdf = pdfVisible(wi, m); }
        else m = sampleAll(smp, pdf);
        return m;
    }
};

struct BSDFSample { V3 wo; float eta; float pdf; bool delta; };

DV V3 rgb(const float *p) { return V3(p[0], p[1], p[2]); }

/* Which leaf models a scene contains: k_shade is instantiated per mask so that a diffuse-only scene does
   not carry (and fetch through the instruction cache) the microfacet and dielectric code. */
enum { MM_ROUGH = 1, MM_DIELECTRIC = 2, MM_ALL = 3 };

/* one-sided leaf models; wi.z sign already resolved by the twosided adapter.
   leafEvalPdf = eval() and pdf() of the same (wi, wo) pair in one pass (diffuse.cpp:110-133,
   roughconductor.cpp:253-337): the two share H, D and G1(wi). */
struct LeafVarying { V3 albedo; float alphaU, alphaV; V3 trans; };   /* the spatially varying parameters at this vertex: constants or texture values */
/* `type` (and the dielectric's eta) by value: BsdfCtx; M is dereferenced by the rough conductor only */
template <int MM> DV V3 leafEvalPdf(uint32_t type, const DevMaterial &M, const LeafVarying &lv, const V3 &wi, const V3 &wo, float &pdf) {
    const V3 &albedo = lv.albedo;
    pdf = 0.0f;
    if (MM == 0 || type == PHIP_BSDF_DIFFUSE) {              /* (MM == 0: the scene's leaves are all diffuse -- `type` is not even fetched) */
        if (cosTheta(wi) <= 0 || cosTheta(wo) <= 0) return V3(0.0f);
        pdf = PT_INV_PI * cosTheta(wo);
        return albedo * (PT_INV_PI * cosTheta(wo));
    } else if ((MM & MM_ROUGH) && type == PHIP_BSDF_ROUGHCONDUCTOR) {
        if (cosTheta(wi) <= 0 || cosTheta(wo) <= 0) return V3(0.0f);
        V3 H = normalize(wo + wi);
        MF distr(M, lv.alphaU, lv.alphaV);
        const float D = distr.eval(H);
        const float G1i = distr.smithG1(wi, H);
        if (M.sampleVisible)
            pdf = D * G1i / (4.0f * cosTheta(wi));
        else
            pdf = (D * cosTheta(H)) / (4 * absDot(wo, H));
        if (D == 0) return V3(0.0f);
        const V3 F = fresnelConductorExact(dot(wi, H), rgb(M.eta), rgb(M.k)) * albedo;     /* m_specularReflectance->eval(bRec.its) */
        const float G = G1i * distr.smithG1(wo, H);
        float model = D * G / (4.0f * cosTheta(wi));
        return F * model;
    }
    return V3(0.0f);
}
template <int MM> DV V3 leafSample(uint32_t type, float eta0, const DevMaterial &M, const LeafVarying &lv, const V3 &wi, const V2 &smp, BSDFSample &bs) {
    const V3 &albedo = lv.albedo;
    bs.eta = 1.0f; bs.delta = false; bs.pdf = 0.0f; bs.wo = V3(0.0f);
    if (MM == 0 || type == PHIP_BSDF_DIFFUSE) {
        if (cosTheta(wi) <= 0) return V3(0.0f);
        bs.wo = squareToCosineHemisphere(smp);
        bs.pdf = PT_INV_PI * cosTheta(bs.wo);
        return albedo;
    } else if ((MM & MM_ROUGH) && type == PHIP_BSDF_ROUGHCONDUCTOR) {
        if (cosTheta(wi) < 0) return V3(0.0f);
        MF distr(M, lv.alphaU, lv.alphaV);
        float pdf;
        V3 m = distr.sample(wi, smp, pdf);
        if (pdf == 0) return V3(0.0f);
        bs.wo = 2 * dot(wi, m) * m - wi;
        if (cosTheta(bs.wo) <= 0) return V3(0.0f);
        V3 F = fresnelConductorExact(dot(wi, m), rgb(M.eta), rgb(M.k)) * albedo;
        float weight;
        if (M.sampleVisible) weight = distr.smithG1(bs.wo, m);
        else weight = distr.eval(m) * distr.G(wi, bs.wo, m) * dot(wi, m) / (pdf * cosTheta(wi));
        pdf /= 4.0f * dot(bs.wo, m);
        bs.pdf = pdf;
        return F * weight;
    } else if ((MM & MM_DIELECTRIC) && type == PHIP_BSDF_DIELECTRIC) {
        const float eta = eta0, invEta = 1 / eta;
        float cosThetaT;
        float F = fresnelDielectricExt(cosTheta(wi), cosThetaT, eta);
        bs.delta = true;
        if (smp.x <= F) {
            bs.wo = V3(-wi.x, -wi.y, wi.z);
            bs.eta = 1.0f; bs.pdf = F;
            return albedo;                                  /* m_specularReflectance->eval(bRec.its) */
        } else {
            float scale = -(cosThetaT < 0 ? invEta : eta);
            bs.wo = V3(scale * wi.x, scale * wi.y, cosThetaT);
            bs.eta = cosThetaT < 0 ? eta : invEta;
            bs.pdf = 1 - F;
            float factor = cosThetaT < 0 ? invEta : eta;
            return lv.trans * (factor * factor);            /* m_specularTransmittance->eval(bRec.its) */
        }
    }
    return V3(0.0f);
}

/* The twosided adapter (twosided.cpp:108-183) resolved ONCE per path vertex: eval, pdf and sample of a
   vertex all see the same wi, so they share the nested model and the flip.  (eval/pdf pick nested0 for
   cosTheta(wi) > 0 and sample for cosTheta(wi) >= 0; at exactly 0 every wrapped model's eval/pdf is zero
   on either side, so one rule serves all three.) */
struct BsdfCtx { const DevMaterial *leaf; V3 wi; bool flip;
                 uint32_t type; float eta0;     /* the leaf's model and (dielectric) its relative index of refraction, by value (a diffuse-only build never fetches `type`: leafEvalPdf) */
                 bool textured;
                 LeafVarying v; /* reflectance (diffuse) / specularReflectance (dielectric, roughconductor), roughness, specularTransmittance at this vertex:
                                   the constants, or the texture values k_shade puts here */
                 DV void constants() { type = leaf->type; eta0 = leaf->eta[0]; v.albedo = rgb(leaf->refl); v.alphaU = leaf->alphaU; v.alphaV = leaf->alphaV; v.trans = rgb(leaf->trans);
                                       textured = (leaf->reflTexture | leaf->alphaUTexture | leaf->alphaVTexture | leaf->transTexture) != 0; } };
DV BsdfCtx bsdfResolve(const DevScene &S, const DevMaterial &M, const V3 &wi) {
    BsdfCtx c; c.leaf = &M; c.wi = wi; c.flip = false;
    if (M.type == PHIP_BSDF_TWOSIDED) {
        c.flip = cosTheta(wi) < 0;
        c.leaf = S.materials + (c.flip ? M.nested1 : M.nested0);
        if (c.flip) c.wi.z = -wi.z;
    }
    c.constants();
    return c;
}
/* same, from a shading record (front/back already are the nested models) */
DV BsdfCtx bsdfResolve(const DevMaterial *materials, const Isect &its) {
    BsdfCtx c; c.wi = its.wi;
    c.flip = (its.flags & TS_TWOSIDED) && cosTheta(its.wi) < 0;
    c.leaf = materials + (c.flip ? its.back : its.front);
    if (c.flip) c.wi.z = -its.wi.z;
    c.constants();
    return c;
}
/* The textured parameters of the vertex's leaf model: texture->eval(its) of every `bitmap` child (reflectance / specularReflectance,
   alpha / alphaU / alphaV as the RGB average, specularTransmittance).  `partials`: the vertex carries UV partials (the first vertex:
   camera-ray differentials, records.inl:69-75) -> MIPMap::eval; otherwise the unfiltered level-0 lookup (bitmap.cpp:431-454). */
DV void bsdfTextures(const DevScene &S, BsdfCtx &c, const V2 &uv, bool partials, float dudx, float dudy, float dvdx, float dvdy) {
    const DevMaterial &M = *c.leaf;
    if (M.reflTexture) c.v.albedo = textureEval(S, M.reflTexture - 1, uv, partials, dudx, dudy, dvdx, dvdy);
    if (M.alphaUTexture) c.v.alphaU = textureEval(S, M.alphaUTexture - 1, uv, partials, dudx, dudy, dvdx, dvdy).average();
    if (M.alphaVTexture) c.v.alphaV = (M.alphaVTexture == M.alphaUTexture) ? c.v.alphaU : textureEval(S, M.alphaVTexture - 1, uv, partials, dudx, dudy, dvdx, dvdy).average();
    if (M.transTexture) c.v.trans = textureEval(S, M.transTexture - 1, uv, partials, dudx, dudy, dvdx, dvdy);
}
template <int MM> DV V3 bsdfEvalPdf(const BsdfCtx &c, V3 wo, float &pdf) {
    if (c.flip) wo.z = -wo.z;
    return leafEvalPdf<MM>(c.type, *c.leaf, c.v, c.wi, wo, pdf);
}
template <int MM> DV V3 bsdfSample(const BsdfCtx &c, const V2 &smp, BSDFSample &bs) {
    V3 result = leafSample<MM>(c.type, c.eta0, *c.leaf, c.v, c.wi, smp, bs);
    if (c.flip && !result.isZero() && bs.pdf != 0)
        bs.wo.z = -bs.wo.z;
    return result;
}

/* BSDF::eval / pdf / sample as the reference exposes them (host-side unit tests) */
DV V3 bsdfEval(const DevScene &S, const DevMaterial &M, V3 wi, V3 wo) {
    float pdf; return bsdfEvalPdf<MM_ALL>(bsdfResolve(S, M, wi), wo, pdf);
}
DV float bsdfPdf(const DevScene &S, const DevMaterial &M, V3 wi, V3 wo) {
    float pdf; bsdfEvalPdf<MM_ALL>(bsdfResolve(S, M, wi), wo, pdf); return pdf;
}
DV V3 bsdfSample(const DevScene &S, const DevMaterial &M, V3 wi, const V2 &smp, BSDFSample &bs) {
    return bsdfSample<MM_ALL>(bsdfResolve(S, M, wi), smp, bs);
}

/* the differential directions of perspective.cpp:293-294 for the sample (sx, sy), before scaleDifferential */
DV void cameraRayDifferentials(const DevCamera &c, float sx, float sy, V3 &rx, V3 &ry) {
    const float px = sx * c.invResX, py = sy * c.invResY, pz = 0.0f;
    const float *m = c.s2c;
    float x = m[0] * px + m[1] * py + m[2] * pz + m[3];
    float y = m[4] * px + m[5] * py + m[6] * pz + m[7];
    float z = m[8] * px + m[9] * py + m[10] * pz + m[11];
    float w = m[12] * px + m[13] * py + m[14] * pz + m[15];
    const V3 nearP = (w == 1.0f) ? V3(x, y, z) : V3(x, y, z) / w;
    const V3 a = normalize(nearP + V3(c.dx[0], c.dx[1], c.dx[2])), b = normalize(nearP + V3(c.dy[0], c.dy[1], c.dy[2]));
    const float *t = c.c2w;
    rx = V3(t[0] * a.x + t[1] * a.y + t[2] * a.z, t[4] * a.x + t[5] * a.y + t[6] * a.z, t[8] * a.x + t[9] * a.y + t[10] * a.z);
    ry = V3(t[0] * b.x + t[1] * b.y + t[2] * b.z, t[4] * b.x + t[5] * b.y + t[6] * b.z, t[8] * b.x + t[9] * b.y + t[10] * b.z);
}

/* perspective.cpp:271-297 */
DV void cameraRay(const DevCamera &c, float sx, float sy, V3 &o, V3 &d, float &mint, float &maxt) {
    const float px = sx * c.invResX, py = sy * c.invResY, pz = 0.0f;
    const float *m = c.s2c;
    float x = m[0] * px + m[1] * py + m[2] * pz + m[3];
    float y = m[4] * px + m[5] * py + m[6] * pz + m[7];
    float z = m[8] * px + m[9] * py + m[10] * pz + m[11];
    float w = m[12] * px + m[13] * py + m[14] * pz + m[15];
    V3 nearP = (w == 1.0f) ? V3(x, y, z) : V3(x, y, z) / w;
    V3 dl = normalize(nearP);
    float invZ = 1.0f / dl.z;
    mint = c.nearClip * invZ;
    maxt = c.farClip * invZ;
    const float *t = c.c2w;
    o = V3(t[0] * 0.0f + t[1] * 0.0f + t[2] * 0.0f + t[3], t[4] * 0.0f + t[5] * 0.0f + t[6] * 0.0f + t[7],
           t[8] * 0.0f + t[9] * 0.0f + t[10] * 0.0f + t[11]);
    d = V3(t[0] * dl.x + t[1] * dl.y + t[2] * dl.z, t[4] * dl.x + t[5] * dl.y + t[6] * dl.z, t[8] * dl.x + t[9] * dl.y + t[10] * dl.z);
}

} // namespace pt
