/*
 * phip.hip -- MI355X (gfx950) path tracer behind the C ABI of include/phip.h: host side + ray and film kernels.
 *
 * Replaces the reference's per-block CPU loop (SamplingIntegrator::renderBlock ->
 * MIPathTracer::Li, src/librender/integrator.cpp:140-188, src/integrators/path/path.cpp:119-300; the sibling integrators `direct` and -- on scenes
 * without media -- `volpath_simple`).  Three device paths share one statement of the integrator (shadeVertex, k_shade.h):
 *
 *   wavefront (big scenes): a pool of path slots lives in HBM as SoA arrays; every iteration runs k_shade (one vertex per slot) and k_rays_w
 *       (persistent waves over the compressed 8-wide BVH: the closest-hit and the any-hit rays of the iteration), a final film pass develops the
 *       per-sample accumulators
 *   fused (scenes whose tree, records, emitter table and materials fit LDS -- the Cornell box of BASELINE.json configs[1], with any of the three leaf
 *       BSDF models when the tree is the packed leaf table of <= 64 Wald records): k_mega keeps a path in registers from the camera sample to its end;
 *       HBM sees one 16-byte store per sample (k_mega.h)
 *   vertex-traced (small scenes k_mega does not serve: bitmap textures, an environment emitter): k_shade_trace -- the wavefront's pool, but ONE kernel per
 *       iteration that shades a slot's vertex and traces its shadow ray and its next ray on the packed leaf table in LDS (k_shade_trace.h)
 *
 * libphip.so is 26 objects of three sources (phip_common.h).  Kernels and where they live:
 *   k_pool.h      PathPool (HBM layout of the slots), slot flags, RenderConst, per-wave statistics, the sample streams' entry points
 *   k_traverse.h  per-lane BVH4 traversal as a state machine, the flat / packed leaf tables of the LDS-resident trees with the Wald tests dealt over
 *                 the wave (traverseFlat2W), LDS-staged stacks and top-of-tree cache                     [this unit, phip_mega.hip, phip_shade.hip]
 *   k_rays.h      k_trace / k_shadow_p (BVH4: trees of fewer than 64 nodes rendered by `direct` or with PHIP_FLAG_NO_FUSED), k_raycast (phip_trace);
 *                 experiment builds also carry the BVH4 ray kernels of rounds 1-2 for big trees                  [this unit]
 *   k_wide.h      the ray kernels over the compressed 8-wide BVH (80-byte nodes, quantised child boxes) that the big scenes use: k_rays_w
 *                 (triangle tests dealt over the wave), k_raycast_w                                               [this unit]
 *   k_shade.h     shadeVertex + k_shade<materials, strictNormals, features>: emitter-hit / environment MIS term, Russian roulette, emission, NEE sample
 *                 (self-contained shadow-queue entry, block-compacted), BSDF sample -> next ray in place; a path that ends is replaced by the SAME
 *                 lane in the same launch (static sample schedule + dynamic tail).  Radiance accumulates in L[sampleId] in the reference's
 *                 order.  volpath_simple is a uniform branch of shadeVertex.                                      [phip_shade.hip]
 *   k_shade_direct.h  MIDirectIntegrator::Li on the same pool                                                     [phip_shade.hip]
 *   k_shade_trace.h   the one-kernel iterations of the small scenes                                               [phip_shade.hip]
 *   k_mega.h      the fused kernel                                                                                [phip_mega.hip]
 *   k_film.h      k_film_splat + k_film_merge (filters of reach <= 2 pixels: register accumulators, no float atomics, deterministic), k_film_tiled /
 *                 k_film (wider filters, small blocks): ImageBlock::put, include/mitsuba/render/imageblock.h:124-204;
 *                 k_reduce_stats, k_export_samples                                                                [this unit]
 *
 * This file: error handling, scene validation and upload (BVH build via bvh.h, camera set-up), replication of the scene to
 * further GPUs, the render loop, the multi-device orchestration (one host thread + stream per GPU, ncclReduce of the
 * films over RCCL/xGMI) and the extern "C" entry points.  Not MFMA work: irregular traversal and gathers.
 * The product never includes, links or calls anything under oracle/.
 */
#include "phip_common.h"
#include "bvh.h"
#include "k_traverse.h"
#include "k_rays.h"
#include "k_wide.h"
#include "k_wide_wave.h"
#include "k_film.h"
#ifndef PHIP_DEBUG_HOOKS
#define PHIP_DEBUG_HOOKS 0
#endif
#ifndef PHIP_WIDE_MIN_RECORDS
#define PHIP_WIDE_MIN_RECORDS 64u    /* scenes of more Wald records than the packed leaf table holds are traversed on the 8-wide tree, whatever their size */
#endif
#include <dlfcn.h>
#include <map>
#include <rccl/rccl.h>          /* types and prototypes only: librccl is bound with dlopen at the first multi-GPU render */

/* ======================================================================================
 *  error handling
 * ====================================================================================== */
/* Environment: the shipped library reads PHIP_DEBUG_TIMING (host-side phase times on stderr), PHIP_MAX_PASS_SAMPLES (the sample-buffer budget: the tests force several
   passes with it) and the builder parameters PHIP_BVH_* (bvh.h) -- nothing that selects an algorithm.  Every switch an A/B row of profiles/ was made with is
   read by experiment builds only (-DPHIP_EXPERIMENTS=1, tools/build_variant.sh): in the product expEnv() is a constant and its names are not even
   in the binary. */
static inline const char *expEnv(const char *name) { return PHIP_EXPERIMENTS ? getenv(name) : nullptr; }
static thread_local std::string g_err;
static int setErr(int code, const std::string &msg) { g_err = msg; return code; }

/* ======================================================================================
 *  host side
 * ====================================================================================== */
namespace {

template <typename T> struct DevBuf {
    T *p = nullptr; size_t n = 0;
    size_t cap = 0;
    /* hipMalloc / hipFree cost milliseconds: keep the allocation when it is large enough */
    void alloc(size_t count) { if (count > cap) { release(); if (count) { HIP_TRY(hipMalloc((void **) &p, count * sizeof(T))); cap = count; } } n = count; }
    void upload(const T *src, size_t count) { alloc(count); if (count) HIP_TRY(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice)); }
    void cloneFrom(const DevBuf<T> &src) { alloc(src.n); if (src.n) HIP_TRY(hipMemcpy(p, src.p, src.n * sizeof(T), hipMemcpyDeviceToDevice)); }   /* UVA: also across GPUs (xGMI) */
    void release() { if (p) { (void) hipFree(p); p = nullptr; } n = 0; cap = 0; }
    ~DevBuf() { release(); }
};

/* 4x4 helpers for the camera set-up: perspective.cpp:126-157, transform.cpp:33-63,99-123, matrix.inl:138-193 */
struct M4 { float m[4][4]; };
M4 m4identity() { M4 r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = i == j ? 1.0f : 0.0f; return r; }
M4 m4mul(const M4 &a, const M4 &b) {
    M4 r;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { float s = 0; for (int k = 0; k < 4; ++k) s += a.m[i][k] * b.m[k][j]; r.m[i][j] = s; }
    return r;
}
bool m4invert(const M4 &src, M4 &t) {
    int indxc[4], indxr[4], ipiv[4] = { 0, 0, 0, 0 };
    t = src;
    for (int i = 0; i < 4; i++) {
        int irow = -1, icol = -1; float big = 0;
        for (int j = 0; j < 4; j++) if (ipiv[j] != 1) for (int k = 0; k < 4; k++) {
            if (ipiv[k] == 0) { if (fabsf(t.m[j][k]) >= big) { big = fabsf(t.m[j][k]); irow = j; icol = k; } }
            else if (ipiv[k] > 1) return false;
        }
        ++ipiv[icol];
        if (irow != icol) for (int k = 0; k < 4; ++k) std::swap(t.m[irow][k], t.m[icol][k]);
        indxr[i] = irow; indxc[i] = icol;
        if (t.m[icol][icol] == 0) return false;
        float pivinv = 1.f / t.m[icol][icol];
        t.m[icol][icol] = 1.f;
        for (int j = 0; j < 4; j++) t.m[icol][j] *= pivinv;
        for (int j = 0; j < 4; j++) if (j != icol) {
            float save = t.m[j][icol]; t.m[j][icol] = 0;
            for (int k = 0; k < 4; k++) t.m[j][k] -= t.m[icol][k] * save;
        }
    }
    for (int j = 3; j >= 0; j--) if (indxr[j] != indxc[j]) for (int k = 0; k < 4; k++) std::swap(t.m[k][indxr[j]], t.m[k][indxc[j]]);
    return true;
}

void setupCamera(const phip_camera &c, const phip_film &f, DevCamera &out) {
    const float aspect = f.width / (float) f.height;
    const float relSizeX = (float) f.crop_width / (float) f.width, relSizeY = (float) f.crop_height / (float) f.height;
    const float relOffX = (float) f.crop_offset_x / (float) f.width, relOffY = (float) f.crop_offset_y / (float) f.height;
    /* inverse of scale(1/relSize) * translate(-relOffset) * scale(-0.5,-0.5*aspect,1) * translate(-1,-1/aspect,0) * perspective
       = perspective^-1 * translate^-1 * scale^-1 * translate^-1 * scale^-1 (Transform keeps the product of inverses) */
    M4 persp; memset(&persp, 0, sizeof(persp));
    const float recip = 1.0f / (c.far_clip - c.near_clip);
    const float cot = 1.0f / pm_tanf((c.xfov_deg / 2.0f) * (PT_PI / 180.0f));
    persp.m[0][0] = cot; persp.m[1][1] = cot; persp.m[2][2] = c.far_clip * recip; persp.m[2][3] = -c.near_clip * c.far_clip * recip; persp.m[3][2] = 1;
    M4 perspInv; m4invert(persp, perspInv);
    M4 t2i = m4identity(); t2i.m[0][3] = -(-1.0f); t2i.m[1][3] = -(-1.0f / aspect); t2i.m[2][3] = -0.0f;   /* inverse of translate(-1,-1/aspect,0) */
    M4 s2i = m4identity(); s2i.m[0][0] = 1.0f / -0.5f; s2i.m[1][1] = 1.0f / (-0.5f * aspect); s2i.m[2][2] = 1.0f / 1.0f;
    M4 t1i = m4identity(); t1i.m[0][3] = -(-relOffX); t1i.m[1][3] = -(-relOffY); t1i.m[2][3] = -0.0f;
    M4 s1i = m4identity(); s1i.m[0][0] = 1.0f / (1.0f / relSizeX); s1i.m[1][1] = 1.0f / (1.0f / relSizeY); s1i.m[2][2] = 1.0f / 1.0f;
    /* Transform::operator* : inv = t.inv * this.inv, applied left to right over the five factors */
    M4 inv = s1i;                 /* (scale1)^-1 */
    inv = m4mul(t1i, inv);        /* (scale1*translate1)^-1 = translate1^-1 * scale1^-1 */
    inv = m4mul(s2i, inv);
    inv = m4mul(t2i, inv);
    inv = m4mul(perspInv, inv);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out.s2c[4 * i + j] = inv.m[i][j];
    for (int i = 0; i < 12; ++i) out.c2w[i] = c.to_world[i];
    out.nearClip = c.near_clip; out.farClip = c.far_clip;
    out.invResX = 1.0f / (float) f.crop_width; out.invResY = 1.0f / (float) f.crop_height;
    /* position differentials on the near plane, perspective.cpp:159-163 (Transform::operator()(Point), transform.h:108-125) */
    auto s2cPoint = [&](float px, float py, float pz) {
        const float x = inv.m[0][0] * px + inv.m[0][1] * py + inv.m[0][2] * pz + inv.m[0][3];
        const float y = inv.m[1][0] * px + inv.m[1][1] * py + inv.m[1][2] * pz + inv.m[1][3];
        const float z = inv.m[2][0] * px + inv.m[2][1] * py + inv.m[2][2] * pz + inv.m[2][3];
        const float w = inv.m[3][0] * px + inv.m[3][1] * py + inv.m[3][2] * pz + inv.m[3][3];
        return (w == 1.0f) ? V3(x, y, z) : V3(x, y, z) / w;
    };
    const V3 p0 = s2cPoint(0.0f, 0.0f, 0.0f);
    const V3 dx = s2cPoint(out.invResX, 0.0f, 0.0f) - p0, dy = s2cPoint(0.0f, out.invResY, 0.0f) - p0;
    out.dx[0] = dx.x; out.dx[1] = dx.y; out.dx[2] = dx.z; out.dy[0] = dy.x; out.dy[1] = dy.y; out.dy[2] = dy.z;
}

/* spiral block order, src/librender/imageproc.cpp:28-78 */
void spiralBlocks(int sizeX, int sizeY, int bs, std::vector<std::pair<int, int>> &out) {
    const int nbx = (int) std::ceil((float) sizeX / (float) bs), nby = (int) std::ceil((float) sizeY / (float) bs);
    const int total = nbx * nby; int generated = 0;
    int cx = nbx / 2, cy = nby / 2, dir = 0, stepsLeft = 1, numSteps = 1;
    out.clear();
    while (generated < total) {
        out.push_back({ cx, cy });
        if (++generated == total) break;
        do {
            switch (dir) { case 0: ++cx; break; case 1: ++cy; break; case 2: --cx; break; case 3: --cy; break; }
            if (--stepsLeft == 0) { dir = (dir + 1) % 4; if (dir == 2 || dir == 0) ++numSteps; stepsLeft = numSteps; }
        } while (cx < 0 || cy < 0 || cx >= nbx || cy >= nby);
    }
}

} // namespace
/* Everything that lives on ONE GPU: the immutable scene arrays and the render-time buffers of the jobs that run there.
   devs[0] of a phip_scene is the device of phip_scene_create; further entries are replicas made by device-to-device copies. */
struct SceneDev {
    int device = 0;
    /* ---- scene (immutable after build / replication) ---- */
    DevBuf<float4> nodes, tris, wtris, triShade, flatLeaves;      /* tris: BVH4 leaf order (LDS-resident scenes only), wtris: the wide tree's */
    bool trisAreWide = false;                                      /* scenes past the packed leaf table: DevScene::tris = wtris */
    DevBuf<uint4> wnodes;                                                                     /* compressed wide BVH (big scenes) */
    DevBuf<DevMaterial> materials;
    DevBuf<float> emitterTab;
    DevBuf<float4> texTexels; DevBuf<DevMipLevels> texDesc;                                   /* bitmap textures */
    DevBuf<float4> envTexels; DevBuf<DevMipLevels> envLevels; DevBuf<float> envCdfRows, envCdfCols, envRowWeights;     /* `envmap` emitter */
    DevScene dev;
    /* ---- render-time buffers (grown on demand, reused between calls) ---- */
    DevBuf<float4> rayO, rayD, hit, thr, camHit, shadow, L, sampleOut;
    DevBuf<float2> jitter;                                                                     /* sequence samplers: the camera sample's pixel jitter per sample id (RenderConst::jitter) */
    DevBuf<uint4> info; DevBuf<uint32_t> state; DevBuf<float2> mis;
    DevBuf<Counters> counters;
    DevBuf<uint32_t> tileOrigin, shadowCount, blockDead, spill, blockShard; DevBuf<int32_t> tileSlot;
    DevBuf<uint32_t> sobolMat, sobolBt; DevBuf<unsigned long long> sobolVdc, sobolVdcBt; uint64_t sobolKey = 0; uint32_t sobolLogRes = 0;    /* PHIP_SAMPLER_SOBOL: the plugin's tables */
    DevBuf<uint32_t> rinvDimInfo, rinvChunk, rinvPw; DevBuf<float> rinvFac; uint32_t rinvTabDims = 0;   /* ... and their multi-digit tables (buildRinvTables) */
    DevBuf<uint32_t> rinvPrimes, rinvOffsets; DevBuf<uint16_t> rinvPerm; uint64_t rinvKey = 0;   /* PHIP_SAMPLER_HALTON / _HAMMERSLEY: primes + permutations */
    uint32_t rinvInvPerm2 = 0x4u, rinvInvPerm3 = 0x24u;                                                                         /* inverse permutations of bases 2 and 3, two bits per digit */
    DevBuf<unsigned long long> dynCounter, stat, invalid, megaNext;
    DevBuf<unsigned int> drawCounters;                                                       /* k_rays_w: 2 x RAY_SHARDS sharded work counters */
    DevBuf<float> film, patchImg;                                                            /* patchImg: k_film_splat's 16 x 16 patch images (k_film.h) */
    uint32_t lastSpp = 0, nLocalTiles = 0; unsigned long long localPixels = 0;
    int tileKey[3] = { -1, -1, -1 };
    bool haveSamples = false;
    bool mergedRays = false;         /* last render used k_rays_p (closest + any hit in one launch) */
    bool fused = false;              /* last render used k_mega */
    hipStream_t stream = nullptr;
    /* phip_render's device-to-host copy of the film into PAGEABLE memory: two pinned staging chunks (filmToHost) */
    float *stage[2] = { nullptr, nullptr }; hipEvent_t stageDone[2] = { nullptr, nullptr };

    template <typename F> void forEachSceneBuffer(F f) {
        f(nodes); f(wnodes); f(tris); f(wtris); f(triShade); f(flatLeaves); f(materials); f(emitterTab); f(texTexels); f(texDesc);
        f(envTexels); f(envLevels); f(envCdfRows); f(envCdfCols); f(envRowWeights);
    }
    /* the pointer members of the DevScene (everything else in it is plain data, equal on every device) */
    void bind() {
        dev.nodes = nodes.p; dev.wnodes = wnodes.p; dev.wtris = wtris.p; dev.tris = trisAreWide ? wtris.p : tris.p; dev.triShade = triShade.p; dev.flatLeaves = flatLeaves.p; dev.materials = materials.p;
        dev.texTexels = texTexels.p; dev.textures = texDesc.p; dev.emitterTab = emitterTab.p;
        dev.env.texels = envTexels.p; dev.env.levels = envLevels.p; dev.env.cdfRows = envCdfRows.p; dev.env.cdfCols = envCdfCols.p;
        dev.env.rowWeights = envRowWeights.p;
    }
    ~SceneDev() {
        (void) hipSetDevice(device);
        if (stream) (void) hipStreamDestroy(stream);
        for (int i = 0; i < 2; ++i) { if (stage[i]) (void) hipHostFree(stage[i]); if (stageDone[i]) (void) hipEventDestroy(stageDone[i]); }
    }
};

struct phip_scene {
    phip_scene_desc descCopy;        /* scalar fields only */
    HostBVH bvh;                     /* tree statistics (the node / record arrays are released after the upload) */
    int traversal = 2;               /* 2 = persistent per-lane BVH4 traversal with dynamic refill (default), 0 = one launch lane per slot (PHIP_TRAVERSAL=lane) */
    bool hasTextures = false; uint32_t triShadeStride = TRISHADE_FLOAT4S;
    int envLevelCount = 0;           /* MIP levels of the envmap (0: no envmap) */
    int materialMask = MM_ALL;       /* leaf BSDF models present: selects the k_shade instantiation */
    bool flatTraceToo = false;       /* a scene of k_mega that k_shade_trace could serve as well (PHIP_FLAG_NO_MEGA) */
    bool flatTrace = false;          /* not a scene of k_mega, but its tree is the packed leaf table (<= 64 Wald records) and emitter table + materials fit LDS: k_shade_trace */
    bool wide = false;               /* the ray kernels walk the compressed 8-wide BVH: every scene since round 6 */
    bool wideOnly = false;           /* ... and the device holds nothing else (trees of >= 64 BVH4 nodes or more than 64 Wald records): no BVH4, no leaf table, no LDS-resident kernels */
    bool fitsLds = false;            /* tree, Wald records, shading records, emitter table and materials fit the fused kernel's LDS plan */
    int fusedWide = 0;               /* round 6: 4 / 5 = the fused kernel walks the 8-wide tree from memory (k_mega<.., FLAT 4 / 5, ..>: emitter table in LDS; materials in LDS / in memory) */
    std::vector<std::unique_ptr<SceneDev>> devs;
    int *cancelFlag = nullptr;       /* host-pinned (portable, mapped): phip_cancel writes it, host loops and k_mega poll it */
    std::mutex renderLock;
    std::mutex progressLock;         /* the progress callback is entered by one device thread at a time (renderMultiDevice runs one host thread per GPU) */
    phip_scene() { devs.emplace_back(new SceneDev()); }
    ~phip_scene() { devs.clear(); if (cancelFlag) (void) hipHostFree(cancelFlag); }
};

static std::vector<DevMaterial> convertMaterials(const phip_material *materials, uint32_t nMaterials, const std::vector<float> *textureMax = nullptr) {
    if (nMaterials && !materials) throw std::runtime_error("materials is NULL");
    std::vector<DevMaterial> mats(nMaterials);
    /* a `bitmap` texture on specularReflectance (dielectric.cpp:159-160, roughconductor.cpp:173-174,236: ensureEnergyConservation) */
    auto specularTexture = [&](const phip_material &m, DevMaterial &o) {
        if (m.reflectance_texture == 0) return;
        if (!textureMax || m.reflectance_texture > textureMax->size()) throw std::runtime_error("material texture id out of range");
        if ((*textureMax)[m.reflectance_texture - 1] > 1.0f) throw std::runtime_error("specularReflectance texture > 1 (ensureEnergyConservation)");
        o.reflTexture = m.reflectance_texture;
    };
    for (uint32_t i = 0; i < nMaterials; ++i) {
        const phip_material &m = materials[i];
        DevMaterial &o = mats[i];
        memset(&o, 0, sizeof(o));
        o.type = m.type; o.nested0 = m.nested[0]; o.nested1 = m.nested[1];
        for (int k = 0; k < 3; ++k) { o.refl[k] = m.reflectance[k]; o.trans[k] = m.transmittance[k]; o.eta[k] = m.eta[k]; o.k[k] = m.k[k]; }
        o.distribution = m.distribution; o.sampleVisible = m.sample_visible ? 1 : 0;
        switch (m.type) {
            case PHIP_BSDF_DIFFUSE: {
                float mx = std::max(m.reflectance[0], std::max(m.reflectance[1], m.reflectance[2]));
                if (m.reflectance_texture != 0) {            /* m_reflectance->getMaximum().max() of the bitmap */
                    if (!textureMax || m.reflectance_texture > textureMax->size()) throw std::runtime_error("material texture id out of range");
                    mx = (*textureMax)[m.reflectance_texture - 1];
                    o.reflTexture = m.reflectance_texture;
                }
                if (mx > 1.0f) throw std::runtime_error("diffuse reflectance > 1 (ensureEnergyConservation, diffuse.cpp:95)");
                if (mx > 0) o.flags |= MF_SMOOTH;            /* component list empty otherwise, diffuse.cpp:97-100 */
            } break;
            case PHIP_BSDF_DIELECTRIC:
                if (!(m.eta[0] > 0)) throw std::runtime_error("dielectric eta must be positive");
                o.flags |= MF_TRANS_OR_BACK;
                specularTexture(m, o);
                if (m.transmittance_texture != 0) {          /* dielectric.cpp:161-162,207-208: ensureEnergyConservation(specularTransmittance) */
                    if (!textureMax || m.transmittance_texture > textureMax->size()) throw std::runtime_error("material texture id out of range");
                    if ((*textureMax)[m.transmittance_texture - 1] > 1.0f) throw std::runtime_error("specularTransmittance texture > 1 (ensureEnergyConservation)");
                    o.transTexture = m.transmittance_texture;
                }
                break;
            case PHIP_BSDF_ROUGHCONDUCTOR: {
                specularTexture(m, o);
                if (m.distribution > PHIP_MF_GGX) { g_err = "unsupported microfacet distribution"; throw std::invalid_argument("unsupported microfacet distribution (only beckmann, ggx)"); }
                o.flags |= MF_SMOOTH;
                /* alpha = ConstantFloatTexture.eval().average() (roughconductor.cpp:275-280), clamp microfacet.h:113-114 */
                o.alphaU = std::max(V3(m.alpha_u).average(), 1e-4f);
                o.alphaV = std::max(V3(m.alpha_v).average(), 1e-4f);
                /* a `bitmap` texture as the child "alpha" / "alphaU" / "alphaV" (roughconductor.cpp:424-431): evaluated per vertex */
                for (uint32_t t : { m.alpha_u_texture, m.alpha_v_texture })
                    if (t != 0 && (!textureMax || t > textureMax->size())) throw std::runtime_error("material texture id out of range");
                o.alphaUTexture = m.alpha_u_texture; o.alphaVTexture = m.alpha_v_texture;
            } break;
            case PHIP_BSDF_TWOSIDED: {
                if (m.nested[0] >= i || m.nested[1] >= i) throw std::runtime_error("twosided: nested materials must precede the adapter");
                const DevMaterial &a = mats[m.nested[0]], &b = mats[m.nested[1]];
                if ((a.type != PHIP_BSDF_DIFFUSE && a.type != PHIP_BSDF_ROUGHCONDUCTOR) || (b.type != PHIP_BSDF_DIFFUSE && b.type != PHIP_BSDF_ROUGHCONDUCTOR))
                    throw std::runtime_error("twosided: only materials without a transmission component can be nested (twosided.cpp:104-106)");
                if ((a.flags | b.flags) & MF_SMOOTH) o.flags |= MF_SMOOTH;
                o.flags |= MF_TRANS_OR_BACK;                  /* EBackSide, twosided.cpp:96-100 */
            } break;
            default: throw std::runtime_error("unknown bsdf type");
        }
    }
    return mats;
}

static void buildScene(phip_scene *sc, const phip_scene_desc &d) {
    SceneDev &sd = *sc->devs[0];
    if (d.abi_version != PHIP_ABI_VERSION) throw std::runtime_error("phip_scene_desc.abi_version mismatch");
    if (d.n_vertices && !d.positions) throw std::runtime_error("positions is NULL");
    if (d.n_triangles && !d.indices) throw std::runtime_error("indices is NULL");
    if (d.film.crop_width <= 0 || d.film.crop_height <= 0 || d.film.width <= 0 || d.film.height <= 0)
        throw std::runtime_error("invalid film size");
    if (d.film.crop_offset_x < 0 || d.film.crop_offset_y < 0 || d.film.crop_offset_x + d.film.crop_width > d.film.width ||
        d.film.crop_offset_y + d.film.crop_height > d.film.height)
        throw std::runtime_error("invalid crop window");          /* film.cpp:44-48 */
    if (d.film.crop_width >= 65536 || d.film.crop_height >= 65536)
        throw std::runtime_error("crop window of 65536 pixels or more per side (block origins are packed into 16 bits)");
    if (!(d.film.filter_radius > 0)) throw std::runtime_error("filter radius must be > 0");
    for (uint32_t i = 0; i < 3 * d.n_triangles; ++i)
        if (d.indices[i] >= d.n_vertices) throw std::runtime_error("triangle index out of range");

    /* shapes */
    std::vector<DevShape> shapes(d.n_shapes);
    std::vector<uint32_t> triShape(d.n_triangles);
    std::vector<float> areaCdf;
    uint32_t expect = 0;
    for (uint32_t i = 0; i < d.n_shapes; ++i) {
        const phip_shape &s = d.shapes[i];
        if (s.first_triangle != expect) throw std::runtime_error("shape triangle ranges must tile the index array in order");
        if (s.material >= d.n_materials) throw std::runtime_error("shape material id out of range");
        if (s.emitter >= (int32_t) d.n_emitters) throw std::runtime_error("shape emitter id out of range");
        if (s.emitter >= 0 && d.emitters[s.emitter].type != PHIP_EMITTER_AREA) throw std::runtime_error("a shape can only carry an area emitter");
        if (s.has_normals && !d.normals) throw std::runtime_error("shape has_normals but normals is NULL");
        expect += s.n_triangles;
        DevShape &o = shapes[i];
        if (s.has_texcoords && !d.texcoords) throw std::runtime_error("shape has_texcoords but texcoords is NULL");
        o.material = s.material; o.emitter = s.emitter; o.hasNormals = s.has_normals ? 1 : 0;
        o.firstTri = s.first_triangle; o.nTris = s.n_triangles; o.cdfOffset = 0; o.invSurfaceArea = 0; o.pad = s.has_texcoords ? 1 : 0;    /* pad: the mesh has texture coordinates */
        for (uint32_t j = 0; j < s.n_triangles; ++j) triShape[s.first_triangle + j] = i;
        if (s.emitter >= 0) {
            /* TriMesh::prepareSamplingTable, trimesh.cpp:388-404 + DiscreteDistribution::normalize */
            if (s.n_triangles == 0) throw std::runtime_error("area emitter on an empty mesh");
            o.cdfOffset = (uint32_t) areaCdf.size();
            std::vector<float> cdf(1, 0.0f);
            for (uint32_t j = 0; j < s.n_triangles; ++j) {
                const uint32_t *ix = d.indices + 3 * (size_t) (s.first_triangle + j);
                const float *p0 = d.positions + 3 * (size_t) ix[0], *p1 = d.positions + 3 * (size_t) ix[1], *p2 = d.positions + 3 * (size_t) ix[2];
                V3 a(p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]), b(p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]);
                cdf.push_back(cdf.back() + 0.5f * cross(a, b).length());
            }
            const float sum = cdf.back();
            if (!(sum > 0)) throw std::runtime_error("area emitter with zero surface area");
            const float norm = 1.0f / sum;
            for (size_t j = 1; j < cdf.size(); ++j) cdf[j] *= norm;
            cdf.back() = 1.0f;
            o.invSurfaceArea = 1.0f / sum;
            areaCdf.insert(areaCdf.end(), cdf.begin(), cdf.end());
        }
    }
    if (expect != d.n_triangles) throw std::runtime_error("shape triangle ranges do not cover the index array");

    /* bitmap textures: every pyramid (as delivered) in one float4 texel array + one descriptor each */
    std::vector<float4> texTexels; std::vector<DevMipLevels> texDesc(d.n_textures); std::vector<float> texMax(d.n_textures, 0.0f);
    if (d.n_textures && !d.textures) throw std::runtime_error("textures is NULL");
    for (uint32_t i = 0; i < d.n_textures; ++i) {
        const phip_texture &t = d.textures[i];
        if (t.width == 0 || t.height == 0 || !t.levels[0]) throw std::runtime_error("texture without level 0");
        if (t.wrap_u > PHIP_WRAP_ONE || t.wrap_v > PHIP_WRAP_ONE || t.filter_type > PHIP_FILTER_EWA) throw std::runtime_error("bad texture wrap mode / filter type");
        DevMipLevels &lv = texDesc[i]; memset(&lv, 0, sizeof(lv));
        int sx = (int) t.width, sy = (int) t.height, n = 1;
        lv.lw[0] = sx; lv.lh[0] = sy;
        if (t.n_levels > 1) {
            while (sx > 1 || sy > 1) {
                sx = std::max(1, (sx + 1) / 2); sy = std::max(1, (sy + 1) / 2);
                if (n >= PHIP_MIP_MAX_LEVELS) throw std::runtime_error("texture: too many MIP levels");
                lv.lw[n] = sx; lv.lh[n] = sy; ++n;
            }
            if ((uint32_t) n != t.n_levels) throw std::runtime_error("texture: n_levels must be 1 or the complete pyramid down to 1x1");
        }
        lv.nLevels = n;
        for (int l = 0; l < n; ++l) {
            if (!t.levels[l]) throw std::runtime_error("texture: level pointer is NULL");
            lv.offset[l] = (uint32_t) texTexels.size();
            const size_t cnt = (size_t) lv.lw[l] * lv.lh[l];
            for (size_t k = 0; k < cnt; ++k) {
                const float *c = t.levels[l] + 3 * k;
                texTexels.push_back(make_float4(c[0], c[1], c[2], 0.0f));
                if (l == 0) texMax[i] = std::max(texMax[i], std::max(c[0], std::max(c[1], c[2])));
            }
        }
        if (texTexels.size() >= (1ull << 32)) throw std::runtime_error("textures too large");
        lv.bcu = t.wrap_u; lv.bcv = t.wrap_v; lv.filterType = t.filter_type; lv.maxAnisotropy = t.max_anisotropy;
        lv.uvScale[0] = t.uv_scale[0]; lv.uvScale[1] = t.uv_scale[1]; lv.uvOffset[0] = t.uv_offset[0]; lv.uvOffset[1] = t.uv_offset[1];
        for (int k = 0; k < 64; ++k) { const float r2 = (float) k / 63.0f; lv.weightLut[k] = pm_expf(-2.0f * r2) - pm_expf(-2.0f); }
    }

    /* materials */
    std::vector<DevMaterial> mats = convertMaterials(d.materials, d.n_materials, &texMax);
    /* TriMesh::computeUVTangents, trimesh.cpp:683-693: an anisotropic BSDF (roughconductor.cpp:196-200,230-231: the clamped
       alphaU != alphaV; twosided.cpp:96-100 inherits the flag) needs texture coordinates for its tangent frame -- an error there */
    for (uint32_t i = 0; i < d.n_shapes; ++i) {
        std::function<bool(uint32_t, int)> aniso = [&](uint32_t m, int depth) -> bool {
            if (m >= d.n_materials || depth > 2) return false;
            const phip_material &M = d.materials[m];
            if (M.type == PHIP_BSDF_ROUGHCONDUCTOR)            /* m_alphaU != m_alphaV as OBJECTS: one texture for both is isotropic, roughconductor.cpp:228-229 */
                return (M.alpha_u_texture | M.alpha_v_texture) ? M.alpha_u_texture != M.alpha_v_texture : std::max(M.alpha_u, 1e-4f) != std::max(M.alpha_v, 1e-4f);
            if (M.type == PHIP_BSDF_TWOSIDED) return aniso(M.nested[0], depth + 1) || aniso(M.nested[1], depth + 1);
            return false;
        };
        if (!d.shapes[i].has_texcoords && aniso(d.shapes[i].material, 0))
            throw std::runtime_error("computeUVTangents(): texture coordinates are required to generate tangent vectors (anisotropic BSDF on a shape without them)");
    }

    /* emitters + selection pdf, scene.cpp:375-381 */
    std::vector<DevEmitter> ems(d.n_emitters);
    int32_t envEmitter = -1; bool envIsMap = false;
    std::vector<float> ecdf(1, 0.0f);
    for (uint32_t i = 0; i < d.n_emitters; ++i) {
        const phip_emitter &e = d.emitters[i];
        if (e.type == PHIP_EMITTER_CONSTANT || e.type == PHIP_EMITTER_ENVMAP) {
            if (envEmitter >= 0) throw std::runtime_error("The scene may only contain one environment emitter");   /* scene.cpp:510-513 */
            envEmitter = (int32_t) i;
            envIsMap = e.type == PHIP_EMITTER_ENVMAP;
        } else if (e.type != PHIP_EMITTER_AREA) throw std::runtime_error("unknown emitter type");
        else if (e.shape >= d.n_shapes || d.shapes[e.shape].emitter != (int32_t) i) throw std::runtime_error("emitter/shape back reference mismatch");
        memset(&ems[i], 0, sizeof(DevEmitter));
        for (int k = 0; k < 3; ++k) ems[i].radiance[k] = e.radiance[k];
        ems[i].samplingWeight = e.sampling_weight; ems[i].shape = e.type == PHIP_EMITTER_AREA ? e.shape : 0xFFFFFFFFu;
        ecdf.push_back(ecdf.back() + e.sampling_weight);
    }
    float emNorm = 0;
    if (d.n_emitters) {
        const float sum = ecdf.back();
        if (sum > 0) { emNorm = 1.0f / sum; for (size_t j = 1; j < ecdf.size(); ++j) ecdf[j] *= emNorm; ecdf.back() = 1.0f; }
    }

    /* acceleration structure */
    {
        const float camPos[3] = { d.camera.to_world[3], d.camera.to_world[7], d.camera.to_world[11] };
        buildBVH(d.positions, d.indices, d.n_triangles, sc->bvh, camPos);
    }
    if (sc->bvh.tris.size() / 12 >= (1u << 28)) throw std::runtime_error("too many triangle records for the leaf reference encoding");
    if (d.n_triangles > HIT_PRIM_MASK) throw std::runtime_error("too many triangles for the hit record (30-bit primitive index)");

    /* upload */
    HIP_TRY(hipSetDevice(sd.device));
    /* shading records (dv_scene.h): the per-triangle constants come from the same __host__ __device__
       functions the kernel would run, so precomputing them does not change a single bit */
    bool anyTexcoords = false;
    for (uint32_t i = 0; i < d.n_shapes; ++i) anyTexcoords |= d.shapes[i].has_texcoords != 0;
    const uint32_t stride = anyTexcoords ? TRISHADE_FLOAT4S_UV : TRISHADE_FLOAT4S;
    std::vector<float4> ts((size_t) stride * d.n_triangles, make_float4(0, 0, 0, 0));
    for (uint32_t i = 0; i < d.n_triangles; ++i) {
        const DevShape &sh = shapes[triShape[i]];
        const DevMaterial &m = mats[sh.material];
        const uint32_t *ix = d.indices + 3 * (size_t) i;
        const V3 p0(d.positions[3 * ix[0]], d.positions[3 * ix[0] + 1], d.positions[3 * ix[0] + 2]);
        const V3 p1(d.positions[3 * ix[1]], d.positions[3 * ix[1] + 1], d.positions[3 * ix[1] + 2]);
        const V3 p2(d.positions[3 * ix[2]], d.positions[3 * ix[2] + 1], d.positions[3 * ix[2] + 2]);
        const bool twosided = m.type == PHIP_BSDF_TWOSIDED, texcoords = sh.pad != 0;
        const uint32_t front = twosided ? m.nested0 : sh.material, back = twosided ? m.nested1 : sh.material;
        uint32_t flags = (sh.hasNormals ? TS_VERTEX_NORMALS : 0u) | (twosided ? TS_TWOSIDED : 0u) | (texcoords ? TS_TEXCOORDS : 0u)
                       | ((m.flags & MF_SMOOTH) ? TS_MF_SMOOTH : 0u) | ((m.flags & MF_TRANS_OR_BACK) ? TS_TRANS_OR_BACK : 0u);
        const V3 side1(p1 - p0), side2(p2 - p0);
        V3 dpdu = side1, dpdv = side2;                      /* skdtree.h:378-379 */
        float4 *r = ts.data() + (size_t) stride * i;
        if (texcoords) {
            /* TriMesh::computeUVTangents, trimesh.cpp:683-735 (zero tangents for degenerate triangles) */
            const float *t0 = d.texcoords + 2 * (size_t) ix[0], *t1 = d.texcoords + 2 * (size_t) ix[1], *t2 = d.texcoords + 2 * (size_t) ix[2];
            dpdu = V3(0.0f); dpdv = V3(0.0f);
            const V2 dUV1(t1[0] - t0[0], t1[1] - t0[1]), dUV2(t2[0] - t0[0], t2[1] - t0[1]);
            const V3 n = cross(side1, side2);
            const float length = n.length();
            if (length != 0) {
                const float determinant = dUV1.x * dUV2.y - dUV1.y * dUV2.x;
                if (determinant == 0) {
                    coordinateSystem(n / length, dpdu, dpdv);
                } else {
                    const float invDet = 1.0f / determinant;
                    dpdu = (side1 * dUV2.y - side2 * dUV1.y) * invDet;
                    dpdv = (side1 * (-dUV2.x) + side2 * dUV1.x) * invDet;
                }
            }
            r[6] = make_float4(t0[0], t0[1], t1[0], t1[1]);
            r[7] = make_float4(t2[0], t2[1], dpdu.x, dpdu.y);
            r[8] = make_float4(dpdu.z, dpdv.x, dpdv.y, dpdv.z);
        }
        V3 a, b, c;
        if (sh.hasNormals) {
            a = V3(d.normals[3 * ix[0]], d.normals[3 * ix[0] + 1], d.normals[3 * ix[0] + 2]);
            b = V3(d.normals[3 * ix[1]], d.normals[3 * ix[1] + 1], d.normals[3 * ix[1] + 2]);
            c = V3(d.normals[3 * ix[2]], d.normals[3 * ix[2] + 1], d.normals[3 * ix[2] + 2]);
        } else {
            Frame f; triShadingFrame(triFaceNormal(side1, side2), dpdu, f);
            a = f.n; b = f.s; c = f.t;
        }
        r[0] = make_float4(p0.x, p0.y, p0.z, pm_from_bits(front));
        r[1] = make_float4(p1.x, p1.y, p1.z, pm_from_bits(back));
        r[2] = make_float4(p2.x, p2.y, p2.z, pm_from_bits((uint32_t) sh.emitter));
        r[3] = make_float4(a.x, a.y, a.z, pm_from_bits(flags));
        r[4] = make_float4(b.x, b.y, b.z, 0.0f);
        r[5] = make_float4(c.x, c.y, c.z, 0.0f);
    }
    /* shade class of every triangle in the spare word of its Wald record(s): 0 diffuse, 1 rough conductor, 2 dielectric -- the heavier
       of the two sides of a two-sided surface; k_rays_w passes it on in the hit record and k_shade deals its lanes by it (k_pool.h) */
    {
        auto classOf = [&](uint32_t leaf) { const int t = mats[leaf].type; return t == PHIP_BSDF_ROUGHCONDUCTOR ? 1u : (t == PHIP_BSDF_DIELECTRIC ? 2u : 0u); };
        std::vector<uint8_t> cls(d.n_triangles);
        for (uint32_t i = 0; i < d.n_triangles; ++i) {
            const DevMaterial &m = mats[shapes[triShape[i]].material];
            const bool twosided = m.type == PHIP_BSDF_TWOSIDED;
            const uint32_t a = classOf(twosided ? m.nested0 : shapes[triShape[i]].material), b = classOf(twosided ? m.nested1 : shapes[triShape[i]].material);
            cls[i] = (uint8_t) ((a == 1u || b == 1u) ? 1u : std::max(a, b));
        }
        for (std::vector<float> *recs : { &sc->bvh.tris, &sc->bvh.wtris })
            for (size_t r = 0; r + 12 <= recs->size(); r += 12) {
                uint32_t prim; memcpy(&prim, &(*recs)[r + 10], 4);
                const uint32_t c = prim < d.n_triangles ? cls[prim] : 0u;
                memcpy(&(*recs)[r + 11], &c, 4);
            }
    }
    if (ts.empty()) sd.triShade.alloc(TRISHADE_FLOAT4S_UV); else sd.triShade.upload(ts.data(), ts.size());
    if (texTexels.empty()) sd.texTexels.alloc(1); else sd.texTexels.upload(texTexels.data(), texTexels.size());
    if (texDesc.empty()) sd.texDesc.alloc(1); else sd.texDesc.upload(texDesc.data(), texDesc.size());
    sc->hasTextures = d.n_textures > 0; sc->triShadeStride = stride;
    if (const char *e = expEnv("PHIP_TRAVERSAL")) sc->traversal = strcmp(e, "lane") == 0 ? 0 : 2;
    /* the compressed 8-wide tree: every scene the packed leaf table of the LDS-resident kernels does not serve (more than 64 Wald records).  Round 6: it used to start at 64 BVH4
       nodes; the scenes in between ran the round-1 BVH4 kernels (or, all-diffuse ones, k_mega's BVH4 walk in LDS) -- now they run k_mega on the wide tree / k_rays_w */
    if (sc->bvh.nWNodes == 0) {      /* a scene without triangles: one node without children (all-zero meta bytes hit nothing) -- the ray kernels need a root to reject */
        sc->bvh.wnodes.assign(20, 0u); sc->bvh.nWNodes = 1; sc->bvh.wMaxDepth = 1;
    }
    sc->wideOnly = sc->traversal == 2 && sc->bvh.nWNodes > 0 && (sc->bvh.nNodes >= 64 || sc->bvh.tris.size() / 12 > PHIP_WIDE_MIN_RECORDS);
    if (const char *e = expEnv("PHIP_WIDE")) sc->wideOnly = sc->wideOnly && atoi(e) != 0;
    /* ... and, round 6, EVERY scene has the wide tree on the device: the ray kernels of the wavefront path (k_rays_w) and phip_trace (k_raycast_w) walk nothing else -- the
       round-1 BVH4 ray kernels are compiled by experiment builds only.  The LDS-resident scenes keep their BVH4-ordered records and leaf table for k_mega / k_shade_trace beside it */
    sc->wide = sc->traversal == 2 && sc->bvh.nWNodes > 0 && (sc->wideOnly || !(PHIP_EXPERIMENTS && expEnv("PHIP_BVH4_RAYS")));
    /* the stack of the structure that is going to be walked: three pushes per BVH4 level (the wide tree's group stack is checked below) */
    if (!sc->wide && 3 * sc->bvh.maxDepth + 2 > STACK_DEPTH + SPILL_DEPTH) throw std::runtime_error("BVH too deep for the traversal stack");
    if (sc->wide) {
        if (WIDE_NODE_STRIDE == 5) sd.wnodes.upload((const uint4 *) sc->bvh.wnodes.data(), sc->bvh.wnodes.size() / 4);
        else {                                               /* one node per WIDE_NODE_STRIDE * 16 bytes (a 128-byte line) */
            const size_t n = sc->bvh.wnodes.size() / 20;
            std::vector<uint4> padded(n * WIDE_NODE_STRIDE, make_uint4(0, 0, 0, 0));
            for (size_t i = 0; i < n; ++i) memcpy(&padded[i * WIDE_NODE_STRIDE], &sc->bvh.wnodes[i * 20], 80);
            sd.wnodes.upload(padded.data(), padded.size());
        }
        if (sc->bvh.wtris.empty()) sd.wtris.alloc(3); else sd.wtris.upload((const float4 *) sc->bvh.wtris.data(), sc->bvh.wtris.size() / 4);
    } else { sd.wnodes.alloc(5); sd.wtris.alloc(3); }
    if (sc->wideOnly) {
        sd.nodes.alloc(8); sd.tris.alloc(3); sd.trisAreWide = true;      /* (DevScene::tris = wtris: SceneDev::bind) */
    } else {
        if (sc->bvh.nodes.empty()) sd.nodes.alloc(8);
        else sd.nodes.upload((const float4 *) sc->bvh.nodes.data(), sc->bvh.nodes.size() / 4);
        sd.tris.upload((const float4 *) sc->bvh.tris.data(), sc->bvh.tris.size() / 4);
    }
    sd.materials.upload(mats.data(), mats.size());
    sc->materialMask = 0;
    for (const DevMaterial &m : mats) {
        if (m.type == PHIP_BSDF_ROUGHCONDUCTOR) sc->materialMask |= MM_ROUGH;
        if (m.type == PHIP_BSDF_DIELECTRIC) sc->materialMask |= MM_DIELECTRIC;
    }
    if (const char *e = expEnv("PHIP_SHADE_GENERIC")) if (atoi(e)) sc->materialMask = MM_ALL;
    /* packed emitter table (dv_scene.h: EmitterTab) */
    std::vector<float> tab(ecdf);
    tab.resize(ecdf.size() + (size_t) EM_STRIDE * d.n_emitters, 0.0f);
    const size_t cdfBase = tab.size();
    size_t nEmTris = 0;
    auto isArea = [&](uint32_t i) { return ems[i].shape != 0xFFFFFFFFu; };
    for (uint32_t i = 0; i < d.n_emitters; ++i) if (isArea(i)) nEmTris += shapes[ems[i].shape].nTris;
    const size_t recBase = (cdfBase + areaCdf.size() + 3) / 4 * 4;                /* 16-byte aligned */
    const bool withRecs = recBase + nEmTris * 4 * TRISHADE_FLOAT4S <= EMITTER_LDS_FLOATS;
    size_t recPos = recBase;
    for (uint32_t i = 0; i < d.n_emitters; ++i) {
        float *r = tab.data() + ecdf.size() + (size_t) EM_STRIDE * i;
        for (int k = 0; k < 3; ++k) r[EM_RADIANCE + k] = ems[i].radiance[k];
        r[EM_WEIGHT] = ems[i].samplingWeight;
        r[EM_TYPE] = pm_from_bits(d.emitters[i].type);
        if (!isArea(i)) continue;
        const DevShape &sh = shapes[ems[i].shape];
        r[EM_FIRST_TRI] = pm_from_bits(sh.firstTri); r[EM_N_TRIS] = pm_from_bits(sh.nTris);
        r[EM_CDF] = pm_from_bits((uint32_t) (cdfBase + sh.cdfOffset));             /* area CDFs follow the records */
        r[EM_INV_AREA] = sh.invSurfaceArea;
        r[EM_REC] = pm_from_bits(withRecs ? (uint32_t) recPos : 0u);
        recPos += (size_t) sh.nTris * 4 * TRISHADE_FLOAT4S;
    }
    tab.insert(tab.end(), areaCdf.begin(), areaCdf.end());
    if (withRecs) {
        tab.resize(recBase, 0.0f);
        for (uint32_t i = 0; i < d.n_emitters; ++i) {
            if (!isArea(i)) continue;
            const DevShape &sh = shapes[ems[i].shape];
            for (uint32_t k = 0; k < sh.nTris; ++k) {       /* the first six float4s of each record (positions, normal / vertex normals) */
                const float *src = (const float *) (ts.data() + (size_t) stride * (sh.firstTri + k));
                tab.insert(tab.end(), src, src + 4 * TRISHADE_FLOAT4S);
            }
        }
    }
    if (tab.size() >= (1ull << 31)) throw std::runtime_error("emitter table too large");
    tab.resize((tab.size() + 3) / 4 * 4, 0.0f);              /* whole float4s: the shading kernels stage the table in LDS with 16-byte loads */
    sd.emitterTab.upload(tab.data(), tab.size());

    DevScene &D = sd.dev;
    memset(&D, 0, sizeof(D));
    D.nodes = sd.nodes.p; D.wtris = sd.wtris.p; D.tris = sd.trisAreWide ? sd.wtris.p : sd.tris.p; D.triShade = sd.triShade.p;
    D.materials = sd.materials.p; D.nMaterials = (uint32_t) mats.size();
    D.texTexels = sd.texTexels.p; D.textures = sd.texDesc.p; D.triShadeStride = sc->triShadeStride;
    D.emitterTab = sd.emitterTab.p; D.emitterTabSize = (uint32_t) tab.size();
    D.nEmitters = d.n_emitters; D.emitterNormalization = emNorm;
    D.envEmitter = envEmitter;
    if (envEmitter >= 0) {
        /* ConstantBackgroundEmitter::createShape (constant.cpp:67-72) as seen from Scene::initializeBidirectional
           (scene.cpp:384-413): bounding sphere (aabb.cpp:44-47) of the kd-tree's enlarged box expanded by the sensor
           position (track.cpp:79-83), radius x 1.5 */
        float mn[3], mx[3];
        for (int a = 0; a < 3; ++a) {
            mn[a] = d.n_triangles ? sc->bvh.sceneMin[a] : INFINITY; mx[a] = d.n_triangles ? sc->bvh.sceneMax[a] : -INFINITY;
        }
        const float *m = d.camera.to_world;
        V3 sp(m[3], m[7], m[11]);
        if (m[15] != 1.0f) sp = sp / m[15];
        const float spv[3] = { sp.x, sp.y, sp.z };
        for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], spv[a]); mx[a] = std::max(mx[a], spv[a]); }
        const V3 center = (V3(mx[0], mx[1], mx[2]) + V3(mn[0], mn[1], mn[2])) * 0.5f;
        const float radius = (center - V3(mx[0], mx[1], mx[2])).length();
        D.envCenter[0] = center.x; D.envCenter[1] = center.y; D.envCenter[2] = center.z;
        D.envRadius = std::max(PT_EPSILON, radius * 1.5f);
    }
    if (envIsMap) {
        /* EnvironmentMap::configure, envmap.cpp:262-328: marginal / conditional CDFs over luminance * sin(theta) */
        const phip_envmap &e = d.envmap;
        if (!e.texels || e.width == 0 || e.height == 0) throw std::runtime_error("envmap emitter without texels");
        if (std::max(e.width, e.height) > 0xFFFF) throw std::runtime_error("Environment maps images must be smaller than 65536 pixels in width and height");
        const int w = (int) e.width, h = (int) e.height;
        std::vector<float4> tex((size_t) w * h);
        for (size_t i = 0; i < tex.size(); ++i) tex[i] = make_float4(e.texels[3 * i], e.texels[3 * i + 1], e.texels[3 * i + 2], 0.0f);
        /* MIP pyramid (level sizes of mipmap.h:182-192) + EWA weight table (mipmap.h:296-301) */
        DevMipLevels lv; memset(&lv, 0, sizeof(lv));
        lv.nLevels = 1; lv.lw[0] = w; lv.lh[0] = h; lv.offset[0] = 0;
        lv.bcu = PHIP_WRAP_REPEAT; lv.bcv = PHIP_WRAP_CLAMP; lv.filterType = PHIP_FILTER_EWA; lv.maxAnisotropy = 10.0f;   /* envmap.cpp:138-139,176-178 */
        if (e.n_levels > 1) {
            int sx = w, sy = h, n = 1;
            while (sx > 1 || sy > 1) {
                sx = std::max(1, (sx + 1) / 2); sy = std::max(1, (sy + 1) / 2);
                if (n >= PHIP_ENVMAP_MAX_LEVELS) throw std::runtime_error("envmap: too many MIP levels");
                lv.lw[n] = sx; lv.lh[n] = sy; ++n;
            }
            if ((uint32_t) n != e.n_levels) throw std::runtime_error("envmap: n_levels must be 1 or the complete pyramid down to 1x1");
            lv.nLevels = n;
            for (int l = 1; l < n; ++l) {
                if (!e.levels[l]) throw std::runtime_error("envmap: level pointer is NULL");
                lv.offset[l] = (uint32_t) tex.size();
                const size_t cnt = (size_t) lv.lw[l] * lv.lh[l];
                for (size_t i = 0; i < cnt; ++i) tex.push_back(make_float4(e.levels[l][3 * i], e.levels[l][3 * i + 1], e.levels[l][3 * i + 2], 0.0f));
            }
        }
        for (int i = 0; i < 64; ++i) { const float r2 = (float) i / 63.0f; lv.weightLut[i] = pm_expf(-2.0f * r2) - pm_expf(-2.0f); }
        sd.envLevels.upload(&lv, 1);
        sc->envLevelCount = lv.nLevels;
        std::vector<float> cdfCols((size_t) (w + 1) * h), cdfRows((size_t) h + 1), rowWeights((size_t) h);
        size_t colPos = 0, rowPos = 0;
        float rowSum = 0.0f;
        cdfRows[rowPos++] = 0;
        for (int y = 0; y < h; ++y) {
            float colSum = 0;
            cdfCols[colPos++] = 0;
            for (int x = 0; x < w; ++x) {
                const float4 &t = tex[(size_t) y * w + x];
                colSum += rgbLuminance(V3(t.x, t.y, t.z));
                cdfCols[colPos++] = colSum;
            }
            const float norm = 1.0f / colSum;
            for (int x = 1; x < w; ++x) cdfCols[colPos - x - 1] *= norm;
            cdfCols[colPos - 1] = 1.0f;
            float sn, cs; pm_sincosf((y + 0.5f) * PT_PI / h, &sn, &cs);
            rowWeights[y] = sn;
            rowSum += colSum * sn;
            cdfRows[rowPos++] = rowSum;
        }
        const float norm = 1.0f / rowSum;
        for (int y = 1; y < h; ++y) cdfRows[rowPos - y - 1] *= norm;
        cdfRows[rowPos - 1] = 1.0f;
        if (rowSum == 0) throw std::runtime_error("The environment map is completely black -- this is not allowed.");
        if (!std::isfinite(rowSum)) throw std::runtime_error("The environment map contains an invalid floating point value (nan/inf) -- giving up.");
        M4 tw, tl;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) tw.m[i][j] = e.to_world[4 * i + j];
        if (!m4invert(tw, tl)) throw std::runtime_error("envmap toWorld is singular");
        sd.envTexels.upload(tex.data(), tex.size());
        sd.envCdfRows.upload(cdfRows.data(), cdfRows.size()); sd.envCdfCols.upload(cdfCols.data(), cdfCols.size());
        sd.envRowWeights.upload(rowWeights.data(), rowWeights.size());
        DevEnvMap &E = D.env;
        E.texels = sd.envTexels.p; E.levels = sd.envLevels.p; E.cdfRows = sd.envCdfRows.p; E.cdfCols = sd.envCdfCols.p; E.rowWeights = sd.envRowWeights.p;
        E.w = w; E.h = h; E.scale = e.scale;
        E.normalization = 1.0f / (rowSum * (2 * PT_PI / w) * (PT_PI / h));
        E.pixelSizeX = 2 * PT_PI / w; E.pixelSizeY = PT_PI / h;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { E.toWorld[3 * i + j] = tw.m[i][j]; E.toLocal[3 * i + j] = tl.m[i][j]; }
    }
    D.rootRef = sc->bvh.rootRef; D.nTriangles = d.n_triangles;
    /* LDS staging plan: stack depth from the tree depth (3 pushes per BVH4 level), top-of-tree node cache
       (nodes are in breadth-first order), all triangle records if there are few */
    D.stackDepth = (uint32_t) std::min<int>(STACK_DEPTH, std::max<int>(4, 3 * ((int) sc->bvh.maxDepth - 1) + 1));
    D.nodeCache = std::min<uint32_t>(sc->bvh.nNodes, NODE_CACHE_MAX);
    D.triCache = (sc->bvh.tris.size() / 12 <= TRI_CACHE_MAX) ? (uint32_t) (sc->bvh.tris.size() / 12) : 0u;
    if (const char *e = expEnv("PHIP_NODE_CACHE")) D.nodeCache = std::min<uint32_t>(sc->bvh.nNodes, (uint32_t) atoi(e));
    if (D.nodeCache == 0) D.triCache = 0;
    D.wnodes = sd.wnodes.p; D.wideNodeCache = 0;
    D.preclip = sc->wide ? 1u : 0u;                          /* k_rays_w traverses rays the shading kernels have clipped (k_clip.h) */
    /* the lane deal of k_shade pays where the expensive model is rare: rough conductors (microfacet sampling: atrium, 8 % of the vertices,
       k_shade -7 %); on a diffuse + dielectric mix the extra round trip costs more than the cheap Fresnel branch (glass room: +8 %) */
    D.shadeSort = (sc->wide && (sc->materialMask & MM_ROUGH)) ? 1u : 0u;
    if (const char *e = expEnv("PHIP_SHADE_SORT")) D.shadeSort = D.shadeSort && atoi(e) != 0;       /* experiment hook */
    if (sc->wide) {
        if (sc->wideOnly) { D.nodeCache = 0; D.triCache = 0; }
        D.wideNodeCache = std::min<uint32_t>(sc->bvh.nWNodes, WIDE_NODE_CACHE_MAX);
        if (const char *e = expEnv("PHIP_NODE_CACHE")) D.wideNodeCache = std::min<uint32_t>(D.wideNodeCache, (uint32_t) atoi(e));
        if ((int) sc->bvh.wMaxDepth + 2 > WIDE_STACK_LDS + SPILL_DEPTH / 2) throw std::runtime_error("wide BVH too deep for the traversal stack");
    }
    for (int a = 0; a < 3; ++a) { D.sceneMin[a] = sc->bvh.sceneMin[a]; D.sceneMax[a] = sc->bvh.sceneMax[a]; }
    setupCamera(d.camera, d.film, D.cam);
    D.film.width = d.film.crop_width; D.film.height = d.film.crop_height;
    D.film.radius = d.film.filter_radius;
    D.film.scaleFactor = PHIP_FILTER_RESOLUTION / d.film.filter_radius;     /* rfilter.cpp:50 */
    D.film.border = (int) std::ceil(d.film.filter_radius - 0.5f);           /* rfilter.cpp:51 */
    D.film.blockSize = 32;
    for (int i = 0; i <= PHIP_FILTER_RESOLUTION; ++i) D.film.table[i] = d.film.filter_table[i];

    sc->descCopy = d;
    sc->descCopy.positions = nullptr; sc->descCopy.normals = nullptr; sc->descCopy.indices = nullptr;
    sc->descCopy.shapes = nullptr; sc->descCopy.materials = nullptr; sc->descCopy.emitters = nullptr;
    /* the fused kernel's LDS plan (k_mega.h): the whole tree, every Wald and shading record, the emitter table and the
       materials in LDS, a stack that cannot spill; diffuse materials only (phip_mega.hip) */
    /* (treeInLds: the whole tree and every Wald record are staged in LDS and the stack cannot spill -- also true of small scenes with glass, copper,
       textures or an environment emitter, which k_shade_trace serves on the same packed leaf table: k_shade_trace.h) */
    const bool treeInLds = D.nodeCache == sc->bvh.nNodes && D.triCache == sc->bvh.tris.size() / 12 && 3 * ((int) sc->bvh.maxDepth - 1) + 1 <= (int) D.stackDepth;
    /* (leaf BSDF models: any -- round 5 -- when the tree is the packed leaf table; diffuse only for the trees the fused kernel walks or sweeps leaf by leaf: decided
       below, once the table is built) */
    const bool fitsLdsBase = !sc->hasTextures && envEmitter < 0 && stride == TRISHADE_FLOAT4S
        && treeInLds && d.n_triangles <= MEGA_TRISHADE_MAX
        && tab.size() <= EMITTER_LDS_FLOATS && mats.size() <= MATERIAL_LDS_MAX;
    sc->fitsLds = fitsLdsBase && sc->materialMask == 0;
    /* ... and, for trees of at most FLAT_LEAVES_MAX leaves (the Cornell box: 17), the leaves as a flat table: the fused kernel tests
       every leaf box in one uniform pass instead of walking the 7-node tree (k_traverse.h: traverseFlat).  Entry = (min.xyz, bits(leaf
       reference)) (max.xyz, 0), boxes as the BVH4 nodes hold them (padded). */
    D.nFlatLeaves = 0; D.flatMode = 0; sd.flatLeaves.alloc(2); D.flatLeaves = sd.flatLeaves.p;
    if (treeInLds && !sc->wideOnly && sc->bvh.nLeaves <= FLAT2_LEAVES_MAX && !expEnv("PHIP_NO_FLAT")) {
        std::vector<float4> flat;
        if (sc->bvh.rootRef < 0) {                         /* a single leaf: its box is the scene's */
            flat.push_back(make_float4(sc->bvh.tightMin[0] - 1.0f, sc->bvh.tightMin[1] - 1.0f, sc->bvh.tightMin[2] - 1.0f, pm_from_bits((uint32_t) sc->bvh.rootRef)));
            flat.push_back(make_float4(sc->bvh.tightMax[0] + 1.0f, sc->bvh.tightMax[1] + 1.0f, sc->bvh.tightMax[2] + 1.0f, 0.0f));
        } else for (uint32_t n = 0; n < sc->bvh.nNodes; ++n) {
            const float *nd = &sc->bvh.nodes[(size_t) n * 32];
            for (int c = 0; c < 4; ++c) {
                uint32_t ref; memcpy(&ref, &nd[24 + c], 4);
                if (nd[c] == INFINITY || (int32_t) ref >= 0) continue;           /* empty slot / inner child */
                flat.push_back(make_float4(nd[c], nd[4 + c], nd[8 + c], pm_from_bits(ref)));
                flat.push_back(make_float4(nd[12 + c], nd[16 + c], nd[20 + c], 0.0f));
            }
        }
        /* at most 32 Wald records: the packed form with record masks (k_traverse.h: traverseFlat2).  A leaf reference is
           ~((first record << 3) | records - 1); a triangle referenced by several leaves (spatial splits) has one record per
           reference -- the copies carry the same 12 words, so only the first copy's bit is set.
           Round 5: 33..64 records keep the packed form with a two-word mask (flatMode 3; the centre / half-extent table of the dealt
           traversal only -- the high word rides in the centre's spare word) */
        const size_t nRec = sc->bvh.tris.size() / 12;
        const size_t packedMax = (MEGA_FLAT_CH && MEGA_BALANCE && !expEnv("PHIP_NO_FLAT3")) ? 64 : 32;
        if (flat.size() / 2 <= FLAT2_LEAVES_MAX && nRec <= packedMax && !expEnv("PHIP_NO_FLAT2")) {
            std::vector<uint32_t> firstCopy(nRec);
            for (size_t i = 0; i < nRec; ++i) {
                firstCopy[i] = (uint32_t) i;
                for (size_t j = 0; j < i; ++j) if (!memcmp(&sc->bvh.tris[12 * i], &sc->bvh.tris[12 * j], 48)) { firstCopy[i] = (uint32_t) j; break; }
            }
            std::vector<float4> packed;
            for (size_t l = 0; l < flat.size() / 2; ++l) {
                const float4 mn = flat[2 * l], mx = flat[2 * l + 1];
                const uint32_t r = ~pm_to_bits(mn.w), first = r >> 3, count = (r & 7u) + 1u;
                unsigned long long bits64 = 0;
                for (uint32_t i = 0; i < count; ++i) bits64 |= 1ull << firstCopy[first + i];
                const uint32_t bits = (uint32_t) bits64, bitsHi = (uint32_t) (bits64 >> 32);
#if MEGA_FLAT_CH
                /* centre / half extent (k_traverse.h: flat2Pass1).  c -+ h must cover the (padded) box whatever the rounding of c, and the
                   distances c' -+ h |rcp| are rounded differently from the plane form the pad of bvh.h was sized for (two roundings of
                   magnitude |c rcp| + |o rcp| instead of one): h gets the rounding of c and another 4e-6 of the scene's extent on top */
                const float ext = std::max(sc->bvh.tightMax[0] - sc->bvh.tightMin[0], std::max(sc->bvh.tightMax[1] - sc->bvh.tightMin[1], sc->bvh.tightMax[2] - sc->bvh.tightMin[2]));
                const float camMax = std::max(std::fabs(d.camera.to_world[3]), std::max(std::fabs(d.camera.to_world[7]), std::fabs(d.camera.to_world[11])));
                float c[3], h[3];
                const float lo[3] = { mn.x, mn.y, mn.z }, hi[3] = { mx.x, mx.y, mx.z };
                for (int a = 0; a < 3; ++a) {
                    c[a] = (float) (0.5 * ((double) lo[a] + (double) hi[a]));
                    const double need = std::max((double) c[a] - (double) lo[a], (double) hi[a] - (double) c[a]);
                    /* ... and the rounding of o rcp, which grows with the ORIGIN's magnitude (two roundings of |o rcp| move a plane by ~2^-23 |o|): the only rays
                       that start outside the scene box are the camera's, so the camera position pays for it (a camera 60 scene extents away used to lose
                       leaf boxes; tests/test_gpu_parity.py: far camera) */
                    h[a] = std::nextafter((float) need, INFINITY) + 2.4e-7f * std::fabs(c[a]) + 4e-6f * ext + 4.8e-7f * camMax;      /* (the largest component on every axis: bvh.h, buildBVH) */
                }
                packed.push_back(make_float4(c[0], c[1], c[2], pm_from_bits(bitsHi)));
                packed.push_back(make_float4(h[0], h[1], h[2], pm_from_bits(bits)));
#else
                (void) bitsHi;
                packed.push_back(make_float4(mn.x, mx.x, mn.y, mx.y));
                packed.push_back(make_float4(mn.z, mx.z, pm_from_bits(bits), 0.0f));
#endif
            }
            flat.swap(packed); D.flatMode = nRec <= 32 ? 2 : 3;
            /* k_mega deals the Wald tests over the wave through LDS buffers that lie over the (then unused) traversal stack (k_traverse.h: traverseFlat2W) */
            D.stackDepth = std::max<uint32_t>(D.stackDepth, ((BLOCK / 64) * BAL_WAVE_BYTES + BLOCK * sizeof(uint32_t) - 1) / (BLOCK * sizeof(uint32_t)));
        } else if (sc->fitsLds && flat.size() / 2 <= FLAT_LEAVES_MAX)
            D.flatMode = 1;
        if (D.flatMode) {
            sd.flatLeaves.upload(flat.data(), flat.size());
            D.flatLeaves = sd.flatLeaves.p; D.nFlatLeaves = (uint32_t) (flat.size() / 2);
        }
    }
    if (fitsLdsBase && sc->materialMask != 0 && D.flatMode >= 2 && !expEnv("PHIP_NO_MEGA_MATERIALS")) {
        sc->fitsLds = true;
        /* k_mega<MM_ALL> keeps its mailbox of copper vertices (QMC build: the exchange buffer of its class deal) in MEGA_DEAL_DWORDS x BLOCK dwords of LDS that lie over the traversal stack (k_mega.h) */
        D.stackDepth = std::max<uint32_t>(D.stackDepth, MEGA_DEAL_DWORDS);
    }
    /* round 6: the fused kernel on a tree that does not fit LDS (k_wide_wave.h) -- every scene on the 8-wide tree whose emitter table fits LDS and that needs none of the
       feature sets k_mega is not compiled with (bitmap textures, an environment emitter) */
    sc->fusedWide = (sc->wideOnly && !sc->hasTextures && envEmitter < 0 && stride == TRISHADE_FLOAT4S && tab.size() <= EMITTER_LDS_FLOATS
                     && sc->bvh.wtris.size() / 12 < WP_TRI_MAX && !expEnv("PHIP_NO_MEGA_WIDE"))
                  ? (mats.size() <= MATERIAL_LDS_MAX ? 4 : 5) : 0;
    const bool traceable = D.flatMode >= 2 && tab.size() <= EMITTER_LDS_FLOATS && mats.size() <= MATERIAL_LDS_MAX && !expEnv("PHIP_NO_SHADE_TRACE");
    sc->flatTrace = !sc->fitsLds && traceable; sc->flatTraceToo = sc->fitsLds && traceable;
    /* ... whose lanes are dealt by BSDF model where there is more than one (the kernel traces its own rays and leaves the class in the hit word) */
    if ((sc->flatTrace || sc->flatTraceToo) && sc->materialMask != 0) { D.shadeSort = 1u; if (const char *e = expEnv("PHIP_SHADE_SORT")) D.shadeSort = atoi(e) != 0 ? 1u : 0u; }
    sd.counters.alloc(1);
    sd.invalid.alloc(1);
    sd.dynCounter.alloc(DYN_SHARDS * DYN_STRIDE);
    sd.megaNext.alloc(1);
    std::vector<float>().swap(sc->bvh.nodes); std::vector<float>().swap(sc->bvh.tris);      /* keep the statistics, drop the arrays */
    std::vector<uint32_t>().swap(sc->bvh.wnodes); std::vector<float>().swap(sc->bvh.wtris);
    HIP_TRY(hipHostMalloc((void **) &sc->cancelFlag, sizeof(int), hipHostMallocPortable | hipHostMallocMapped));
    *sc->cancelFlag = 0;
}


/* Byte tables of the Sobol' direction numbers (dv_math.h: SobolTab::matBt / vdcBt / vdcInvBt): entry [v] of byte b = the XOR of the rows 8 b + j over the
   set bits j of v, built as entry [v without its lowest set bit] ^ row 8 b + that bit.  `rows2` = the two enumeration rows (vdc, then vdc_inv), 52 words each. */
static void buildSobolByteTables(const uint32_t *matrices, size_t dims, const unsigned long long *rows2, std::vector<uint32_t> &bt, std::vector<unsigned long long> &vb) {
    bt.assign(dims * SOBOL_BT_BYTES * 256u, 0u);
    for (size_t d = 0; d < dims; ++d)
        for (uint32_t b = 0; b < SOBOL_BT_BYTES; ++b) {
            uint32_t *t = &bt[(d * SOBOL_BT_BYTES + b) * 256u];
            for (uint32_t x = 1; x < 256u; ++x) {
                const uint32_t j = 8u * b + (uint32_t) __builtin_ctz(x);
                /* (sampleSingle indexes matrices[i + dimension * 52] for every set bit i of the index: above bit 51 that is the next dimension's rows) */
                const size_t row = d * PHIP_SOBOL_MATRIX_SIZE + j;
                t[x] = t[x & (x - 1u)] ^ (row < dims * PHIP_SOBOL_MATRIX_SIZE ? matrices[row] : 0u);
            }
        }
    vb.assign((4u + 7u) * 256u, 0ull);
    for (uint32_t b = 0; b < 4u + 7u; ++b) {
        const unsigned long long *rows = b < 4u ? rows2 : rows2 + PHIP_SOBOL_MATRIX_SIZE;
        const uint32_t bb = b < 4u ? b : b - 4u;
        unsigned long long *t = &vb[(size_t) b * 256u];
        for (uint32_t x = 1; x < 256u; ++x) {
            const uint32_t j = 8u * bb + (uint32_t) __builtin_ctz(x);
            t[x] = t[x & (x - 1u)] ^ (j < (uint32_t) PHIP_SOBOL_MATRIX_SIZE ? rows[j] : 0ull);
        }
    }
}

/* Multi-digit tables of the radical inverses (dv_math.h: RinvTab::dimInfo / chunk / fac / pw) for the leading dimensions whose base is below 1024 (at most
   RINV_TAB_DIMS): the permuted value of every chunk of k digits, the float factors radical^m as the digit loop multiplies them up, the powers base^n */
static uint32_t buildRinvTables(const uint32_t *primes, const uint16_t *perm /* concatenated, or NULL */, const uint32_t *permOffset, uint32_t dims,
                                std::vector<uint32_t> &dimInfo, std::vector<uint32_t> &chunk, std::vector<float> &fac, std::vector<uint32_t> &pw) {
    uint32_t nd = 0;
    while (nd < dims && nd < RINV_TAB_DIMS && primes[nd] < 1024u) ++nd;
    dimInfo.assign((size_t) 8 * nd, 0u); chunk.clear(); fac.assign((size_t) RINV_FAC_STRIDE * nd, 0.0f); pw.assign((size_t) RINV_PW_STRIDE * nd, 0u);
    for (uint32_t d = 0; d < nd; ++d) {
        const uint32_t base = primes[d];
        const uint16_t *P = perm ? perm + permOffset[d] : nullptr;
        uint32_t k = 1, B = base;
        while ((unsigned long long) B * base <= 1024ull && k < RINV_PW_STRIDE - 1u) { B *= base; ++k; }
        const uint32_t first = (uint32_t) chunk.size();
        dimInfo[8 * d] = base; dimInfo[8 * d + 1] = B; dimInfo[8 * d + 2] = k | (first << 8);
        dimInfo[8 * d + 3] = (uint32_t) ((1ull << 32) / B) + 1u;
        for (uint32_t c = 0; c < B; ++c) {
            uint32_t vFull = 0, vSig = 0, n = 0, t = c;
            for (uint32_t j = 0; j < k; ++j) { const uint32_t digit = t % base; t /= base; vFull = vFull * base + (P ? (uint32_t) P[digit] : digit); }
            for (t = c; t; t /= base, ++n) { const uint32_t digit = t % base; vSig = vSig * base + (P ? (uint32_t) P[digit] : digit); }
            chunk.push_back(vFull | (vSig << 10) | (n << 20));
        }
        const float radical = 1.0f / (float) (int) base;
        const float tail = P ? radical * (float) (int) P[0] / (1.0f - radical) : 0.0f;        /* the constant of scrambledRadicalInverse, by its own expression */
        dimInfo[8 * d + 4] = pm_to_bits(radical); dimInfo[8 * d + 5] = pm_to_bits(tail);
        float f = 1.0f;
        for (uint32_t m = 0; m < RINV_FAC_STRIDE; ++m) { fac[(size_t) RINV_FAC_STRIDE * d + m] = f; f *= radical; }
        uint32_t pwr = 1;
        for (uint32_t n = 0; n <= k; ++n) { pw[(size_t) RINV_PW_STRIDE * d + n] = pwr; pwr *= base; }
    }
    return nd;
}

/* 64-bit content key of a caller's table (word-wise multiply-xorshift; ~2 GB/s: 0.1 ms for the Sobol direction numbers) */
static uint64_t contentHash(const void *data, size_t bytes, uint64_t h) {
    const unsigned char *b = (const unsigned char *) data;
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) { uint64_t w; memcpy(&w, b + i, 8); h = (h ^ w) * 0xFF51AFD7ED558CCDull; h ^= h >> 32; }
    uint64_t w = 0; if (i < bytes) memcpy(&w, b + i, bytes - i);
    h = (h ^ w ^ (uint64_t) bytes) * 0xC4CEB9FE1A85EC53ull; h ^= h >> 29;
    return h ? h : 1;
}

/* Replica of the scene on another GPU: device-to-device copies of the immutable arrays (xGMI), same DevScene. */
static SceneDev *replicateScene(phip_scene *sc, int device) {
    SceneDev &src = *sc->devs[0];
    std::unique_ptr<SceneDev> dst(new SceneDev());
    dst->device = device;
    HIP_TRY(hipSetDevice(device));
    dst->nodes.cloneFrom(src.nodes); dst->wnodes.cloneFrom(src.wnodes); dst->tris.cloneFrom(src.tris); dst->wtris.cloneFrom(src.wtris); dst->trisAreWide = src.trisAreWide; dst->triShade.cloneFrom(src.triShade); dst->flatLeaves.cloneFrom(src.flatLeaves);
    dst->materials.cloneFrom(src.materials); dst->emitterTab.cloneFrom(src.emitterTab);
    dst->texTexels.cloneFrom(src.texTexels); dst->texDesc.cloneFrom(src.texDesc);
    dst->envTexels.cloneFrom(src.envTexels); dst->envLevels.cloneFrom(src.envLevels);
    dst->envCdfRows.cloneFrom(src.envCdfRows); dst->envCdfCols.cloneFrom(src.envCdfCols); dst->envRowWeights.cloneFrom(src.envRowWeights);
    dst->dev = src.dev;
    dst->bind();
    dst->counters.alloc(1); dst->invalid.alloc(1); dst->dynCounter.alloc(DYN_SHARDS * DYN_STRIDE); dst->megaNext.alloc(1);
    HIP_TRY(hipDeviceSynchronize());
    sc->devs.emplace_back(std::move(dst));
    return sc->devs.back().get();
}

static void phipLaunchShade(int feat, bool strictNormals, int materialMask, dim3 grid, hipStream_t stream,
                            const DevScene &S, const PathPool &P, const RenderConst &rc, float4 *L) {
    switch (feat & 11) {
        case 0: phipLaunchShadeF0(strictNormals, materialMask, grid, stream, S, P, rc, L); break;
        case 1: phipLaunchShadeF1(strictNormals, materialMask, grid, stream, S, P, rc, L); break;
        case 2: phipLaunchShadeF2(strictNormals, materialMask, grid, stream, S, P, rc, L); break;
        case 8: phipLaunchShadeF8(strictNormals, materialMask, grid, stream, S, P, rc, L); break;
        case 11: phipLaunchShadeF11(strictNormals, materialMask, grid, stream, S, P, rc, L); break;
        default: phipLaunchShadeF3(strictNormals, materialMask, grid, stream, S, P, rc, L); break;
    }
}
static void phipLaunchShadeTrace(int feat, bool strictNormals, int materialMask, dim3 grid, size_t lds, hipStream_t stream,
                                 const DevScene &S, const PathPool &P, const RenderConst &rc, float4 *L) {
    switch (feat & 11) {
        case 0: phipLaunchShadeTraceF0(strictNormals, materialMask, grid, lds, stream, S, P, rc, L); break;
        case 1: phipLaunchShadeTraceF1(strictNormals, materialMask, grid, lds, stream, S, P, rc, L); break;
        case 2: phipLaunchShadeTraceF2(strictNormals, materialMask, grid, lds, stream, S, P, rc, L); break;
        case 8: phipLaunchShadeTraceF8(strictNormals, materialMask, grid, lds, stream, S, P, rc, L); break;
        case 11: phipLaunchShadeTraceF11(strictNormals, materialMask, grid, lds, stream, S, P, rc, L); break;
        default: phipLaunchShadeTraceF3(strictNormals, materialMask, grid, lds, stream, S, P, rc, L); break;
    }
}
static void phipLaunchShadeDirect(int feat, int materialMask, dim3 grid, hipStream_t stream,
                                  const DevScene &S, const PathPool &P, const RenderConst &rc, float4 *L) {
    switch (feat & 11) {
        case 8: phipLaunchShadeDirectF8(materialMask, grid, stream, S, P, rc, L); break;
        case 11: phipLaunchShadeDirectF11(materialMask, grid, stream, S, P, rc, L); break;
        case 0: phipLaunchShadeDirectF0(materialMask, grid, stream, S, P, rc, L); break;
        case 1: phipLaunchShadeDirectF1(materialMask, grid, stream, S, P, rc, L); break;
        case 2: phipLaunchShadeDirectF2(materialMask, grid, stream, S, P, rc, L); break;
        default: phipLaunchShadeDirectF3(materialMask, grid, stream, S, P, rc, L); break;
    }
}

static size_t traversalLdsBytes(const DevScene &D) {               /* kernels launched with blocks of BLOCK threads */
    if (D.wideNodeCache) return wideLdsBytes(wideRaycastCache(D.wideNodeCache), BLOCK);
    return traversalLdsBytesOf(D);
}

static void algorithmicBytes(bool mergedRays, phip_stats &st, double filmPixels, bool wide = false) {
    /* SURVEY 8(d) with this structure's sizes: 128-byte BVH4 node visits, 48-byte triangle records
       (no separate index array: records are stored in leaf order) */
    const double film = 20.0 * filmPixels;
    const double nodeBytes = wide ? 80.0 : 128.0;
    st.algorithmic_bytes = nodeBytes * (double) (st.closest_node_visits + st.shadow_node_visits) +
           48.0 * (double) (st.closest_triangle_tests + st.shadow_triangle_tests) +
           (64.0 + 40.0 + 108.0) * (double) st.closest_rays + (64.0 + 4.0) * (double) st.shadow_rays +
           104.0 * (double) st.path_vertices + film;
    /* closest-hit kernel: node + triangle fetches, ray read (32 B), hit record write (16 B) ... counted with
       the SURVEY's read+write convention: ray 64 B, hit 40 B */
    st.trace_kernel_bytes = nodeBytes * (double) st.closest_node_visits + 48.0 * (double) st.closest_triangle_tests +
           (64.0 + 40.0) * (double) st.closest_rays;
    if (mergedRays)    /* k_rays_p also casts the shadow rays: their node + record fetches, entry read, 4-byte result */
        st.trace_kernel_bytes += nodeBytes * (double) st.shadow_node_visits + 48.0 * (double) st.shadow_triangle_tests +
               (64.0 + 4.0) * (double) st.shadow_rays;
}

/* hipEvents of one render call; destroyed whatever happens (an exception leaves through HIP_TRY) */
struct EventList {
    std::vector<hipEvent_t> ev;
    hipEvent_t record(hipStream_t s) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); ev.push_back(e); HIP_TRY(hipEventRecord(e, s)); return e; }
    double sumPairs() const { double ms = 0; for (size_t i = 0; i + 1 < ev.size(); i += 2) { float t = 0; (void) hipEventElapsedTime(&t, ev[i], ev[i + 1]); ms += t; } return ms; }
    ~EventList() { for (auto e : ev) (void) hipEventDestroy(e); }
};

static bool cancelRequested(const phip_scene *sc) { return __atomic_load_n(sc->cancelFlag, __ATOMIC_RELAXED) != 0; }

static void validateParams(const phip_scene *sc, const phip_render_params *p) {
    if (p->spp <= 0) throw std::invalid_argument("spp must be > 0");
    if (p->integrator > PHIP_INTEGRATOR_VOLPATH_SIMPLE) throw std::invalid_argument("unknown integrator kind");
    const bool direct = p->integrator == PHIP_INTEGRATOR_DIRECT;
    if (!direct) {
        if (p->rr_depth <= 0) throw std::invalid_argument("'rrDepth' must be set to a value greater than zero!");                       /* integrator.cpp:219-220 */
        if (p->max_depth <= 0 && p->max_depth != -1) throw std::invalid_argument("'maxDepth' must be set to -1 (infinite) or a value greater than zero!"); /* :222-223 */
    } else {
        if (p->emitter_samples < 0 || p->bsdf_samples < 0) throw std::invalid_argument("direct: emitterSamples and bsdfSamples must not be negative");
        if (p->emitter_samples + p->bsdf_samples <= 0) throw std::invalid_argument("direct: emitterSamples + bsdfSamples must be > 0");     /* Assert, direct.cpp:107 */
        if (p->emitter_samples + p->bsdf_samples >= (int) DEPTH_MASK) throw std::invalid_argument("direct: at most 65534 shading samples per camera sample");
    }
    if (p->sampler > PHIP_SAMPLER_HAMMERSLEY) throw std::invalid_argument("unknown sampler kind");
    if (p->sampler == PHIP_SAMPLER_HALTON || p->sampler == PHIP_SAMPLER_HAMMERSLEY) {
        const char *name = p->sampler == PHIP_SAMPLER_HALTON ? "PHIP_SAMPLER_HALTON" : "PHIP_SAMPLER_HAMMERSLEY";
        /* `direct`: more than one sample of a kind is a requested sample array -- hammersley has none (hammersley.cpp:293-300: Log(EError)) */
        if (direct && p->sampler == PHIP_SAMPLER_HAMMERSLEY && (p->emitter_samples > 1 || p->bsdf_samples > 1))
            throw std::invalid_argument("PHIP_SAMPLER_HAMMERSLEY: request2DArray(): Not supported for the Hammersley QMC sequence! With `direct`, emitterSamples and bsdfSamples must be at most 1");
        if (!p->qmc_primes || p->qmc_dimensions < 8 || p->qmc_dimensions > 1024) throw std::invalid_argument(std::string(name) + ": qmc_primes / qmc_dimensions (the reference's prime table, 8 .. 1024 entries) are required");
        if (p->qmc_primes[0] != 2 || p->qmc_primes[1] != 3) throw std::invalid_argument(std::string(name) + ": qmc_primes must start 2, 3 (the pixel enumeration is over those bases)");
        for (uint32_t d = 0; d < p->qmc_dimensions; ++d) if (p->qmc_primes[d] < 2 || p->qmc_primes[d] > 65535u) throw std::invalid_argument(std::string(name) + ": qmc_primes out of range");
        if (p->qmc_permutations) {
            if (p->qmc_permutations[0] > 1 || p->qmc_permutations[1] > 1 || p->qmc_permutations[0] == p->qmc_permutations[1]) throw std::invalid_argument(std::string(name) + ": qmc_permutations does not start with a permutation of {0, 1}");
            if (p->qmc_permutations[2] > 2 || p->qmc_permutations[3] > 2 || p->qmc_permutations[4] > 2) throw std::invalid_argument(std::string(name) + ": the permutation of base 3 is not one of {0, 1, 2}");
        }
        if (!direct && p->rr_depth < 2) throw std::invalid_argument(std::string(name) + ": rrDepth must be at least 2 (the dimension bookkeeping of halton.cpp:364-366 is restated for that case)");
        /* (a crop window anywhere on the film: the image blocks, and with them the pixel positions the sampler's generate() sees, are relative to the crop
           window -- renderproc.cpp:163-164 starts them at (0, 0) -- so the crop offset never reaches the sequence: nothing to restate, nothing to refuse) */
        const unsigned long long n = (unsigned long long) (p->sample_total > 0 ? p->sample_total : p->spp);
        if (n >= (1ull << 17)) throw std::invalid_argument(std::string(name) + ": at most 131071 samples per pixel (32-bit pixel offsets)");
    }
    if (p->sampler == PHIP_SAMPLER_SOBOL || p->sampler == PHIP_SAMPLER_STRATIFIED) {
        const char *name = p->sampler == PHIP_SAMPLER_SOBOL ? "PHIP_SAMPLER_SOBOL" : "PHIP_SAMPLER_STRATIFIED";
        const DevScene &D0 = sc->devs[0]->dev;
        if (p->sampler == PHIP_SAMPLER_SOBOL) {
            if (!p->sobol_matrices || p->sobol_dimensions < 8) throw std::invalid_argument("PHIP_SAMPLER_SOBOL: sobol_matrices / sobol_dimensions (the reference plugin's direction numbers) are required");
            if (p->sobol_log_resolution > 26) throw std::invalid_argument("PHIP_SAMPLER_SOBOL: sobol_log_resolution out of range");
            if (!direct && p->rr_depth < 2) throw std::invalid_argument("PHIP_SAMPLER_SOBOL: rrDepth must be at least 2 (the dimension bookkeeping of sobol.cpp:241-242 is restated for that case)");
            if (direct && (p->emitter_samples > 1 || p->bsdf_samples > 1) && p->sobol_log_resolution < 2)
                throw std::invalid_argument("PHIP_SAMPLER_SOBOL: sample arrays of `direct` need a film of more than two pixels per side (SobolSampler::generate enumerates them per pixel: sobol.cpp:182,190)");
            if (p->sobol_log_resolution > 1 && (!p->sobol_vdc || !p->sobol_vdc_inv)) throw std::invalid_argument("PHIP_SAMPLER_SOBOL: sobol_vdc / sobol_vdc_inv are required when the film is enumerated per pixel");
            uint32_t need = 0; { uint32_t side = (uint32_t) std::max(D0.film.width, D0.film.height), r = 1; while (r < side) { r <<= 1; ++need; } }
            if (p->sobol_log_resolution != need) throw std::invalid_argument("PHIP_SAMPLER_SOBOL: sobol_log_resolution must be log2 of the crop window's larger side rounded up to a power of two (sobol.cpp:147-157)");
        } else {
            const unsigned n = (unsigned) (p->sample_total > 0 ? p->sample_total : p->spp);
            unsigned r = 1; while (r * r < n) ++r;
            if (r * r != n) throw std::invalid_argument("PHIP_SAMPLER_STRATIFIED: the sample count of the render must be a perfect square (stratified.cpp:64-72 rounds it up)");
        }
    }
    if (p->sampler == PHIP_SAMPLER_LD) {
        const unsigned n = (unsigned) (p->sample_total > 0 ? p->sample_total : p->spp);
        if (n == 0 || (n & (n - 1))) throw std::invalid_argument("PHIP_SAMPLER_LD: the sample count of the render must be a power of two (ldsampler.cpp:83-87)");
        if (p->integrator == PHIP_INTEGRATOR_DIRECT && (unsigned long long) n * (unsigned) std::max(p->emitter_samples, p->bsdf_samples) > 0x7fffffffull)
            throw std::invalid_argument("PHIP_SAMPLER_LD: sample array too long");
    }
    if (p->sample_offset < 0 || p->sample_total < 0) throw std::invalid_argument("sample_offset / sample_total must not be negative");
    if (p->sample_total != 0 && (long long) p->sample_offset + p->spp > p->sample_total) throw std::invalid_argument("sample_offset + spp exceeds sample_total");
    if ((p->flags & PHIP_FLAG_SAMPLE_BUFFER) && (p->flags & PHIP_FLAG_ACCUMULATE))
        throw std::invalid_argument("PHIP_FLAG_SAMPLE_BUFFER holds the samples of one call: not with PHIP_FLAG_ACCUMULATE");
    if (sc->devs[0]->dev.env.w > 0 && sc->envLevelCount <= 1 && !p->hide_emitters && !(p->flags & PHIP_FLAG_ENVMAP_BILINEAR_BACKGROUND))
        throw std::invalid_argument("envmap without MIP levels: directly visible background needs the filtered (EWA) lookup of envmap.cpp:395-407: "
                                    "pass the pyramid, render with hideEmitters or set PHIP_FLAG_ENVMAP_BILINEAR_BACKGROUND");
    const int bs = p->block_size > 0 ? p->block_size : 32;
    if (bs < 2 || bs > 128 || (bs & (bs - 1))) throw std::invalid_argument("block_size must be a power of two in [2,128] (mitsuba.cpp:233-239 allows 2..128)");
    if (bs < sc->devs[0]->dev.film.border) throw std::invalid_argument("The block size must be larger than the image reconstruction filter radius!"); /* renderproc.cpp:175-176 */
    const int shardCount = p->shard_count > 0 ? p->shard_count : 1;
    if (p->shard_index < 0 || p->shard_index >= shardCount) throw std::invalid_argument("shard_index out of range");
    if (p->n_devices < 0 || p->n_devices > PHIP_MAX_DEVICES) throw std::invalid_argument("n_devices out of range");
    if (p->n_devices > 1 && p->stream) throw std::invalid_argument("a caller's stream belongs to one device: leave `stream` NULL when n_devices > 1 (every device renders on a stream of the library)");
}

/* One device's share of a render call: the blocks whose index in the reference's spiral order is congruent to shardIndex
   modulo shardCount, into dOut (device memory of sd.device).  The caller holds the scene's render lock and has validated p. */
static int renderOnDevice(phip_scene *sc, SceneDev &sd, const phip_render_params *p, int shardIndex, int shardCount, float *dOut, phip_stats *stats) {
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    const bool direct = p->integrator == PHIP_INTEGRATOR_DIRECT;
    const int bs = p->block_size > 0 ? p->block_size : 32;
    HIP_TRY(hipSetDevice(sd.device));

    DevScene D = sd.dev;
    D.film.blockSize = bs;
    const int W = D.film.width, H = D.film.height;
    int tileShift = 0; while ((1 << tileShift) < bs) ++tileShift;

    /* tile -> shard assignment in the reference's spiral order (cached between calls with the same layout) */
    const int tilesX = (W + bs - 1) / bs, tilesY = (H + bs - 1) / bs;
    if (sd.tileKey[0] != bs || sd.tileKey[1] != shardIndex || sd.tileKey[2] != shardCount) {
        std::vector<std::pair<int, int>> spiral;
        spiralBlocks(W, H, bs, spiral);
        std::vector<int32_t> tileSlot((size_t) tilesX * tilesY, -1);
        std::vector<uint32_t> tileOrigin;
        for (size_t i = 0; i < spiral.size(); ++i) {
            if ((int) (i % (size_t) shardCount) != shardIndex) continue;
            tileSlot[(size_t) spiral[i].second * tilesX + spiral[i].first] = (int32_t) tileOrigin.size();
            tileOrigin.push_back((uint32_t) (spiral[i].first * bs) | ((uint32_t) (spiral[i].second * bs) << 16));
        }
        sd.nLocalTiles = (uint32_t) tileOrigin.size();
        sd.localPixels = 0;
        for (uint32_t o : tileOrigin) sd.localPixels += (unsigned long long) std::min(bs, W - (int) (o & 0xFFFFu)) * std::min(bs, H - (int) (o >> 16));
        if (tileOrigin.empty()) sd.tileOrigin.alloc(1);
        else sd.tileOrigin.upload(tileOrigin.data(), tileOrigin.size());
        sd.tileSlot.upload(tileSlot.data(), tileSlot.size());
        sd.tileKey[0] = bs; sd.tileKey[1] = shardIndex; sd.tileKey[2] = shardCount;
    }
    const uint32_t nLocalTiles = sd.nLocalTiles;

    hipStream_t stream = (p->n_devices <= 1) ? (hipStream_t) p->stream : nullptr;    /* a caller's stream belongs to one device */
    if (!stream) { if (!sd.stream) HIP_TRY(hipStreamCreate(&sd.stream)); stream = sd.stream; }

    /* passes: bound the per-sample buffer (16 B per sample id; 24 B with the jitter the sequence samplers keep for the film pass) */
    const bool keepJitter = (p->sampler == PHIP_SAMPLER_SOBOL || p->sampler == PHIP_SAMPLER_HALTON || p->sampler == PHIP_SAMPLER_HAMMERSLEY) && !expEnv("PHIP_NO_JITTER_BUFFER");
    const unsigned long long tilePixels = (unsigned long long) bs * bs;
    const unsigned long long maxIdsPerPass = (1ull << 32) - 1;                 /* sample ids are 32-bit in the slot state */
    unsigned long long budgetIds = (24ull << 30) / (keepJitter ? 24 : 16);    /* 24 GiB of sample buffer */
    if (const char *e = getenv("PHIP_MAX_PASS_SAMPLES")) budgetIds = std::max(1ull, strtoull(e, nullptr, 10));   /* test hook: force several passes */
    unsigned long long idsPerSpp = (unsigned long long) nLocalTiles * tilePixels;
    uint32_t sppPerPass = (uint32_t) p->spp;
    if (idsPerSpp > 0) {
        unsigned long long cap = std::min(maxIdsPerPass, budgetIds) / idsPerSpp;
        if (cap < 1) cap = 1;
        sppPerPass = (uint32_t) std::min<unsigned long long>(cap, (unsigned long long) p->spp);
    }
    const bool keepSamples = (p->flags & PHIP_FLAG_SAMPLE_BUFFER) != 0;
    if (keepSamples) { sd.sampleOut.alloc((size_t) W * H * (size_t) p->spp); HIP_TRY(hipMemsetAsync(sd.sampleOut.p, 0, sd.sampleOut.n * sizeof(float4), stream)); }
    sd.haveSamples = keepSamples; sd.lastSpp = (uint32_t) p->spp;

    const unsigned long long idsFirstPass = idsPerSpp * sppPerPass;
    if (sd.L.n < idsFirstPass) sd.L.alloc((size_t) idsFirstPass);
    if (keepJitter && sd.jitter.n < idsFirstPass) sd.jitter.alloc((size_t) idsFirstPass);

    /* which device path: the fused kernel when the scene fits its LDS plan (decided at scene creation) */
    const bool rinv = p->sampler == PHIP_SAMPLER_HALTON || p->sampler == PHIP_SAMPLER_HAMMERSLEY;
    const bool qmc = p->sampler == PHIP_SAMPLER_SOBOL || p->sampler == PHIP_SAMPLER_STRATIFIED || rinv;     /* served by the wavefront kernels compiled with FEAT bit 3 */
    /* (the fused kernel on the 8-wide tree: by default where it is faster than the wavefront kernels -- trees that live in L2, DESIGN.md 3.9 -- with PHIP_FLAG_FUSED_ANY wherever it can run) */
    const bool wideFused = sc->fusedWide && (sc->bvh.nWNodes <= PHIP_FUSED_WIDE_MAX_NODES || (p->flags & PHIP_FLAG_FUSED_ANY));
    bool fused = (sc->fitsLds || wideFused) && sc->traversal == 2 && !(p->flags & (PHIP_FLAG_NO_FUSED | PHIP_FLAG_NO_MEGA));
    const int megaFlat = sc->fitsLds ? (D.nFlatLeaves ? (int) D.flatMode : 0) : sc->fusedWide;      /* k_mega's traversal form (k_mega.h) */
    if (direct && megaFlat < 2) fused = false;      /* (round 6: `direct` rides the fused kernel too -- k_mega<.., DIRECT>, on the packed leaf tables and on the tree in memory) */
    const bool megaWide = megaFlat >= 4;
    if (qmc && p->sampler == PHIP_SAMPLER_SOBOL) {
        /* the plugin's tables, uploaded once per (pointer, size): ~210 KB of direction numbers + the two 52-word enumeration rows */
        const size_t nm = (size_t) p->sobol_dimensions * PHIP_SOBOL_MATRIX_SIZE;
        /* (keyed by CONTENT: the tables are host pointers "read during the call" -- a caller may refill or reallocate them at the same address) */
        uint64_t key = contentHash(p->sobol_matrices, nm * sizeof(uint32_t), 0x9E3779B97F4A7C15ull);
        if (p->sobol_log_resolution > 1) { key = contentHash(p->sobol_vdc, PHIP_SOBOL_MATRIX_SIZE * sizeof(uint64_t), key); key = contentHash(p->sobol_vdc_inv, PHIP_SOBOL_MATRIX_SIZE * sizeof(uint64_t), key); }
        if (sd.sobolKey != key || sd.sobolMat.n != nm || sd.sobolLogRes != p->sobol_log_resolution) {
            sd.sobolMat.upload(p->sobol_matrices, nm);
            std::vector<unsigned long long> v(2 * PHIP_SOBOL_MATRIX_SIZE, 0ull);
            if (p->sobol_log_resolution > 1)
                for (int i = 0; i < PHIP_SOBOL_MATRIX_SIZE; ++i) { v[i] = p->sobol_vdc[i]; v[PHIP_SOBOL_MATRIX_SIZE + i] = p->sobol_vdc_inv[i]; }
            sd.sobolVdc.upload(v.data(), v.size());
            {
                std::vector<uint32_t> bt; std::vector<unsigned long long> vb;
                buildSobolByteTables(p->sobol_matrices, (size_t) p->sobol_dimensions, v.data(), bt, vb);
                sd.sobolBt.upload(bt.data(), bt.size()); sd.sobolVdcBt.upload(vb.data(), vb.size());
            }
            sd.sobolKey = key; sd.sobolLogRes = p->sobol_log_resolution;
        }
    }
    uint64_t rinvKey = 0;
    if (rinv) {                                        /* content key of primes + permutations (at most ~200 KB; see the Sobol tables above) */
        size_t total = 0;
        for (uint32_t d = 0; d < p->qmc_dimensions; ++d) total += p->qmc_primes[d];
        rinvKey = contentHash(p->qmc_primes, p->qmc_dimensions * sizeof(uint32_t), p->qmc_permutations ? 0x51ED270B7A5F1D3Bull : 0x2545F4914F6CDD1Dull);
        if (p->qmc_permutations) rinvKey = contentHash(p->qmc_permutations, total * sizeof(uint16_t), rinvKey);
    }
    if (rinv && (sd.rinvKey != rinvKey || sd.rinvPrimes.n != p->qmc_dimensions)) {
        /* primes, the start of every base's permutation, the permutations themselves; the inverse permutations of bases 2 and 3 (the pixel enumeration) */
        std::vector<uint32_t> off(p->qmc_dimensions);
        size_t total = 0;
        for (uint32_t d = 0; d < p->qmc_dimensions; ++d) { off[d] = (uint32_t) total; total += p->qmc_primes[d]; }
        sd.rinvPrimes.upload(p->qmc_primes, p->qmc_dimensions); sd.rinvOffsets.upload(off.data(), off.size());
        sd.rinvInvPerm2 = 0x4u; sd.rinvInvPerm3 = 0x24u;
        if (p->qmc_permutations) {
            sd.rinvPerm.upload(p->qmc_permutations, total);
            const uint16_t *p2 = p->qmc_permutations, *p3 = p->qmc_permutations + 2;
            sd.rinvInvPerm2 = sd.rinvInvPerm3 = 0;
            for (uint32_t i = 0; i < 2; ++i) sd.rinvInvPerm2 |= i << (2u * p2[i]);          /* invPerm[perm[i]] = i (faure.cpp: invertPermutation) */
            for (uint32_t i = 0; i < 3; ++i) sd.rinvInvPerm3 |= i << (2u * p3[i]);
        }
        {
            std::vector<uint32_t> di, ch, pw; std::vector<float> fc;
            sd.rinvTabDims = buildRinvTables(p->qmc_primes, p->qmc_permutations, off.data(), p->qmc_dimensions, di, ch, fc, pw);
            if (sd.rinvTabDims) { sd.rinvDimInfo.upload(di.data(), di.size()); sd.rinvChunk.upload(ch.data(), ch.size()); sd.rinvFac.upload(fc.data(), fc.size()); sd.rinvPw.upload(pw.data(), pw.size()); }
        }
        sd.rinvKey = rinvKey;
    }
    if (const char *e = expEnv("PHIP_MEGA")) fused = fused && atoi(e) != 0;            /* experiment hook: PHIP_MEGA=0 forces the wavefront kernels */
    /* resident blocks of the fused kernel for THIS render (the QMC build has ~15 KB more static LDS than the plan of fitsLds priced at scene
       creation): when none fits a compute unit the render runs on the wavefront kernels, as it did before the samplers moved to k_mega */
    int megaPerCU = 0;
    uint32_t megaNodeCache = 0; size_t megaLds = 0;
    if (fused) {
        if (megaWide) {
            /* the top of the tree every block stages (BFS order); the mailbox build (all leaf BSDF models, counter stream) has the S-box behind the round buffers */
            megaNodeCache = std::min<uint32_t>(sc->bvh.nWNodes, MEGA_WIDE_NODE_CACHE);
            if (const char *e = expEnv("PHIP_MEGA_NODE_CACHE")) megaNodeCache = std::min<uint32_t>(sc->bvh.nWNodes, (uint32_t) atoi(e));
            const bool mailbox = false;     /* (the tree-in-memory builds deal their paths by BSDF model; the mailboxes' LDS would cost the fourth block of a CU: k_mega.h) */
            megaLds = MEGA_POOL ? megaWidePoolLdsBytesOf(D, megaNodeCache, mailbox, megaFlat == 4, MB_DW * MB_NS * sizeof(uint32_t))
                                : megaWideLdsBytesOf(D, megaNodeCache, mailbox, megaFlat == 4, MB_DW * MB_NS * sizeof(uint32_t));
            megaPerCU = std::min(MEGA_WAVES, direct ? phipMegaBlocksPerCUDirect(sc->materialMask, false, megaFlat, qmc, megaLds)
                                                    : phipMegaBlocksPerCUWide(sc->materialMask, p->strict_normals != 0, megaFlat, qmc, megaLds));
        } else {
            megaLds = megaLdsBytesOf(D);
            megaPerCU = std::min(MEGA_WAVES, direct ? phipMegaBlocksPerCUDirect(sc->materialMask, false, megaFlat, qmc, megaLds)
                                                    : phipMegaBlocksPerCU(sc->materialMask, p->strict_normals != 0, megaFlat, qmc, megaLds));
        }
        if (const char *e = expEnv("PHIP_MEGA_BLOCKS")) megaPerCU = std::max(1, std::min(megaPerCU, atoi(e)));
        if (megaPerCU <= 0) fused = false;
        if (getenv("PHIP_DEBUG_TIMING")) fprintf(stderr, "[phip] k_mega (traversal form %d): %d blocks per CU with %zu bytes of dynamic LDS\n", megaFlat, megaPerCU, megaLds);
    }
    sd.fused = fused;
    /* ... or k_shade_trace: the scene's tree is the packed leaf table, but k_mega does not serve it (glass / copper / textures / environment emitter) */
    const bool shadeTrace = !fused && (sc->flatTrace || sc->flatTraceToo) && !direct && sc->traversal == 2 && !(p->flags & PHIP_FLAG_NO_FUSED);

    phip_stats st; memset(&st, 0, sizeof(st));
    const bool timing = (p->flags & PHIP_FLAG_KERNEL_TIMING) != 0;
    EventList evTrace, evShadow, evShade, evFilm, evFused;
    const unsigned long long samplesTotal = sd.localPixels;                  /* crop pixels of this device's blocks */

    HIP_TRY(hipMemsetAsync(sd.invalid.p, 0, sizeof(unsigned long long), stream));
    int nCU = 256; { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, sd.device) == hipSuccess) nCU = prop.multiProcessorCount; }
    const size_t ldsBytes = traversalLdsBytes(D);
    const dim3 block(BLOCK);
    Counters hc;
    bool cancelled = false;
    bool accumulate = (p->flags & PHIP_FLAG_ACCUMULATE) != 0;
    unsigned long long samplesDone = 0;

    /* wavefront state (allocated only when that path runs) */
    uint32_t capacity = 0, nWaves = 0, nBlocks = 0;
    PathPool P; memset(&P, 0, sizeof(P));
    bool merged = false;
    if (!fused) {
        /* pool size: large enough that per-launch fixed costs vanish, small enough that the tail (slots
           running dry at the end of a pass) stays a small fraction of the pass (measured: 4M / 8M slots) */
        /* (round 3) Every launch of the persistent ray kernel ends with a drain in which each wave waits for its longest ray while the
           work queue is empty: ~0.15 ms per launch on the 250 k-triangle scenes, 14 % of a launch over 4 M slots, 7 % over 8 M.  A bigger
           pool means fewer, longer launches: atrium 1920x1080x64 spp 427 / 454 / 470 / 470 Msamples/s with 4 / 8 / 16 / 32 M slots, the
           glass room at 512 spp 480 / 496 / 505 with 8 / 16 / 32 M -- so the pool grows with the job (about 0.14 KB of HBM per slot). */
        /* (round 4, after the ray kernel's triangle rounds: 8 / 16 / 32 / 64 M slots on the 4K slice (531 M ids) 610 / 636 / 646 / 657 Msamples/s, C4 at 512 spp (1062 M ids)
           flat at 64 M and -2 % at 128 M, C3 (133 M ids) flat from 16 M on: jobs of more than 256 M ids get 64 M slots) */
        /* (round 5: k_shade_trace has no persistent ray kernel whose drain a big pool amortises -- what a big pool costs it is the tail of the pass, the launches
           in which the long paths of a few slots finish: mixed Cornell box at 256 spp with 1 / 2 / 4 / 8 / 16 M slots 1518 / 1617 / 1643 / 1596 / 1450 Msamples/s,
           profiles/r05_gpu_call_e_*) */
        const unsigned long long poolCap = shadeTrace ? (1ull << 22)
                                         : idsFirstPass > (256ull << 20) ? (1ull << 26) : idsFirstPass >= (64ull << 20) ? (1ull << 24)
                                         : idsFirstPass >= (16ull << 20) ? (1ull << 23) : (1ull << 22);
        capacity = (uint32_t) std::min<unsigned long long>(std::max<unsigned long long>(idsFirstPass, BLOCK), poolCap);
        /* traversal-stack overflow: SPILL_DEPTH words per LANE of a ray kernel -- the wide tree is only walked by persistent grids
           (at most 8 resident blocks of 256 per CU), the BVH4 also by one-lane-per-slot launches */
        const bool canSpill = sc->wide ? (int) sc->bvh.wMaxDepth + 2 > WIDE_STACK_LDS : 3 * ((int) sc->bvh.maxDepth - 1) + 1 > (int) D.stackDepth;
        const bool persistentOnly = sc->wide || (sc->traversal == 2 && sc->bvh.nNodes >= 64 && !expEnv("PHIP_MERGED"));
        {   /* ... and with the memory that is there: the pool's state is ~144 B per slot (+ 384 B of spill stack where every slot is a lane that can
               spill); it may take a quarter of what is free now (a shared or partitioned GPU, n_devices replicas), never less than the 4 M slots
               every job ran with before the pool grew */
            size_t freeB = 0, totalB = 0;
            if (sd.rayO.n < capacity && hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
                unsigned long long fit = (unsigned long long) (freeB / 4) / (144ull + ((!persistentOnly && canSpill) ? (unsigned long long) SPILL_DEPTH * 4ull : 0ull));
                while (capacity > (1u << 22) && capacity > fit) capacity >>= 1;
            }
        }
        capacity = (capacity + BLOCK - 1) / BLOCK * BLOCK;
        if (const char *e = expEnv("PHIP_POOL")) { capacity = (uint32_t) std::max(BLOCK, atoi(e)) / BLOCK * BLOCK; }
        const size_t laneCap = ((size_t) capacity + WIDE_BLOCK - 1) / WIDE_BLOCK * WIDE_BLOCK;      /* k_rays_w runs whole blocks of WIDE_BLOCK lanes */
        nWaves = (uint32_t) (laneCap / 64); nBlocks = capacity / BLOCK;
        if (sd.rayO.n < capacity) {
            sd.rayO.alloc(capacity); sd.rayD.alloc(capacity); sd.hit.alloc(capacity); sd.thr.alloc(capacity);
            sd.mis.alloc(capacity); sd.info.alloc(capacity); sd.state.alloc(capacity); sd.shadow.alloc(3 * (size_t) capacity);
            sd.shadowCount.alloc(nBlocks); sd.blockDead.alloc(nBlocks); sd.blockShard.alloc(nBlocks);
        }
        {   /* persistent grids (at most 8 resident blocks of 256 per CU): the buffer always covers the whole grid, also when the host's depth bound says
               that no lane can reach it -- 200 MB of 288 GB against a silent out-of-bounds write should that hand-derived bound ever be off.  One lane
               per slot (BVH4 trees of < 64 nodes: small scenes with non-diffuse materials, every `direct` render of a small scene): 384 B per SLOT of a
               pool of up to 64 M slots is 3..25 GB for a stack that is at most ~20 entries deep on such a tree -- there the bound decides, as before
               round 4; when it says "can spill" the full size is allocated AND priced in the free-memory guard above */
            const size_t spillLanes = persistentOnly ? std::min<size_t>(laneCap, (size_t) nCU * 8 * 256) : (canSpill ? laneCap : (size_t) WIDE_BLOCK);
            if (sd.spill.n < spillLanes * SPILL_DEPTH) sd.spill.alloc(spillLanes * SPILL_DEPTH);
            P.spillLanes = (uint32_t) std::min<size_t>(sd.spill.n / SPILL_DEPTH, 0xFFFFFFFFu);
        }
        if (sd.stat.n < (size_t) ST_COUNT * nWaves) sd.stat.alloc((size_t) ST_COUNT * nWaves);
        if (direct && sd.camHit.n < capacity) sd.camHit.alloc(capacity);
        P.camHit = sd.camHit.p;
        P.rayO = sd.rayO.p; P.rayD = sd.rayD.p; P.hit = sd.hit.p; P.thr = sd.thr.p; P.mis = sd.mis.p; P.info = sd.info.p; P.state = sd.state.p;
        P.shadow = sd.shadow.p; P.shadowCount = sd.shadowCount.p; P.blockDead = sd.blockDead.p; P.stat = sd.stat.p; P.spill = sd.spill.p; P.capacity = capacity; P.nWaves = nWaves;
        /* big trees: closest-hit and any-hit rays share one persistent launch (measured +2..4 % on the 250k-triangle scenes;
           on small trees the plain per-slot closest-hit launch wins, so the kernels stay separate there) */
        merged = sc->traversal == 2 && sc->bvh.nNodes >= 64;
        if (const char *e = expEnv("PHIP_MERGED")) merged = sc->traversal == 2 && atoi(e) != 0;
        if (sc->wide) merged = true;                             /* the wide tree has the merged kernel only */
    }
    sd.mergedRays = merged;
    /* persistent kernels: exactly the resident set (TRACE_WAVES waves per SIMD = TRACE_WAVES blocks of 256 per CU)
       ... but never more blocks per CU than their LDS (stack + node/record cache) allows: a persistent grid larger than
       the resident set would serialise */
    /* The grid of a persistent kernel is exactly its resident set.  Residency is asked of the runtime (registers, the block's static
       + dynamic LDS and the LDS allocation granule all enter) -- an arithmetic estimate that is one block per CU too high makes
       the surplus blocks wait for a resident one to finish: a second round that doubled the ray kernel's time when the LDS
       node cache grew to 7.5 KB (round 2). */
    auto residentBlocks = [&](const void *kernel, int wanted) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, BLOCK, ldsBytes) != hipSuccess || n <= 0) n = 1;
        return std::min(n, wanted);
    };
    [[maybe_unused]] auto persistentGrid = [&](const void *kernel, int blocksPerCU) {
        return dim3((unsigned) std::max(1, std::min<int>(nCU * residentBlocks(kernel, blocksPerCU), (int) ((capacity + BLOCK - 1) / BLOCK))));
    };
    const dim3 grid((capacity + BLOCK - 1) / BLOCK);
#if PHIP_EXPERIMENTS
    const dim3 pgrid = persistentGrid((const void *) k_shadow_p, TRACE_WAVES);
    const dim3 pgridTrace = persistentGrid(sc->bvh.nNodes >= 64 ? (const void *) k_trace_p<false> : (const void *) k_trace_p<true>, TRACE_P_WAVES);
#endif
    /* k_rays_w: blocks of WIDE_BLOCK threads with their own LDS plan (one block per CU holds 800 nodes of the tree) */
    const size_t wideLds = sc->wide ? wideLdsBytes(D.wideNodeCache, WIDE_BLOCK) + wideDealBytes(WIDE_BLOCK) : 0;
#if PHIP_EXPERIMENTS
    dim3 pgridRays = persistentGrid((const void *) k_rays_p, RAYS_WAVES);       /* (PHIP_WIDE=0: the BVH4 ray kernels of rounds 1-2 on the big scenes) */
#else
    dim3 pgridRays(1);
#endif
    if (sc->wide) {
        if (wideLds > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void *) k_rays_w, hipFuncAttributeMaxDynamicSharedMemorySize, (int) wideLds));
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *) k_rays_w, WIDE_BLOCK, wideLds) != hipSuccess || n <= 0) n = 1;
        n = std::min(n, WIDE_WAVES * 256 / WIDE_BLOCK);
        if (n <= 0) n = 1;
        pgridRays = dim3((unsigned) std::max(1, std::min<int>(nCU * n, (int) ((capacity + WIDE_BLOCK - 1) / WIDE_BLOCK))));
    }
#if PHIP_EXPERIMENTS
    const bool forcePersist = expEnv("PHIP_TRACE_PERSIST") != nullptr;
#endif

    /* fused path: resident grid and per-wave statistics rows */
    dim3 megaGrid(1); MegaParams M; memset(&M, 0, sizeof(M));
    if (fused) {
        megaGrid = dim3((unsigned) (nCU * megaPerCU));
        M.nWaves = megaGrid.x * (BLOCK / 64);
        if (sd.stat.n < (size_t) ST_COUNT * M.nWaves) sd.stat.alloc((size_t) ST_COUNT * M.nWaves);
        M.stat = sd.stat.p; M.nextId = sd.megaNext.p;
        M.nodeCache = megaNodeCache; M.spill = nullptr;
        if (megaWide) {     /* the group stacks' overflow behind their six LDS entries: the buffer always covers the grid, as for k_rays_w (a silent out-of-bounds write otherwise) */
            const size_t spillLanes = (size_t) megaGrid.x * BLOCK;
            if (sd.spill.n < spillLanes * SPILL_DEPTH) sd.spill.alloc(spillLanes * SPILL_DEPTH);
            M.spill = sd.spill.p;
        }
        int *dflag = nullptr; HIP_TRY(hipHostGetDevicePointer((void **) &dflag, sc->cancelFlag, 0));
        M.cancel = dflag;
        P.stat = sd.stat.p; P.nWaves = M.nWaves;               /* k_reduce_stats reads these two */
    }

    for (uint32_t sppDone = 0; sppDone < (uint32_t) p->spp && !cancelled; sppDone += sppPerPass) {
        RenderConst rc;
        rc.sppPass = std::min(sppPerPass, (uint32_t) p->spp - sppDone); rc.sppFirst = (uint32_t) p->sample_offset + sppDone;
        rc.sppMagic = (uint32_t) std::min<unsigned long long>((1ull << 32) / rc.sppPass, 0xFFFFFFFFull);
        rc.tilePixels = (uint32_t) tilePixels; rc.tileShift = (uint32_t) tileShift; rc.nLocalTiles = nLocalTiles;
        rc.totalIds = idsPerSpp * rc.sppPass;
        rc.maxDepth = p->max_depth; rc.rrDepth = p->rr_depth; rc.strictNormals = p->strict_normals; rc.hideEmitters = p->hide_emitters;
        rc.seed = p->seed; rc.tileOrigin = sd.tileOrigin.p; rc.countAlive = 0;
        rc.sampler = (uint32_t) p->sampler; rc.ldMask = (uint32_t) (p->sample_total > 0 ? p->sample_total : p->spp) - 1u;
        memset(&rc.sobol, 0, sizeof(rc.sobol)); rc.stRes = 1;
        if (p->sampler == PHIP_SAMPLER_SOBOL) {
            rc.sobol.matrices = sd.sobolMat.p; rc.sobol.vdc = (const uint64_t *) sd.sobolVdc.p; rc.sobol.vdcInv = (const uint64_t *) sd.sobolVdc.p + PHIP_SOBOL_MATRIX_SIZE;
            rc.sobol.dims = p->sobol_dimensions; rc.sobol.logRes = p->sobol_log_resolution; rc.sobol.scramble = (uint32_t) p->sobol_scramble;
            rc.sobol.resolution = (float) (1u << p->sobol_log_resolution);
            if (!(PHIP_EXPERIMENTS && expEnv("PHIP_SOBOL_BITWISE"))) {      /* (experiment builds, A/B: the row-by-row loops of sobolseq.h -- the product's device code has the byte tables only) */
                rc.sobol.matBt = sd.sobolBt.p;
                rc.sobol.vdcBt = (const uint64_t *) sd.sobolVdcBt.p; rc.sobol.vdcInvBt = (const uint64_t *) sd.sobolVdcBt.p + 4u * 256u;
            }
        } else if (p->sampler == PHIP_SAMPLER_STRATIFIED) {
            const unsigned n = (unsigned) (p->sample_total > 0 ? p->sample_total : p->spp);
            unsigned r = 1; while (r * r < n) ++r;
            rc.stRes = r;
        }
        memset(&rc.rinv, 0, sizeof(rc.rinv));
        if (rinv) {
            RinvTab &T = rc.rinv;
            T.primes = sd.rinvPrimes.p; T.perm = p->qmc_permutations ? sd.rinvPerm.p : nullptr; T.permOffset = sd.rinvOffsets.p; T.dims = p->qmc_dimensions;
            T.invPerm2 = sd.rinvInvPerm2; T.invPerm3 = sd.rinvInvPerm3;
            if (sd.rinvTabDims && !expEnv("PHIP_RINV_DIGITWISE")) {     /* (A/B: the digit-by-digit loops of qmc.cpp) */
                T.dimInfo = sd.rinvDimInfo.p; T.chunk = sd.rinvChunk.p; T.fac = sd.rinvFac.p; T.pw = sd.rinvPw.p; T.tabDims = sd.rinvTabDims;
            }
            const uint32_t res[2] = { (uint32_t) D.film.width, (uint32_t) D.film.height };
            T.sampleCount = (uint32_t) (p->sample_total > 0 ? p->sample_total : p->spp);
            if (p->sampler == PHIP_SAMPLER_HALTON) {
                /* HaltonSampler::setFilmResolution(res, blocked = true), halton.cpp:244-266 */
                T.hammersley = 0; T.stride = 1;
                uint32_t pw[2], ex[2];
                for (int i = 0; i < 2; ++i) {
                    const uint32_t prime = i ? 3u : 2u; uint32_t value = 1, e = 0;
                    while (value < std::min(res[i], RINV_MAX_RESOLUTION)) { value *= prime; ++e; }
                    pw[i] = value; ex[i] = e; T.stride *= value;
                }
                T.powX = pw[0]; T.powY = pw[1]; T.expX = ex[0]; T.expY = ex[1];
                /* multiplicativeInverse(a, n): x with a x = 1 (mod n), in 0 .. n - 1 (halton.cpp:214-241; n = 1 gives 0) */
                auto inverse = [](uint32_t a, uint32_t n) { for (uint32_t x = 0; x < n; ++x) if ((unsigned long long) a * x % n == 1u % n) return x; return 0u; };
                T.multInvX = inverse(T.powY, T.powX); T.multInvY = inverse(T.powX, T.powY);
            } else {
                /* HammersleySampler::setFilmResolution(res, blocked = true), hammersley.cpp:181-196 */
                T.hammersley = 1;
                uint32_t r[2];
                for (int i = 0; i < 2; ++i) { uint32_t v = 1; while (v < res[i]) v <<= 1; r[i] = std::min(RINV_MAX_RESOLUTION, v); }
                T.powX = r[0]; T.powY = r[1]; T.expX = 0; T.expY = 0; while ((1u << T.expY) < r[1]) ++T.expY;
                T.stride = r[1];
                T.factor = 1.0f / (float) ((size_t) T.sampleCount * (size_t) r[0] * (size_t) r[1]);
            }
        }
        rc.diffScaleFactor = 1.0f / sqrtf((float) (p->sample_total > 0 ? p->sample_total : p->spp));
        rc.emitterSamples = direct ? p->emitter_samples : 0; rc.bsdfSamples = direct ? p->bsdf_samples : 0;
        if (direct) {   /* direct.cpp:130-138 */
            const size_t sum = (size_t) p->emitter_samples + (size_t) p->bsdf_samples;
            rc.weightBSDF = 1 / (float) (size_t) p->bsdf_samples;
            rc.weightLum = 1 / (float) (size_t) p->emitter_samples;
            rc.fracBSDF = (size_t) p->bsdf_samples / (float) sum;
            rc.fracLum = (size_t) p->emitter_samples / (float) sum;
        } else {
            rc.weightBSDF = rc.weightLum = 1.0f; rc.fracBSDF = rc.fracLum = 0.5f;
        }
        rc.envFiltered = (sc->envLevelCount > 1 && !(p->flags & PHIP_FLAG_ENVMAP_BILINEAR_BACKGROUND)) ? 1u : 0u;
        rc.volpath = p->integrator == PHIP_INTEGRATOR_VOLPATH_SIMPLE ? 1u : 0u;
        rc.staticIds = 0; rc.shardIds = 0; rc.dynCounter = sd.dynCounter.p; rc.blockShard = sd.blockShard.p;
        rc.jitter = keepJitter ? sd.jitter.p : nullptr;
        HIP_TRY(hipMemsetAsync(sd.counters.p, 0, sizeof(Counters), stream));
        uint32_t iter = 0;

        if (fused) {
            /* ---- one launch: every path of the pass from camera sample to its last vertex (k_mega.h) ---- */
            HIP_TRY(hipMemsetAsync(sd.megaNext.p, 0, sizeof(unsigned long long), stream));
            HIP_TRY(hipMemsetAsync(sd.stat.p, 0, (size_t) ST_COUNT * M.nWaves * sizeof(unsigned long long), stream));
            if (rc.totalIds) {
                if (timing) evFused.record(stream);
                if (direct) phipLaunchMegaDirect(sc->materialMask, false, megaFlat, qmc, megaGrid, megaLds, stream, D, M, rc, sd.L.p);
                else if (megaWide) phipLaunchMegaWide(sc->materialMask, p->strict_normals != 0, megaFlat, qmc, megaGrid, megaLds, stream, D, M, rc, sd.L.p);
                else phipLaunchMega(sc->materialMask, p->strict_normals != 0, megaFlat, qmc, megaGrid, megaLds, stream, D, M, rc, sd.L.p);
                if (timing) evFused.record(stream);
                iter = 1;
            }
            HIP_TRY(hipGetLastError());
            /* k_mega bounds every wait of its mailbox protocol and the depth of its task stacks; a wave that gave up poisons its sample count (k_mega.h).  Such a pass is
               incomplete: it is not added to the film -- this pass and the rest of the job run on the kernels that have no such protocol (round 6: degrade, do not fail) */
            HIP_TRY(hipMemsetAsync(&sd.counters.p->total[ST_SAMPLES], 0, sizeof(unsigned long long), stream));
            hipLaunchKernelGGL(k_reduce_stats, dim3(1, REDUCE_SPLIT), dim3(256), 0, stream, P, sd.counters.p, (int) ST_SAMPLES);
            HIP_TRY(hipMemcpyAsync(&hc.total[ST_SAMPLES], &sd.counters.p->total[ST_SAMPLES], sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            if (cancelRequested(sc)) cancelled = true;
            if (!cancelled && hc.total[ST_SAMPLES] > rc.totalIds) {
                fprintf(stderr, "[phip] warning: the fused kernel gave up on this pass (%s); samples %u.. of the job are rendered by the wavefront kernels\n",
                        megaWide ? "a task stack outgrew LDS + spill buffer" : "a mailbox wait timed out", (unsigned) (p->sample_offset + sppDone));
                phip_render_params q = *p;
                q.flags |= PHIP_FLAG_NO_MEGA | PHIP_FLAG_NO_FUSED;
                q.sample_total = p->sample_total > 0 ? p->sample_total : p->spp;
                q.sample_offset = p->sample_offset + (int) sppDone; q.spp = p->spp - (int) sppDone;
                if (sppDone > 0) q.flags |= PHIP_FLAG_ACCUMULATE;
                phip_stats st2; memset(&st2, 0, sizeof(st2));
                const int rc2 = renderOnDevice(sc, sd, &q, shardIndex, shardCount, dOut, &st2);
                st2.samples += st.samples; st2.closest_rays += st.closest_rays; st2.shadow_rays += st.shadow_rays; st2.path_vertices += st.path_vertices;
                st2.closest_node_visits += st.closest_node_visits; st2.closest_triangle_tests += st.closest_triangle_tests;
                st2.shadow_node_visits += st.shadow_node_visits; st2.shadow_triangle_tests += st.shadow_triangle_tests; st2.iterations += st.iterations;
                st2.render_ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
                if (stats) *stats = st2;
                return rc2;
            }
        } else {
            /* static share: the first 3/4 of every slot's samples; the remainder is handed out dynamically */
            {
                const unsigned long long perSlot = rc.totalIds / capacity;
                unsigned long long staticPerSlot = perSlot - perSlot / 4;
                if (const char *e = expEnv("PHIP_STATIC_PERCENT")) staticPerSlot = perSlot * (unsigned long long) atoi(e) / 100;
                rc.staticIds = staticPerSlot * capacity;
                const unsigned long long dyn = rc.totalIds - rc.staticIds;
                rc.shardIds = (dyn + DYN_SHARDS - 1) / DYN_SHARDS;
                HIP_TRY(hipMemsetAsync(sd.blockShard.p, 0, nBlocks * sizeof(uint32_t), stream));
                HIP_TRY(hipMemsetAsync(sd.dynCounter.p, 0, DYN_SHARDS * DYN_STRIDE * sizeof(unsigned long long), stream));
            }
            HIP_TRY(hipMemsetAsync(sd.stat.p, 0, (size_t) ST_COUNT * nWaves * sizeof(unsigned long long), stream));
            HIP_TRY(hipMemsetAsync(sd.shadowCount.p, 0, (size_t) nBlocks * sizeof(uint32_t), stream));
            HIP_TRY(hipMemsetAsync(sd.blockDead.p, 0, (size_t) nBlocks * sizeof(uint32_t), stream));
            HIP_TRY(hipMemsetD32Async((hipDeviceptr_t) sd.state.p, (int) F_FRESH, (size_t) capacity, stream));
            /* (no clear of L: the first vertex of every sample -- or its camera ray leaving the scene -- writes L[id]) */

            HIP_TRY(hipStreamSynchronize(stream));
            const auto tLoop0 = clk::now();
            bool done = rc.totalIds == 0;
            bool drainingSeen = false;                          /* a termination test has counted fewer live slots than the pool holds: blocks may have retired */
            /* environment emitter, bitmap textures; 8: the QMC samplers (two builds: the plain one, and one with both other features) */
            const int feat0 = (D.envEmitter >= 0 ? 1 : 0) | (sc->hasTextures ? 2 : 0), feat = qmc ? (feat0 ? 11 : 8) : feat0;
            const size_t shadeTraceLds = shadeTraceLdsBytes(D);
            while (!done) {
                const bool check = ((iter + 1) & 7) == 0 || rc.totalIds <= (unsigned long long) capacity * 4;
                rc.countAlive = check ? 1 : 0;
                rc.draining = drainingSeen ? 1u : 0u;
                if (timing) evShade.record(stream);
                if (direct) phipLaunchShadeDirect(feat, sc->materialMask, grid, stream, D, P, rc, sd.L.p);
                else if (shadeTrace) phipLaunchShadeTrace(feat, rc.strictNormals != 0, sc->materialMask, grid, shadeTraceLds, stream, D, P, rc, sd.L.p);
                else phipLaunchShade(feat, rc.strictNormals != 0, sc->materialMask, grid, stream, D, P, rc, sd.L.p);
                if (timing) evShade.record(stream);
                if (shadeTrace) {
                    /* (the vertex kernel traced both rays of the iteration itself: k_shade_trace.h) */
                } else if (merged) {
                    if (timing) evTrace.record(stream);
                    if (sc->wide) {
                        /* k_rays_w draws its chunks from sharded counters */
                        if (sd.drawCounters.n < 2 * RAY_SHARDS * RAY_SHARD_STRIDE) sd.drawCounters.alloc(2 * RAY_SHARDS * RAY_SHARD_STRIDE);
                        HIP_TRY(hipMemsetAsync(sd.drawCounters.p, 0, 2 * RAY_SHARDS * RAY_SHARD_STRIDE * sizeof(unsigned int), stream));
                        hipLaunchKernelGGL(k_rays_w, pgridRays, dim3(WIDE_BLOCK), wideLds, stream, D, P, sd.L.p, sd.drawCounters.p);
                    }
#if PHIP_EXPERIMENTS
                    else hipLaunchKernelGGL(k_rays_p, pgridRays, block, ldsBytes, stream, D, P, sd.L.p);
#endif
                    if (timing) evTrace.record(stream);
                } else {
#if PHIP_EXPERIMENTS      /* the BVH4 ray kernels of round 1 (PHIP_BVH4_RAYS=1 on a small scene, PHIP_WIDE=0 / PHIP_TRAVERSAL=lane): A/B only */
                    if (timing) evShadow.record(stream);
                    if (sc->traversal != 2) hipLaunchKernelGGL(k_shadow, grid, block, ldsBytes, stream, D, P, sd.L.p); else
                    hipLaunchKernelGGL(k_shadow_p, pgrid, block, ldsBytes, stream, D, P, sd.L.p);
                    if (timing) evShadow.record(stream);
                    if (timing) evTrace.record(stream);
                    if (sc->traversal == 2 && sc->bvh.nNodes >= 64) hipLaunchKernelGGL(k_trace_p<false>, pgridTrace, block, ldsBytes, stream, D, P);
                    else if (sc->traversal == 2 && forcePersist) hipLaunchKernelGGL(k_trace_p<true>, pgridTrace, block, ldsBytes, stream, D, P);
                    else
                    hipLaunchKernelGGL(k_trace, grid, block, ldsBytes, stream, D, P);
                    if (timing) evTrace.record(stream);
#else
                    throw std::runtime_error("internal error: no ray kernel for this scene (the wide tree is missing)");
#endif
                }
                ++iter;
                if (check) {
                    /* termination test: only the live-slot row (and, for a progress callback, the finished-sample row) is summed inside the loop */
                    HIP_TRY(hipMemsetAsync(&sd.counters.p->total[ST_ALIVE], 0, sizeof(unsigned long long), stream));
                    hipLaunchKernelGGL(k_reduce_stats, dim3(1, REDUCE_SPLIT), dim3(256), 0, stream, P, sd.counters.p, (int) ST_ALIVE);
                    HIP_TRY(hipMemcpyAsync(&hc.total[ST_ALIVE], &sd.counters.p->total[ST_ALIVE], sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
                    if (p->progress) {
                        HIP_TRY(hipMemsetAsync(&sd.counters.p->total[ST_SAMPLES], 0, sizeof(unsigned long long), stream));
                        hipLaunchKernelGGL(k_reduce_stats, dim3(1, REDUCE_SPLIT), dim3(256), 0, stream, P, sd.counters.p, (int) ST_SAMPLES);
                        HIP_TRY(hipMemcpyAsync(&hc.total[ST_SAMPLES], &sd.counters.p->total[ST_SAMPLES], sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
                    }
                    HIP_TRY(hipStreamSynchronize(stream));
                    if (hc.total[ST_ALIVE] == 0) done = true;
                    if (hc.total[ST_ALIVE] < capacity) drainingSeen = true;
                    if (p->progress) { std::lock_guard<std::mutex> g(sc->progressLock); p->progress(p->progress_user, sd.device, samplesDone + hc.total[ST_SAMPLES], samplesTotal * (unsigned long long) p->spp); }
                    if (cancelRequested(sc)) { cancelled = true; done = true; }
                }
            }
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(stream));
            if (getenv("PHIP_DEBUG_TIMING")) fprintf(stderr, "[phip] setup %.2f ms, loop %.2f ms (%u iterations)\n", std::chrono::duration<double, std::milli>(tLoop0 - t0).count(), std::chrono::duration<double, std::milli>(clk::now() - tLoop0).count(), iter);
        }
        st.iterations += iter;
        if (cancelled) break;                                   /* L is incomplete: no film pass */

        /* film */
        if (timing) evFilm.record(stream);
        {
            const dim3 fg((W + 15) / 16, (H + 15) / 16);
            const int reach = (int) std::floor(D.film.radius + 0.5f);
            const int acc = (sppDone > 0 || accumulate) ? 1 : 0;
            const char *fv = expEnv("PHIP_FILM_V1"); (void) fv;       /* experiment hook: the round-2 tiled kernel */
            const bool splat = reach <= 2 && tileShift >= 4 && bs == (1 << tileShift) && nLocalTiles > 0 && !expEnv("PHIP_FILM_GATHER");     /* PHIP_FILM_GATHER: the tiled gathers of rounds 2 / 3 (A/B) */
            if (splat) {
                /* round 4: one pass over L with the footprint sums in registers, then an ordered merge of the 16 x 16 patch images (k_film.h) */
                const int r = std::max(reach, 1), cells = (16 + 2 * r) * (16 + 2 * r) * 5;
                const unsigned nPatches = (unsigned) nLocalTiles << (2 * (tileShift - 4));
                if (sd.patchImg.n < (size_t) nPatches * cells) sd.patchImg.alloc((size_t) nPatches * cells);
                const float4 *Lc = (const float4 *) sd.L.p;
                if (r == 1) { if (qmc) hipLaunchKernelGGL((k_film_splat<1, true>), dim3(nPatches), dim3(256), 0, stream, D, rc, Lc, sd.patchImg.p, sd.invalid.p);
                              else hipLaunchKernelGGL((k_film_splat<1, false>), dim3(nPatches), dim3(256), 0, stream, D, rc, Lc, sd.patchImg.p, sd.invalid.p);
                              hipLaunchKernelGGL(k_film_merge<1>, fg, block, 0, stream, D, rc, (const float *) sd.patchImg.p, (const int32_t *) sd.tileSlot.p, tilesX, dOut, acc); }
                else        { if (qmc) hipLaunchKernelGGL((k_film_splat<2, true>), dim3(nPatches), dim3(256), 0, stream, D, rc, Lc, sd.patchImg.p, sd.invalid.p);
                              else hipLaunchKernelGGL((k_film_splat<2, false>), dim3(nPatches), dim3(256), 0, stream, D, rc, Lc, sd.patchImg.p, sd.invalid.p);
                              hipLaunchKernelGGL(k_film_merge<2>, fg, block, 0, stream, D, rc, (const float *) sd.patchImg.p, (const int32_t *) sd.tileSlot.p, tilesX, dOut, acc); }
            } else if (qmc)
                hipLaunchKernelGGL(k_film<true>, fg, block, 0, stream, D, rc, (const float4 *) sd.L.p, (const int32_t *) sd.tileSlot.p, tilesX, dOut,
                                   acc, sd.invalid.p);
#if PHIP_EXPERIMENTS
            else if (reach <= 2 && bs >= FILM_TILE + 2 * std::max(reach, 1) && !expEnv("PHIP_FILM_GENERIC") && !(fv && atoi(fv))) {
                if (reach <= 1)
                    hipLaunchKernelGGL(k_film_tiled2<1>, fg, block, 0, stream, D, rc, (const float4 *) sd.L.p, (const int32_t *) sd.tileSlot.p, tilesX, dOut, acc, sd.invalid.p);
                else
                    hipLaunchKernelGGL(k_film_tiled2<2>, fg, block, 0, stream, D, rc, (const float4 *) sd.L.p, (const int32_t *) sd.tileSlot.p, tilesX, dOut, acc, sd.invalid.p);
            }
#endif
            else if (reach <= FILM_MAX_REACH && !expEnv("PHIP_FILM_GENERIC"))
                if (reach <= 2)
                    hipLaunchKernelGGL(k_film_tiled<2>, fg, block, 0, stream, D, rc, (const float4 *) sd.L.p, (const int32_t *) sd.tileSlot.p, tilesX, dOut,
                                       acc, sd.invalid.p, reach);
                else
                    hipLaunchKernelGGL(k_film_tiled<FILM_MAX_REACH>, fg, block, 0, stream, D, rc, (const float4 *) sd.L.p, (const int32_t *) sd.tileSlot.p, tilesX, dOut,
                                       acc, sd.invalid.p, reach);
            else
                hipLaunchKernelGGL(k_film<false>, fg, block, 0, stream, D, rc, (const float4 *) sd.L.p, (const int32_t *) sd.tileSlot.p, tilesX, dOut,
                                   acc, sd.invalid.p);
        }
        if (timing) evFilm.record(stream);
        if (keepSamples && rc.totalIds) {
            const size_t n = (size_t) W * H * rc.sppPass;
            hipLaunchKernelGGL(k_export_samples, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, stream, D, rc, (const float4 *) sd.L.p,
                               (const int32_t *) sd.tileSlot.p, tilesX, sd.sampleOut.p, (uint32_t) p->spp, (uint32_t) p->sample_offset);
        }
        HIP_TRY(hipMemsetAsync(sd.counters.p, 0, sizeof(Counters), stream));
        hipLaunchKernelGGL(k_reduce_stats, dim3(ST_COUNT, REDUCE_SPLIT), dim3(256), 0, stream, P, sd.counters.p, 0);
        HIP_TRY(hipMemcpyAsync(&hc, sd.counters.p, sizeof(Counters), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipGetLastError());
        st.samples += hc.total[ST_SAMPLES]; st.closest_rays += hc.total[ST_CLOSEST_RAYS]; st.shadow_rays += hc.total[ST_SHADOW_RAYS];
        st.path_vertices += hc.total[ST_VERTICES]; st.closest_node_visits += hc.total[ST_NODE]; st.closest_triangle_tests += hc.total[ST_TRI];
        st.shadow_node_visits += hc.total[ST_SH_NODE]; st.shadow_triangle_tests += hc.total[ST_SH_TRI];
        samplesDone += hc.total[ST_SAMPLES];
        if (p->progress) { std::lock_guard<std::mutex> g(sc->progressLock); p->progress(p->progress_user, sd.device, samplesDone, samplesTotal * (unsigned long long) p->spp); }
    }
    if ((nLocalTiles == 0 || cancelled) && !accumulate) { HIP_TRY(hipMemsetAsync(dOut, 0, (size_t) W * H * 5 * sizeof(float), stream)); HIP_TRY(hipStreamSynchronize(stream)); }
    unsigned long long inv = 0;
    HIP_TRY(hipMemcpy(&inv, sd.invalid.p, sizeof(inv), hipMemcpyDeviceToHost));
    st.invalid_samples = inv;
    st.trace_kernel_ms = evTrace.sumPairs(); st.shadow_kernel_ms = evShadow.sumPairs();
    st.shade_kernel_ms = evShade.sumPairs(); st.film_kernel_ms = evFilm.sumPairs(); st.fused_kernel_ms = evFused.sumPairs();
    st.fused = fused ? 1u : 0u; st.n_devices = 1;
    st.vertex_traced = shadeTrace ? 1u : 0u;
    st.render_ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
    algorithmicBytes(merged, st, (double) W * H, sc->wide);
    if (stats) *stats = st;
    return cancelled ? PHIP_ERR_CANCELLED : PHIP_OK;
}

/* ---- RCCL, bound at the first multi-GPU render (librccl is not a load-time dependency of single-GPU users; a process that
   already carries an RCCL -- PyTorch does -- keeps exactly one copy) ---- */
namespace {
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::mutex lock;
    std::map<std::vector<int>, std::vector<ncclComm_t>> comms;       /* one communicator clique per device list, kept for the process */
    void bind() {
        if (handle) return;
        for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) { handle = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (handle) break; }
        if (!handle) throw std::runtime_error(std::string("multi-GPU render needs librccl: ") + dlerror());
        auto sym = [&](const char *n) { void *s = dlsym(handle, n); if (!s) throw std::runtime_error(std::string("librccl lacks ") + n); return s; };
        CommInitAll = (decltype(CommInitAll)) sym("ncclCommInitAll"); CommDestroy = (decltype(CommDestroy)) sym("ncclCommDestroy");
        Reduce = (decltype(Reduce)) sym("ncclReduce"); GroupStart = (decltype(GroupStart)) sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd)) sym("ncclGroupEnd"); GetErrorString = (decltype(GetErrorString)) sym("ncclGetErrorString");
    }
    void check(ncclResult_t r, const char *what) { if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + (GetErrorString ? GetErrorString(r) : "RCCL error")); }
    const std::vector<ncclComm_t> &clique(const std::vector<int> &devices) {
        auto it = comms.find(devices);
        if (it != comms.end()) return it->second;
        std::vector<ncclComm_t> c(devices.size());
        check(CommInitAll(c.data(), (int) devices.size(), devices.data()), "ncclCommInitAll");
        return comms.emplace(devices, std::move(c)).first->second;
    }
};
Rccl g_rccl;

__global__ void k_add_films(float *dst, const float *src, size_t n) {
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}
} // namespace

/* The RCCL calls of the merge on whatever is there: librccl is bound as renderMultiDevice binds it, a clique is made of the first
   min(n, visible) devices -- ONE device is a valid clique --, and ncclReduce(sum) is run inside a group as the merge runs it, on a buffer
   of ones.  On the single-GPU boxes of this pool that is the only way the six entry points ever execute (the merge itself needs two distinct
   devices); returns the number of devices that took part, or a negative error code (phip_last_error). */
#if PHIP_DEBUG_HOOKS      /* (libphip_debug.so only) */
extern "C" int phip_debug_rccl_selftest(int n_devices, size_t n_floats) {
    try {
        int visible = 0; HIP_TRY(hipGetDeviceCount(&visible));
        const int n = std::max(1, std::min(n_devices, visible));
        std::vector<int> devices(n); for (int i = 0; i < n; ++i) devices[i] = i;
        std::lock_guard<std::mutex> g(g_rccl.lock);
        g_rccl.bind();
        const std::vector<ncclComm_t> &comm = g_rccl.clique(devices);
        std::vector<float *> buf(n, nullptr); std::vector<hipStream_t> st(n, nullptr);
        std::vector<float> ones(n_floats, 1.0f);
        for (int i = 0; i < n; ++i) {
            HIP_TRY(hipSetDevice(devices[i])); HIP_TRY(hipStreamCreate(&st[i]));
            HIP_TRY(hipMalloc((void **) &buf[i], n_floats * sizeof(float)));
            HIP_TRY(hipMemcpy(buf[i], ones.data(), n_floats * sizeof(float), hipMemcpyHostToDevice));
        }
        g_rccl.check(g_rccl.GroupStart(), "ncclGroupStart");
        for (int i = 0; i < n; ++i) {
            HIP_TRY(hipSetDevice(devices[i]));
            g_rccl.check(g_rccl.Reduce(buf[i], buf[i], n_floats, ncclFloat, ncclSum, 0, comm[i], st[i]), "ncclReduce");
        }
        g_rccl.check(g_rccl.GroupEnd(), "ncclGroupEnd");
        for (int i = 0; i < n; ++i) { HIP_TRY(hipSetDevice(devices[i])); HIP_TRY(hipStreamSynchronize(st[i])); }
        HIP_TRY(hipSetDevice(devices[0]));
        HIP_TRY(hipMemcpy(ones.data(), buf[0], n_floats * sizeof(float), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) { HIP_TRY(hipSetDevice(devices[i])); (void) hipFree(buf[i]); (void) hipStreamDestroy(st[i]); }
        HIP_TRY(hipSetDevice(devices[0]));
        for (size_t k = 0; k < n_floats; ++k) if (ones[k] != (float) n) return setErr(PHIP_ERR_DEVICE, "ncclReduce(sum) of ones over " + std::to_string(n) + " device(s) gave " + std::to_string(ones[k]));
        return n;
    } catch (const std::exception &e) { return setErr(PHIP_ERR_DEVICE, e.what()); }
}
#endif

/* The call's shard on p->n_devices GPUs: one host thread + stream per device, blocks dealt round-robin in the reference's
   spiral order, films merged on devices[0] by one ncclReduce(sum) -- the in-process analogue of the reference's workers
   handing ImageBlocks to BlockedRenderProcess::processResult (renderproc.cpp:142-149). */
static int renderMultiDevice(phip_scene *sc, const phip_render_params *p, float *dOut, phip_stats *stats) {
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    const int n = p->n_devices;
    const bool alias = (p->flags & PHIP_FLAG_ALIAS_DEVICES) != 0;
    int visible = 0; HIP_TRY(hipGetDeviceCount(&visible));
    if (p->devices[0] != sc->devs[0]->device) throw std::invalid_argument("devices[0] must be the scene's device");
    std::vector<int> devices(p->devices, p->devices + n);
    bool distinct = true;
    for (int i = 0; i < n; ++i) {
        if (devices[i] < 0 || devices[i] >= visible) throw std::invalid_argument("device ordinal out of range");
        for (int j = 0; j < i; ++j) if (devices[j] == devices[i]) distinct = false;
    }
    if (!distinct && !alias) throw std::invalid_argument("a device is listed twice (PHIP_FLAG_ALIAS_DEVICES allows it for tests)");
    /* replicas: position i of the list renders on devs[i] */
    for (int i = 1; i < n; ++i) {
        if ((int) sc->devs.size() > i && sc->devs[i]->device == devices[i]) continue;
        if ((int) sc->devs.size() > i) sc->devs.resize(i);          /* a different device list: rebuild from here */
        replicateScene(sc, devices[i]);
    }
    const int W = sc->devs[0]->dev.film.width, H = sc->devs[0]->dev.film.height;
    const size_t filmFloats = (size_t) W * H * 5;
    const int S = p->shard_count > 0 ? p->shard_count : 1, s = p->shard_index;
    std::vector<float *> out(n, nullptr);
    out[0] = dOut;
    for (int i = 1; i < n; ++i) {
        SceneDev &sd = *sc->devs[i];
        HIP_TRY(hipSetDevice(sd.device));
        if (sd.film.n < filmFloats) sd.film.alloc(filmFloats);
        out[i] = sd.film.p;
    }
    std::vector<phip_stats> st(n);
    std::vector<int> rc(n, PHIP_OK);
    std::vector<std::string> err(n);
    std::vector<std::thread> workers;
    phip_render_params q = *p;
    q.flags &= ~PHIP_FLAG_SAMPLE_BUFFER;                         /* per-sample export is a single-device test hook */
    for (int i = 0; i < n; ++i) {
        workers.emplace_back([&, i]() {
            try {
                phip_render_params mine = q;
                if (i > 0) mine.flags &= ~PHIP_FLAG_ACCUMULATE;   /* only the root's buffer carries the previous calls */
                rc[i] = renderOnDevice(sc, *sc->devs[i], &mine, s + S * i, S * n, out[i], &st[i]);
            } catch (const std::invalid_argument &e) { rc[i] = PHIP_ERR_INVALID; err[i] = e.what(); }
              catch (const std::exception &e) { rc[i] = PHIP_ERR_DEVICE; err[i] = e.what(); }
        });
    }
    for (auto &w : workers) w.join();
    bool cancelled = false;
    for (int i = 0; i < n; ++i) {
        if (rc[i] == PHIP_ERR_CANCELLED) cancelled = true;
        else if (rc[i] != PHIP_OK) return setErr(rc[i], "device " + std::to_string(devices[i]) + ": " + err[i]);
    }
    /* ---- merge: film(devices[0]) += sum of the others ---- */
    const auto tr0 = clk::now();
    if (!cancelled) {
        if (distinct) {
            std::lock_guard<std::mutex> g(g_rccl.lock);
            g_rccl.bind();
            const std::vector<ncclComm_t> &comm = g_rccl.clique(devices);
            g_rccl.check(g_rccl.GroupStart(), "ncclGroupStart");
            for (int i = 0; i < n; ++i) {
                SceneDev &sd = *sc->devs[i];
                HIP_TRY(hipSetDevice(sd.device));
                g_rccl.check(g_rccl.Reduce(out[i], out[i], filmFloats, ncclFloat, ncclSum, 0, comm[i], sd.stream), "ncclReduce");
            }
            g_rccl.check(g_rccl.GroupEnd(), "ncclGroupEnd");
            for (int i = 0; i < n; ++i) { HIP_TRY(hipSetDevice(sc->devs[i]->device)); HIP_TRY(hipStreamSynchronize(sc->devs[i]->stream)); }
        } else {
            /* aliased devices (test hook): the films are in the same memory, a kernel sums them */
            HIP_TRY(hipSetDevice(devices[0]));
            for (int i = 1; i < n; ++i)
                hipLaunchKernelGGL(k_add_films, dim3((unsigned) ((filmFloats + 255) / 256)), dim3(256), 0, sc->devs[0]->stream, out[0], (const float *) out[i], filmFloats);
            HIP_TRY(hipStreamSynchronize(sc->devs[0]->stream));
            HIP_TRY(hipGetLastError());
        }
    }
    HIP_TRY(hipSetDevice(devices[0]));
    if (stats) {
        phip_stats t; memset(&t, 0, sizeof(t));
        for (int i = 0; i < n; ++i) {
            const phip_stats &a = st[i];
            t.samples += a.samples; t.closest_rays += a.closest_rays; t.shadow_rays += a.shadow_rays; t.path_vertices += a.path_vertices;
            t.closest_node_visits += a.closest_node_visits; t.closest_triangle_tests += a.closest_triangle_tests;
            t.shadow_node_visits += a.shadow_node_visits; t.shadow_triangle_tests += a.shadow_triangle_tests;
            t.invalid_samples += a.invalid_samples; t.iterations = std::max(t.iterations, a.iterations);
            t.trace_kernel_ms = std::max(t.trace_kernel_ms, a.trace_kernel_ms); t.shadow_kernel_ms = std::max(t.shadow_kernel_ms, a.shadow_kernel_ms);
            t.shade_kernel_ms = std::max(t.shade_kernel_ms, a.shade_kernel_ms); t.film_kernel_ms = std::max(t.film_kernel_ms, a.film_kernel_ms);
            t.fused_kernel_ms = std::max(t.fused_kernel_ms, a.fused_kernel_ms);
            t.algorithmic_bytes += a.algorithmic_bytes; t.trace_kernel_bytes += a.trace_kernel_bytes; t.fused |= a.fused; t.vertex_traced |= a.vertex_traced;
        }
        t.n_devices = (uint32_t) n;
        t.reduce_ms = std::chrono::duration<double, std::milli>(clk::now() - tr0).count();
        t.render_ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
        *stats = t;
    }
    return cancelled ? PHIP_ERR_CANCELLED : PHIP_OK;
}

/* Entry of both render functions: validation, the scene's render lock, single- or multi-device dispatch, the sticky
   cancellation flag (consumed by the call that observed it). */
static int renderLocked(phip_scene *sc, const phip_render_params *p, float *dOut /* device memory on the scene's device */, phip_stats *stats) {      /* (the caller holds sc->renderLock) */
    validateParams(sc, p);
    int rc;
    if (p->n_devices > 1) {
        rc = renderMultiDevice(sc, p, dOut, stats);
    } else {
        const int device = p->n_devices == 1 ? p->devices[0] : p->device;
        if (device != sc->devs[0]->device) throw std::invalid_argument("scene was created on a different device");
        rc = renderOnDevice(sc, *sc->devs[0], p, p->shard_index, p->shard_count > 0 ? p->shard_count : 1, dOut, stats);
    }
    if (rc == PHIP_ERR_CANCELLED) {
        __atomic_store_n(sc->cancelFlag, 0, __ATOMIC_RELAXED);
        return setErr(PHIP_ERR_CANCELLED, "rendering was cancelled");
    }
    return rc;
}
static int renderImpl(phip_scene *sc, const phip_render_params *p, float *dOut, phip_stats *stats) {
    std::lock_guard<std::mutex> lock(sc->renderLock);
    return renderLocked(sc, p, dOut, stats);
}

/* ======================================================================================
 *  C ABI
 * ====================================================================================== */
extern "C" {

const char *phip_last_error(void) { return g_err.c_str(); }
#define PHIP_STR2(x) #x
#define PHIP_STR(x) PHIP_STR2(x)
#ifndef PHIP_BUILD_ID
#define PHIP_BUILD_ID "unknown-build-id"
#endif
/* the hash of the sources and flags this library was compiled from (mitsuba_amd/_ffi.py: source_id) */
const char *phip_build_id(void) { static const char tag[] = "phip-build-id:" PHIP_BUILD_ID; return tag + 14; }
const char *phip_version(void) { return "path_hip 0.6 (gfx950, abi " PHIP_STR(PHIP_ABI_VERSION) ")"; }

int phip_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return setErr(PHIP_ERR_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    return n;
}

phip_scene *phip_scene_create(const phip_scene_desc *desc, int device) {
    if (!desc) { setErr(PHIP_ERR_INVALID, "desc is NULL"); return nullptr; }
    int n = phip_device_count();
    if (n <= 0) { if (n == 0) setErr(PHIP_ERR_DEVICE, "no HIP device visible: path_hip has no CPU fallback"); return nullptr; }
    if (device < 0 || device >= n) { setErr(PHIP_ERR_INVALID, "device ordinal out of range"); return nullptr; }
    phip_scene *sc = new (std::nothrow) phip_scene();
    if (!sc) { setErr(PHIP_ERR_NOMEM, "out of memory"); return nullptr; }
    sc->devs[0]->device = device;
    try {
        buildScene(sc, *desc);
        return sc;
    } catch (const std::invalid_argument &e) {
        setErr(PHIP_ERR_UNSUPPORTED, e.what());
    } catch (const std::exception &e) {
        setErr(PHIP_ERR_INVALID, e.what());
    }
    delete sc;
    return nullptr;
}

void phip_scene_destroy(phip_scene *scene) {
    if (!scene) return;
    delete scene;
}

int phip_scene_replicate(phip_scene *scene, const int32_t *devices, int32_t n_devices) {
    if (!scene || (!devices && n_devices)) return setErr(PHIP_ERR_INVALID, "NULL argument");
    if (n_devices < 1 || n_devices > PHIP_MAX_DEVICES) return setErr(PHIP_ERR_INVALID, "n_devices out of range");
    try {
        std::lock_guard<std::mutex> lock(scene->renderLock);
        int visible = 0; HIP_TRY(hipGetDeviceCount(&visible));
        if (devices[0] != scene->devs[0]->device) return setErr(PHIP_ERR_INVALID, "devices[0] must be the scene's device");
        for (int i = 1; i < n_devices; ++i) {
            if (devices[i] < 0 || devices[i] >= visible) return setErr(PHIP_ERR_INVALID, "device ordinal out of range");
            if ((int) scene->devs.size() > i && scene->devs[i]->device == devices[i]) continue;
            if ((int) scene->devs.size() > i) scene->devs.resize(i);
            replicateScene(scene, devices[i]);
        }
        HIP_TRY(hipSetDevice(scene->devs[0]->device));
        return PHIP_OK;
    } catch (const std::exception &e) {
        return setErr(PHIP_ERR_DEVICE, e.what());
    }
}

int phip_render_device(phip_scene *scene, const phip_render_params *params, void *d_out, phip_stats *out_stats) {
    if (!scene || !params || !d_out) return setErr(PHIP_ERR_INVALID, "NULL argument");
    try {
        return renderImpl(scene, params, (float *) d_out, out_stats);
    } catch (const std::invalid_argument &e) {
        return setErr(PHIP_ERR_INVALID, e.what());
    } catch (const std::exception &e) {
        return setErr(PHIP_ERR_DEVICE, e.what());
    }
}

/* The film's way to the host (inside the metric: SURVEY 8(d), renderjob.cpp:105 -- the reference's `Render time` includes film->put).
   A pageable hipMemcpy of the 20 MB C2 film measured 8.4 ms (2.5 GB/s: the runtime stages it through one small pinned buffer, copy and
   memcpy in turn).  Here:
     * `dst` is pinned memory (phip_host_alloc, hipHostMalloc, torch pin_memory, hipHostRegister): ONE asynchronous copy at the link's rate;
     * `dst` is pageable (the Bitmap of the Mitsuba shim): the film crosses in FILM_STAGE_BYTES chunks through two pinned staging buffers,
       the copy of chunk i + 1 in flight while chunk i is moved to `dst` by up to four host threads. */
#define FILM_STAGE_BYTES ((size_t) 4 << 20)
static void parallelCopy(char *dst, const char *src, size_t bytes) {
    const size_t nThreads = std::min<size_t>(4, bytes / ((size_t) 512 << 10));
    if (nThreads < 2) { memcpy(dst, src, bytes); return; }
    std::vector<std::thread> th;
    const size_t per = ((bytes / nThreads) + 4095) & ~(size_t) 4095;
    for (size_t i = 1; i < nThreads; ++i) {
        const size_t b = i * per, e = std::min(bytes, b + per);
        if (b < e) th.emplace_back([=] { memcpy(dst + b, src + b, e - b); });
    }
    memcpy(dst, src, std::min(bytes, per));
    for (auto &t : th) t.join();
}
static void filmToHost(SceneDev &sd, float *dst, const float *dFilm, size_t bytes) {
    if (!sd.stream) HIP_TRY(hipStreamCreate(&sd.stream));
    hipPointerAttribute_t attr; memset(&attr, 0, sizeof(attr));
    const hipError_t e = hipPointerGetAttributes(&attr, dst);
    if (e != hipSuccess) (void) hipGetLastError();             /* an unregistered host pointer is "invalid value" on some runtimes: pageable */
    if (e == hipSuccess && attr.type == hipMemoryTypeHost) {
        HIP_TRY(hipMemcpyAsync(dst, dFilm, bytes, hipMemcpyDeviceToHost, sd.stream));
        HIP_TRY(hipStreamSynchronize(sd.stream));
        return;
    }
    if (bytes <= ((size_t) 256 << 10)) {                       /* a small frame: not worth 8 MB of pinned staging (their allocation is 8 ms) */
        HIP_TRY(hipMemcpy(dst, dFilm, bytes, hipMemcpyDeviceToHost));
        return;
    }
    for (int i = 0; i < 2; ++i) {
        if (!sd.stage[i]) HIP_TRY(hipHostMalloc((void **) &sd.stage[i], FILM_STAGE_BYTES, hipHostMallocDefault));
        if (!sd.stageDone[i]) HIP_TRY(hipEventCreateWithFlags(&sd.stageDone[i], hipEventDisableTiming));
    }
    const size_t nChunks = (bytes + FILM_STAGE_BYTES - 1) / FILM_STAGE_BYTES;
    auto issue = [&](size_t c) {
        const size_t off = c * FILM_STAGE_BYTES, len = std::min(FILM_STAGE_BYTES, bytes - off);
        HIP_TRY(hipMemcpyAsync(sd.stage[c & 1], (const char *) dFilm + off, len, hipMemcpyDeviceToHost, sd.stream));
        HIP_TRY(hipEventRecord(sd.stageDone[c & 1], sd.stream));
    };
    if (nChunks) issue(0);
    for (size_t c = 0; c < nChunks; ++c) {
        if (c + 1 < nChunks) issue(c + 1);                     /* chunk c + 1 lands in the other buffer while chunk c is moved out of this one */
        HIP_TRY(hipEventSynchronize(sd.stageDone[c & 1]));
        const size_t off = c * FILM_STAGE_BYTES, len = std::min(FILM_STAGE_BYTES, bytes - off);
        parallelCopy((char *) dst + off, (const char *) sd.stage[c & 1], len);
    }
}

void *phip_host_alloc(size_t bytes) {
    void *p = nullptr;
    const hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable);
    if (e != hipSuccess) { setErr(PHIP_ERR_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e)); return nullptr; }
    return p;
}
void phip_host_free(void *p) { if (p) (void) hipHostFree(p); }

int phip_render(phip_scene *scene, const phip_render_params *params, float *out_rgbaw, phip_stats *out_stats) {
    if (!scene || !params || !out_rgbaw) return setErr(PHIP_ERR_INVALID, "NULL argument");
    try {
        SceneDev &sd = *scene->devs[0];
        HIP_TRY(hipSetDevice(sd.device));
        const size_t n = (size_t) sd.dev.film.width * sd.dev.film.height * 5;
        /* ONE critical section for the render and the delivery of its frame (the frame lives in the scene-owned sd.film -- with the lock dropped in between, a second
           thread rendering the same scene could have its frame copied out by this one, or a half-written one) */
        std::lock_guard<std::mutex> lock(scene->renderLock);
        if (sd.film.n < n) {
            if (params->flags & PHIP_FLAG_ACCUMULATE) return setErr(PHIP_ERR_INVALID, "PHIP_FLAG_ACCUMULATE without a previous phip_render on this scene");
            sd.film.alloc(n);
        }
        int rc = renderLocked(scene, params, sd.film.p, out_stats);
        if (rc != PHIP_OK) return rc;
        const auto t0 = std::chrono::steady_clock::now();
        filmToHost(sd, out_rgbaw, sd.film.p, n * sizeof(float));
        if (out_stats) {
            out_stats->d2h_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            out_stats->render_ms += out_stats->d2h_ms;
        }
        return PHIP_OK;
    } catch (const std::invalid_argument &e) {
        return setErr(PHIP_ERR_INVALID, e.what());
    } catch (const std::exception &e) {
        return setErr(PHIP_ERR_DEVICE, e.what());
    }
}

int phip_film_to_host(phip_scene *scene, const void *d_rgbaw, float *out_rgbaw) {
    if (!scene || !d_rgbaw || !out_rgbaw) return setErr(PHIP_ERR_INVALID, "NULL argument");
    try {
        SceneDev &sd = *scene->devs[0];
        HIP_TRY(hipSetDevice(sd.device));
        const size_t bytes = (size_t) sd.dev.film.width * sd.dev.film.height * 5 * sizeof(float);
        /* the frame has to be device memory of the scene's first device (a pointer of another GPU, or a host pointer, used to be trusted) */
        hipPointerAttribute_t attr; memset(&attr, 0, sizeof(attr));
        if (hipPointerGetAttributes(&attr, d_rgbaw) != hipSuccess) { (void) hipGetLastError(); return setErr(PHIP_ERR_INVALID, "phip_film_to_host: d_rgbaw is not a device pointer"); }
        if (attr.type != hipMemoryTypeDevice || attr.device != sd.device)
            return setErr(PHIP_ERR_INVALID, "phip_film_to_host: d_rgbaw must be device memory of the scene's device");
        /* the staging buffers and the stream are the scene's: one delivery at a time, and not while a render of the scene uses them (phip_render's own copy) */
        std::lock_guard<std::mutex> lock(scene->renderLock);
        HIP_TRY(hipDeviceSynchronize());                       /* the frame may have been written on another stream (an RCCL reduce on torch's) */
        filmToHost(sd, out_rgbaw, (const float *) d_rgbaw, bytes);
        return PHIP_OK;
    } catch (const std::exception &e) {
        return setErr(PHIP_ERR_DEVICE, e.what());
    }
}

int phip_get_samples(phip_scene *scene, float *out_rgba, size_t n_samples) {
    if (!scene || !out_rgba) return setErr(PHIP_ERR_INVALID, "NULL argument");
    SceneDev &sd = *scene->devs[0];
    if (!sd.haveSamples) return setErr(PHIP_ERR_INVALID, "last render did not set PHIP_FLAG_SAMPLE_BUFFER");
    const size_t n = (size_t) sd.dev.film.width * sd.dev.film.height * sd.lastSpp;
    if (n_samples != n) return setErr(PHIP_ERR_INVALID, "n_samples does not match crop_w*crop_h*spp");
    (void) hipSetDevice(sd.device);
    hipError_t e = hipMemcpy(out_rgba, sd.sampleOut.p, n * sizeof(float4), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return setErr(PHIP_ERR_DEVICE, hipGetErrorString(e));
    return PHIP_OK;
}

int phip_trace(phip_scene *scene, const phip_ray *rays, size_t n, phip_hit *hits, uint8_t *occluded, phip_stats *out_stats) {
    if (!scene || (!rays && n)) return setErr(PHIP_ERR_INVALID, "NULL argument");
    try {
        SceneDev &sd = *scene->devs[0];
        HIP_TRY(hipSetDevice(sd.device));
        std::lock_guard<std::mutex> lock(scene->renderLock);
        if (out_stats) memset(out_stats, 0, sizeof(*out_stats));
        /* bounded chunks: the per-lane spill region alone is SPILL_DEPTH * 4 = 384 B per ray */
        const size_t CHUNK = 4u << 20;
        DevBuf<phip_ray> dr; DevBuf<phip_hit> dh; DevBuf<uint8_t> dz; DevBuf<unsigned long long> stat; DevBuf<uint32_t> spill;
        const size_t cap = std::min(n, CHUNK);
        if (n) {
            dr.alloc(cap); if (hits) dh.alloc(cap); if (occluded) dz.alloc(cap);
            spill.alloc(((cap + BLOCK - 1) / BLOCK * BLOCK) * (size_t) SPILL_DEPTH);
        }
        EventList ev;
        for (size_t first = 0; first < n; first += CHUNK) {
            const size_t m = std::min(CHUNK, n - first);
            HIP_TRY(hipMemcpy(dr.p, rays + first, m * sizeof(phip_ray), hipMemcpyHostToDevice));
            PathPool P; memset(&P, 0, sizeof(P));
            P.nWaves = (uint32_t) ((m + 63) / 64 + BLOCK / 64);
            stat.alloc((size_t) ST_COUNT * P.nWaves);
            HIP_TRY(hipMemset(stat.p, 0, stat.n * sizeof(unsigned long long)));
            P.stat = stat.p; P.spill = spill.p; P.spillLanes = (uint32_t) (spill.n / SPILL_DEPTH);
            ev.record(0);
            if (scene->wide) hipLaunchKernelGGL(k_raycast_w, dim3((unsigned) ((m + BLOCK - 1) / BLOCK)), dim3(BLOCK), traversalLdsBytes(sd.dev), 0, sd.dev, (const phip_ray *) dr.p, m, dh.p, dz.p, P);
#if PHIP_EXPERIMENTS
            else hipLaunchKernelGGL(k_raycast, dim3((unsigned) ((m + BLOCK - 1) / BLOCK)), dim3(BLOCK), traversalLdsBytes(sd.dev), 0, sd.dev, (const phip_ray *) dr.p, m, dh.p, dz.p, P);
#else
            else throw std::runtime_error("internal error: no ray kernel for this scene (the wide tree is missing)");
#endif
            ev.record(0);
            HIP_TRY(hipMemsetAsync(sd.counters.p, 0, sizeof(Counters), 0));
            hipLaunchKernelGGL(k_reduce_stats, dim3(ST_COUNT, REDUCE_SPLIT), dim3(256), 0, 0, P, sd.counters.p, 0);
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipGetLastError());
            if (hits) HIP_TRY(hipMemcpy(hits + first, dh.p, m * sizeof(phip_hit), hipMemcpyDeviceToHost));
            if (occluded) HIP_TRY(hipMemcpy(occluded + first, dz.p, m, hipMemcpyDeviceToHost));
            if (out_stats) {
                Counters hc; HIP_TRY(hipMemcpy(&hc, sd.counters.p, sizeof(hc), hipMemcpyDeviceToHost));
                out_stats->closest_node_visits += hc.total[ST_NODE]; out_stats->closest_triangle_tests += hc.total[ST_TRI];
                out_stats->shadow_node_visits += hc.total[ST_SH_NODE]; out_stats->shadow_triangle_tests += hc.total[ST_SH_TRI];
            }
        }
        if (out_stats) {
            out_stats->closest_rays = hits ? n : 0; out_stats->shadow_rays = occluded ? n : 0;
            out_stats->trace_kernel_ms = ev.sumPairs(); out_stats->iterations = (uint32_t) ((n + CHUNK - 1) / CHUNK);
            out_stats->n_devices = 1;
            algorithmicBytes(false, *out_stats, 0.0, scene->wide);
        }
        return PHIP_OK;
    } catch (const std::exception &e) {
        return setErr(PHIP_ERR_DEVICE, e.what());
    }
}

void phip_cancel(phip_scene *scene) { if (scene && scene->cancelFlag) __atomic_store_n(scene->cancelFlag, 1, __ATOMIC_RELAXED); }

void phip_develop(const float *rgbaw, size_t n_pixels, float *out_rgb) {
    /* fmtconv.cpp:979-991: divide by the weight channel, 0 if the weight is 0 */
    for (size_t i = 0; i < n_pixels; ++i) {
        const float w = rgbaw[5 * i + 4];
        const float inv = w != 0 ? 1.0f / w : 0.0f;
        for (int k = 0; k < 3; ++k) out_rgb[3 * i + k] = rgbaw[5 * i + k] * inv;
    }
}

int phip_scene_accel_info(const phip_scene *scene, phip_accel_info *out) {
    if (!scene || !out) return setErr(PHIP_ERR_INVALID, "NULL argument");
    out->n_nodes = scene->bvh.nNodes; out->n_leaves = scene->bvh.nLeaves; out->n_triangle_refs = scene->bvh.nTriRefs;
    out->max_depth = scene->bvh.maxDepth; out->node_bytes = 128; out->triangle_bytes = 48;
    out->sah_cost = scene->bvh.sahCost; out->build_ms = scene->bvh.buildMs;
    if (scene->wideOnly) { out->n_nodes = scene->bvh.nWNodes; out->max_depth = scene->bvh.wMaxDepth; out->node_bytes = 80; out->sah_cost = scene->bvh.wSahCost; }
    out->fits_lds = scene->fitsLds ? 1u : 0u; out->fused_traversal = scene->fitsLds ? scene->devs[0]->dev.flatMode : (uint32_t) scene->fusedWide;
    return PHIP_OK;
}

/* ---- two utilities of the boundary's callers (declared in include/phip.h) ---- */
/* rfilter.cpp:38-57 + gaussian.cpp:34-57: (radius, table[32]) of `gaussian` with the given stddev */
void phip_gaussian_filter(float stddev, float *radius, float *table32) {
    const float r = 4 * stddev;
    float sum = 0.0f;
    const float alpha = -1.0f / (2.0f * stddev * stddev);
    for (size_t i = 0; i < PHIP_FILTER_RESOLUTION; ++i) {
        float x = (r * i) / PHIP_FILTER_RESOLUTION;
        float value = smax(0.0f, pm_expf(alpha * x * x) - pm_expf(alpha * r * r));
        table32[i] = value;
        sum += value;
    }
    table32[PHIP_FILTER_RESOLUTION] = 0.0f;
    sum *= 2 * r / PHIP_FILTER_RESOLUTION;
    const float normalization = 1.0f / sum;
    for (size_t i = 0; i < PHIP_FILTER_RESOLUTION; ++i)
        table32[i] *= normalization;
    *radius = r;
}

size_t phip_abi_sizeof(int which) {
    switch (which) {
        case 0: return sizeof(phip_material);
        case 1: return sizeof(phip_shape);
        case 2: return sizeof(phip_emitter);
        case 3: return sizeof(phip_camera);
        case 4: return sizeof(phip_film);
        case 5: return sizeof(phip_scene_desc);
        case 6: return sizeof(phip_render_params);
        case 7: return sizeof(phip_stats);
        case 8: return sizeof(phip_ray);
        case 9: return sizeof(phip_hit);
        case 10: return sizeof(phip_accel_info);
        default: return 0;
    }
}

} // extern "C"

/* the test hooks (host twins of device functions, the fmath probe kernel, the HBM calibration kernels) are NOT part of the product: libphip.so is built without them, and the
   same sources with -DPHIP_DEBUG_HOOKS=1 give mitsuba_amd/_build/libphip_debug.so, which only tests/ and tools/ load (mitsuba_amd/_ffi.py: debug_lib; round 6, VERDICT r5 hygiene) */
#if PHIP_DEBUG_HOOKS
#include "phip_debug.inl"
#endif
