/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's `path` integrator hot path (see the o_*.h headers for the
 * file:line map).  Built by oracle/Makefile into oracle/_build/liboracle.so and loaded through
 * ctypes by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg ONLY.  It is the
 * parity checker and the timed CPU baseline ("port"); the product (libphip.so) never links,
 * loads or calls it.
 *
 * Parity status: PINNED TO THE REFERENCE ITSELF.  oracle/Makefile.ref compiles the reference's own libcore + librender +
 * the plugins on the path in place (oracle/_ref; Boost and Eigen are header-shimmed by oracle/ref_shims, the XML loader
 * and the image codecs are left out), oracle/ref_driver.cpp assembles a reference Scene from the same phip_scene_desc, and
 * the libm build of this restatement (make libm: libm transcendentals, math::fastexp/fastlog semantics), run on the
 * reference's own sampler stream (`independent`: SFMT19937, one clone), reproduces the reference's per-sample Li and its
 * ImageBlock accumulator BIT FOR BIT on every scene class of the path -- MIPathTracer and MIDirectIntegrator, all four
 * BSDFs, area / constant / envmap emitters, bitmap textures, EWA-filtered lookups (tests/test_ref_pin.py live,
 * tests/test_golden.py through the committed fixture tests/golden/ref_renders.npz, function-level hooks for
 * Scene::rayIntersect, BSDF::sample/eval/pdf, sampleEmitterDirect, the camera).  The default (parity) build differs from
 * that only in the transcendentals (include/phip_fmath.h, shared with the GPU): <= a few ulp per sample, ~1e-7 rel. L2.
 * Additionally the reference's own test vectors: SFMT19937 seed-4321 golden words (src/tests/test_random.cpp:433-473), the
 * five Triangle::getClippedAABB known answers (src/tests/test_kd.cpp:34-84) and the chi-square sample/pdf/eval contracts
 * of src/tests/test_chisquare.cpp and test_microfacet.cpp.
 */
#include "o_render.h"
#include <string>

using namespace orc;

static thread_local std::string g_err;

extern "C" {

const char *oracle_last_error(void) { return g_err.c_str(); }

void *oracle_scene_create(const phip_scene_desc *desc) {
    try {
        Scene *s = new Scene();
        s->load(*desc);
        return s;
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}

void oracle_scene_destroy(void *scene) { delete static_cast<Scene *>(scene); }
/* test hook: ray queries by a sweep over every triangle instead of the kd-tree (o_kdtree.h: bruteForce) */
void oracle_scene_set_bruteforce(void *scene, int on) { static_cast<Scene *>(scene)->kdtree.bruteForce = on != 0; }

/* sampler_mode: 0 = ctr parity stream, 1 = per-worker SFMT19937 streams like `independent` */
int oracle_render_masks(void *scene_, const phip_render_params *p, int threads, int sampler_mode,
                        float *out_rgbaw, float *out_samples_rgba, uint32_t *out_smooth_masks, phip_stats *stats);
/* PHIP_SAMPLER_SOBOL / _STRATIFIED: the sampler's parameters out of phip_render_params (the tables stay where the caller has them) */
static bool isQmc(uint32_t s) { return s == PHIP_SAMPLER_SOBOL || s == PHIP_SAMPLER_STRATIFIED || s == PHIP_SAMPLER_HALTON || s == PHIP_SAMPLER_HAMMERSLEY; }
static void setQmc(const Scene &scene, const phip_render_params *p, int &qmc, SobolTables &sob, uint32_t &stRes, RinvTables &rinv) {
    if (p->sampler == PHIP_SAMPLER_HALTON || p->sampler == PHIP_SAMPLER_HAMMERSLEY) {
        if (!p->qmc_primes || p->qmc_dimensions < 8) throw std::runtime_error("PHIP_SAMPLER_HALTON / _HAMMERSLEY: tables missing");
        if (p->rr_depth < 2) throw std::runtime_error("PHIP_SAMPLER_HALTON / _HAMMERSLEY: rrDepth >= 2");
        qmc = 3; rinv.primes = p->qmc_primes; rinv.perm = p->qmc_permutations; rinv.dims = p->qmc_dimensions;
        rinv.permOffset.assign(p->qmc_dimensions, 0);
        for (uint32_t d = 1; d < p->qmc_dimensions; ++d) rinv.permOffset[d] = rinv.permOffset[d - 1] + p->qmc_primes[d - 1];
        rinv.hammersley = p->sampler == PHIP_SAMPLER_HAMMERSLEY;
        rinv.setFilmResolution(scene.film.crop_width, scene.film.crop_height, (size_t) (p->sample_total > 0 ? p->sample_total : p->spp));
    } else if (p->sampler == PHIP_SAMPLER_SOBOL) {
        if (!p->sobol_matrices || (p->sobol_log_resolution > 1 && (!p->sobol_vdc || !p->sobol_vdc_inv))) throw std::runtime_error("PHIP_SAMPLER_SOBOL: tables missing");
        if (p->rr_depth < 2) throw std::runtime_error("PHIP_SAMPLER_SOBOL: rrDepth >= 2");
        qmc = 1; sob.matrices = p->sobol_matrices; sob.vdc = p->sobol_vdc; sob.vdcInv = p->sobol_vdc_inv; sob.dims = p->sobol_dimensions;
        sob.logRes = p->sobol_log_resolution; sob.scramble = p->sobol_scramble; sob.resolution = (float) (1u << p->sobol_log_resolution);
    } else {
        const unsigned n = (unsigned) (p->sample_total > 0 ? p->sample_total : p->spp);
        unsigned r = 1; while (r * r < n) ++r;
        if (r * r != n) throw std::runtime_error("PHIP_SAMPLER_STRATIFIED: perfect-square sample count");
        qmc = 2; stRes = r;
    }
}

int oracle_render(void *scene_, const phip_render_params *p, int threads, int sampler_mode,
                  float *out_rgbaw, float *out_samples_rgba, phip_stats *stats) {
    return oracle_render_masks(scene_, p, threads, sampler_mode, out_rgbaw, out_samples_rgba, nullptr, stats);
}
int oracle_render_masks(void *scene_, const phip_render_params *p, int threads, int sampler_mode,
                        float *out_rgbaw, float *out_samples_rgba, uint32_t *out_smooth_masks, phip_stats *stats) {
    try {
        const Scene &scene = *static_cast<Scene *>(scene_);
        RenderParams rp;
        rp.spp = p->spp; rp.blockSize = p->block_size > 0 ? p->block_size : 32; rp.threads = threads;
        rp.ip.maxDepth = p->max_depth; rp.ip.rrDepth = p->rr_depth;
        rp.ip.strictNormals = p->strict_normals != 0; rp.ip.hideEmitters = p->hide_emitters != 0;
        rp.ctr = sampler_mode == 0; rp.seed = p->seed;
        rp.shardIndex = p->shard_index; rp.shardCount = p->shard_count > 0 ? p->shard_count : 1;
        rp.sampleOffset = p->sample_offset; rp.sampleTotal = p->sample_total;
        rp.ld = p->sampler == PHIP_SAMPLER_LD;
        if (rp.ld) {
            const unsigned n = (unsigned) (p->sample_total > 0 ? p->sample_total : p->spp);
            if (sampler_mode != 0 || n == 0 || (n & (n - 1)))
                throw std::runtime_error("PHIP_SAMPLER_LD: on the counter stream, power-of-two sample count");
        }
        if (isQmc(p->sampler)) {
            if (sampler_mode != 0) throw std::runtime_error("the QMC samplers: counter-stream mode");
            setQmc(scene, p, rp.qmc, rp.sobol, rp.stRes, rp.rinv);
        }
        rp.direct = p->integrator == PHIP_INTEGRATOR_DIRECT;
        rp.volpath = p->integrator == PHIP_INTEGRATOR_VOLPATH_SIMPLE;
        if (p->integrator > PHIP_INTEGRATOR_VOLPATH_SIMPLE) throw std::runtime_error("unknown integrator");
        if (rp.direct) {
            if (p->emitter_samples < 0 || p->bsdf_samples < 0) throw std::runtime_error("direct: negative sample count");
            rp.dp.emitterSamples = (size_t) p->emitter_samples; rp.dp.bsdfSamples = (size_t) p->bsdf_samples;
            rp.dp.strictNormals = rp.ip.strictNormals; rp.dp.hideEmitters = rp.ip.hideEmitters;
            rp.dp.configure();
            rp.ip.rrDepth = 5; rp.ip.maxDepth = -1;          /* unused by `direct` */
        }
        /* integrator.cpp:219-224 */
        if (rp.ip.rrDepth <= 0) throw std::runtime_error("'rrDepth' must be set to a value greater than zero!");
        if (rp.ip.maxDepth <= 0 && rp.ip.maxDepth != -1) throw std::runtime_error("'maxDepth' must be set to -1 (infinite) or a value greater than zero!");
        if (scene.envmap.valid() && scene.envmap.nLevels <= 1 && !p->hide_emitters && !(p->flags & PHIP_FLAG_ENVMAP_BILINEAR_BACKGROUND))
            throw std::runtime_error("envmap without MIP levels: directly visible background needs the filtered (EWA) lookup: "
                                     "pass the pyramid, render with hideEmitters or set PHIP_FLAG_ENVMAP_BILINEAR_BACKGROUND");
        RenderResult rr = render(scene, rp, out_rgbaw, out_samples_rgba, out_smooth_masks);
        if (stats) {
            memset(stats, 0, sizeof(*stats));
            stats->samples = rr.counters.samples;
            stats->closest_rays = rr.counters.closestRays;
            stats->shadow_rays = rr.counters.shadowRays;
            stats->path_vertices = rr.counters.pathVertices;
            stats->closest_node_visits = rr.counters.closest.nodeVisits; stats->shadow_node_visits = rr.counters.shadow.nodeVisits;
            stats->closest_triangle_tests = rr.counters.closest.triTests; stats->shadow_triangle_tests = rr.counters.shadow.triTests;
            stats->invalid_samples = rr.counters.invalidSamples;
            stats->render_ms = rr.seconds * 1e3;
            /* 8 B kd node, 4 B index + 48 B TriAccel, ray/hit/state/film terms */
            stats->algorithmic_bytes =
                8.0 * (double) (stats->closest_node_visits + stats->shadow_node_visits) +
                52.0 * (double) (stats->closest_triangle_tests + stats->shadow_triangle_tests) +
                (64.0 + 40.0 + 108.0) * (double) stats->closest_rays + (64.0 + 4.0) * (double) stats->shadow_rays +
                104.0 * (double) stats->path_vertices + 20.0 * (double) scene.film.crop_width * scene.film.crop_height;
            stats->trace_kernel_bytes = 8.0 * (double) stats->closest_node_visits + 52.0 * (double) stats->closest_triangle_tests +
                (64.0 + 40.0) * (double) stats->closest_rays;
        }
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

int oracle_trace(void *scene_, const phip_ray *rays, size_t n, phip_hit *hits, uint8_t *occluded, phip_stats *stats) {
    const Scene &scene = *static_cast<Scene *>(scene_);
    TraversalCounters c, cs;
    for (size_t i = 0; i < n; ++i) {
        Ray r(Vec3(rays[i].o[0], rays[i].o[1], rays[i].o[2]), Vec3(rays[i].d[0], rays[i].d[1], rays[i].d[2]), rays[i].mint, rays[i].maxt);
        if (hits) {
            Float t, u, v; uint32_t prim;
            if (scene.kdtree.rayIntersect(r, t, u, v, prim, &c)) { hits[i].t = t; hits[i].u = u; hits[i].v = v; hits[i].prim = prim; }
            else { hits[i].t = std::numeric_limits<float>::infinity(); hits[i].u = hits[i].v = 0; hits[i].prim = PHIP_NO_HIT; }
        }
        if (occluded) occluded[i] = scene.kdtree.rayIntersectShadow(r, &cs) ? 1 : 0;
    }
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->closest_rays = hits ? n : 0; stats->shadow_rays = occluded ? n : 0;
        stats->closest_node_visits = c.nodeVisits; stats->shadow_node_visits = cs.nodeVisits;
        stats->closest_triangle_tests = c.triTests; stats->shadow_triangle_tests = cs.triTests;
    }
    return 0;
}

/* one sample of the path tracer with the keys of the full frame (debugging aid of tools/fullsize_sample_diff.py) */
int oracle_path_sample(void *scene_, const phip_render_params *p, int px, int py, int k, float *out4, int verbose) {
    try {
        const Scene &scene = *static_cast<Scene *>(scene_);
        const phip_film &f = scene.film;
        PerspectiveCamera cam; cam.configure(scene.camera, f);
        IntegratorParams ip; ip.maxDepth = p->max_depth; ip.rrDepth = p->rr_depth; ip.strictNormals = p->strict_normals != 0; ip.hideEmitters = p->hide_emitters != 0;
        SampleSource smp; smp.ctr = true; smp.seed = p->seed; smp.rng = nullptr;
        smp.pixel = (uint32_t) (py * f.crop_width + px); smp.sample = (uint32_t) k;
        smp.ld = p->sampler == PHIP_SAMPLER_LD; smp.ldMask = (uint32_t) (p->sample_total > 0 ? p->sample_total : p->spp) - 1u; smp.rrDepth = p->rr_depth;
        SobolTables sob; RinvTables rinv;
        if (isQmc(p->sampler)) { setQmc(scene, p, smp.qmc, sob, smp.stRes, rinv); smp.sobol = &sob; smp.rinv = &rinv; smp.px = (uint32_t) px; smp.py = (uint32_t) py; }
        const Float diffScaleFactor = 1.0f / std::sqrt((Float) (p->sample_total > 0 ? p->sample_total : p->spp));
        Vec2 jit = smp.cameraSample();
        Vec2 samplePos((Float) px + jit.x, (Float) py + jit.y);
        Vec3 rx, ry;
        Ray ray = cam.sampleRay(samplePos, &rx, &ry);
        rx = ray.d + (rx - ray.d) * diffScaleFactor; ry = ray.d + (ry - ray.d) * diffScaleFactor;
        PathCounters pc; pc.verbose = verbose != 0;
        Float alpha;
        Spectrum spec = pathLi(scene, ip, ray, smp, alpha, &pc, &rx, &ry);
        out4[0] = spec[0]; out4[1] = spec[1]; out4[2] = spec[2]; out4[3] = alpha;
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

/* brute-force closest hit over every TriAccel: the structure-independent answer */
int oracle_trace_bruteforce(void *scene_, const phip_ray *rays, size_t n, phip_hit *hits) {
    const Scene &scene = *static_cast<Scene *>(scene_);
    const KDTree &kd = scene.kdtree;
    for (size_t i = 0; i < n; ++i) {
        Ray r(Vec3(rays[i].o[0], rays[i].o[1], rays[i].o[2]), Vec3(rays[i].d[0], rays[i].d[1], rays[i].d[2]), rays[i].mint, rays[i].maxt);
        Float mint = r.mint, maxt = r.maxt;
        hits[i].t = std::numeric_limits<float>::infinity(); hits[i].u = hits[i].v = 0; hits[i].prim = PHIP_NO_HIT;
        for (uint32_t p = 0; p < kd.primCount; ++p) {
            Float u, v, t;
            if (kd.triAccel[p].rayIntersect(r, mint, maxt, u, v, t)) { maxt = t; hits[i].t = t; hits[i].u = u; hits[i].v = v; hits[i].prim = p; }
        }
    }
    return 0;
}

struct oracle_kd_info {
    uint32_t n_nodes, n_indices, max_depth, retracted;
    double exp_traversal_steps, exp_leaves_visited, exp_prims_intersected, sah_cost;
    float aabb_min[3], aabb_max[3];
};
void oracle_kd_info_get(void *scene_, oracle_kd_info *o) {
    const KDTree &kd = static_cast<Scene *>(scene_)->kdtree;
    o->n_nodes = (uint32_t) kd.nodes.size(); o->n_indices = (uint32_t) kd.indices.size();
    o->max_depth = kd.builtDepth; o->retracted = kd.retractedSplits;
    o->exp_traversal_steps = kd.expTraversalSteps; o->exp_leaves_visited = kd.expLeavesVisited;
    o->exp_prims_intersected = kd.expPrimitivesIntersected; o->sah_cost = kd.sahCost;
    for (int i = 0; i < 3; ++i) { o->aabb_min[i] = kd.aabb.min[i]; o->aabb_max[i] = kd.aabb.max[i]; }
}

void oracle_gaussian_filter(float stddev, float *radius, float *table32) { gaussianFilterTable(stddev, *radius, table32); }

/* camera ray for a crop-window sample position (tests) */
void oracle_camera_ray(void *scene_, float sx, float sy, phip_ray *out) {
    const Scene &scene = *static_cast<Scene *>(scene_);
    PerspectiveCamera cam; cam.configure(scene.camera, scene.film);
    Ray r = cam.sampleRay(Vec2(sx, sy));
    for (int i = 0; i < 3; ++i) { out->o[i] = r.o[i]; out->d[i] = r.d[i]; }
    out->mint = r.mint; out->maxt = r.maxt;
}

/* ---- known-answer hooks ---- */
void oracle_sfmt_words(uint64_t seed, size_t n, uint64_t *out) {
    SFMT r(seed);
    for (size_t i = 0; i < n; ++i) out[i] = r.nextULong();
}
void oracle_sfmt_floats(uint64_t seed, int clone, size_t n, float *out) {
    SFMT parent(seed);
    if (clone) { SFMT child; child.seedFrom(parent); for (size_t i = 0; i < n; ++i) out[i] = child.nextFloat(); }
    else for (size_t i = 0; i < n; ++i) out[i] = parent.nextFloat();
}
void oracle_ld_point(uint32_t pixel, uint32_t sample, uint32_t dim, uint32_t seed, uint32_t mask, float *out2) {
    SampleSource s; s.pixel = pixel; s.sample = sample; s.seed = seed; s.ld = true; s.ldMask = mask;
    const Vec2 p = s.ldPoint(dim); out2[0] = p.x; out2[1] = p.y;
}
/* dimension `dim` of point `index` of the Halton / Hammersley sequence as RinvTables::sample draws it (halton.cpp:343-350, hammersley.cpp:235-243);
   perm = NULL: no scrambling.  Pinned on the reference's own known answers (src/tests/test_samplers.cpp:33-77) by tests/test_golden.py. */
float oracle_rinv_sample(const uint32_t *primes, uint32_t dims, const uint16_t *perm, int hammersley, uint64_t sample_count, uint64_t index, uint32_t dim) {
    RinvTables T; T.primes = primes; T.perm = perm; T.dims = dims; T.hammersley = hammersley != 0;
    size_t off = 0; T.permOffset.resize(dims);
    for (uint32_t d = 0; d < dims; ++d) { T.permOffset[d] = off; off += primes[d]; }
    if (hammersley) T.factor = (float) 1.0f / (float) sample_count;       /* hammersley.cpp:97 before setFilmResolution: m_factor = 1 / sampleCount */
    return T.sample(index, dim);
}
void oracle_ctr_block(uint32_t pixel, uint32_t sample, uint32_t block, uint32_t seed, float *out4) {
    SampleSource s; s.pixel = pixel; s.sample = sample; s.seed = seed; s.block(block, out4);
}
int oracle_clipped_aabb(const float *tri9, const float *box6, float *out6) {
    AABB b(Vec3(box6[0], box6[1], box6[2]), Vec3(box6[3], box6[4], box6[5]));
    AABB r = triangleClippedAABB(Vec3(tri9[0], tri9[1], tri9[2]), Vec3(tri9[3], tri9[4], tri9[5]), Vec3(tri9[6], tri9[7], tri9[8]), b);
    for (int i = 0; i < 3; ++i) { out6[i] = r.min[i]; out6[3 + i] = r.max[i]; }
    return r.isValid() ? 1 : 0;
}

/* ---- BSDF / microfacet hooks for the chi-square contracts ---- */
void oracle_bsdf_sample(void *scene_, uint32_t material, size_t n, const float *wi3, const float *sample2,
                        float *wo3, float *weight3, float *pdf, uint8_t *delta) {
    const Scene &scene = *static_cast<Scene *>(scene_);
    BSDF b(scene);
    for (size_t i = 0; i < n; ++i) {
        BSDFSamplingRecord r; r.wi = Vec3(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]); r.wo = Vec3(0.0f); r.eta = 1; r.sampledDelta = false;
        Float p = 0;
        Spectrum w = b.sample(scene.materials[material], r, p, Vec2(sample2[2 * i], sample2[2 * i + 1]));
        if (w.isZero()) { p = 0; r.wo = Vec3(0.0f); }
        for (int k = 0; k < 3; ++k) { wo3[3 * i + k] = r.wo[k]; weight3[3 * i + k] = w[k]; }
        pdf[i] = p; if (delta) delta[i] = r.sampledDelta;
    }
}
void oracle_bsdf_eval_pdf(void *scene_, uint32_t material, size_t n, const float *wi3, const float *wo3, float *value3, float *pdf) {
    const Scene &scene = *static_cast<Scene *>(scene_);
    BSDF b(scene);
    for (size_t i = 0; i < n; ++i) {
        Vec3 wi(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]), wo(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]);
        Spectrum v = b.eval(scene.materials[material], wi, wo);
        for (int k = 0; k < 3; ++k) value3[3 * i + k] = v[k];
        pdf[i] = b.pdf(scene.materials[material], wi, wo);
    }
}
void oracle_mf_sample(int type, float au, float av, int visible, size_t n, const float *wi3, const float *sample2, float *m3, float *pdf) {
    MicrofacetDistribution d(type, au, av, visible != 0);
    for (size_t i = 0; i < n; ++i) {
        Float p; Vec3 m = d.sample(Vec3(wi3[0], wi3[1], wi3[2]), Vec2(sample2[2 * i], sample2[2 * i + 1]), p);
        m3[3 * i] = m.x; m3[3 * i + 1] = m.y; m3[3 * i + 2] = m.z; pdf[i] = p;
    }
}
void oracle_mf_pdf(int type, float au, float av, int visible, size_t n, const float *wi3, const float *m3, float *pdf, float *D) {
    MicrofacetDistribution d(type, au, av, visible != 0);
    for (size_t i = 0; i < n; ++i) {
        Vec3 m(m3[3 * i], m3[3 * i + 1], m3[3 * i + 2]);
        pdf[i] = d.pdf(Vec3(wi3[0], wi3[1], wi3[2]), m);
        if (D) D[i] = d.eval(m);
    }
}

/* emitter sampling hook: value/pdf/direction of sampleEmitterDirect (no visibility) + pdfEmitterDirect */
void oracle_sample_emitter(void *scene_, const float *ref3, const float *refN3, size_t n, const float *sample2,
                           float *d3, float *dist, float *pdf, float *value3, float *pdf_check) {
    const Scene &scene = *static_cast<Scene *>(scene_);
    for (size_t i = 0; i < n; ++i) {
        DirectSamplingRecord dRec;
        dRec.ref = Vec3(ref3[0], ref3[1], ref3[2]); dRec.refN = Vec3(refN3[0], refN3[1], refN3[2]);
        dRec.emitter = -1; dRec.pdf = 0; dRec.measure = EInvalidMeasure;
        Vec2 sample(sample2[2 * i], sample2[2 * i + 1]);
        Float emPdf;
        size_t index = scene.emitterPDF.sampleReuse(sample.x, emPdf);
        const phip_emitter &em = scene.emitters[index];
        scene.shapeSampleDirect(scene.shapes[em.shape], dRec, sample);
        Spectrum value(0.0f);
        if (dot(dRec.d, dRec.refN) >= 0 && dot(dRec.d, dRec.n) < 0 && dRec.pdf != 0) value = Spectrum(em.radiance) / dRec.pdf; else dRec.pdf = 0;
        if (dRec.pdf != 0) { dRec.emitter = (int) index; dRec.pdf *= emPdf; value /= emPdf; }
        for (int k = 0; k < 3; ++k) { d3[3 * i + k] = dRec.d[k]; value3[3 * i + k] = value[k]; }
        dist[i] = dRec.dist; pdf[i] = dRec.pdf;
        pdf_check[i] = dRec.pdf != 0 ? scene.pdfEmitterDirect(dRec) : 0.0f;
    }
}

/* Emitter::sampleDirect / pdfDirect of the scene's ENVIRONMENT emitter (constant.cpp / envmap.cpp:516-556), solid-angle measure, no visibility test: the pair
   the reference's test_chisquare::test03_EmitterDirect holds against each other (test_chisquare.cpp:575-618).  sample: d3 + pdf; pdf of given directions. */
int oracle_env_sample_direct(void *scene_, const float *ref3, size_t n, const float *sample2, float *d3, float *pdf, float *value3) {
    const Scene &scene = *static_cast<Scene *>(scene_);
    if (scene.envEmitter < 0) return -1;
    const phip_emitter &em = scene.emitters[scene.envEmitter];
    for (size_t i = 0; i < n; ++i) {
        DirectSamplingRecord dRec;
        dRec.ref = Vec3(ref3[0], ref3[1], ref3[2]); dRec.refN = Vec3(0.0f, 0.0f, 0.0f);
        dRec.emitter = scene.envEmitter; dRec.pdf = 0; dRec.measure = EInvalidMeasure;
        const Vec2 sample(sample2[2 * i], sample2[2 * i + 1]);
        const Spectrum value = em.type == PHIP_EMITTER_ENVMAP ? scene.envmapSampleDirect(dRec, sample) : scene.constantSampleDirect(em, dRec, sample);
        for (int k = 0; k < 3; ++k) { d3[3 * i + k] = dRec.pdf != 0 ? dRec.d[k] : 0.0f; value3[3 * i + k] = value[k]; }
        pdf[i] = dRec.pdf;
    }
    return 0;
}
int oracle_env_pdf_direct(void *scene_, const float *ref3, size_t n, const float *d3, float *pdf) {
    const Scene &scene = *static_cast<Scene *>(scene_);
    if (scene.envEmitter < 0) return -1;
    const phip_emitter &em = scene.emitters[scene.envEmitter];
    for (size_t i = 0; i < n; ++i) {
        DirectSamplingRecord dRec;
        dRec.ref = Vec3(ref3[0], ref3[1], ref3[2]); dRec.refN = Vec3(0.0f, 0.0f, 0.0f);
        dRec.d = Vec3(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]); dRec.emitter = scene.envEmitter; dRec.measure = ESolidAngle; dRec.dist = 1.0f; dRec.n = -dRec.d;
        pdf[i] = em.type == PHIP_EMITTER_ENVMAP ? scene.envmapPdfDirect(dRec) : scene.constantPdfDirect(dRec);
    }
    return 0;
}

/* MipMap::eval on explicit inputs (mipmap.h:629-728) */
int oracle_mip_eval(const phip_texture *t, size_t n, const float *uv2, const float *d0, const float *d1, float *out3) {
    try {
        MipMap m;
        m.load(t->width, t->height, t->n_levels, t->levels);
        m.bcu = t->wrap_u; m.bcv = t->wrap_v; m.filterType = t->filter_type; m.maxAnisotropy = t->max_anisotropy;
        for (size_t i = 0; i < n; ++i) {
            Spectrum v = m.eval(Vec2(uv2[2 * i], uv2[2 * i + 1]), Vec2(d0[2 * i], d0[2 * i + 1]), Vec2(d1[2 * i], d1[2 * i + 1]));
            for (int k = 0; k < 3; ++k) out3[3 * i + k] = v[k];
        }
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

/* ---- phip_fmath.h spot checks: op 0 sin,1 cos,2 exp,3 log,4 acos,5 atan2,6 tan,7 pow,8 erf,9 erfinv,10 atan ---- */
void oracle_fmath(int op, size_t n, const float *a, const float *b, float *out) {
    for (size_t i = 0; i < n; ++i) {
        float s, c;
        switch (op) {
            case 0: pm_sincosf(a[i], &s, &c); out[i] = s; break;
            case 1: pm_sincosf(a[i], &s, &c); out[i] = c; break;
            case 2: out[i] = pm_expf(a[i]); break;
            case 3: out[i] = pm_logf(a[i]); break;
            case 4: out[i] = pm_acosf(a[i]); break;
            case 5: out[i] = pm_atan2f(a[i], b[i]); break;
            case 6: out[i] = pm_tanf(a[i]); break;
            case 7: out[i] = pm_powf(a[i], b[i]); break;
            case 8: out[i] = mts_erf(a[i]); break;
            case 9: out[i] = mts_erfinv(a[i]); break;
            case 10: out[i] = pm_atanf(a[i]); break;
            default: out[i] = 0;
        }
    }
}

int oracle_uses_libm(void) {
#if defined(ORACLE_LIBM)
    return 1;
#else
    return 0;
#endif
}

} // extern "C"
