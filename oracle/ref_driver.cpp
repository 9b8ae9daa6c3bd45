/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * ref_driver.cpp: a C-ABI driver around the REFERENCE ITSELF (oracle/_ref/libmtsref.so = the reference's libcore +
 * librender + plugins compiled from /root/reference by oracle/Makefile.ref).  It assembles a reference `Scene` from a
 * phip_scene_desc the way the XML loader would (PluginManager::createObject + addChild + configure,
 * src/librender/scenehandler.cpp) and exposes what the oracle restates, so that tests can pin the restatement to the real code:
 *     ref_render            SamplingIntegrator::renderBlock semantics (integrator.cpp:140-188) around the reference's
 *                           MIPathTracer::Li / MIDirectIntegrator::Li, one sampler clone, row-major pixel order
 *     ref_render_job        the reference's complete multi-threaded render (RenderJob -> BlockedRenderProcess on the
 *                           Scheduler's LocalWorkers): the CPU baseline of bench.py ("kind": "reference")
 *     ref_trace             Scene::rayIntersect
 *     ref_bsdf_*            BSDF::sample / eval / pdf of the scene's materials
 *     ref_sample_emitter    Scene::sampleEmitterDirect (no visibility test) + pdfEmitterDirect
 *     ref_camera_ray        Sensor::sampleRayDifferential
 *     ref_mip_*             TMIPMap (render/mipmap.h): the MIP pyramid exactly as envmap.cpp / bitmap.cpp build it, MIPMap::eval
 * Nothing here is on the product path; nothing of it exists on the GPU box unless oracle/_ref was built beforehand.
 */
#include <mitsuba/render/scene.h>
#include <mitsuba/render/renderproc.h>
#include <mitsuba/core/plugin.h>
#include <mitsuba/core/statistics.h>
#include <mitsuba/core/fstream.h>
#include <mitsuba/core/qmc.h>
#include "faure.h"                      /* src/samplers/faure.h (plugin-local): PermutationStorage */
#include <mitsuba/core/bitmap.h>
#include <mitsuba/core/sched.h>
#include <mitsuba/core/appender.h>
#include <mitsuba/render/trimesh.h>
#include <mitsuba/render/mipmap.h>
#include <mitsuba/render/renderjob.h>
#include <mitsuba/render/renderqueue.h>
#include <chrono>
#include <atomic>
#include <mutex>
#include "phip.h"
#include <execinfo.h>
#include <signal.h>

using namespace mitsuba;

namespace {

std::string g_err;
bool g_init = false;
int g_samplerKind = 0;      /* 0 = the reference's `independent`; 1 = oracle/ref_glue/ctr_sampler.cpp (the parity stream); 2 / 3 / 4 / 5 / 6 = the reference's `ldsampler` / `sobol` / `stratified` / `halton` / `hammersley` */
/* shapes the NEXT ref_scene_create loads through one of the reference's own mesh-loader plugins (shapes/obj.cpp): (plugin, file,
   material id of the description, toWorld) -- a real asset enters the scene exactly as the XML loader would add it */
struct FileShape { std::string plugin, filename; uint32_t material; float toWorld[16]; };
std::vector<FileShape> g_fileShapes;
int g_analyticRectangles = 0;   /* build exact rectangles as the reference's analytic `rectangle` shape instead of a two-triangle mesh */

struct RefScene {
    ref<Scene> scene;
    std::vector<ref<BSDF> > materials;     /* by phip material id */
    std::vector<ref<Bitmap> > keep;
    std::vector<ref<Texture> > textures;
    int width, height;
};

/* the scene's sampler for one render: `independent`, or the parity stream (ctr_sampler.cpp: defined by call order, so it needs to know
   nothing about the scene) */
Sampler *makeSampler(const Scene *scene, const phip_render_params *p) {
    Properties smp(g_samplerKind == 1 ? "ctr" : g_samplerKind == 2 ? "ldsampler" : g_samplerKind == 3 ? "sobol" : g_samplerKind == 4 ? "stratified" : g_samplerKind == 5 ? "halton" : g_samplerKind == 6 ? "hammersley" : "independent");
    smp.setSize("sampleCount", (size_t) p->spp);
    if (g_samplerKind == 1) {
        smp.setInteger("seed", (int) p->seed);
        smp.setInteger("cropWidth", scene->getFilm()->getCropSize().x);
        smp.setString("mode", p->integrator == PHIP_INTEGRATOR_DIRECT ? "direct" : "path");
        smp.setSize("emitterSamples", (size_t) std::max(0, p->emitter_samples)); smp.setSize("bsdfSamples", (size_t) std::max(0, p->bsdf_samples));
        smp.setInteger("rrDepth", p->rr_depth);          /* the depth of the first Russian-roulette request (path.cpp:276-283) */
        smp.setBoolean("ld", p->sampler == PHIP_SAMPLER_LD);   /* the ldsampler construction on the counter-based generator (include/phip.h) */
        smp.setSize("sampleTotal", (size_t) (p->sample_total > 0 ? p->sample_total : p->spp));
        smp.setBoolean("stratified", p->sampler == PHIP_SAMPLER_STRATIFIED);   /* the construction of `stratified` on the counter stream (include/phip.h) */
    }
    if ((g_samplerKind == 5 || g_samplerKind == 6) && p->seed) smp.setInteger("scramble", (int) p->seed - 2);   /* seed 1 -> -1 (Faure, the default), 2 -> 0 (none), n -> n - 2 */
    if (g_samplerKind == 3 && p->seed) smp.setSize("scramble", (size_t) p->seed);   /* the plugin's frame number; phip_render_params.sobol_scramble holds what the plugin makes of it (sobol.cpp:92-102) */
    Sampler *s = static_cast<Sampler *>(PluginManager::getInstance()->createObject(MTS_CLASS(Sampler), smp));
    s->configure();
    return s;
}

Spectrum rgb(const float *v) { Spectrum s; s.fromLinearRGB(v[0], v[1], v[2]); return s; }

ConfigurableObject *create(const Class *cls, const Properties &p) {
    return PluginManager::getInstance()->createObject(cls, p);
}

/* what the XML loader does for every child element (scenehandler.cpp:762-773): the child has been configured already */
void attach(ConfigurableObject *parent, const std::string &name, ConfigurableObject *child) {
    parent->addChild(name, child);
    child->setParent(parent);
}

Transform toTransform(const float *m16) {
    Matrix4x4 m;
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) m(r, c) = m16[4 * r + c];
    return Transform(m);
}

ref<Bitmap> rgbBitmap(const float *texels, uint32_t w, uint32_t h) {
    ref<Bitmap> b = new Bitmap(Bitmap::ERGB, Bitmap::EFloat32, Vector2i((int) w, (int) h));
    memcpy(b->getFloat32Data(), texels, (size_t) w * h * 3 * sizeof(float));
    return b;
}

const char *wrapName(uint32_t m) {
    switch (m) { case PHIP_WRAP_CLAMP: return "clamp"; case PHIP_WRAP_MIRROR: return "mirror"; case PHIP_WRAP_ZERO: return "zero"; case PHIP_WRAP_ONE: return "one"; default: return "repeat"; }
}

/* a flat-shaded two-triangle shape whose four vertices form an exact rectangle -> the reference's `rectangle` plugin, else NULL */
Shape *makeRectangle(const phip_scene_desc &d, const phip_shape &s) {
    if (s.n_triangles != 2 || s.n_vertices != 4 || (s.has_normals && d.normals) || (s.has_texcoords && d.texcoords)) return NULL;
    Point p[4];
    for (int i = 0; i < 4; ++i) { const float *q = d.positions + 3 * (size_t) (s.first_vertex + i); p[i] = Point(q[0], q[1], q[2]); }
    const uint32_t *ix = d.indices + 3 * (size_t) s.first_triangle;
    const Vector nTri = cross(p[ix[1] - s.first_vertex] - p[ix[0] - s.first_vertex], p[ix[2] - s.first_vertex] - p[ix[0] - s.first_vertex]);
    Vector u = (p[1] - p[0]) * 0.5f, v = (p[3] - p[0]) * 0.5f;
    const Float scale = std::max(u.length(), v.length());
    if ((p[2] - (p[1] + (p[3] - p[0]))).length() > 1e-5f * scale) return NULL;                   /* not a parallelogram */
    if (std::abs(dot(normalize(u), normalize(v))) > 1e-6f) return NULL;                          /* sheared: rectangle.cpp:91-92 refuses it */
    if (dot(cross(u, v), nTri) < 0) std::swap(u, v);                                             /* the front side of the mesh */
    const Vector n = normalize(cross(u, v));
    const Point c = p[0] + (p[2] - p[0]) * 0.5f;
    Matrix4x4 m(u.x, v.x, n.x, c.x,  u.y, v.y, n.y, c.y,  u.z, v.z, n.z, c.z,  0, 0, 0, 1);
    Properties props("rectangle");
    props.setTransform("toWorld", Transform(m));
    return static_cast<Shape *>(PluginManager::getInstance()->createObject(MTS_CLASS(Shape), props));
}

/* <texture type="bitmap"> from the in-memory image (bitmap.cpp:187-190); the plugin builds its own MIP pyramid */
Texture *makeTexture(RefScene *rs, const phip_scene_desc &d, uint32_t id) {
    const phip_texture &t = d.textures[id];
    ref<Bitmap> bmp = rgbBitmap(t.levels[0], t.width, t.height);
    rs->keep.push_back(bmp);
    Properties tp("bitmap");
    Properties::Data data; data.ptr = (uint8_t *) bmp.get(); data.size = sizeof(Bitmap);
    tp.setData("bitmap", data);
    static const char *filters[] = { "nearest", "bilinear", "trilinear", "ewa" };
    tp.setString("filterType", filters[t.filter_type & 3]);
    tp.setString("wrapModeU", wrapName(t.wrap_u)); tp.setString("wrapModeV", wrapName(t.wrap_v));
    tp.setFloat("maxAnisotropy", t.max_anisotropy);
    tp.setFloat("gamma", 1.0f);
    tp.setFloat("uscale", t.uv_scale[0]); tp.setFloat("vscale", t.uv_scale[1]);
    tp.setFloat("uoffset", t.uv_offset[0]); tp.setFloat("voffset", t.uv_offset[1]);
    Texture *tex = static_cast<Texture *>(create(MTS_CLASS(Texture), tp));
    tex->incRef();
    tex->configure();
    rs->textures.push_back(tex);
    tex->decRef(false);
    return tex;
}

BSDF *makeBSDF(RefScene *rs, const phip_scene_desc &d, uint32_t id) {
    if (rs->materials[id]) return rs->materials[id];
    const phip_material &m = d.materials[id];
    ref<BSDF> bsdf;
    if (m.type == PHIP_BSDF_DIFFUSE) {
        Properties p("diffuse");
        if (!m.reflectance_texture) p.setSpectrum("reflectance", rgb(m.reflectance));
        bsdf = static_cast<BSDF *>(create(MTS_CLASS(BSDF), p));
        if (m.reflectance_texture) attach(bsdf, "reflectance", makeTexture(rs, d, m.reflectance_texture - 1));
    } else if (m.type == PHIP_BSDF_DIELECTRIC) {
        Properties p("dielectric");
        p.setFloat("intIOR", m.eta[0]); p.setFloat("extIOR", 1.0f);
        if (!m.reflectance_texture) p.setSpectrum("specularReflectance", rgb(m.reflectance));
        p.setSpectrum("specularTransmittance", rgb(m.transmittance));
        bsdf = static_cast<BSDF *>(create(MTS_CLASS(BSDF), p));
        if (m.reflectance_texture) attach(bsdf, "specularReflectance", makeTexture(rs, d, m.reflectance_texture - 1));
        if (m.transmittance_texture) attach(bsdf, "specularTransmittance", makeTexture(rs, d, m.transmittance_texture - 1));
    } else if (m.type == PHIP_BSDF_ROUGHCONDUCTOR) {
        Properties p("roughconductor");
        p.setString("material", "none");              /* no data/ior lookup (roughconductor.cpp:176-190): eta and k are given */
        p.setSpectrum("eta", rgb(m.eta)); p.setSpectrum("k", rgb(m.k)); p.setFloat("extEta", 1.0f);
        if (!m.reflectance_texture) p.setSpectrum("specularReflectance", rgb(m.reflectance));
        p.setString("distribution", m.distribution == PHIP_MF_GGX ? "ggx" : "beckmann");
        p.setFloat("alphaU", m.alpha_u); p.setFloat("alphaV", m.alpha_v);
        p.setBoolean("sampleVisible", m.sample_visible != 0);
        bsdf = static_cast<BSDF *>(create(MTS_CLASS(BSDF), p));
        if (m.reflectance_texture) attach(bsdf, "specularReflectance", makeTexture(rs, d, m.reflectance_texture - 1));
        /* roughconductor.cpp:424-431: one texture as "alpha" serves both axes (isotropic); "alphaU" / "alphaV" are separate objects */
        if (m.alpha_u_texture && m.alpha_u_texture == m.alpha_v_texture) attach(bsdf, "alpha", makeTexture(rs, d, m.alpha_u_texture - 1));
        else {
            if (m.alpha_u_texture) attach(bsdf, "alphaU", makeTexture(rs, d, m.alpha_u_texture - 1));
            if (m.alpha_v_texture) attach(bsdf, "alphaV", makeTexture(rs, d, m.alpha_v_texture - 1));
        }
    } else if (m.type == PHIP_BSDF_TWOSIDED) {
        Properties p("twosided");
        bsdf = static_cast<BSDF *>(create(MTS_CLASS(BSDF), p));
        attach(bsdf, "", makeBSDF(rs, d, m.nested[0]));
        if (m.nested[1] != m.nested[0]) attach(bsdf, "", makeBSDF(rs, d, m.nested[1]));
    } else {
        throw std::runtime_error("ref_driver: unknown material type");
    }
    bsdf->configure();
    rs->materials[id] = bsdf;
    return bsdf;
}

} // namespace

extern "C" {

const char *ref_last_error(void) { return g_err.c_str(); }

/* the start-up sequence of src/mitsuba/mitsuba.cpp:422-433 (minus SHVector and the XML SceneHandler) */
static void segvHandler(int sig) {
    void *frames[64];
    int n = backtrace(frames, 64);
    fprintf(stderr, "ref_driver: signal %d, backtrace:\n", sig);
    backtrace_symbols_fd(frames, n, 2);
    _exit(139);
}
/* the last warning / error the reference logged (a RenderJob that fails -- e.g. a plugin shim that refuses a scene -- reports through the log only) and how the job ended */
static std::string g_lastWarning;
static std::mutex g_logLock;
class CaptureAppender : public Appender {
public:
    void append(ELogLevel level, const std::string &text) { if (level >= EWarn) { std::lock_guard<std::mutex> g(g_logLock); g_lastWarning = text; } }
    void logProgress(Float, const std::string &, const std::string &, const std::string &, const void *) { }
};
class JobStatusListener : public RenderListener {
public:
    std::atomic<int> cancelled{0};
    void finishJobEvent(const RenderJob *, bool c) { if (c) cancelled = 1; }
};

int ref_init(void) {
    if (g_init) return 0;
    if (getenv("REF_DRIVER_BACKTRACE")) signal(SIGSEGV, segvHandler);
    try {
        Class::staticInitialization();
        Object::staticInitialization();
        PluginManager::staticInitialization();
        Statistics::staticInitialization();
        Thread::staticInitialization();
        Logger::staticInitialization();
        FileStream::staticInitialization();
        Spectrum::staticInitialization();
        Bitmap::staticInitialization();
        Scheduler::staticInitialization();
        Thread::getThread()->getLogger()->setLogLevel(EWarn);
        if (!getenv("REF_DRIVER_VERBOSE"))
            Thread::getThread()->getLogger()->clearAppenders();      /* no progress bars; Log(EError) still throws */
        Thread::getThread()->getLogger()->addAppender(new CaptureAppender());      /* a RenderJob catches what its integrator throws and only LOGS it (renderjob.cpp:96-120): keep the text */
        g_init = true;
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

/* 0 = `independent` (default), 1 = the counter-based parity stream for the following renders */
void ref_set_sampler(int kind) { g_samplerKind = kind; }
void ref_set_analytic_rectangles(int on) { g_analyticRectangles = on; }
void ref_add_shape_file(const char *plugin, const char *filename, uint32_t material, const float *to_world16) {
    FileShape f; f.plugin = plugin; f.filename = filename; f.material = material;
    for (int i = 0; i < 16; ++i) f.toWorld[i] = to_world16 ? to_world16[i] : (i % 5 == 0 ? 1.0f : 0.0f);
    g_fileShapes.push_back(f);
}

/* the data of the radical-inverse samplers (`halton`, `hammersley`): the first `dims` primes of the reference's table (qmc.cpp:27-81) and, for
   scramble != 0, the digit permutations its own PermutationStorage builds (src/samplers/faure.cpp: -1 = Faure's, any other value = pseudorandom
   ones), concatenated -- the permutation of dimension d starts at the sum of the primes before it.  perm_out: sum(primes[0..dims)) entries. */
int ref_qmc_tables(int scramble, uint32_t dims, uint32_t *primes_out, uint16_t *perm_out) {
    try {
        if (dims > primeTableSize) throw std::runtime_error("ref_qmc_tables: at most 1024 dimensions");
        ref<PermutationStorage> ps = scramble != 0 ? new PermutationStorage(scramble) : NULL;
        size_t off = 0;
        for (uint32_t d = 0; d < dims; ++d) {
            primes_out[d] = (uint32_t) primeTable[d];
            if (ps != NULL && perm_out) memcpy(perm_out + off, ps->getPermutation(d), sizeof(uint16_t) * (size_t) primeTable[d]);
            off += (size_t) primeTable[d];
        }
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

/* writes shape `si` of a scene description as a .serialized mesh file -- by the reference's own TriMesh::serialize (trimesh.cpp:1131-1180: header, zlib
   stream, version 4), so that the reference's `serialized` loader plugin (src/shapes/serialized.cpp) has a file of its own format to read */
int ref_write_serialized(const phip_scene_desc *dp, uint32_t si, const char *path) {
    try {
        const phip_scene_desc &d = *dp;
        if (si >= d.n_shapes) throw std::runtime_error("ref_write_serialized: no such shape");
        const phip_shape &s = d.shapes[si];
        const bool hasN = s.has_normals && d.normals, hasUV = s.has_texcoords && d.texcoords;
        ref<TriMesh> mesh = new TriMesh(formatString("shape%u", si), s.n_triangles, s.n_vertices, hasN, hasUV, false, false, !hasN);
        for (uint32_t v = 0; v < s.n_vertices; ++v) {
            const float *p = d.positions + 3 * (size_t) (s.first_vertex + v);
            mesh->getVertexPositions()[v] = Point(p[0], p[1], p[2]);
            if (hasN) { const float *n = d.normals + 3 * (size_t) (s.first_vertex + v); mesh->getVertexNormals()[v] = Normal(n[0], n[1], n[2]); }
            if (hasUV) { const float *t = d.texcoords + 2 * (size_t) (s.first_vertex + v); mesh->getVertexTexcoords()[v] = Point2(t[0], t[1]); }
        }
        for (uint32_t t = 0; t < s.n_triangles; ++t)
            for (int k = 0; k < 3; ++k)
                mesh->getTriangles()[t].idx[k] = d.indices[3 * (size_t) (s.first_triangle + t) + k] - s.first_vertex;
        ref<FileStream> fsOut = new FileStream(fs::path(path), FileStream::ETruncReadWrite);
        fsOut->setByteOrder(Stream::ELittleEndian);
        mesh->serialize(fsOut);
        fsOut->close();
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

/* stops the Scheduler's worker threads (they would keep the process alive at exit) */
void ref_shutdown(void) {
    if (!g_init) return;
    Scheduler *sched = Scheduler::getInstance();
    if (sched && sched->isRunning()) sched->stop();
}

/* filter: 0 = gaussian (stddev), 1 = box.  integrator: "path" / "direct" parameters arrive with ref_render. */
void *ref_scene_create(const phip_scene_desc *dp, float gaussian_stddev) {
    try {
        if (ref_init() != 0) return NULL;
        const phip_scene_desc &d = *dp;
        RefScene *rs = new RefScene();
        rs->materials.resize(d.n_materials);
        rs->width = d.film.crop_width; rs->height = d.film.crop_height;
        ref<Scene> scene = new Scene(Properties("scene"));

        /* ---- sensor (+ sampler, film, rfilter) ---- */
        Properties sp("perspective");
        sp.setTransform("toWorld", toTransform(d.camera.to_world));
        sp.setFloat("fov", d.camera.xfov_deg); sp.setString("fovAxis", "x");
        sp.setFloat("nearClip", d.camera.near_clip); sp.setFloat("farClip", d.camera.far_clip);
        ref<Sensor> sensor = static_cast<Sensor *>(create(MTS_CLASS(Sensor), sp));
        Properties fp("hdrfilm");
        fp.setInteger("width", d.film.width); fp.setInteger("height", d.film.height);
        fp.setInteger("cropOffsetX", d.film.crop_offset_x); fp.setInteger("cropOffsetY", d.film.crop_offset_y);
        fp.setInteger("cropWidth", d.film.crop_width); fp.setInteger("cropHeight", d.film.crop_height);
        fp.setString("pixelFormat", "rgba");      /* so that the alpha channel is computed (integrator.cpp:160-161) */
        fp.setBoolean("banner", false);
        ref<Film> film = static_cast<Film *>(create(MTS_CLASS(Film), fp));
        Properties rp("gaussian"); rp.setFloat("stddev", gaussian_stddev);
        ref<ReconstructionFilter> rf = static_cast<ReconstructionFilter *>(create(MTS_CLASS(ReconstructionFilter), rp));
        rf->configure();
        attach(film, "", rf); film->configure();
        Properties smp("independent"); smp.setInteger("sampleCount", 1);
        ref<Sampler> sampler = static_cast<Sampler *>(create(MTS_CLASS(Sampler), smp));
        sampler->configure();
        attach(sensor, "", film); attach(sensor, "", sampler); sensor->configure();
        attach(scene, "", sensor);

        /* ---- environment emitters first: Scene::getEmitters() lists them before the shapes' area emitters
                (scene.cpp:496-516 vs 570-571); the description has to be in that order ---- */
        bool areaSeen = false;
        for (uint32_t i = 0; i < d.n_emitters; ++i) {
            const phip_emitter &e = d.emitters[i];
            if (e.type == PHIP_EMITTER_AREA) { areaSeen = true; continue; }
            if (areaSeen) throw std::runtime_error("ref_driver: list environment emitters before area emitters (the order of Scene::getEmitters())");
            ref<Emitter> em;
            if (e.type == PHIP_EMITTER_CONSTANT) {
                Properties p("constant"); p.setSpectrum("radiance", rgb(e.radiance)); p.setFloat("samplingWeight", e.sampling_weight);
                em = static_cast<Emitter *>(create(MTS_CLASS(Emitter), p));
            } else {
                ref<Bitmap> bmp = rgbBitmap(d.envmap.texels, d.envmap.width, d.envmap.height);
                rs->keep.push_back(bmp);
                Properties p("envmap");
                Properties::Data data; data.ptr = (uint8_t *) bmp.get(); data.size = sizeof(Bitmap);
                p.setData("bitmap", data);
                p.setFloat("scale", d.envmap.scale); p.setFloat("gamma", 1.0f);
                p.setFloat("samplingWeight", e.sampling_weight);
                p.setTransform("toWorld", toTransform(d.envmap.to_world));
                em = static_cast<Emitter *>(create(MTS_CLASS(Emitter), p));
            }
            em->configure();
            attach(scene, "", em);
        }

        /* ---- shapes ---- */
        uint32_t nextArea = 0;
        for (uint32_t i = 0; i < d.n_emitters; ++i) if (d.emitters[i].type != PHIP_EMITTER_AREA) ++nextArea;
        for (uint32_t si = 0; si < d.n_shapes; ++si) {
            const phip_shape &s = d.shapes[si];
            const bool hasN = s.has_normals && d.normals, hasUV = s.has_texcoords && d.texcoords;
            ref<Shape> analytic = g_analyticRectangles ? makeRectangle(d, s) : NULL;
            if (analytic) {
                /* an analytic shape of the reference (shapes/rectangle.cpp): its own intersection and sampling routines on the CPU;
                   the plugin shim turns it into a mesh through Shape::createTriMesh (rectangle.cpp:170-203) */
                attach(analytic, "", makeBSDF(rs, d, s.material));
                if (s.emitter >= 0) {
                    if ((uint32_t) s.emitter != nextArea) throw std::runtime_error("ref_driver: area emitters must be listed in shape order (Scene::getEmitters())");
                    ++nextArea;
                    const phip_emitter &e = d.emitters[s.emitter];
                    Properties p("area"); p.setSpectrum("radiance", rgb(e.radiance)); p.setFloat("samplingWeight", e.sampling_weight);
                    ref<Emitter> em = static_cast<Emitter *>(create(MTS_CLASS(Emitter), p));
                    em->configure();
                    attach(analytic, "", em);
                }
                analytic->configure();
                attach(scene, "", analytic);
                continue;
            }
            ref<TriMesh> mesh = new TriMesh(formatString("shape%u", si), s.n_triangles, s.n_vertices, hasN, hasUV, false, false, !hasN);
            for (uint32_t v = 0; v < s.n_vertices; ++v) {
                const float *p = d.positions + 3 * (size_t) (s.first_vertex + v);
                mesh->getVertexPositions()[v] = Point(p[0], p[1], p[2]);
                if (hasN) { const float *n = d.normals + 3 * (size_t) (s.first_vertex + v); mesh->getVertexNormals()[v] = Normal(n[0], n[1], n[2]); }
                if (hasUV) { const float *t = d.texcoords + 2 * (size_t) (s.first_vertex + v); mesh->getVertexTexcoords()[v] = Point2(t[0], t[1]); }
            }
            for (uint32_t t = 0; t < s.n_triangles; ++t)
                for (int k = 0; k < 3; ++k)
                    mesh->getTriangles()[t].idx[k] = d.indices[3 * (size_t) (s.first_triangle + t) + k] - s.first_vertex;
            attach(mesh, "", makeBSDF(rs, d, s.material));
            if (s.emitter >= 0) {
                if ((uint32_t) s.emitter != nextArea) throw std::runtime_error("ref_driver: area emitters must be listed in shape order (Scene::getEmitters())");
                ++nextArea;
                const phip_emitter &e = d.emitters[s.emitter];
                Properties p("area"); p.setSpectrum("radiance", rgb(e.radiance)); p.setFloat("samplingWeight", e.sampling_weight);
                ref<Emitter> em = static_cast<Emitter *>(create(MTS_CLASS(Emitter), p));
                em->configure();
                attach(mesh, "", em);
            }
            mesh->configure();
            attach(scene, "", mesh);
        }
        /* ---- shapes from files, through the reference's loader plugins (after the description's shapes: Scene::getShapes() order) ---- */
        {
            std::vector<FileShape> files; files.swap(g_fileShapes);
            for (const FileShape &f : files) {
                Properties p(f.plugin);
                p.setString("filename", f.filename);
                p.setTransform("toWorld", toTransform(f.toWorld));
                p.setBoolean("faceNormals", true);            /* flat shading: what the harness mesh without vertex normals gets (trimesh.cpp) */
                ref<Shape> shape = static_cast<Shape *>(create(MTS_CLASS(Shape), p));
                attach(shape, "", makeBSDF(rs, d, f.material));
                shape->configure();
                attach(scene, "", shape);
            }
        }
        for (uint32_t i = 0; i < d.n_materials; ++i) makeBSDF(rs, d, i);

        /* a default integrator so that Scene::configure does not create one with its own ideas */
        Properties ip("path");
        ref<Integrator> integ = static_cast<Integrator *>(create(MTS_CLASS(Integrator), ip));
        integ->configure();
        attach(scene, "", integ);
        scene->configure();
        scene->initialize();
        rs->scene = scene;
        return rs;
    } catch (const std::exception &e) {
        g_err = e.what();
        return NULL;
    }
}

void ref_scene_destroy(void *h) { delete static_cast<RefScene *>(h); }

/*
 * Renders the crop window as ONE image block with the reference integrator selected by p->integrator.
 * out_samples: [y][x][spp][4] (R,G,B,alpha) per-sample Li; out_film: [y][x][5] = the block's (R,G,B,alpha,weight)
 * accumulator after ImageBlock::put of every sample (interior of the block bitmap).
 * The loop around Li restates integrator.cpp:140-188 (it has to, to get at the individual samples); when
 * out_samples == NULL the reference's own SamplingIntegrator::renderBlock is called instead.
 * Sampler: `independent`, sampleCount = spp, one clone (renderjob.cpp:58-69) -- the oracle's sampler="sfmt", threads=1.
 */
int ref_render(void *h, const phip_render_params *p, float *out_samples, float *out_film) {
    try {
        RefScene *rs = static_cast<RefScene *>(h);
        Scene *scene = rs->scene;
        Sensor *sensor = scene->getSensor();
        Film *film = sensor->getFilm();
        const Vector2i size = film->getCropSize();
        if (size.x > 255 || size.y > 255) throw std::runtime_error("ref_render: at most 255 x 255 pixels (one image block)");

        Properties ip(p->integrator == PHIP_INTEGRATOR_DIRECT ? "direct" : p->integrator == PHIP_INTEGRATOR_VOLPATH_SIMPLE ? "volpath_simple" : "path");
        if (p->integrator == PHIP_INTEGRATOR_DIRECT) {
            ip.setSize("emitterSamples", (size_t) p->emitter_samples); ip.setSize("bsdfSamples", (size_t) p->bsdf_samples);
        } else {
            ip.setInteger("maxDepth", p->max_depth); ip.setInteger("rrDepth", p->rr_depth);
        }
        ip.setBoolean("strictNormals", p->strict_normals != 0); ip.setBoolean("hideEmitters", p->hide_emitters != 0);
        ref<SamplingIntegrator> integ = static_cast<SamplingIntegrator *>(create(MTS_CLASS(Integrator), ip));
        integ->configure();

        ref<Sampler> parent = makeSampler(scene, p);
        parent->setFilmResolution(film->getCropSize(), true);   /* SamplingIntegrator::preprocess, integrator.cpp:40-41 (`sobol` enumerates the sequence per pixel) */
        integ->configureSampler(scene, parent);          /* requests the sample arrays of `direct` */
        ref<Sampler> sampler = parent->clone();          /* worker 0's sampler */

        ref<ImageBlock> block = new ImageBlock(Bitmap::ESpectrumAlphaWeight, size, film->getReconstructionFilter());
        block->setOffset(Point2i(0, 0));
        std::vector<TPoint2<uint8_t> > points;
        for (int y = 0; y < size.y; ++y) for (int x = 0; x < size.x; ++x) points.push_back(TPoint2<uint8_t>((uint8_t) x, (uint8_t) y));
        bool stop = false;

        if (!out_samples) {
            integ->renderBlock(scene, sensor, sampler, block, stop, points);
        } else {
            const Float diffScaleFactor = 1.0f / std::sqrt((Float) sampler->getSampleCount());
            RadianceQueryRecord rRec(scene, sampler);
            Point2 apertureSample(0.5f); Float timeSample = 0.5f;
            RayDifferential sensorRay;
            block->clear();
            const uint32_t queryType = RadianceQueryRecord::ESensorRay;
            for (size_t i = 0; i < points.size(); ++i) {
                Point2i offset = Point2i(points[i]) + Vector2i(block->getOffset());
                sampler->generate(offset);
                for (size_t j = 0; j < sampler->getSampleCount(); j++) {
                    rRec.newQuery(queryType, sensor->getMedium());
                    Point2 samplePos(Point2(offset) + Vector2(rRec.nextSample2D()));
                    Spectrum spec = sensor->sampleRayDifferential(sensorRay, samplePos, apertureSample, timeSample);
                    sensorRay.scaleDifferential(diffScaleFactor);
                    spec *= integ->Li(sensorRay, rRec);
                    block->put(samplePos, spec, rRec.alpha);
                    sampler->advance();
                    float *o = out_samples + (((size_t) offset.y * size.x + offset.x) * p->spp + j) * 4;
                    Float r, g, b; spec.toLinearRGB(r, g, b);
                    o[0] = r; o[1] = g; o[2] = b; o[3] = rRec.alpha;
                }
            }
        }
        if (out_film) {
            const Bitmap *bmp = block->getBitmap();
            const int border = block->getBorderSize(), bw = bmp->getWidth(), ch = bmp->getChannelCount();
            const Float *src = bmp->getFloatData();
            for (int y = 0; y < size.y; ++y) for (int x = 0; x < size.x; ++x)
                for (int c = 0; c < 5; ++c)
                    out_film[((size_t) y * size.x + x) * 5 + c] = src[((size_t) (y + border) * bw + (x + border)) * ch + c];
        }
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

/*
 * The reference's own render, start to finish, as `mitsuba scene.xml` runs it (src/mitsuba/mitsuba.cpp:330-372):
 * RenderJob (renderjob.cpp) -> Scene::preprocess / render -> SamplingIntegrator::render (integrator.cpp:95-129) ->
 * BlockedRenderProcess (renderproc.cpp: 32x32 blocks in spiral order, Hilbert-curve pixel order, one sampler clone per
 * worker) on `threads` LocalWorkers -> Film::put.  out_rgb (optional): the developed crop window, H x W x 3.
 * The worker count is fixed by the first call.
 */
int ref_render_job_plugin(void *h, const phip_render_params *p, const char *integrator_plugin, int threads, float *out_rgb, double *seconds);
int ref_render_job(void *h, const phip_render_params *p, int threads, float *out_rgb, double *seconds) {
    return ref_render_job_plugin(h, p, NULL, threads, out_rgb, seconds);
}
/* integrator_plugin: NULL = "path" / "direct" by p->integrator, else the plugin to instantiate with the same parameters
   -- "path_hip" / "direct_hip": the product's shims, loaded by the reference's PluginManager like any other plugin */
int ref_render_job_plugin(void *h, const phip_render_params *p, const char *integrator_plugin, int threads, float *out_rgb, double *seconds) {
    try {
        RefScene *rs = static_cast<RefScene *>(h);
        Scene *scene = rs->scene;
        Scheduler *sched = Scheduler::getInstance();
        if (sched->getWorkerCount() == 0) {
            for (int i = 0; i < std::max(1, threads); ++i)
                sched->registerWorker(new LocalWorker(i, formatString("wrk%i", i)));
            sched->start();
        }
        Properties ip(integrator_plugin ? integrator_plugin : (p->integrator == PHIP_INTEGRATOR_DIRECT ? "direct" : p->integrator == PHIP_INTEGRATOR_VOLPATH_SIMPLE ? "volpath_simple" : "path"));
        if (p->integrator == PHIP_INTEGRATOR_DIRECT) {
            ip.setSize("emitterSamples", (size_t) p->emitter_samples); ip.setSize("bsdfSamples", (size_t) p->bsdf_samples);
        } else {
            ip.setInteger("maxDepth", p->max_depth); ip.setInteger("rrDepth", p->rr_depth);
        }
        ip.setBoolean("strictNormals", p->strict_normals != 0); ip.setBoolean("hideEmitters", p->hide_emitters != 0);
        ref<Integrator> integ = static_cast<Integrator *>(create(MTS_CLASS(Integrator), ip));
        integ->configure();
        ref<Sampler> sampler = makeSampler(scene, p);
        integ->configureSampler(scene, sampler);
        scene->setIntegrator(integ);
        scene->setSampler(sampler);
        scene->setBlockSize(p->block_size > 0 ? (uint32_t) p->block_size : 32u);
        scene->getFilm()->clear();

        ref<RenderQueue> queue = new RenderQueue();
        ref<JobStatusListener> status = new JobStatusListener();
        queue->registerListener(status);
        { std::lock_guard<std::mutex> g(g_logLock); g_lastWarning.clear(); }
        ref<RenderJob> job = new RenderJob("rend", scene, queue, -1, -1, -1, false, false);
        const auto t0 = std::chrono::steady_clock::now();
        job->start();
        queue->waitLeft(0);
        queue->join();
        queue->unregisterListener(status);
        if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (status->cancelled) {          /* the job ended without a frame: its integrator threw (Log(EError)) or returned false */
            std::lock_guard<std::mutex> g(g_logLock);
            g_err = "the render job failed: " + (g_lastWarning.empty() ? std::string("(nothing was logged)") : g_lastWarning);
            return -2;
        }
        if (out_rgb) {
            const Vector2i size = scene->getFilm()->getCropSize();
            ref<Bitmap> target = new Bitmap(Bitmap::ERGB, Bitmap::EFloat32, size);
            scene->getFilm()->develop(Point2i(0, 0), size, Point2i(0, 0), target);
            memcpy(out_rgb, target->getFloat32Data(), (size_t) size.x * size.y * 3 * sizeof(float));
        }
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

/* rays: n x (o.xyz, mint, d.xyz, maxt); hits: n x (t, u, v, shapeIndex or -1, primIndex) */
int ref_trace(void *h, const float *rays, size_t n, float *hits5) {
    try {
        RefScene *rs = static_cast<RefScene *>(h);
        const ref_vector<Shape> &shapes = rs->scene->getShapes();
        for (size_t i = 0; i < n; ++i) {
            const float *r = rays + 8 * i;
            Ray ray(Point(r[0], r[1], r[2]), Vector(r[4], r[5], r[6]), r[3], r[7], 0.0f);
            Intersection its;
            float *o = hits5 + 5 * i;
            if (rs->scene->rayIntersect(ray, its)) {
                int si = -1;
                for (size_t k = 0; k < shapes.size(); ++k) if (shapes[k].get() == its.shape) { si = (int) k; break; }
                o[0] = its.t; o[1] = its.uv.x; o[2] = its.uv.y; o[3] = (float) si; o[4] = (float) its.primIndex;
            } else { o[0] = std::numeric_limits<float>::infinity(); o[1] = o[2] = 0; o[3] = o[4] = -1; }
        }
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

/* full intersection record: p, geoN, shN, dpdu, dpdv, uv, wi -- 20 floats per ray (t = inf: miss) */
int ref_intersect(void *h, const float *rays, size_t n, float *out20) {
    try {
        RefScene *rs = static_cast<RefScene *>(h);
        for (size_t i = 0; i < n; ++i) {
            const float *r = rays + 8 * i;
            Ray ray(Point(r[0], r[1], r[2]), Vector(r[4], r[5], r[6]), r[3], r[7], 0.0f);
            Intersection its;
            float *o = out20 + 20 * i;
            for (int k = 0; k < 20; ++k) o[k] = 0;
            if (!rs->scene->rayIntersect(ray, its)) { o[0] = std::numeric_limits<float>::infinity(); continue; }
            o[0] = its.t;
            for (int k = 0; k < 3; ++k) { o[1 + k] = its.p[k]; o[4 + k] = its.geoFrame.n[k]; o[7 + k] = its.shFrame.n[k]; o[10 + k] = its.dpdu[k]; o[13 + k] = its.dpdv[k]; }
            o[16] = its.uv.x; o[17] = its.uv.y; o[18] = its.wi.z; o[19] = its.isEmitter() ? 1.0f : 0.0f;
        }
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

static void fakeIts(Intersection &its, const Vector &wi) {
    its.p = Point(0.0f); its.t = 1.0f; its.wi = wi; its.uv = Point2(0.0f);
    its.geoFrame = Frame(Normal(0, 0, 1)); its.shFrame = its.geoFrame;
    its.dpdu = Vector(1, 0, 0); its.dpdv = Vector(0, 1, 0); its.hasUVPartials = false; its.time = 0; its.shape = NULL;
}

int ref_bsdf_sample(void *h, uint32_t material, size_t n, const float *wi3, const float *sample2,
                    float *wo3, float *weight3, float *pdf, uint8_t *delta) {
    try {
        RefScene *rs = static_cast<RefScene *>(h);
        const BSDF *bsdf = rs->materials.at(material);
        for (size_t i = 0; i < n; ++i) {
            Intersection its; fakeIts(its, Vector(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]));
            BSDFSamplingRecord bRec(its, NULL, ERadiance);
            Float p = 0;
            Spectrum w = bsdf->sample(bRec, p, Point2(sample2[2 * i], sample2[2 * i + 1]));
            if (w.isZero()) { p = 0; bRec.wo = Vector(0.0f); }
            Float r, g, b; w.toLinearRGB(r, g, b);
            wo3[3 * i] = bRec.wo.x; wo3[3 * i + 1] = bRec.wo.y; wo3[3 * i + 2] = bRec.wo.z;
            weight3[3 * i] = r; weight3[3 * i + 1] = g; weight3[3 * i + 2] = b;
            pdf[i] = p; if (delta) delta[i] = (!w.isZero() && (bRec.sampledType & BSDF::EDelta)) ? 1 : 0;
        }
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

int ref_bsdf_eval_pdf(void *h, uint32_t material, size_t n, const float *wi3, const float *wo3, float *value3, float *pdf) {
    try {
        RefScene *rs = static_cast<RefScene *>(h);
        const BSDF *bsdf = rs->materials.at(material);
        for (size_t i = 0; i < n; ++i) {
            Intersection its; fakeIts(its, Vector(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]));
            BSDFSamplingRecord bRec(its, Vector(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]), ERadiance);
            Spectrum v = bsdf->eval(bRec);
            Float r, g, b; v.toLinearRGB(r, g, b);
            value3[3 * i] = r; value3[3 * i + 1] = g; value3[3 * i + 2] = b;
            pdf[i] = bsdf->pdf(bRec);
        }
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

/* Scene::sampleEmitterDirect(dRec, sample, testVisibility = false) + pdfEmitterDirect, like oracle_sample_emitter */
int ref_sample_emitter(void *h, const float *ref3, const float *refN3, size_t n, const float *sample2,
                       float *d3, float *dist, float *pdf, float *value3, float *pdf_check) {
    try {
        RefScene *rs = static_cast<RefScene *>(h);
        for (size_t i = 0; i < n; ++i) {
            Intersection its; fakeIts(its, Vector(0, 0, 1));
            its.p = Point(ref3[0], ref3[1], ref3[2]);
            DirectSamplingRecord dRec(its);
            dRec.refN = Normal(refN3[0], refN3[1], refN3[2]);
            Spectrum value = rs->scene->sampleEmitterDirect(dRec, Point2(sample2[2 * i], sample2[2 * i + 1]), false);
            Float r, g, b; value.toLinearRGB(r, g, b);
            d3[3 * i] = dRec.d.x; d3[3 * i + 1] = dRec.d.y; d3[3 * i + 2] = dRec.d.z;
            value3[3 * i] = r; value3[3 * i + 1] = g; value3[3 * i + 2] = b;
            dist[i] = dRec.dist; pdf[i] = dRec.pdf;
            pdf_check[i] = (dRec.pdf != 0 && !value.isZero()) ? rs->scene->pdfEmitterDirect(dRec) : 0.0f;
        }
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

/* out: o.xyz, d.xyz, mint, maxt, rx.xyz, ry.xyz (differential directions before scaleDifferential) */
int ref_camera_ray(void *h, float sx, float sy, float *out14) {
    try {
        RefScene *rs = static_cast<RefScene *>(h);
        RayDifferential ray;
        rs->scene->getSensor()->sampleRayDifferential(ray, Point2(sx, sy), Point2(0.5f), 0.5f);
        for (int k = 0; k < 3; ++k) { out14[k] = ray.o[k]; out14[3 + k] = ray.d[k]; out14[8 + k] = ray.rxDirection[k]; out14[11 + k] = ray.ryDirection[k]; }
        out14[6] = ray.mint; out14[7] = ray.maxt;
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}


/* ---- MIP pyramids: kind 0 = EnvironmentMap's (envmap.cpp:178-181: Spectrum / half precision, repeat x clamp, illuminant
 * intent, no clamping), kind 1 = BitmapTexture's (bitmap.cpp:175-177,298-300: Color3 / half precision, reflectance intent, max value 1) ---- */
struct RefMip {
    typedef TMIPMap<Spectrum, TSpectrum<half, SPECTRUM_SAMPLES> > EnvMip;
    typedef TSpectrum<Float, 3> TexColor3;           /* bitmap.cpp:172-177: BitmapTexture's own Color3 / Color3h */
    typedef TSpectrum<half, 3> TexColor3h;
    typedef TMIPMap<TexColor3, TexColor3h> TexMip;
    ref<EnvMip> env; ref<TexMip> tex;
    std::vector<ref<Bitmap> > levels;
};

void *ref_mip_build(int kind, const float *texels, uint32_t w, uint32_t h, uint32_t wrap_u, uint32_t wrap_v, uint32_t filter_type, float max_anisotropy) {
    try {
        if (ref_init() != 0) return NULL;
        ref<Bitmap> bmp = rgbBitmap(texels, w, h);
        Properties rp("lanczos"); rp.setInteger("lobes", 2);
        ref<ReconstructionFilter> rf = static_cast<ReconstructionFilter *>(create(MTS_CLASS(ReconstructionFilter), rp));
        rf->configure();
        RefMip *m = new RefMip();
        const EMIPFilterType ft = (EMIPFilterType) filter_type;      /* ENearest, EBilinear, ETrilinear, EEWA = phip_filter_type */
        const Float aniso = ft == EEWA ? max_anisotropy : 1.0f;      /* bitmap.cpp:234-235 */
        int n;
        if (kind == 0) {
            m->env = new RefMip::EnvMip(bmp, Bitmap::ESpectrum, Bitmap::EFloat, rf, ReconstructionFilter::ERepeat, ReconstructionFilter::EClamp,
                                        ft, aniso, fs::path(), 0, std::numeric_limits<Float>::infinity(), Spectrum::EIlluminant);
            n = m->env->getLevels();
            for (int l = 0; l < n; ++l) m->levels.push_back(m->env->toBitmap(l)->convert(Bitmap::ERGB, Bitmap::EFloat32));
        } else {
            m->tex = new RefMip::TexMip(bmp, Bitmap::ERGB, Bitmap::EFloat, rf, (ReconstructionFilter::EBoundaryCondition) wrap_u,
                                        (ReconstructionFilter::EBoundaryCondition) wrap_v, ft, aniso, fs::path(), 0);
            n = m->tex->getLevels();
            for (int l = 0; l < n; ++l) m->levels.push_back(m->tex->toBitmap(l)->convert(Bitmap::ERGB, Bitmap::EFloat32));
        }
        return m;
    } catch (const std::exception &e) { g_err = e.what(); return NULL; }
}
int ref_mip_levels(void *h) { return (int) static_cast<RefMip *>(h)->levels.size(); }
void ref_mip_level_size(void *h, int l, int *w, int *ht) { const Bitmap *b = static_cast<RefMip *>(h)->levels.at(l); *w = b->getWidth(); *ht = b->getHeight(); }
void ref_mip_level_data(void *h, int l, float *out) {
    const Bitmap *b = static_cast<RefMip *>(h)->levels.at(l);
    memcpy(out, b->getFloat32Data(), (size_t) b->getWidth() * b->getHeight() * 3 * sizeof(float));
}
/* MIPMap::eval(uv, d0, d1), mipmap.h:629-728 */
int ref_mip_eval(void *h, size_t n, const float *uv2, const float *d0, const float *d1, float *out3) {
    try {
        RefMip *m = static_cast<RefMip *>(h);
        for (size_t i = 0; i < n; ++i) {
            const Point2 uv(uv2[2 * i], uv2[2 * i + 1]); const Vector2 a(d0[2 * i], d0[2 * i + 1]), b(d1[2 * i], d1[2 * i + 1]);
            Float r, g, bl;
            if (m->env) { Spectrum v = m->env->eval(uv, a, b); v.toLinearRGB(r, g, bl); }
            else { RefMip::TexColor3 v = m->tex->eval(uv, a, b); r = v[0]; g = v[1]; bl = v[2]; }
            out3[3 * i] = r; out3[3 * i + 1] = g; out3[3 * i + 2] = bl;
        }
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}
void ref_mip_destroy(void *h) { delete static_cast<RefMip *>(h); }

} // extern "C"
