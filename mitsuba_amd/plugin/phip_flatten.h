/*
 * phip_flatten.h -- shared by the `path_hip` and `direct_hip` plugin shims: Scene -> phip_scene_desc -> phip_scene.
 * (See path_hip.cpp for the build recipe; INTEGRATION.md lists the accessors the stand-in classes below assume.)
 */
#pragma once
#include <mitsuba/render/scene.h>
#include <mitsuba/render/renderproc.h>
#include <mitsuba/core/plugin.h>
#include <mitsuba/core/fresolver.h>
#include <mitsuba/render/mipmap.h>
#include <mitsuba/core/qmc.h>
#include <boost/algorithm/string.hpp>
#include <dlfcn.h>
#include <link.h>
#include "../../bsdfs/microfacet.h"      /* MicrofacetDistribution: plugin-local header of src/bsdfs */
#include "../../bsdfs/ior.h"             /* lookupIOR */
#include "../../samplers/faure.cpp"      /* PermutationStorage (faure.h, which has no include guard, comes with it): plugin-local to src/samplers and compiled into every plugin
                                            that uses it (src/samplers/SConscript:5-6), so into this one too -- the digit permutations of <sampler type="halton"/> /
                                            "hammersley" are handed to the device as data */
#include "phip.h"

/* How the shim gets at four plugin-local classes of the reference (TwoSidedBRDF's children, SmoothDiffuse's reflectance
 * texture, BitmapTexture's and EnvironmentMap's MIP pyramids and lookup parameters), which have no public getters:
 *   (default)                     not at all: stock Mitsuba, public getters only -- diffuse with a constant reflectance,
 *                                 dielectric, roughconductor, area + constant emitters; anything else is an EError
 *   -DPHIP_REFERENCE_ACCESSORS    the five one-line accessors of INTEGRATION.md section 2 have been added to the plugins
 *   -DPHIP_REFERENCE_SOURCES      no change to Mitsuba at all: the plugins' own source files are #included here for their class
 *                                 definitions and the members are read directly (compile with -fno-access-control); the
 *                                 copies of the plugin code this drags into path_hip.so are never instantiated
 */
#if defined(PHIP_REFERENCE_SOURCES)
#define CreateInstance CreateInstance_phip_twosided
#define GetDescription GetDescription_phip_twosided
#include "../../bsdfs/twosided.cpp"
#undef CreateInstance
#undef GetDescription
#define CreateInstance CreateInstance_phip_diffuse
#define GetDescription GetDescription_phip_diffuse
#include "../../bsdfs/diffuse.cpp"
#undef CreateInstance
#undef GetDescription
#define CreateInstance CreateInstance_phip_roughconductor
#define GetDescription GetDescription_phip_roughconductor
#include "../../bsdfs/roughconductor.cpp"
#undef CreateInstance
#undef GetDescription
#define CreateInstance CreateInstance_phip_dielectric
#define GetDescription GetDescription_phip_dielectric
#include "../../bsdfs/dielectric.cpp"
#undef CreateInstance
#undef GetDescription
#define CreateInstance CreateInstance_phip_bitmap
#define GetDescription GetDescription_phip_bitmap
#include "../../textures/bitmap.cpp"
#undef CreateInstance
#undef GetDescription
#define CreateInstance CreateInstance_phip_envmap
#define GetDescription GetDescription_phip_envmap
#include "../../emitters/envmap.cpp"
#undef CreateInstance
#undef GetDescription
#endif

#if SPECTRUM_SAMPLES != 3 || !defined(SINGLE_PRECISION)
#error "path_hip computes float32 linear RGB: build Mitsuba with -DSINGLE_PRECISION -DSPECTRUM_SAMPLES=3 (its default configuration)"
#endif

MTS_NAMESPACE_BEGIN

#if defined(PHIP_REFERENCE_ACCESSORS)
/* What the shim needs of src/emitters/envmap.cpp's EnvironmentMap (a plugin-local class): its MIP pyramid.  With the
   accessor of INTEGRATION.md added there and the class declaration moved to a header, this stand-in goes away. */
class EnvironmentMapAccess : public Emitter {
public:
    typedef TSpectrum<half, SPECTRUM_SAMPLES> SpectrumHalf;
    typedef TMIPMap<Spectrum, SpectrumHalf> MIPMap;
    virtual const MIPMap *getMIPMap() const = 0;
};

/* Same for src/bsdfs/diffuse.cpp's SmoothDiffuse (its reflectance texture) and src/textures/bitmap.cpp's BitmapTexture
   (MIP pyramid and lookup parameters): plugin-local classes without getters for what the device needs. */
class SmoothDiffuseAccess : public BSDF {
public:
    virtual const Texture *getReflectanceTexture() const = 0;
};
class BitmapTextureAccess : public Texture2D {
public:
    typedef TSpectrum<Float, 3> Color3;            /* bitmap.cpp:172-177 */
    typedef TSpectrum<half, 3> Color3h;
    typedef TMIPMap<Color3, Color3h> MIPMap3;
    virtual const MIPMap3 *getMIPMap3() const = 0;
    virtual ReconstructionFilter::EBoundaryCondition getWrapModeU() const = 0;
    virtual ReconstructionFilter::EBoundaryCondition getWrapModeV() const = 0;
    virtual Float getMaxAnisotropy() const = 0;
    virtual Vector2 getUVScale() const = 0;
    virtual Point2 getUVOffset() const = 0;
};

/* src/bsdfs/{roughconductor,dielectric}.cpp: the specularReflectance texture */
class SpecularReflectanceAccess : public BSDF {
public:
    virtual const Texture *getSpecularReflectanceTexture() const = 0;
};

/* ... and the second accessor of each of the two, declared AFTER getSpecularReflectanceTexture in the plugin source:
   src/bsdfs/roughconductor.cpp: the roughness textures (children "alpha" / "alphaU" / "alphaV"), axis 0 = m_alphaU, 1 = m_alphaV;
   src/bsdfs/dielectric.cpp: the specularTransmittance texture */
class RoughConductorAccess : public SpecularReflectanceAccess {
public:
    virtual const Texture *getAlphaTexture(int axis) const = 0;
};
class DielectricAccess : public SpecularReflectanceAccess {
public:
    virtual const Texture *getSpecularTransmittanceTexture() const = 0;
};

/* src/bsdfs/twosided.cpp's TwoSidedBRDF: its two children */
class TwoSidedAccess : public BSDF {
public:
    virtual const BSDF *getNestedBRDF(int i) const = 0;
};

#endif

#if defined(PHIP_REFERENCE_ACCESSORS) || defined(PHIP_REFERENCE_SOURCES)
#define PHIP_HAVE_INTERNALS 1
/* one spelling for both ways in */
struct PhipBitmapInfo {
    const TMIPMap<TSpectrum<Float, 3>, TSpectrum<half, 3> > *mip;
    ReconstructionFilter::EBoundaryCondition wrapU, wrapV;
    Float maxAnisotropy; Vector2 uvScale; Point2 uvOffset;
};
#if defined(PHIP_REFERENCE_ACCESSORS)
inline const BSDF *phipNested(const BSDF *b, int i) { return static_cast<const TwoSidedAccess *>(b)->getNestedBRDF(i); }
inline const Texture *phipReflectanceTexture(const BSDF *b) { return static_cast<const SmoothDiffuseAccess *>(b)->getReflectanceTexture(); }
inline const Texture *phipSpecularTexture(const BSDF *b, bool) { return static_cast<const SpecularReflectanceAccess *>(b)->getSpecularReflectanceTexture(); }
inline const Texture *phipAlphaTexture(const BSDF *b, int axis) { return static_cast<const RoughConductorAccess *>(b)->getAlphaTexture(axis); }
inline const Texture *phipTransmittanceTexture(const BSDF *b) { return static_cast<const DielectricAccess *>(b)->getSpecularTransmittanceTexture(); }
inline const TMIPMap<Spectrum, TSpectrum<half, SPECTRUM_SAMPLES> > *phipEnvMip(const Emitter *e) { return static_cast<const EnvironmentMapAccess *>(e)->getMIPMap(); }
inline PhipBitmapInfo phipBitmap(const Texture *t) {
    const BitmapTextureAccess *b = static_cast<const BitmapTextureAccess *>(t);
    PhipBitmapInfo i = { b->getMIPMap3(), b->getWrapModeU(), b->getWrapModeV(), b->getMaxAnisotropy(), b->getUVScale(), b->getUVOffset() };
    return i;
}
/* (the accessor patch goes through virtual functions of the plugins' own classes: nothing to probe) */
inline void phipLayoutProbeDiffuse(const BSDF *) { }
inline void phipLayoutProbeTwoSided(const BSDF *) { }
inline void phipLayoutProbeRoughConductor(const BSDF *) { }
inline void phipLayoutProbeDielectric(const BSDF *) { }
inline void phipLayoutProbeBitmap(const Texture *) { }
#else
inline const BSDF *phipNested(const BSDF *b, int i) { return static_cast<const TwoSidedBRDF *>(b)->m_nestedBRDF[i].get(); }
inline const Texture *phipReflectanceTexture(const BSDF *b) { return static_cast<const SmoothDiffuse *>(b)->m_reflectance.get(); }
inline const Texture *phipSpecularTexture(const BSDF *b, bool conductor) {
    return conductor ? static_cast<const RoughConductor *>(b)->m_specularReflectance.get() : static_cast<const SmoothDielectric *>(b)->m_specularReflectance.get();
}
inline const Texture *phipAlphaTexture(const BSDF *b, int axis) { const RoughConductor *r = static_cast<const RoughConductor *>(b); return axis ? r->m_alphaV.get() : r->m_alphaU.get(); }
inline const Texture *phipTransmittanceTexture(const BSDF *b) { return static_cast<const SmoothDielectric *>(b)->m_specularTransmittance.get(); }
inline const TMIPMap<Spectrum, TSpectrum<half, SPECTRUM_SAMPLES> > *phipEnvMip(const Emitter *e) { return static_cast<const EnvironmentMap *>(e)->m_mipmap; }
inline PhipBitmapInfo phipBitmap(const Texture *t) {
    const BitmapTexture *b = static_cast<const BitmapTexture *>(t);
    PhipBitmapInfo i = { b->m_mipmap3.get(), b->m_wrapModeU, b->m_wrapModeV, b->m_maxAnisotropy, b->m_uvScale, b->m_uvOffset };
    return i;
}
/* -DPHIP_REFERENCE_SOURCES reads members of objects that live in OTHER shared objects (diffuse.so, twosided.so, ...) through the class definitions compiled into this
   one: the same source files, but nothing in the language checks that both were compiled to the same layout (an ODR / layout hazard with no guard).
   Two guards (round 6):
     (a) compile time -- the members the shim reads lie where the reference's sources, as included here, put them: a change of those sources that moves them fails the build
         of the shim instead of reading garbage (sizes in terms of the public base classes, so that a different Float / Spectrum configuration fails too);
     (b) run time, per object and before anything is read out of it -- every value the shim is about to take through a private member is ALSO reachable through a public
         virtual of the base class (BSDF::getDiffuseReflectance / getRoughness, Texture::getResolution / getMaximum, Object::getClass of a nested object): the two ways must
         agree on a probe intersection, or the plugin that owns the object was built with another layout than this shim and the render is refused (phipLayoutProbe*). */
static_assert(sizeof(TwoSidedBRDF) == sizeof(BSDF) + 2 * sizeof(ref<BSDF>), "twosided.cpp: TwoSidedBRDF is BSDF + m_nestedBRDF[2]");
static_assert(sizeof(SmoothDiffuse) == sizeof(BSDF) + sizeof(ref<Texture>), "diffuse.cpp: SmoothDiffuse is BSDF + m_reflectance");
static_assert(sizeof(ref<Texture>) == sizeof(void *) && sizeof(ref<BSDF>) == sizeof(void *), "ref<T> is one pointer");
inline Intersection phipProbeIntersection() {
    Intersection its;
    its.p = Point(0.0f); its.t = 1.0f; its.uv = Point2(0.37f, 0.61f); its.wi = Vector(0, 0, 1);
    its.geoFrame = its.shFrame = Frame(Normal(0, 0, 1));
    its.dudx = its.dudy = its.dvdx = its.dvdy = 0; its.hasUVPartials = false; its.time = 0; its.shape = NULL; its.instance = NULL;
    return its;
}
inline bool phipSameSpectrum(const Spectrum &a, const Spectrum &b) { for (int i = 0; i < SPECTRUM_SAMPLES; ++i) if (!(a[i] == b[i])) return false; return true; }
inline void phipLayoutProbeDiffuse(const BSDF *b) {
    const Intersection its = phipProbeIntersection();
    const Texture *t = phipReflectanceTexture(b);
    if (!t || !t->getClass()->derivesFrom(MTS_CLASS(Texture)) || !phipSameSpectrum(t->eval(its), b->getDiffuseReflectance(its)))
        SLog(EError, "path_hip: the 'diffuse' plugin that owns this BSDF was not built from the sources this shim was compiled against (SmoothDiffuse::m_reflectance is not where diffuse.cpp puts it)");
}
inline void phipLayoutProbeTwoSided(const BSDF *b) {
    const Intersection its = phipProbeIntersection();          /* wi.z > 0: TwoSidedBRDF::getDiffuseReflectance asks m_nestedBRDF[0] (twosided.cpp:198-203) */
    const BSDF *n0 = phipNested(b, 0), *n1 = phipNested(b, 1);
    if (!n0 || !n1 || !n0->getClass()->derivesFrom(MTS_CLASS(BSDF)) || !n1->getClass()->derivesFrom(MTS_CLASS(BSDF))
        || !phipSameSpectrum(n0->getDiffuseReflectance(its), b->getDiffuseReflectance(its)) || n0->getRoughness(its, 0) != b->getRoughness(its, 0))
        SLog(EError, "path_hip: the 'twosided' plugin that owns this BSDF was not built from the sources this shim was compiled against (TwoSidedBRDF::m_nestedBRDF is not where twosided.cpp puts it)");
}
inline void phipLayoutProbeRoughConductor(const BSDF *b) {
    const Intersection its = phipProbeIntersection();          /* RoughConductor::getRoughness = 0.5 (alphaU + alphaV) of the textures' averages (roughconductor.cpp:434-437) */
    const Texture *tu = phipAlphaTexture(b, 0), *tv = phipAlphaTexture(b, 1), *ts = phipSpecularTexture(b, true);
    if (!tu || !tv || !ts || !tu->getClass()->derivesFrom(MTS_CLASS(Texture)) || !tv->getClass()->derivesFrom(MTS_CLASS(Texture)) || !ts->getClass()->derivesFrom(MTS_CLASS(Texture))
        || 0.5f * (tu->eval(its).average() + tv->eval(its).average()) != b->getRoughness(its, 0))
        SLog(EError, "path_hip: the 'roughconductor' plugin that owns this BSDF was not built from the sources this shim was compiled against (m_alphaU / m_alphaV / m_specularReflectance are not where roughconductor.cpp puts them)");
}
inline void phipLayoutProbeDielectric(const BSDF *b) {
    const Texture *tr = phipSpecularTexture(b, false), *tt = phipTransmittanceTexture(b);
    if (!tr || !tt || !tr->getClass()->derivesFrom(MTS_CLASS(Texture)) || !tt->getClass()->derivesFrom(MTS_CLASS(Texture)))
        SLog(EError, "path_hip: the 'dielectric' plugin that owns this BSDF was not built from the sources this shim was compiled against (m_specularReflectance / m_specularTransmittance are not where dielectric.cpp puts them)");
}
inline void phipLayoutProbeBitmap(const Texture *t) {
    const PhipBitmapInfo i = phipBitmap(t);                     /* BitmapTexture::getResolution = the MIP pyramid's level-0 size (bitmap.cpp) */
    const Vector3i res = t->getResolution();
    if (!i.mip || i.mip->getWidth() != res.x || i.mip->getHeight() != res.y || !(i.maxAnisotropy >= 0))
        SLog(EError, "path_hip: the 'bitmap' plugin that owns this texture was not built from the sources this shim was compiled against (BitmapTexture::m_mipmap3 is not where bitmap.cpp puts it)");
}
#endif
#endif

/* Owns the device scene of one integrator instance. */
class PhipSceneHolder {
public:
    PhipSceneHolder() : m_qmcScramble(0), m_scene(NULL), m_device(0), m_deviceCount(1) { }
    ~PhipSceneHolder() { if (m_scene) phip_scene_destroy(m_scene); }
    phip_scene *get() const { return m_scene; }
    void setDevice(int device) { m_device = device; }
    int getDevice() const { return m_device; }

    /* Devices of the job: `devices` = how many GPUs the render spreads over (default 1; 0 = every visible GPU), starting at
       `device`.  With more than one, libphip runs one host thread + stream per GPU and merges the films with ncclReduce. */
    void setDeviceCount(int n) { m_deviceCount = n; }
    int getDeviceCount() const { return m_deviceCount; }

    /* The scene's <sampler> (src/librender/integrator.cpp:104,169 clone it per worker): `independent` is honoured as "independent
       uniform samples" -- its SFMT stream is one sequential generator per worker thread (independent.cpp:71-103), which no
       parallel schedule reproduces, not even the reference's own from run to run (independent.cpp:42-45) -- so the device's
       counter-based stream stands in, which is said once.  `ldsampler` maps to PHIP_SAMPLER_LD: the same construction (scrambled
       (0,2)-sequences in a random order per pixel and dimension for the first four 1D / 2D requests of a sample, ldsampler.cpp:151-226)
       with the scrambles and the order drawn from the counter-based generator instead of the worker's Random -- default `dimension` only.  `stratified` maps to PHIP_SAMPLER_STRATIFIED the same way; `sobol`, `halton` and `hammersley` are deterministic and map to PHIP_SAMPLER_SOBOL / _HALTON / _HAMMERSLEY: the plugins' own numbers. */
    static int checkSampler(const Sampler *sampler, const char *name) {
        const std::string cls = sampler->getClass()->getName();
        if (cls == "LowDiscrepancySampler") {
            if (sampler->getProperties().getSize("dimension", 4) != 4)
                SLog(EError, "%s: ldsampler with dimension != 4 is not supported", name);
            static bool toldLD = false;
            if (!toldLD) {
                toldLD = true;
                SLog(EWarn, "%s: 'ldsampler' is served by the device's low-discrepancy stream (the same scrambled (0,2)-sequences per pixel and "
                            "dimension, scrambles from the counter-based generator): the same stratification, not the same numbers as the CPU integrator", name);
            }
            return PHIP_SAMPLER_LD;
        }
        if (cls == "IndependentSampler") {
            static bool told = false;
            if (!told) {
                told = true;
                SLog(EWarn, "%s: the 'independent' sampler is served by the device's counter-based stream (pcg4d(pixel, sample, dimension), seed 0): "
                            "statistically equivalent, not the same random numbers as the CPU integrator", name);
            }
            return PHIP_SAMPLER_CTR;
        }
        if (cls == "StratifiedSampler") {
            if (sampler->getProperties().getInteger("dimension", 4) != 4)
                SLog(EError, "%s: stratified sampler with dimension != 4 is not supported", name);
            static bool toldST = false;
            if (!toldST) {
                toldST = true;
                SLog(EWarn, "%s: 'stratified' is served by the device's stratified stream (one cell of the sampleCount grid per sample and dimension in a "
                            "permuted order, jittered: stratified.cpp:147-200 with the permutations and jitter from the counter-based generator): the same stratification, "
                            "not the same numbers as the CPU integrator", name);
            }
            return PHIP_SAMPLER_STRATIFIED;
        }
        if (cls == "SobolSampler")
            return PHIP_SAMPLER_SOBOL;          /* the plugin's own sequence, number for number: setSobol() below */
        if (cls == "HaltonSampler") return PHIP_SAMPLER_HALTON;              /* likewise: setRadicalInverse() below */
        if (cls == "HammersleySampler") return PHIP_SAMPLER_HAMMERSLEY;
        SLog(EError, "%s: sampler \"%s\" is not supported ('independent', 'ldsampler', 'stratified', 'sobol', 'halton', 'hammersley')", name, cls.c_str());
        return PHIP_SAMPLER_CTR;
    }

    /* PHIP_SAMPLER_SOBOL: the direction numbers are the `sobol` plugin's (sobol::Matrices, src/samplers/sobolseq.cpp). The scene's sampler
       is an instance of it, so the plugin is loaded (PluginManager dlopens plugins RTLD_LOCAL, plugin.cpp:71): find it among the loaded
       objects, take a second handle on it and read the three tables as data.  Dimensions are consumed in call order exactly as
       SobolSampler does (sobol.cpp:218-247), so path_hip + sobol renders the image `path` + sobol renders, sample for sample. */
    struct FindPlugin { const char *suffix; std::string path; };
    static int findPluginCb(struct dl_phdr_info *info, size_t, void *data) {
        FindPlugin *f = (FindPlugin *) data;
        const std::string n = info->dlpi_name ? info->dlpi_name : "";
        const size_t l = strlen(f->suffix);
        if (n.size() >= l && n.compare(n.size() - l, l, f->suffix) == 0) { f->path = n; return 1; }
        return 0;
    }
    /* PHIP_SAMPLER_HALTON / _HAMMERSLEY: the reference's prime table (libcore, exported) and the digit permutations its PermutationStorage builds for the
       sampler's `scramble` property (-1, the default: Faure's; 0: none; otherwise pseudorandom ones -- halton.cpp:124,184-194), concatenated; built once
       per scramble value like the plugins' own m_globalPermutations */
    void setRadicalInverse(const Sampler *sampler, phip_render_params &rp) {
        const int scramble = sampler->getProperties().getInteger("scramble", -1);
        if (m_qmcPrimes.empty() || m_qmcScramble != scramble) {
            m_qmcPrimes.assign(primeTable, primeTable + primeTableSize);
            m_qmcPerm.clear();
            if (scramble != 0) {
                ref<PermutationStorage> ps = new PermutationStorage(scramble);
                for (size_t d = 0; d < primeTableSize; ++d)
                    m_qmcPerm.insert(m_qmcPerm.end(), ps->getPermutation((uint32_t) d), ps->getPermutation((uint32_t) d) + primeTable[d]);
            }
            m_qmcScramble = scramble;
        }
        rp.qmc_primes = (const uint32_t *) &m_qmcPrimes[0];             /* (primeTable is `const int[]`: the values are positive) */
        rp.qmc_permutations = m_qmcPerm.empty() ? NULL : &m_qmcPerm[0];
        rp.qmc_dimensions = (uint32_t) primeTableSize;
    }
    std::vector<int> m_qmcPrimes; std::vector<uint16_t> m_qmcPerm; int m_qmcScramble;

    static void setSobol(const Sampler *sampler, const Vector2i &cropSize, phip_render_params &rp, const char *name) {
        FindPlugin f; f.suffix = "/sobol.so";
        dl_iterate_phdr(findPluginCb, &f);
        void *h = f.path.empty() ? NULL : dlopen(f.path.c_str(), RTLD_LAZY | RTLD_NOLOAD);
        const uint32_t *m32 = h ? (const uint32_t *) dlsym(h, "_ZN5sobol8Matrices10matrices32E") : NULL;
        const uint64_t *vdc = h ? (const uint64_t *) dlsym(h, "_ZN5sobol8Matrices18vdc_sobol_matricesE") : NULL;
        const uint64_t *inv = h ? (const uint64_t *) dlsym(h, "_ZN5sobol8Matrices22vdc_sobol_matrices_invE") : NULL;
        if (!m32 || !vdc || !inv)
            SLog(EError, "%s: the direction numbers of the loaded 'sobol' plugin were not found (%s)", name, f.path.empty() ? "plugin not among the loaded objects" : f.path.c_str());
        uint32_t res = 1, m = 0;
        while (res < (uint32_t) std::max(cropSize.x, cropSize.y)) { res <<= 1; ++m; }        /* SobolSampler::setFilmResolution, sobol.cpp:147-157 */
        rp.sobol_matrices = m32; rp.sobol_dimensions = 1024;                                 /* sobol::Matrices::num_dimensions */
        rp.sobol_log_resolution = m;
        rp.sobol_vdc = m > 1 ? vdc + (size_t) (m - 1) * PHIP_SOBOL_MATRIX_SIZE : NULL;       /* vdc_sobol_matrices[m - 1] (sobolseq.h:104-107) */
        rp.sobol_vdc_inv = m > 1 ? inv + (size_t) (m - 1) * PHIP_SOBOL_MATRIX_SIZE : NULL;
        uint64_t scramble = (uint64_t) sampler->getProperties().getSize("scramble", 0);
        if (scramble) { union { uint64_t ui64; uint32_t v[2]; } u = { scramble }; scramble = sampleTEA(u.v[0], u.v[1]); }   /* sobol.cpp:92-102 */
        rp.sobol_scramble = scramble;
    }

    /* SamplingIntegrator::render (integrator.cpp:95-129) + BlockedRenderProcess (renderproc.cpp:142-176): the job runs as a few
       progressive passes over the whole frame (sample indices [k0, k1) of every pixel per pass), so that listeners see what the
       reference's block stream gives them: signalWorkBegin when a pass starts, film->put + signalWorkEnd + a ProgressReporter
       update when it ends, and a cancellation point in between. */
    bool render(Scene *scene, RenderQueue *queue, const RenderJob *job, phip_render_params &rp, const char *name) {
        ref<Sensor> sensor = scene->getSensor();
        ref<Film> film = sensor->getFilm();
        const Vector2i size = film->getCropSize();
        const Sampler *sampler = scene->getSampler();
        const int samplerKind = checkSampler(sampler, name);
        const size_t spp = sampler->getSampleCount();
        SLog(EInfo, "Starting render job (%ix%i, " SIZE_T_FMT " samples, %s) ..", size.x, size.y, spp, phip_version());
        rp.block_size = (int32_t) scene->getBlockSize();
        if (samplerKind == PHIP_SAMPLER_SOBOL) setSobol(sampler, size, rp, name);
        if (samplerKind == PHIP_SAMPLER_HALTON || samplerKind == PHIP_SAMPLER_HAMMERSLEY) setRadicalInverse(sampler, rp);
        rp.sampler = samplerKind; rp.seed = 0; rp.shard_index = 0; rp.shard_count = 1; rp.device = m_device;
        int nDev = m_deviceCount == 0 ? phip_device_count() - m_device : m_deviceCount;
        if (nDev > PHIP_MAX_DEVICES) nDev = PHIP_MAX_DEVICES;
        if (nDev > 1) { rp.n_devices = nDev; for (int i = 0; i < nDev; ++i) rp.devices[i] = m_device + i; }
        /* passes of at most ~128 M camera samples: a fraction of a second each on one MI355X */
        const double pixels = (double) size.x * (double) size.y;
        size_t sppPerPass = (size_t) std::max(1.0, std::floor(128e6 * std::max(1, nDev) / pixels));
        if (const char *e = getenv("PHIP_SHIM_PASS_SPP")) sppPerPass = (size_t) std::max(1, atoi(e));
        const size_t nPasses = (spp + sppPerPass - 1) / sppPerPass;
        /* crop-relative (renderproc.cpp:160-173); no border: border pixels are already folded in by the device film pass */
        ref<Bitmap> target = new Bitmap(Bitmap::ESpectrumAlphaWeight, Bitmap::EFloat32, size);   /* = the film storage's format: setBitmap is a memcpy */
        ref<ImageBlock> block = new ImageBlock(Bitmap::ESpectrumAlphaWeight, size, NULL);          /* what listeners are handed (no border) */
        block->setOffset(Point2i(0, 0));
        ref<RectangularWorkUnit> wu = new RectangularWorkUnit(); wu->setOffset(Point2i(0, 0)); wu->setSize(size);
        ProgressReporter progress("Rendering", nPasses, job);
        phip_stats total; memset(&total, 0, sizeof(total));
        for (size_t pass = 0, k0 = 0; pass < nPasses; ++pass, k0 += sppPerPass) {
            rp.spp = (int32_t) std::min(sppPerPass, spp - k0);
            rp.sample_offset = (int32_t) k0; rp.sample_total = (int32_t) spp;
            rp.flags = (rp.flags & ~PHIP_FLAG_ACCUMULATE) | (pass > 0 ? PHIP_FLAG_ACCUMULATE : 0);
            queue->signalWorkBegin(job, wu.get(), 0);
            phip_stats st;
            int rc = phip_render(m_scene, &rp, target->getFloat32Data(), &st);
            if (rc == PHIP_ERR_CANCELLED) {
                queue->signalWorkCanceled(job, Point2i(0, 0), size);
                return false;
            }
            if (rc != PHIP_OK)
                SLog(EError, "%s: %s", name, phip_last_error());   /* throws std::runtime_error, caught by RenderJob::run */
            film->setBitmap(target);                              /* hdrfilm.cpp:395-425: replaces m_storage's bitmap with the accumulated frame */
            memcpy(block->getBitmap()->getFloat32Data(), target->getFloat32Data(), target->getBufferSize());
            queue->signalWorkEnd(job, block, false);
            progress.update(pass + 1);
            total.samples += st.samples; total.closest_rays += st.closest_rays; total.shadow_rays += st.shadow_rays; total.render_ms += st.render_ms;
        }
        queue->signalRefresh(job);
        SLog(EInfo, "%s: %.1f Msamples/s, %.1f Mrays/s (%i GPU%s, %i pass%s)", name, total.samples / 1e3 / total.render_ms,
            (total.closest_rays + total.shadow_rays) / 1e3 / total.render_ms, std::max(1, nDev), nDev > 1 ? "s" : "", (int) nPasses, nPasses > 1 ? "es" : "");
        return true;
    }

    /* ---- Scene -> phip_scene_desc ---- */
    void flatten(const Scene *scene) {
        m_textures.clear(); m_textureIds.clear(); m_textureLevels.clear(); m_envLevels.clear();
        std::vector<float> positions, normals, texcoords; std::vector<uint32_t> indices;
        bool anyTexcoords = false;
        std::vector<phip_shape> shapes; std::vector<phip_material> materials; std::vector<phip_emitter> emitters;
        std::map<const BSDF *, uint32_t> bsdfIds;
        std::map<const Shape *, uint32_t> shapeIds;
        phip_envmap envmap; memset(&envmap, 0, sizeof(envmap));
        bool anyNormals = false;

        const ref_vector<Shape> &list = scene->getShapes();
        for (size_t i = 0; i < list.size(); ++i) {
            const Shape *shape = list[i].get();
            ref<TriMesh> mesh;
            /* outside the path: an error, never a silent approximation */
            if (shape->hasSubsurface())
                SLog(EError, "path_hip: shape \"%s\" has a subsurface integrator -- not supported", shape->getName().c_str());
            if (shape->isMediumTransition())
                SLog(EError, "path_hip: shape \"%s\" marks a participating-medium transition -- media are not supported", shape->getName().c_str());
            if (shape->getClass()->derivesFrom(MTS_CLASS(TriMesh)))
                mesh = const_cast<TriMesh *>(static_cast<const TriMesh *>(shape));
            else {
                mesh = const_cast<Shape *>(shape)->createTriMesh();       /* rectangle.cpp:170-203, cube, disk, sphere ... */
                const std::string scls = shape->getClass()->getName();
                if (scls != "Rectangle" && scls != "Cube")                  /* planar shapes tessellate exactly */
                    SLog(EWarn, "path_hip: analytic shape \"%s\" (%s) is rendered as its createTriMesh() tessellation: silhouette and "
                                "area-emitter sampling differ from the CPU integrator", shape->getName().c_str(), scls.c_str());
            }
            if (!mesh)
                SLog(EError, "path_hip: shape \"%s\" cannot be converted to a triangle mesh", shape->getName().c_str());
            phip_shape s; memset(&s, 0, sizeof(s));
            s.first_vertex = (uint32_t) (positions.size() / 3); s.n_vertices = (uint32_t) mesh->getVertexCount();
            s.first_triangle = (uint32_t) (indices.size() / 3); s.n_triangles = (uint32_t) mesh->getTriangleCount();
            s.has_normals = mesh->getVertexNormals() ? 1 : 0; anyNormals |= s.has_normals != 0;
            s.has_texcoords = mesh->getVertexTexcoords() ? 1 : 0; anyTexcoords |= s.has_texcoords != 0;
            for (size_t v = 0; v < mesh->getVertexCount(); ++v) {
                const Point &p = mesh->getVertexPositions()[v];
                positions.push_back(p.x); positions.push_back(p.y); positions.push_back(p.z);
                Normal n = mesh->getVertexNormals() ? mesh->getVertexNormals()[v] : Normal(0.0f);
                normals.push_back(n.x); normals.push_back(n.y); normals.push_back(n.z);
                Point2 uv = mesh->getVertexTexcoords() ? mesh->getVertexTexcoords()[v] : Point2(0.0f);
                texcoords.push_back(uv.x); texcoords.push_back(uv.y);
            }
            for (size_t t = 0; t < mesh->getTriangleCount(); ++t)
                for (int k = 0; k < 3; ++k) indices.push_back(s.first_vertex + mesh->getTriangles()[t].idx[k]);
            s.material = convertBSDF(shape->getBSDF(), materials, bsdfIds);
            s.emitter = -1;
            shapeIds[shape] = (uint32_t) shapes.size();
            shapes.push_back(s);
        }
        /* emitters in the order of Scene::getEmitters(): the selection PDF (scene.cpp:375-381) is built in that order */
        const ref_vector<Emitter> &ems = scene->getEmitters();
        for (size_t i = 0; i < ems.size(); ++i) {
            const Emitter *e = ems[i].get();
            const std::string cls = e->getClass()->getName();
            phip_emitter pe; memset(&pe, 0, sizeof(pe));
            Spectrum rad = e->getProperties().getSpectrum("radiance", Spectrum::getD65());   /* area.cpp:80, constant.cpp:48 */
            Float r, g, b; rad.toLinearRGB(r, g, b);
            pe.radiance[0] = r; pe.radiance[1] = g; pe.radiance[2] = b;
            pe.sampling_weight = e->getSamplingWeight();
            if (cls == "AreaLight") {
                std::map<const Shape *, uint32_t>::const_iterator it = shapeIds.find(e->getShape());
                if (it == shapeIds.end())
                    SLog(EError, "path_hip: area emitter without a shape in the scene");
                pe.type = PHIP_EMITTER_AREA; pe.shape = it->second;
                shapes[it->second].emitter = (int32_t) emitters.size();
            } else if (cls == "ConstantBackgroundEmitter") {
                pe.type = PHIP_EMITTER_CONSTANT; pe.shape = 0xFFFFFFFFu;     /* the library derives m_sceneBSphere itself */
            } else if (cls == "EnvironmentMap") {
                /* the MIP pyramid exactly as the plugin built and stores it (half precision, read back as float RGB): level 0
                   drives the illumination (envmap.cpp:516-632), all levels the EWA lookup of directly visible pixels
                   (envmap.cpp:395-407).  EnvironmentMap keeps m_mipmap private: INTEGRATION.md lists the one-line accessor
                   `const MIPMap *getMIPMap() const { return m_mipmap; }` this needs. */
#if defined(PHIP_HAVE_INTERNALS)
                pe.type = PHIP_EMITTER_ENVMAP; pe.shape = 0xFFFFFFFFu;
                const int nLevels = phipEnvMip(e)->getLevels();
                if (nLevels > PHIP_ENVMAP_MAX_LEVELS) SLog(EError, "path_hip: environment map with too many MIP levels");
                m_envLevels.clear();
                for (int l = 0; l < nLevels; ++l) {
                    m_envLevels.push_back(phipEnvMip(e)->toBitmap(l)->convert(Bitmap::ERGB, Bitmap::EFloat32));
                    envmap.levels[l] = m_envLevels[l]->getFloat32Data();
                }
                envmap.n_levels = (uint32_t) nLevels;
                envmap.texels = m_envLevels[0]->getFloat32Data();
                envmap.width = (uint32_t) m_envLevels[0]->getWidth(); envmap.height = (uint32_t) m_envLevels[0]->getHeight();
                envmap.scale = e->getProperties().getFloat("scale", 1.0f);
                const Matrix4x4 tw = e->getWorldTransform()->eval(0).getMatrix();     /* a copy: eval() returns a temporary */
                for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) envmap.to_world[4 * r + c] = tw(r, c);
#else
                SLog(EError, "path_hip: the envmap emitter needs -DPHIP_REFERENCE_SOURCES or the accessor patch of INTEGRATION.md (-DPHIP_REFERENCE_ACCESSORS)");
#endif
            } else {
                SLog(EError, "path_hip: emitter \"%s\" is not supported (area, constant, envmap)", cls.c_str());
            }
            emitters.push_back(pe);
        }

        phip_scene_desc d; memset(&d, 0, sizeof(d));
        d.abi_version = PHIP_ABI_VERSION;
        d.n_vertices = (uint32_t) (positions.size() / 3); d.positions = positions.data(); d.normals = anyNormals ? normals.data() : NULL;
        d.n_triangles = (uint32_t) (indices.size() / 3); d.indices = indices.data();
        d.n_shapes = (uint32_t) shapes.size(); d.shapes = shapes.data();
        d.n_materials = (uint32_t) materials.size(); d.materials = materials.data();
        d.n_emitters = (uint32_t) emitters.size(); d.emitters = emitters.data();
        d.texcoords = anyTexcoords ? texcoords.data() : NULL;
        d.n_textures = (uint32_t) m_textures.size(); d.textures = m_textures.empty() ? NULL : &m_textures[0];
        d.envmap = envmap;

        const Sensor *sensor = scene->getSensor();
        if (sensor->getClass()->getName() != "PerspectiveCameraImpl")
            SLog(EError, "path_hip: only the 'perspective' sensor is supported");
        const PerspectiveCamera *cam = static_cast<const PerspectiveCamera *>(sensor);
        const Matrix4x4 m = cam->getWorldTransform(0).getMatrix();         /* a copy: getWorldTransform(t) returns a temporary */
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) d.camera.to_world[4 * r + c] = m(r, c);
        d.camera.xfov_deg = cam->getXFov(); d.camera.near_clip = cam->getNearClip(); d.camera.far_clip = cam->getFarClip();

        const Film *film = sensor->getFilm();
        d.film.width = film->getSize().x; d.film.height = film->getSize().y;
        d.film.crop_offset_x = film->getCropOffset().x; d.film.crop_offset_y = film->getCropOffset().y;
        d.film.crop_width = film->getCropSize().x; d.film.crop_height = film->getCropSize().y;
        const ReconstructionFilter *rf = film->getReconstructionFilter();
        d.film.filter_radius = rf->getRadius();
        for (int i = 0; i <= PHIP_FILTER_RESOLUTION; ++i)      /* evalDiscretized(x) = m_values[min(int(|x| * res/radius), res)] */
            d.film.filter_table[i] = rf->evalDiscretized((i + 0.5f) * rf->getRadius() / PHIP_FILTER_RESOLUTION);

        if (const char *dump = getenv("PHIP_SHIM_DUMP")) {      /* debugging aid: the flattened description as raw arrays */
            FILE *f = fopen(dump, "wb");
            if (f) {
                uint32_t hdr[6] = { d.n_vertices, d.n_triangles, d.n_shapes, d.n_materials, d.n_emitters, d.n_textures };
                fwrite(hdr, sizeof(hdr), 1, f);
                fwrite(d.positions, sizeof(float), 3 * (size_t) d.n_vertices, f);
                fwrite(normals.data(), sizeof(float), 3 * (size_t) d.n_vertices, f);
                fwrite(d.indices, sizeof(uint32_t), 3 * (size_t) d.n_triangles, f);
                fwrite(d.shapes, sizeof(phip_shape), d.n_shapes, f);
                fwrite(d.materials, sizeof(phip_material), d.n_materials, f);
                fwrite(d.emitters, sizeof(phip_emitter), d.n_emitters, f);
                fwrite(&d.camera, sizeof(d.camera), 1, f);
                fwrite(&d.film, sizeof(d.film), 1, f);
                fclose(f);
            }
        }
        if (m_scene) phip_scene_destroy(m_scene);
        m_scene = phip_scene_create(&d, m_device);
        if (!m_scene)
            SLog(EError, "path_hip: %s", phip_last_error());
    }

    static void rgb(const Spectrum &s, float out[3]) { Float r, g, b; s.toLinearRGB(r, g, b); out[0] = r; out[1] = g; out[2] = b; }

#if defined(PHIP_HAVE_INTERNALS)
    /* <texture type="bitmap">: the RGB MIP pyramid as the plugin built and stores it + the lookup parameters
       (bitmap.cpp: wrapModeU/V, filterType, maxAnisotropy; Texture2D: uscale/vscale/uoffset/voffset) */
    uint32_t convertBitmap(const Texture *tex) {
        std::map<const Texture *, uint32_t>::iterator it = m_textureIds.find(tex);
        if (it != m_textureIds.end()) return it->second;
        phipLayoutProbeBitmap(tex);
        const PhipBitmapInfo info = phipBitmap(tex);
        if (!info.mip) SLog(EError, "path_hip: only RGB bitmap textures are supported");
        phip_texture t; memset(&t, 0, sizeof(t));
        t.n_levels = (uint32_t) info.mip->getLevels();
        for (int l = 0; l < info.mip->getLevels(); ++l) {
            m_textureLevels.push_back(info.mip->toBitmap(l)->convert(Bitmap::ERGB, Bitmap::EFloat32));
            t.levels[l] = m_textureLevels.back()->getFloat32Data();
        }
        t.width = (uint32_t) info.mip->getWidth(); t.height = (uint32_t) info.mip->getHeight();
        t.wrap_u = (uint32_t) info.wrapU; t.wrap_v = (uint32_t) info.wrapV;       /* phip_wrap_mode = EBoundaryCondition, value for value */
        t.filter_type = (uint32_t) info.mip->getFilterType(); t.max_anisotropy = info.maxAnisotropy;
        t.uv_scale[0] = info.uvScale.x; t.uv_scale[1] = info.uvScale.y;
        t.uv_offset[0] = info.uvOffset.x; t.uv_offset[1] = info.uvOffset.y;
        uint32_t id = (uint32_t) m_textures.size();
        m_textures.push_back(t); m_textureIds[tex] = id;
        return id;
    }
#endif

    /* a `bitmap` texture on specularReflectance (a <texture name="specularReflectance"> child; the property is then absent) */
    void specularTexture(const BSDF *bsdf, bool conductor, phip_material &m) {
#if defined(PHIP_HAVE_INTERNALS)
        const Texture *tex = phipSpecularTexture(bsdf, conductor);
        if (tex->getClass()->getName() == "BitmapTexture")
            m.reflectance_texture = 1 + convertBitmap(tex);
        else if (!tex->isConstant())
            SLog(EError, "path_hip: texture \"%s\" is not supported (constant, bitmap)", tex->getClass()->getName().c_str());
#endif
    }

    uint32_t convertBSDF(const BSDF *bsdf, std::vector<phip_material> &materials, std::map<const BSDF *, uint32_t> &ids) {
        std::map<const BSDF *, uint32_t>::iterator it = ids.find(bsdf);
        if (it != ids.end()) return it->second;
        phip_material m; memset(&m, 0, sizeof(m));
        const std::string cls = bsdf->getClass()->getName();
        const Properties &props = bsdf->getProperties();
        Intersection its;       /* constant textures only: any intersection record evaluates to the same value */
        if (cls == "SmoothDiffuse") {
            m.type = PHIP_BSDF_DIFFUSE;
#if defined(PHIP_HAVE_INTERNALS)
            phipLayoutProbeDiffuse(bsdf);
            const Texture *tex = phipReflectanceTexture(bsdf);
            if (tex->getClass()->getName() == "BitmapTexture")
                m.reflectance_texture = 1 + convertBitmap(tex);
            else if (tex->isConstant())
                rgb(bsdf->getDiffuseReflectance(its), m.reflectance);
            else
                SLog(EError, "path_hip: texture \"%s\" is not supported (constant, bitmap)", tex->getClass()->getName().c_str());
#else
            if (bsdf->getType() & BSDF::ESpatiallyVarying)
                SLog(EError, "path_hip: a textured reflectance needs -DPHIP_REFERENCE_SOURCES or the accessor patch of INTEGRATION.md (-DPHIP_REFERENCE_ACCESSORS)");
            rgb(bsdf->getDiffuseReflectance(its), m.reflectance);
#endif
        } else if (cls == "SmoothDielectric") {
#if defined(PHIP_HAVE_INTERNALS)
            phipLayoutProbeDielectric(bsdf);
#endif
            m.type = PHIP_BSDF_DIELECTRIC; m.eta[0] = bsdf->getEta();
            rgb(props.getSpectrum("specularReflectance", Spectrum(1.0f)), m.reflectance);
            specularTexture(bsdf, false, m);
            rgb(props.getSpectrum("specularTransmittance", Spectrum(1.0f)), m.transmittance);
#if defined(PHIP_HAVE_INTERNALS)
            if (const Texture *tex = phipTransmittanceTexture(bsdf)) {       /* a <texture name="specularTransmittance"> child */
                if (tex->getClass()->getName() == "BitmapTexture")
                    m.transmittance_texture = 1 + convertBitmap(tex);
                else if (!tex->isConstant())
                    SLog(EError, "path_hip: texture \"%s\" is not supported (constant, bitmap)", tex->getClass()->getName().c_str());
            }
#endif
        } else if (cls == "RoughConductor") {
#if defined(PHIP_HAVE_INTERNALS)
            phipLayoutProbeRoughConductor(bsdf);
#endif
            m.type = PHIP_BSDF_ROUGHCONDUCTOR;
            /* roughconductor.cpp:176-190: eta / k from data/ior/<material>.{eta,k}.spd unless given explicitly */
            ref<FileResolver> fResolver = Thread::getThread()->getFileResolver();
            std::string material = props.getString("material", "Cu");
            Spectrum intEta, intK;
            if (boost::to_lower_copy(material) == "none") { intEta = Spectrum(0.0f); intK = Spectrum(1.0f); }
            else {
                intEta.fromContinuousSpectrum(InterpolatedSpectrum(fResolver->resolve("data/ior/" + material + ".eta.spd")));
                intK.fromContinuousSpectrum(InterpolatedSpectrum(fResolver->resolve("data/ior/" + material + ".k.spd")));
            }
            Float extEta = lookupIOR(props, "extEta", "air");
            rgb(props.getSpectrum("eta", intEta) / extEta, m.eta); rgb(props.getSpectrum("k", intK) / extEta, m.k);
            /* (BSDF::getSpecularReflectance folds the Fresnel term at its.wi in, roughconductor.cpp:252-257: not the parameter) */
            rgb(props.getSpectrum("specularReflectance", Spectrum(1.0f)), m.reflectance);
            specularTexture(bsdf, true, m);
            MicrofacetDistribution distr(props);
            if (distr.getType() == MicrofacetDistribution::EPhong)
                SLog(EError, "path_hip: the phong/as microfacet distribution is not supported");
            m.distribution = distr.getType() == MicrofacetDistribution::EGGX ? PHIP_MF_GGX : PHIP_MF_BECKMANN;
            m.alpha_u = distr.getAlphaU(); m.alpha_v = distr.getAlphaV(); m.sample_visible = distr.getSampleVisible() ? 1 : 0;
            /* a TEXTURE on alpha / alphaU / alphaV is a child object (roughconductor.cpp:424-431), invisible in the Properties */
            bool roughnessDone = false;
#if defined(PHIP_HAVE_INTERNALS)
            if (const Texture *tu = phipAlphaTexture(bsdf, 0)) {
                const Texture *tv = phipAlphaTexture(bsdf, 1);
                const Texture *axes[2] = { tu, tv };
                uint32_t *ids2[2] = { &m.alpha_u_texture, &m.alpha_v_texture };
                for (int a = 0; a < 2; ++a) {
                    if (axes[a]->getClass()->getName() == "BitmapTexture")
                        *ids2[a] = 1 + convertBitmap(axes[a]);         /* (one object for both axes -> one id: isotropic) */
                    else if (!axes[a]->isConstant())
                        SLog(EError, "path_hip: texture \"%s\" on the roughness is not supported (constant, bitmap)", axes[a]->getClass()->getName().c_str());
                }
                roughnessDone = true;
            }
#endif
            if (!roughnessDone) {
                /* no way in (stock build / an accessor set without getAlphaTexture): a textured roughness must not slip through behind a
                   textured specularReflectance in the ESpatiallyVarying test below.  Probed through the public interface: the roughness
                   must not depend on the texture coordinates. */
                Intersection probe; probe.p = Point(0.0f); probe.geoFrame = probe.shFrame = Frame(Normal(0, 0, 1)); probe.wi = Vector(0, 0, 1);
                probe.hasUVPartials = false; probe.dudx = probe.dudy = probe.dvdx = probe.dvdy = 0;
                const Float uvs[4][2] = { { 0.13f, 0.71f }, { 0.62f, 0.29f }, { 0.91f, 0.87f }, { 0.37f, 0.05f } };
                Float first = 0;
                for (int k = 0; k < 4; ++k) {
                    probe.uv = Point2(uvs[k][0], uvs[k][1]);
                    const Float r = bsdf->getRoughness(probe, 0);
                    if (k == 0) first = r;
                    else if (r != first)
                        SLog(EError, "path_hip: a textured roughness (alpha) on roughconductor needs -DPHIP_REFERENCE_SOURCES or the getAlphaTexture accessor of INTEGRATION.md");
                }
            }
        } else if (cls == "TwoSidedBRDF") {
            /* twosided.cpp keeps its children in m_nestedBRDF[2]; they are reachable as named children */
#if defined(PHIP_HAVE_INTERNALS)
            phipLayoutProbeTwoSided(bsdf);
            std::vector<const BSDF *> nested = getNestedBSDFs(bsdf);
            uint32_t a = convertBSDF(nested[0], materials, ids), b = nested.size() > 1 ? convertBSDF(nested[1], materials, ids) : a;
            m.type = PHIP_BSDF_TWOSIDED; m.nested[0] = a; m.nested[1] = b;
#else
            SLog(EError, "path_hip: the twosided adapter needs -DPHIP_REFERENCE_SOURCES or the accessor patch of INTEGRATION.md (-DPHIP_REFERENCE_ACCESSORS)");
#endif
        } else {
            SLog(EError, "path_hip: BSDF '%s' is outside the supported set (diffuse, dielectric, roughconductor, twosided)", cls.c_str());
        }
        if ((bsdf->getType() & BSDF::ESpatiallyVarying) && m.type != PHIP_BSDF_TWOSIDED
            && !(m.reflectance_texture | m.alpha_u_texture | m.alpha_v_texture | m.transmittance_texture))
            SLog(EError, "path_hip: only reflectance / specularReflectance / specularTransmittance / alpha can be textured (bitmap)");
        materials.push_back(m);
        ids[bsdf] = (uint32_t) materials.size() - 1;
        return ids[bsdf];
    }

#if defined(PHIP_HAVE_INTERNALS)
    static std::vector<const BSDF *> getNestedBSDFs(const BSDF *bsdf) {
        std::vector<const BSDF *> out;
        out.push_back(phipNested(bsdf, 0));
        out.push_back(phipNested(bsdf, 1));       /* twosided.cpp:87-88: the front BRDF again when only one was given */
        return out;
    }
#endif

private:
    phip_scene *m_scene;
    int m_device;
    int m_deviceCount;           /* GPUs the render spreads over (0 = all visible) */
    std::vector<phip_texture> m_textures; std::map<const Texture *, uint32_t> m_textureIds;
    std::vector<ref<Bitmap> > m_textureLevels;   /* float RGB copies of the textures' MIP levels */
    std::vector<ref<Bitmap> > m_envLevels;       /* float RGB copies of the environment map's MIP levels (alive until phip_scene_create) */
};

MTS_NAMESPACE_END
