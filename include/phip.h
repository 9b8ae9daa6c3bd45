/*
 * phip.h -- C ABI of the MI355X-native `path_hip` integrator back end (libphip.so).
 *
 * This is the drop-in boundary: it is exactly what a Mitsuba 0.6
 * integrator plugin `path_hip.so` (class PathHIP : MonteCarloIntegrator, see
 * mitsuba_amd/plugin/path_hip.cpp and INTEGRATION.md) binds in place of the CPU
 * per-block worker loop.  Reference interfaces replaced (file:line under /root/reference):
 *
 *   phip_scene_create   <- Scene::initialize + ShapeKDTree::build     src/librender/scene.cpp:322-384,
 *                                                                      src/librender/skdtree.cpp:68-110
 *   phip_render         <- SamplingIntegrator::render + BlockRenderer::process + renderBlock
 *                          + MIPathTracer::Li + ImageBlock::put        src/librender/integrator.cpp:95-188,
 *                                                                      src/integrators/path/path.cpp:119-300,
 *                                                                      include/mitsuba/render/imageblock.h:103-204
 *   phip_render_device  <- same, result left in device memory so the caller can RCCL-reduce it
 *                          (replaces StreamBackend::sendWorkResult,    src/libcore/sched_remote.cpp:519-532)
 *   phip_render_params.n_devices / devices[]  <- the Scheduler's worker set: one host thread + stream per GPU, the
 *                          reference's spiral block list dealt over them, the per-device films merged by one
 *                          ncclReduce(sum) (replaces BlockedRenderProcess::processResult's film->put under a mutex,
 *                          src/librender/renderproc.cpp:142-149, and sched_remote.cpp:519-532)
 *   phip_render_params.progress  <- ProgressReporter::update / RenderQueue::signalWorkEnd
 *                                                                      src/librender/renderproc.cpp:146-154
 *   phip_trace          <- ShapeKDTree::rayIntersect(ray, its) / (ray) src/librender/skdtree.cpp:112-142,207-226
 *   phip_cancel         <- SamplingIntegrator::cancel                  src/librender/integrator.cpp:90-93
 *   phip_develop        <- HDRFilm::develop weight normalisation       src/libcore/fmtconv.cpp:979-991
 *
 * Conventions: plain C, plain pointers and sizes, no exceptions cross the boundary.  Every
 * function that can fail returns 0 on success and a negative phip_status otherwise;
 * phip_last_error() returns a thread-local human-readable message.  All pointers passed in are
 * borrowed for the duration of the call only (phip_scene_create deep-copies).  All arithmetic on
 * the path is float32 / RGB (the reference's -DSINGLE_PRECISION -DSPECTRUM_SAMPLES=3 build).
 */
#ifndef PHIP_H
#define PHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHIP_ABI_VERSION 7

typedef enum phip_status {
    PHIP_OK              =  0,
    PHIP_ERR_INVALID     = -1,  /* bad argument / malformed scene description            */
    PHIP_ERR_UNSUPPORTED = -2,  /* feature outside the path (e.g. phong microfacets)     */
    PHIP_ERR_DEVICE      = -3,  /* HIP runtime error, no device, kernel failure          */
    PHIP_ERR_CANCELLED   = -4,  /* phip_cancel() was called while rendering              */
    PHIP_ERR_NOMEM       = -5
} phip_status;

/* ---- materials: src/bsdfs/{diffuse,dielectric,roughconductor,twosided}.cpp ---- */
typedef enum phip_bsdf_type {
    PHIP_BSDF_DIFFUSE        = 0,  /* SmoothDiffuse, diffuse.cpp:110-150                 */
    PHIP_BSDF_DIELECTRIC     = 1,  /* SmoothDielectric, dielectric.cpp:217-387           */
    PHIP_BSDF_ROUGHCONDUCTOR = 2,  /* RoughConductor, roughconductor.cpp:253-415         */
    PHIP_BSDF_TWOSIDED       = 3   /* TwoSidedBRDF adapter, twosided.cpp:108-183         */
} phip_bsdf_type;

typedef enum phip_microfacet_type {   /* MicrofacetDistribution::EType, microfacet.h:47-56 */
    PHIP_MF_BECKMANN = 0,
    PHIP_MF_GGX      = 1
} phip_microfacet_type;

typedef struct phip_material {
    uint32_t type;              /* phip_bsdf_type                                          */
    uint32_t nested[2];         /* TWOSIDED: material ids of the front / back BRDF
                                   (equal when only one was nested, twosided.cpp:87-88)  */
    float    reflectance[3];    /* DIFFUSE: reflectance; DIELECTRIC / ROUGHCONDUCTOR:
                                   specularReflectance                                   */
    float    transmittance[3];  /* DIELECTRIC: specularTransmittance                     */
    float    eta[3];            /* DIELECTRIC: eta[0] = intIOR/extIOR (dielectric.cpp:158);
                                   ROUGHCONDUCTOR: eta/extEta as linear RGB              */
    float    k[3];              /* ROUGHCONDUCTOR: k/extEta as linear RGB                */
    float    alpha_u, alpha_v;  /* ROUGHCONDUCTOR roughness (clamped to >= 1e-4 inside)  */
    uint32_t distribution;      /* phip_microfacet_type                                  */
    uint32_t sample_visible;    /* microfacet.h:144, default true                        */
    uint32_t reflectance_texture; /* 0 = the constant `reflectance`, else 1 + id of a `bitmap` texture on DIFFUSE `reflectance`
                                   (diffuse.cpp:110-150) or on DIELECTRIC / ROUGHCONDUCTOR `specularReflectance`
                                   (dielectric.cpp:300, roughconductor.cpp:297-298,373-374): texture->eval(its)  */
    uint32_t alpha_u_texture, alpha_v_texture;   /* ROUGHCONDUCTOR (ABI 5): 0 = the constants alpha_u / alpha_v, else 1 + id of a `bitmap` texture
                                   given as the child "alpha" (then both ids are equal), "alphaU" or "alphaV" (roughconductor.cpp:424-431);
                                   the roughness at a vertex is texture->eval(its).average() (roughconductor.cpp:275-280), clamped to >= 1e-4
                                   (microfacet.h:113-114).  Two different textures make the BSDF anisotropic (roughconductor.cpp:228-229) */
    uint32_t transmittance_texture; /* DIELECTRIC (ABI 5): 0 or 1 + id of a `bitmap` texture on specularTransmittance (dielectric.cpp:301,348) */
} phip_material;

/* ---- `bitmap` texture (src/textures/bitmap.cpp, Texture2D::eval texture.cpp:112-121): an RGB MIP pyramid exactly as
 * the plugin built and stores it (level 0 = the image; every further level, sizes max(1, (n+1)/2) down to 1x1,
 * mipmap.h:182-192 -- filterType nearest / bilinear: level 0 only).  Lookups without UV partials read level 0
 * bilinearly (bitmap.cpp:431-454); the first path vertex has partials (camera-ray differentials,
 * intersection.cpp:5-76) and uses MIPMap::eval (mipmap.h:629-833). ---- */
typedef enum phip_wrap_mode {     /* = ReconstructionFilter::EBoundaryCondition, value for value (rfilter.h:53-64) */
    PHIP_WRAP_CLAMP = 0, PHIP_WRAP_REPEAT = 1, PHIP_WRAP_MIRROR = 2, PHIP_WRAP_ZERO = 3, PHIP_WRAP_ONE = 4
} phip_wrap_mode;
typedef enum phip_filter_type {   /* EMIPFilterType, mipmap.h:40-52 */
    PHIP_FILTER_NEAREST = 0, PHIP_FILTER_BILINEAR = 1, PHIP_FILTER_TRILINEAR = 2, PHIP_FILTER_EWA = 3
} phip_filter_type;
#define PHIP_MIP_MAX_LEVELS 17
typedef struct phip_texture {
    uint32_t width, height;
    uint32_t n_levels;                    /* 1 or the complete pyramid                     */
    const float *levels[PHIP_MIP_MAX_LEVELS];   /* levels[l]: RGB floats, row-major        */
    uint32_t wrap_u, wrap_v;              /* phip_wrap_mode ('wrapModeU/V', default repeat) */
    uint32_t filter_type;                 /* phip_filter_type ('filterType', default EWA)  */
    float    max_anisotropy;              /* 'maxAnisotropy', default 20                   */
    float    uv_scale[2], uv_offset[2];   /* Texture2D 'uscale/vscale', 'uoffset/voffset'  */
} phip_texture;

/* ---- shapes: every shape is a TriMesh (analytic shapes go through Shape::createTriMesh) ---- */
typedef struct phip_shape {
    uint32_t first_vertex, n_vertices;   /* range in positions[] / normals[]              */
    uint32_t first_triangle, n_triangles;/* range in indices[] (indices are GLOBAL vertex ids) */
    uint32_t material;                   /* id into materials[]                           */
    int32_t  emitter;                    /* id into emitters[] or -1                      */
    uint32_t has_normals;                /* 0: shading normal = face normal (skdtree.h:393)*/
    uint32_t has_texcoords;              /* 1: its.uv from texcoords[], dpdu/dpdv from TriMesh::computeUVTangents (trimesh.cpp:683-735) */
} phip_shape;

/* ---- emitters: `area` (src/emitters/area.cpp) and the `constant` environment (src/emitters/constant.cpp).
 * Emitters are listed in the order of Scene::getEmitters() (the selection PDF of scene.cpp:375-381 depends on
 * it).  At most one environment emitter (scene.cpp:510-513).  The bounding sphere of the constant emitter
 * is derived by the library exactly like ConstantBackgroundEmitter::createShape does (constant.cpp:67-72:
 * 1.5 x the bounding sphere of the kd-tree's box expanded by the sensor position); `envmap` has the same sphere. ---- */
enum { PHIP_EMITTER_AREA = 0, PHIP_EMITTER_CONSTANT = 1, PHIP_EMITTER_ENVMAP = 2 };
typedef struct phip_emitter {
    float    radiance[3];
    float    sampling_weight;            /* Emitter::getSamplingWeight, default 1         */
    uint32_t shape;                      /* area: the parent shape (area.cpp:185-203); constant: ignored */
    uint32_t type;                       /* PHIP_EMITTER_*                                 */
    uint32_t reserved[2];
} phip_emitter;

/* ---- sensor: `perspective` pinhole (src/sensors/perspective.cpp:126-180,271-297) ---- */
typedef struct phip_camera {
    float to_world[16];      /* row-major camera-to-world, no scale (perspective.cpp:116-118)  */
    float xfov_deg;          /* horizontal field of view in degrees (sensor.cpp:244-278 resolved)*/
    float near_clip, far_clip;
    float reserved;
} phip_camera;

/* ---- film + reconstruction filter (films/hdrfilm.cpp, libcore/rfilter.cpp:38-57) ---- */
#define PHIP_FILTER_RESOLUTION 31
typedef struct phip_film {
    int32_t width, height;               /* full film size                                */
    int32_t crop_offset_x, crop_offset_y;
    int32_t crop_width, crop_height;     /* the rendered window; output is crop-sized     */
    float   filter_radius;               /* ReconstructionFilter::getRadius               */
    float   filter_table[PHIP_FILTER_RESOLUTION + 1]; /* m_values[], last entry 0         */
} phip_film;

/* ---- `envmap` (src/emitters/envmap.cpp): latitude-longitude radiance map.  `texels` is MIP level 0 exactly as the
 * reference stores it (MIPMap::getArray(): RGB, already rounded to half precision by the plugin) -- the illumination
 * code reads only that level: sampleDirect / pdfDirect (envmap.cpp:516-632) and evalEnvironment for rays without
 * differentials (envmap.cpp:380-394, MIPMap::evalBilinear, mipmap.h:575-596).  Directly visible background pixels
 * (camera rays carry differentials, envmap.cpp:395-407) use the EWA-filtered lookup of MIPMap::eval (mipmap.h:629-833)
 * over the plugin's MIP pyramid: pass every level as the reference built and stored it (`levels`, sizes halve with
 * max(1, (n+1)/2) down to 1x1, mipmap.h:182-192).  With n_levels <= 1 a render that shows the environment needs
 * hideEmitters (those pixels are then never evaluated, path.cpp:139-141) or PHIP_FLAG_ENVMAP_BILINEAR_BACKGROUND. */
#define PHIP_ENVMAP_MAX_LEVELS 17        /* images are smaller than 65536 pixels (envmap.cpp:163-165) */
typedef struct phip_envmap {
    const float *texels;                 /* level 0: height x width x 3 floats (RGB), row 0 = +Y pole; NULL: no envmap */
    uint32_t width, height;
    float    scale;                      /* 'scale' property                               */
    float    to_world[16];               /* row-major emitter-to-world transform ('toWorld') */
    uint32_t n_levels;                   /* 0 / 1: level 0 only; else the complete pyramid  */
    const float *levels[PHIP_ENVMAP_MAX_LEVELS];   /* levels[l], l >= 1: RGB floats of level l ([0] is ignored) */
} phip_envmap;

typedef struct phip_scene_desc {
    uint32_t abi_version;                /* PHIP_ABI_VERSION                              */
    uint32_t n_vertices;
    const float    *positions;           /* 3*n_vertices, world space                     */
    const float    *normals;             /* 3*n_vertices or NULL (then no shape has normals)*/
    uint32_t n_triangles;
    const uint32_t *indices;             /* 3*n_triangles, global vertex ids              */
    uint32_t n_shapes;
    const phip_shape    *shapes;         /* triangle ranges must tile [0,n_triangles) in order */
    uint32_t n_materials;
    const phip_material *materials;
    uint32_t n_emitters;
    const phip_emitter  *emitters;
    phip_camera camera;
    phip_film   film;
    phip_envmap envmap;                  /* used by the emitter of type PHIP_EMITTER_ENVMAP */
    const float *texcoords;              /* 2*n_vertices or NULL (then no shape has texcoords) */
    uint32_t n_textures;
    const phip_texture *textures;
} phip_scene_desc;

/* ---- integrator parameters: MonteCarloIntegrator (src/librender/integrator.cpp:190-225) ---- */
typedef enum phip_sampler_kind {
    PHIP_SAMPLER_CTR = 0     /* counter-based (pixel, sample, dimension) stream -- the parity stream.  `seed` selects the
                                stream: the Mitsuba shim maps the scene's `independent` sampler here (its SFMT stream is
                                a per-worker sequential generator, src/samplers/independent.cpp:71-103, that no parallel
                                schedule can reproduce). */
    , PHIP_SAMPLER_LD = 1    /* (ABI 5) the construction of `ldsampler` (src/samplers/ldsampler.cpp): the first 4 2D requests of a sample
                                (pixel jitter, emitter / BSDF samples of the first vertices) and its first 4 1D requests (Russian
                                roulette) are points of randomly scrambled (0,2)-sequences (core/qmc.h) visited in a random order per
                                pixel and dimension, later requests fall back to the counter stream -- with the scrambles and the
                                order taken from the counter-based generator instead of the worker's sequential Random, so the
                                stream is addressable and reproducible.  The sample count of the whole render (`sample_total`,
                                else `spp`) must be a power of two (ldsampler.cpp:83-87 rounds it up).  `direct`: its sample arrays
                                (direct.cpp:139-146) are one scrambled sequence of sampleCount x N points each, in a random order
                                (ldsampler.cpp:193-197); single shading samples are the sample's next 2D requests. */
    , PHIP_SAMPLER_SOBOL = 2 /* (ABI 6, `path` only) the reference's `sobol` sampler (src/samplers/sobol.cpp): its stream is addressable as it
                                stands -- sample k of pixel (x, y) is point sobol::look_up(m, k, x, y) of the global Sobol' sequence
                                (Gruenschloss' enumeration of elementary intervals, sobolseq.h:94-130), dimension d of that point is
                                sobol::sampleSingle(index, d) (sobolseq.h:42-58), dimensions are consumed in call order (camera sample 0-1,
                                then every 2D request two and every 1D request one; as the plugin stands no 2D request is served from
                                dimension 4, sobol.cpp:241-242 with m_arrayStartDim = m_arrayEndDim = 5: the third 2D request of a sample
                                starts at 5 -- restated for rr_depth >= 2, where the first two requests after the camera sample are 2D;
                                rr_depth 1 is PHIP_ERR_INVALID) -- so the device reproduces it bit for bit, given the
                                plugin's direction numbers as DATA: phip_render_params.sobol_* (the Mitsuba shim reads them out of the
                                loaded plugin, the test harness out of oracle/_ref/plugins/sobol.so or the fixture tests/golden/sobol_tables.npz made from it).
                                Requests beyond sobol_dimensions fall back to the counter stream (the reference stops with an error there).
                                The film's crop window must start at the origin. */
    , PHIP_SAMPLER_STRATIFIED = 3 /* (ABI 6; every integrator since round 5) the construction of `stratified` (src/samplers/stratified.cpp:147-200): the first 4 2D
                                requests of a sample (the pixel jitter is the first) and its first 4 1D requests are jittered points of a
                                res x res (res^2 x 1) grid whose cells the samples of a pixel visit in a random order per dimension, later
                                requests are independent -- with the order a keyed permutation (as PHIP_SAMPLER_LD) and the jitter the
                                counter stream's number for that request, instead of the worker's sequential Random.  The sample count of
                                the render must be a perfect square (stratified.cpp:64-72 rounds it up).  With PHIP_INTEGRATOR_DIRECT a sample array of
                                more than one shading sample per kind is one Latin hypercube over all its entries (stratified.cpp:160-164), single samples
                                are the sample's next 2D requests. */
    , PHIP_SAMPLER_HALTON = 4     /* (ABI 6; `direct` too since round 4) the reference's `halton` sampler as it stands (src/samplers/halton.cpp): sample k of pixel (x, y) is
                                point offset(x mod 128, y mod 128) + stride * k of the Halton sequence (Gruenschloss' enumeration over bases 2 and 3,
                                halton.cpp:244-296: stride = 2^a 3^b, the offset by the Chinese remainder theorem), dimension d is the (scrambled) radical
                                inverse in the d-th prime (qmc.cpp:141-166); dimensions are consumed exactly as by PHIP_SAMPLER_SOBOL (the same bookkeeping,
                                incl. the 2D request that skips dimension 4).  The primes and the digit permutations (Faure's by default, `scramble` = -1;
                                none for 0; pseudorandom ones otherwise) are DATA: phip_render_params.qmc_*. */
    , PHIP_SAMPLER_HAMMERSLEY = 5 /* (ABI 6; `direct` with single shading samples since round 4) the reference's `hammersley` sampler (src/samplers/hammersley.cpp): dimension 0 is index * 1 / (sampleCount
                                resX resY), dimension d > 0 the radical inverse in the (d-1)-th prime, index = offset(x mod 128, y mod 128) + resY * k
                                (hammersley.cpp:181-222); the rest as PHIP_SAMPLER_HALTON.  The sample count is that of the whole render (`sample_total`). */
} phip_sampler_kind;
#define PHIP_SOBOL_MATRIX_SIZE 52    /* words per dimension of sobol::Matrices (sobolseq.h:30) */

/* which SamplingIntegrator::Li the call evaluates */
typedef enum phip_integrator_kind {
    PHIP_INTEGRATOR_PATH = 0,    /* MIPathTracer, src/integrators/path/path.cpp:119-300                                     */
    PHIP_INTEGRATOR_DIRECT = 1,  /* MIDirectIntegrator, src/integrators/direct/direct.cpp:149-312 (max_depth / rr_depth unused) */
    PHIP_INTEGRATOR_VOLPATH_SIMPLE = 2   /* SimpleVolumetricPathTracer on a scene WITHOUT participating media, src/integrators/path/volpath_simple.cpp:88-318: the path tracer
                                            without multiple importance sampling (parameters as PHIP_INTEGRATOR_PATH).  Media are not part of the scene description: the
                                            plugin shim refuses a scene that has any (mitsuba_amd/plugin/volpath_simple_hip.cpp) */
} phip_integrator_kind;

#define PHIP_MAX_DEVICES 16
typedef struct phip_render_params {
    int32_t  spp;                /* sampler sampleCount                                   */
    int32_t  max_depth;          /* -1 = infinite (integrator.cpp:197)                    */
    int32_t  rr_depth;           /* default 5                                             */
    int32_t  strict_normals;     /* default 0                                             */
    int32_t  hide_emitters;      /* default 0                                             */
    int32_t  block_size;         /* Scene::getBlockSize, default 32 (mitsuba.cpp:144)     */
    uint32_t sampler;            /* phip_sampler_kind                                     */
    uint32_t seed;
    /* block sharding: this call renders the blocks whose index in the reference's spiral order
       (imageproc.cpp:43-78) is congruent to shard_index modulo shard_count. 0/1 = whole image. */
    int32_t  shard_index;
    int32_t  shard_count;
    int32_t  device;             /* HIP device ordinal                                    */
    int32_t  flags;              /* PHIP_FLAG_*                                           */
    void    *stream;             /* hipStream_t to launch on, NULL = library-owned stream; must be NULL when n_devices > 1 (PHIP_ERR_INVALID otherwise) */
    uint32_t integrator;         /* phip_integrator_kind                                  */
    int32_t  emitter_samples;    /* `direct`: emitterSamples (direct.cpp:98-99), default 1 */
    int32_t  bsdf_samples;       /* `direct`: bsdfSamples (direct.cpp:100-101), default 1  */
    /* Multi-GPU inside the call (ABI 4).  n_devices <= 1: the scene's device (`device`).  n_devices > 1: the blocks of this
       call's shard are dealt round-robin, in the reference's spiral order, over devices[0..n_devices): one host thread and
       one stream per device, the scene is replicated to a device on first use (phip_scene_replicate does it ahead of time),
       and the per-device (R,G,B,alpha,weight) films are merged on devices[0] with one ncclReduce(sum) (RCCL over xGMI).
       devices[0] must be the scene's device; phip_render_device's buffer lives there.  A device may not be listed twice
       unless PHIP_FLAG_ALIAS_DEVICES is set (test hook for single-GPU boxes: the films are then summed by a kernel). */
    int32_t  n_devices;
    int32_t  devices[PHIP_MAX_DEVICES];
    /* Progressive rendering (ABI 4): this call renders sample indices [sample_offset, sample_offset + spp) of a render of
       sample_total samples per pixel (0 = spp; only the camera-ray differential scale 1/sqrt(sampleCount),
       integrator.cpp:144-145, depends on it); with PHIP_FLAG_ACCUMULATE the result is ADDED to the film left by the
       previous call (phip_render: the library-owned film; phip_render_device: the caller's buffer). */
    int32_t  sample_offset;
    int32_t  sample_total;
    /* Progress (ABI 4): called whenever the job's live-path count is polled (about every 8 wavefront iterations) and at the
       end, with the camera samples finished so far by that device.  With n_devices > 1 the calls come from the library's
       per-device host threads -- never two at a time (they are serialised), but not on the caller's thread.
       May be NULL.  The callback may call phip_cancel. */
    void   (*progress)(void *user, int32_t device, uint64_t samples_done, uint64_t samples_total);
    void    *progress_user;
    /* PHIP_SAMPLER_SOBOL (ABI 6): host pointers, read during the call.  sobol_matrices = sobol::Matrices::matrices32, sobol_dimensions x 52
       words; sobol_vdc / sobol_vdc_inv = rows [m - 1] of sobol::Matrices::vdc_sobol_matrices / _inv (52 words each), m = sobol_log_resolution =
       log2 of the larger side of the crop window rounded up to a power of two (SobolSampler::setFilmResolution, sobol.cpp:147-157; m <= 1:
       no per-pixel enumeration, the tables may be NULL); sobol_scramble = the sampler's m_scramble (0 unless the scene sets `scramble`: then sampleTEA of it, sobol.cpp:92-102). */
    const uint32_t *sobol_matrices;
    const uint64_t *sobol_vdc;
    const uint64_t *sobol_vdc_inv;
    uint32_t sobol_dimensions;
    uint32_t sobol_log_resolution;
    uint64_t sobol_scramble;
    /* PHIP_SAMPLER_HALTON / _HAMMERSLEY (ABI 6): host pointers, read during the call.  qmc_primes = the first qmc_dimensions entries of the
       reference's primeTable (qmc.cpp:27-81); qmc_permutations = the digit permutations of its PermutationStorage (src/samplers/faure.cpp)
       for those bases, concatenated -- the one of dimension d starts at primes[0] + ... + primes[d - 1] -- or NULL for `scramble` = 0. */
    const uint32_t *qmc_primes;
    const uint16_t *qmc_permutations;
    uint32_t qmc_dimensions;
    uint32_t qmc_reserved;
} phip_render_params;

#define PHIP_FLAG_KERNEL_TIMING 1   /* bracket the kernels with hipEvents, fill phip_stats.*_ms */
#define PHIP_FLAG_SAMPLE_BUFFER 2   /* keep per-sample radiance for phip_get_samples (tests)    */
#define PHIP_FLAG_ENVMAP_BILINEAR_BACKGROUND 4   /* directly visible envmap pixels: unfiltered level-0 lookup instead of the reference's EWA filter (deviation!) */
#define PHIP_FLAG_ACCUMULATE 8      /* add to the film of the previous call instead of overwriting it (progressive rendering) */
#define PHIP_FLAG_ALIAS_DEVICES 16  /* devices[] may name one GPU several times (exercises the multi-device path on a 1-GPU box) */
#define PHIP_FLAG_NO_FUSED 32       /* never use the fused single-kernel path (k_mega) nor the one-kernel iterations (k_shade_trace) -- A/B and parity tests of the wavefront kernels on small scenes */
#define PHIP_FLAG_NO_MEGA 64        /* not k_mega, but k_shade_trace where the scene admits it (a small scene with glass / copper runs k_mega since round 5: how the tests still reach
                                       k_shade_trace on it; the kernel's own clients are small scenes with textures or an environment emitter) */
#define PHIP_FLAG_FUSED_ANY 128     /* the fused kernel on EVERY scene it admits (round 6: k_mega walks the 8-wide tree from memory, phip_accel_info.fused_traversal 4 / 5) -- by default
                                       only trees of at most PHIP_FUSED_WIDE_MAX_NODES wide nodes run it, the wavefront kernels are faster beyond; parity tests and A/B */
#define PHIP_FUSED_WIDE_MAX_NODES 4096

typedef struct phip_stats {
    uint64_t samples;                /* camera samples rendered by this call                  */
    uint64_t closest_rays;           /* "Normal rays traced" (skdtree.cpp:46)                 */
    uint64_t shadow_rays;            /* "Shadow rays traced" (skdtree.cpp:47)                 */
    uint64_t path_vertices;          /* sum of path depths, "Average path length" (path.cpp:24)*/
    uint64_t closest_node_visits;    /* accel nodes fetched by closest-hit queries            */
    uint64_t closest_triangle_tests; /* TriAccel tests by closest-hit queries                 */
    uint64_t shadow_node_visits;     /* same for any-hit (shadow) queries                     */
    uint64_t shadow_triangle_tests;
    uint64_t invalid_samples;        /* rejected by the ImageBlock::put validity check        */
    uint32_t iterations;             /* wavefront iterations = launches of each kernel        */
    uint32_t vertex_traced;          /* 1: the iterations ran k_shade_trace -- vertex, shadow ray and next ray of a slot in ONE kernel per iteration (small scenes
                                        that are not k_mega's: <= 64 Wald records, any material); trace / shadow ms are then 0.  (`reserved`, always 0, before round 5) */
    double   render_ms;              /* host wall clock of the call                           */
    double   trace_kernel_ms;        /* sum of HIP-event durations of the closest-hit kernel  */
    double   shadow_kernel_ms;       /* ... of the any-hit kernel                             */
    double   shade_kernel_ms;
    double   film_kernel_ms;
    double   algorithmic_bytes;      /* SURVEY 8(d) bytes of the whole call, from the counters */
    double   trace_kernel_bytes;     /* the closest-hit kernel's share: node + triangle + ray + hit bytes */
    double   fused_kernel_ms;        /* ... of k_mega (scenes that fit LDS: the whole path in one kernel) */
    double   reduce_ms;              /* multi-device: ncclReduce of the films + its synchronisation, host wall clock */
    uint32_t fused;                  /* 1: the call ran the fused kernel (then trace/shadow/shade ms are 0) */
    uint32_t n_devices;              /* devices that rendered (counters are summed over them, *_ms are the maximum) */
    double   d2h_ms;                 /* (ABI 7) phip_render: the film's device-to-host copy, host wall clock; included in render_ms */
} phip_stats;

typedef struct phip_ray  { float o[3]; float mint; float d[3]; float maxt; } phip_ray;
typedef struct phip_hit  { float t, u, v; uint32_t prim; /* global triangle id, 0xFFFFFFFF = miss */ } phip_hit;
#define PHIP_NO_HIT 0xFFFFFFFFu

typedef struct phip_scene phip_scene;   /* opaque */

/* number of HIP devices visible, or a negative phip_status */
int          phip_device_count(void);
const char  *phip_last_error(void);
const char  *phip_version(void);
const char  *phip_build_id(void);         /* hash of the sources + flags the library was built from (mitsuba_amd/_ffi.py: source_id); "unknown-build-id" outside the in-tree build */

/* Flattens the scene, builds the acceleration structure on the host and uploads it to `device`. */
phip_scene  *phip_scene_create(const phip_scene_desc *desc, int device);
void         phip_scene_destroy(phip_scene *scene);

/* Renders the crop window; out_rgbaw = crop_h*crop_w*5 float32 (R,G,B,alpha,weight) on the HOST,
   accumulated filter-weighted sums exactly like the film's ESpectrumAlphaWeight bitmap.
   out_rgbaw may be any host memory.  Pinned memory (phip_host_alloc, hipHostMalloc, a registered range) receives the film in one
   asynchronous copy at the link's rate; pageable memory receives it in chunks through the library's pinned staging buffers. */
int  phip_render(phip_scene *scene, const phip_render_params *params,
                 float *out_rgbaw, phip_stats *out_stats);

/* Same, but d_out_rgbaw is DEVICE memory on params->device (e.g. a torch tensor's data_ptr());
   the buffer is overwritten.  Completion is synchronous on return. */
int  phip_render_device(phip_scene *scene, const phip_render_params *params,
                        void *d_out_rgbaw, phip_stats *out_stats);

/* (ABI 7) A device-resident frame of this scene's crop size -- phip_render_device's output, e.g. after the caller's own RCCL reduce of
   per-process films (bench.py --gpus N under torchrun) -- to host memory, the way phip_render delivers it: one asynchronous copy into pinned
   memory, staged chunks into pageable memory.  The reference's analogue is the master's film->put (src/librender/renderproc.cpp:142-149). */
int  phip_film_to_host(phip_scene *scene, const void *d_rgbaw, float *out_rgbaw);

/* Per-sample radiance of the last render with PHIP_FLAG_SAMPLE_BUFFER: n = crop_w*crop_h*spp
   entries of (R,G,B,alpha) ordered [y][x][sample].  Test/diagnostic hook. */
int  phip_get_samples(phip_scene *scene, float *out_rgba, size_t n_samples);

/* Ray-cast entry (the reference's kdbench / test_kd workload): closest hit into hits[],
   and/or any-hit into occluded[] (either may be NULL).  Host pointers. */
int  phip_trace(phip_scene *scene, const phip_ray *rays, size_t n,
                phip_hit *hits, uint8_t *occluded, phip_stats *out_stats);

/* Replicates the device-resident scene (geometry, acceleration structure, materials, textures) from the scene's device
   to each listed device (device-to-device copies over xGMI), so that a later multi-device render starts immediately. */
int  phip_scene_replicate(phip_scene *scene, const int32_t *devices, int32_t n_devices);

/* Thread-safe and sticky, like Scheduler::cancel on a process (src/libcore/sched.cpp): a running phip_render returns
   PHIP_ERR_CANCELLED; a request that arrives before the render starts cancels that render.  The flag is consumed by the
   call that observes it. */
void phip_cancel(phip_scene *scene);

/* (ABI 7) Page-locked host memory for phip_render's out_rgbaw (hipHostMalloc, portable); NULL + phip_last_error() on failure.
   Replaces nothing in the reference: its film lives in the process that renders (src/librender/film.cpp). */
void *phip_host_alloc(size_t bytes);
void  phip_host_free(void *p);

/* RGB = sum/weight (0 where weight == 0): rgbaw[n*5] -> rgb[n*3]. Host-side helper. */
void phip_develop(const float *rgbaw, size_t n_pixels, float *out_rgb);

/* Acceleration-structure facts for DESIGN.md / bench accounting. */
typedef struct phip_accel_info {
    uint32_t n_nodes, n_leaves, n_triangle_refs, max_depth;
    uint32_t node_bytes, triangle_bytes;
    float    sah_cost;
    float    build_ms;
    uint32_t fits_lds;       /* 1: tree, records, emitter table and materials fit the fused kernel's LDS plan (k_mega) */
    uint32_t fused_traversal; /* how k_mega traverses the scene: 0 = it walks the BVH4, 1 = flat table of leaf boxes, 2 / 3 = packed table with masks of
                                 <= 32 / <= 64 Wald records, tests dealt over the wave (was `reserved`, always 0, before round 5: same layout); round 6, fits_lds = 0:
                                 4 / 5 = k_mega can walk the compressed 8-wide tree from memory (materials in LDS / in memory; PHIP_FLAG_FUSED_ANY) */
} phip_accel_info;
int  phip_scene_accel_info(const phip_scene *scene, phip_accel_info *out);

/* Utilities for callers that fill a phip_scene_desc without Mitsuba at hand (the test harness, bench.py): the table of the reference's default `gaussian` reconstruction
 * filter with the given standard deviation -- (radius, PHIP_FILTER_RESOLUTION + 1 values), rfilter.cpp:38-57 + gaussian.cpp:34-57; the Mitsuba shim copies the scene's own
 * filter instead -- and sizeof of the ABI's structs by index (0 phip_material, 1 phip_shape, 2 phip_emitter, 3 phip_camera, 4 phip_film, 5 phip_scene_desc,
 * 6 phip_render_params, 7 phip_stats, 8 phip_ray, 9 phip_hit, 10 phip_accel_info), for bindings that mirror them (tests/test_abi.py holds the ctypes mirror against it). */
void   phip_gaussian_filter(float stddev, float *radius, float *table);
size_t phip_abi_sizeof(int which);

#ifdef __cplusplus
}
#endif
#endif /* PHIP_H */
