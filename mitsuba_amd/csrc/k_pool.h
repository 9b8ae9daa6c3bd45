/*
 * k_pool.h -- path-pool layout in HBM, slot flags, render constants, per-wave statistics, small device helpers
 * Included by every translation unit through phip_common.h; see the header of phip.hip for the kernel overview.
 */

/* ======================================================================================
 *  device-side state
 * ====================================================================================== */
#define BLOCK 256
#ifndef STACK_DEPTH
#define STACK_DEPTH 24          /* LDS entries per lane (96 B): 6 waves/SIMD fit in 160 KB; deeper entries spill to HBM */
#endif
#ifndef NODE_CACHE_MAX
#define NODE_CACHE_MAX 48            /* BVH4 nodes staged in LDS per block (144 B each): 16 -> 48 measured -2 % traversal time; 64 costs a block of occupancy */
#endif
#ifndef TRI_CACHE_MAX
#define TRI_CACHE_MAX 96        /* triangle records staged in LDS when the whole scene has at most this many */
#endif
#ifndef TRACE_WAVES
#define TRACE_WAVES 6           /* __launch_bounds__ second argument (waves per SIMD) for the traversal kernels */
#endif

enum : uint32_t {
    F_ALIVE = 1u << 16, F_SCATTERED = 1u << 17, F_EMITTED = 1u << 18, F_PREV_DELTA = 1u << 19, F_FIRST = 1u << 20, F_DEAD = 1u << 21, F_FRESH = 1u << 22, F_DYNAMIC = 1u << 23,
    F_REFN_ZERO = 1u << 24,             /* DirectSamplingRecord::refN of the vertex the ray left is zero (BSDF with a back side / transmission) */
    F_NOTRACE = 1u << 25,               /* `direct`: the slot is alive but has no closest-hit query in flight this iteration */
    F_TRACE_MASK = F_ALIVE | F_NOTRACE, /* the traversal kernels trace a slot iff (state & F_TRACE_MASK) == F_ALIVE */
    DEPTH_MASK = 0xFFFFu,
    NS_SHIFT = 26                       /* bits 26..31: non-smooth vertices of the path so far, modulo 64 (call-order parity stream, k_shade.h) */
};

struct PathPool {
    float4 *rayO;     /* o.xyz, mint */
    float4 *rayD;     /* d.xyz, maxt */
    float4 *hit;      /* t, u, v, bits(prim) */
    float4 *thr;      /* throughput rgb, eta */
    float2 *mis;      /* bsdfPdf of the sampled direction, dot(direction, refN): all the emitter-hit MIS term needs (8 B instead of refN + pdf = 16 B) */
    float4 *camHit;   /* `direct` only: the camera ray's hit record, kept while the vertex's BSDF-sampled rays are traced */
    uint4 *info;      /* sampleId, pixel, sampleIndex, - : written when the slot starts a sample, read-only afterwards */
    uint32_t *state;  /* depth | flags: the only per-iteration slot header (4 B instead of rewriting 16 B) */
    float4 *shadow;   /* 3 float4 per entry: (o.xyz,maxt) (d.xyz,0) (contrib.rgb,bits(sampleId)) -- clipped, (o,maxt') (d,mint'), with DevScene::preclip (k_clip.h);
                         block b's entries are compacted at [b*BLOCK, b*BLOCK + shadowCount[b]) */
    uint32_t *shadowCount;            /* per block of BLOCK slots */
    uint32_t *blockDead;              /* per block: every slot is F_DEAD and nothing is queued any more -- the drain phase of a pass skips these blocks */
    unsigned long long *stat;         /* ST_COUNT arrays of nWaves entries */
    uint32_t *spill;                  /* traversal-stack overflow: SPILL_DEPTH entries per lane */
    uint32_t capacity, nWaves;
    uint32_t spillLanes;              /* lanes the spill buffer covers (the host sizes it by its bound on the stack depth: phip.hip) */
};
/* the spill region of a lane of a BVH4 kernel, or NULL when the host's depth bound said that it cannot spill and the buffer does not cover the lane: a push beyond the LDS
   entries then traps (TravStack::push) instead of writing out of bounds */
__device__ __forceinline__ uint32_t *spillOf(const PathPool &P, size_t lane) { return lane < P.spillLanes ? P.spill + lane * 96u /* SPILL_DEPTH */ : nullptr; }

/* The closest-hit record of a slot is (t, u, v, w) with w = bits(prim) | shade class << 30 (PHIP_NO_HIT stays all ones): the ray kernel
 * of the big scenes (k_rays_w) passes on the class it finds in the spare word of the Wald record it hit -- 0 diffuse, 1 rough
 * conductor, 2 dielectric (the heavier of the two sides of a two-sided surface) -- so that k_shade can deal a block's slots to its
 * lanes by BSDF model before it has fetched anything else (k_shade.h).  The other ray kernels leave the class 0. */
#define HIT_CLASS_SHIFT 30
#define HIT_PRIM_MASK 0x3FFFFFFFu
__host__ __device__ __forceinline__ uint32_t hitPrim(uint32_t w) { return w == PHIP_NO_HIT ? w : (w & HIT_PRIM_MASK); }
__host__ __device__ __forceinline__ uint32_t hitClass(uint32_t w) { return w == PHIP_NO_HIT ? 0u : (w >> HIT_CLASS_SHIFT); }

/* Work counters are kept per wave (one owner, plain read-modify-write, no atomics: a single
 * contended word saturates at ~88 atomics/us on MI355X) in SoA arrays stat[k][waveId] and summed
 * by k_reduce_stats when the host wants them. */
enum { ST_CLOSEST_RAYS = 0, ST_NODE, ST_TRI, ST_SHADOW_RAYS, ST_SH_NODE, ST_SH_TRI, ST_VERTICES, ST_SAMPLES, ST_ALIVE, ST_COUNT };

#define EMITTER_LDS_FLOATS 1024      /* emitter table staged in LDS by the shading kernels when it has at most this many floats (4 KB) */
#define MATERIAL_LDS_MAX 48          /* ... and the materials when there are at most this many (4.5 KB) */
#ifndef MEGA_WAVES
#define MEGA_WAVES 4                 /* k_mega: waves per SIMD (= blocks of 256 per CU): 128 VGPRs, no scratch -- with MachineLICM off for that unit (_ffi.py);
                                        with it on the kernel needs 168 VGPRs (3 waves: measured 2020 vs 2171 Msamples/s at 4 waves even with 148 B of scratch) */
#endif
#ifndef MEGA_MAILBOX
#define MEGA_MAILBOX 1               /* k_mega<MM_ALL>, counter stream: a serving wave behind two LDS mailboxes (k_mega.h) */
#endif
#define MB_NS 64u                    /* entries of the S-box (dynamic LDS) and dwords per entry: the host sizes k_mega's launch with them */
#define MB_DW 24u                    /* (22 words of path state + the sample's 64-bit sequence index, which the QMC builds carry along) */
#ifndef MEGA_POOL
#define MEGA_POOL 1                  /* k_mega<.., FLAT >= 4, ..>: one shared task stack per wave (k_wide_wave.h: traceWidePool); the host sizes the launch's LDS by it */
#endif
#ifndef MEGA_WIDE_NODE_CACHE
#define MEGA_WIDE_NODE_CACHE 48u     /* k_mega<.., FLAT >= 4, ..>: top-of-tree nodes (BFS order) a block stages in LDS (80 B each) */
#endif
#define FLAT_LEAVES_MAX 32           /* k_mega: trees of at most this many leaves are traversed as a flat table of leaf boxes (one bit per leaf) */
#define FLAT2_LEAVES_MAX 64          /* ... of the packed table with record masks (flatMode 2 / 3: at most 32 / 64 Wald records, the mask is over records, not leaves) */
#define MEGA_TRISHADE_MAX 96         /* k_mega: shading records staged in LDS (9 KB) */
#define DYN_SHARDS 8                    /* one dynamic-sample counter per XCD-sized group of blocks */
#define DYN_STRIDE 16                   /* unsigned long longs between counters (128 B) */

/* k_mega (k_mega.h): the fused single-kernel path for scenes that fit LDS */
struct MegaParams {
    unsigned long long *nextId;      /* the pass's sample-id counter (zeroed by the host) */
    const int *cancel;               /* host-pinned flag polled when a wave draws a chunk of ids (phip_cancel) */
    unsigned long long *stat;        /* ST_COUNT rows of nWaves entries */
    uint32_t nWaves;
    /* k_mega<.., FLAT >= 4, ..> (the tree in memory: k_wide_wave.h) */
    uint32_t nodeCache;              /* top-of-tree nodes (BFS order) every block stages in LDS */
    uint32_t *spill;                 /* overflow of the group stacks: SPILL_DEPTH words per lane of the grid */
};

struct Counters {
    unsigned long long total[ST_COUNT];   /* written by k_reduce_stats */
};

struct RenderConst {
    unsigned long long totalIds;      /* ids in this pass = nLocalTiles * sppPass * tilePixels */
    uint32_t sppPass, sppFirst;       /* samples in this pass, first sample index of the pass */
    uint32_t sppMagic;                /* min(floor(2^32 / sppPass), 2^32 - 1): division by sppPass = one mulhi + one correction */
    uint32_t tilePixels, tileShift;   /* blockSize^2, log2(blockSize) */
    uint32_t nLocalTiles;
    int maxDepth, rrDepth, strictNormals, hideEmitters;
    uint32_t seed;
    uint32_t sampler, ldMask;         /* phip_sampler_kind; PHIP_SAMPLER_LD: sampleCount - 1 of the whole render (a power of two) */
    SobolTab sobol;                   /* PHIP_SAMPLER_SOBOL: the reference plugin's tables in device memory (dv_math.h) */
    RinvTab rinv;                     /* PHIP_SAMPLER_HALTON / _HAMMERSLEY: primes, permutations, the partition of the sequence over the pixels */
    uint32_t stRes;                   /* PHIP_SAMPLER_STRATIFIED: sqrt of the render's sample count */
    float diffScaleFactor;            /* 1 / sqrt(spp) of the whole render: RayDifferential::scaleDifferential, integrator.cpp:144-145,181 */
    /* `direct` (direct.cpp:130-138): sample counts, MIS fractions and per-sample weights */
    int emitterSamples, bsdfSamples;
    float fracLum, fracBSDF, weightLum, weightBSDF;
    uint32_t envFiltered;             /* camera rays that miss use the envmap's EWA lookup (pyramid present) */
    uint32_t volpath;                 /* PHIP_INTEGRATOR_VOLPATH_SIMPLE: shadeVertex runs volpath_simple's loop (k_shade.h) */
    const uint32_t *tileOrigin;       /* per local tile: x | y << 16 (crop-relative) */
    uint32_t countAlive;              /* this iteration records the number of live slots */
    uint32_t draining;                /* some slots of the pool have died (the host has seen a live count below the capacity): the shading kernels test their block's retired flag before anything else */
    unsigned long long staticIds;     /* ids [0, staticIds) follow the static slot schedule, the rest is handed out dynamically */
    unsigned long long shardIds;      /* dynamic ids per counter shard */
    unsigned long long *dynCounter;   /* DYN_SHARDS counters, one 128-byte line each */
    uint32_t *blockShard;             /* per block: the counter shard it currently draws from */
    float2 *jitter;                   /* sequence samplers (sobol / halton / hammersley): the camera sample's pixel jitter per sample id, written by the kernel that
                                         starts the path and read by the film pass -- re-deriving it there costs a look_up + a 2D number per sample (sobol: 4.6 of
                                         6.7 ms per C2 frame, halton 16.8 of 18.9); NULL: the film pass derives it (the counter stream: one hash) */
};


/* ======================================================================================
 *  small device helpers
 * ====================================================================================== */
__device__ __forceinline__ uint32_t compactBits(uint32_t x) {   /* even bits of x -> low 16 bits */
    x &= 0x55555555u;
    x = (x ^ (x >> 1)) & 0x33333333u;
    x = (x ^ (x >> 2)) & 0x0f0f0f0fu;
    x = (x ^ (x >> 4)) & 0x00ff00ffu;
    x = (x ^ (x >> 8)) & 0x0000ffffu;
    return x;
}
__host__ __device__ __forceinline__ uint32_t spreadBits(uint32_t x) {
    x &= 0x0000ffffu;
    x = (x ^ (x << 8)) & 0x00ff00ffu;
    x = (x ^ (x << 4)) & 0x0f0f0f0fu;
    x = (x ^ (x << 2)) & 0x33333333u;
    x = (x ^ (x << 1)) & 0x55555555u;
    return x;
}

/* ---- the sample stream (HISTORY.md 3.5): what the `c`-th request of a sample returns.  PHIP_SAMPLER_CTR: words of pcg4d blocks;
 *      PHIP_SAMPLER_LD: the first LD_DIMENSIONS 2D requests (the pixel jitter is request 0) and 1D requests of a sample come from
 *      scrambled (0,2)-sequences, later ones from the counter stream -- next1D / next2D of ldsampler.cpp:212-226 ---- */
/* The reference's deterministic sequence samplers (sobol, halton, hammersley) share their consumption: a point index per (pixel, sample), one
   dimension after the other in call order (k_shade.h restates the bookkeeping of sobol.cpp:218-247 = halton.cpp:352-384 = hammersley.cpp:245-280) */
__device__ __forceinline__ bool isSequenceSampler(uint32_t s) { return s == PHIP_SAMPLER_SOBOL || s == PHIP_SAMPLER_HALTON || s == PHIP_SAMPLER_HAMMERSLEY; }
__device__ __forceinline__ uint64_t seqIndex(const RenderConst &rc, uint32_t k, uint32_t px, uint32_t py) {
    return rc.sampler == PHIP_SAMPLER_SOBOL ? sobolSampleIndex(rc.sobol, k, px, py) : rinvSampleIndex(rc.rinv, k, px, py);
}
__device__ __forceinline__ float seqSample(const RenderConst &rc, uint64_t idx, uint32_t dim) {
    return rc.sampler == PHIP_SAMPLER_SOBOL ? sobolSample(rc.sobol, idx, dim) : rinvSample(rc.rinv, idx, dim);
}
/* a 2D request: dimensions dim, dim + 1 of the point */
__device__ __forceinline__ V2 seqSample2(const RenderConst &rc, uint64_t idx, uint32_t dim) {
    float a, b;
    if (rc.sampler == PHIP_SAMPLER_SOBOL) sobolSample2(rc.sobol, idx, dim, a, b); else rinvSample2(rc.rinv, idx, dim, a, b);
    return V2(a, b);
}
/* dimensions the tables hold (hammersley: its dimension d > 0 uses prime d - 1) */
__device__ __forceinline__ uint32_t seqDims(const RenderConst &rc) {
    return rc.sampler == PHIP_SAMPLER_SOBOL ? rc.sobol.dims : rc.rinv.dims + rc.rinv.hammersley;
}

/* QMC (a compile-time switch: the kernels of the metric's configurations are compiled without this code): PHIP_SAMPLER_SOBOL / _HALTON / _HAMMERSLEY / _STRATIFIED */
template <bool QMC = false>
__device__ __forceinline__ V2 streamJitter(const RenderConst &rc, uint32_t pixel, uint32_t k, uint32_t filmWidth = 1u) {
    if (rc.sampler == PHIP_SAMPLER_LD) { float x, y; ldPoint(pixel, k, 0u, rc.seed, rc.ldMask, x, y); return V2(x, y); }
    if (QMC && rc.sampler == PHIP_SAMPLER_SOBOL) {
        float x, y; sobolCameraSample(rc.sobol, k, pixel % filmWidth, pixel / filmWidth, x, y); return V2(x, y);
    }
    if (QMC && (rc.sampler == PHIP_SAMPLER_HALTON || rc.sampler == PHIP_SAMPLER_HAMMERSLEY)) {
        float x, y; rinvCameraSample(rc.rinv, k, pixel % filmWidth, pixel / filmWidth, x, y); return V2(x, y);
    }
    const U4 h = pcg4d(pixel, k, 0, rc.seed);
    if (QMC && rc.sampler == PHIP_SAMPLER_STRATIFIED) {       /* 2D request 0 of the sample (stratified.cpp:177-189) */
        float x, y; stPoint2D(pixel, k, 0u, rc.seed, rc.stRes, u32ToFloat(h.x), u32ToFloat(h.y), x, y); return V2(x, y);
    }
    return V2(u32ToFloat(h.x), u32ToFloat(h.y));
}
/* the camera sample's jitter as the film pass gets it: read back (rc.jitter) or derived again */
template <bool QMC> __device__ __forceinline__ V2 filmJitter(const RenderConst &rc, unsigned long long id, uint32_t pixel, uint32_t k, uint32_t filmWidth) {
    if (QMC && rc.jitter) { const float2 j = rc.jitter[id]; return V2(j.x, j.y); }
    return streamJitter<QMC>(rc, pixel, k, filmWidth);
}

/* `direct`: shading sample i of kind `which` (0: emitter sample, direct.cpp:212-216; 1: BSDF sample, :251-255) of camera sample k.
   PHIP_SAMPLER_CTR: block 1 + i holds (emitter sample i, BSDF sample i).  PHIP_SAMPLER_LD: more than one sample of a kind is a
   requested 2D array (direct.cpp:139-146; the emitter array first), a single one the next 2D request of the sample (the jitter is 0) */
/* QMC, the sequence samplers (sobol.cpp:170-196,226-257 = halton.cpp:274-328,352-384 = hammersley.cpp:206-280): the integrator's requests are, in
   this order, the emitter samples and the BSDF samples (direct.cpp:139-146, 212-216, 251-255).
     more than one sample of a kind = a requested 2D ARRAY: array a (the emitter array first) owns dimensions 5 + 2 a, 6 + 2 a, and its element
       k * count + i is THE POINT OF SAMPLE k * count + i OF THE PIXEL in those dimensions (generate(): look_up(j, pixel) / offset + j * stride) --
       hammersley refuses arrays (hammersley.cpp:293-300), so does validateParams;
     a single sample = the sample's next 2D request: the emitter sample at dimensions (2, 3) behind the camera sample; the BSDF sample at (2, 3)
       when the emitter samples are an array, else at (5, 6) -- next2D() never hands out dimension 4 (`m_dimension + 1 >= m_arrayStartDim &&
       m_dimension < m_arrayEndDim` with the arrays' range [5, 5) when none is requested: the skip k_shade.h restates for `path`).
   A single request is made even when it is not used (zero samples of that kind, a BSDF without smooth component). */
template <bool QMC = false>
__device__ __forceinline__ V2 streamDirectSample(const RenderConst &rc, uint32_t pixel, uint32_t k, int which, uint32_t i, uint32_t filmWidth = 1u) {
    if (QMC && isSequenceSampler(rc.sampler)) {
        const uint32_t E = (uint32_t) rc.emitterSamples, B = (uint32_t) rc.bsdfSamples, count = which ? B : E;
        const uint32_t px = pixel % filmWidth, py = pixel / filmWidth;
        uint32_t dim; uint64_t idx;
        if (count > 1u) {
            dim = 5u + 2u * (which ? (E > 1u ? 1u : 0u) : 0u);
            const uint32_t j = k * count + i;
            idx = rc.sampler == PHIP_SAMPLER_SOBOL ? sobolLookUp(rc.sobol, j, px, py) : rinvSampleIndex(rc.rinv, j, px, py);
        } else {
            dim = which == 0 ? 2u : (E > 1u ? 2u : 5u);
            idx = seqIndex(rc, k, px, py);
        }
        if (dim + 1u < seqDims(rc)) return seqSample2(rc, idx, dim);
        /* (beyond the tables -- the reference stops with an error there -- the counter stream below) */
    }
    if (rc.sampler == PHIP_SAMPLER_LD) {
        const uint32_t E = (uint32_t) rc.emitterSamples, B = (uint32_t) rc.bsdfSamples, count = which ? B : E;
        float x, y;
        if (count > 1) ldArrayPoint(pixel, which ? (E > 1 ? 1u : 0u) : 0u, (k & rc.ldMask) * count + i, (rc.ldMask + 1u) * count, rc.seed, x, y);
        else ldPoint(pixel, k, 2u * (which ? (E > 1 ? 1u : 2u) : 1u), rc.seed, rc.ldMask, x, y);
        return V2(x, y);
    }
    const U4 h = pcg4d(pixel, k, 1 + i, rc.seed);
    const V2 u = which ? V2(u32ToFloat(h.z), u32ToFloat(h.w)) : V2(u32ToFloat(h.x), u32ToFloat(h.y));
    if (QMC && rc.sampler == PHIP_SAMPLER_STRATIFIED) {
        /* round 5: `stratified` with `direct`.  More than one sample of a kind: a requested array = one Latin hypercube over its stRes^2 * count entries
           (stArrayPoint); a single one: the sample's next 2D request (the camera sample was request 0) -- one cell of the stRes x stRes grid, as for `path` */
        const uint32_t E = (uint32_t) rc.emitterSamples, B = (uint32_t) rc.bsdfSamples, count = which ? B : E, n = rc.stRes * rc.stRes;
        float x, y;
        if (count > 1) stArrayPoint(pixel, which ? (E > 1 ? 1u : 0u) : 0u, (k % n) * count + i, n * count, rc.seed, u.x, u.y, x, y);
        else stPoint2D(pixel, k, which ? (E > 1 ? 1u : 2u) : 1u, rc.seed, rc.stRes, u.x, u.y, x, y);
        return V2(x, y);
    }
    return u;
}

/* sample id -> (local tile, sample-in-pass, pixel); ids are tile-major, then sample, then the
   Morton index of the pixel inside the tile so that a wave covers an 8x8 pixel patch */
__device__ __forceinline__ bool decodeId(const RenderConst &rc, const DevFilm &film, unsigned long long id,
                                         uint32_t &px, uint32_t &py, uint32_t &k) {
    const uint32_t m = (uint32_t) (id & (rc.tilePixels - 1));
    const uint32_t r = (uint32_t) (id >> (2 * rc.tileShift));   /* ids of a pass are < 2^32 */
    uint32_t tile = __umulhi(r, rc.sppMagic);                  /* floor(r / sppPass) or one less */
    k = r - tile * rc.sppPass;
    if (k >= rc.sppPass) { k -= rc.sppPass; ++tile; }
    const uint32_t org = rc.tileOrigin[tile];
    px = (org & 0xFFFFu) + compactBits(m);
    py = (org >> 16) + compactBits(m >> 1);
    k += rc.sppFirst;
    return px < (uint32_t) film.width && py < (uint32_t) film.height;
}

/* per-wave statistics slot: wave-reduce v, lane 0 accumulates into stat[k][waveId] (unique owner).  The accumulation is an atomic add WITHOUT
   return value -- not for exclusion (the slot has one owner) but because it is fire-and-forget: `*p += v` is load - add - store, and a wave
   of a shading kernel that ends on three of them waits three memory round trips before it frees its registers and LDS */
__device__ __forceinline__ void waveStat(const PathPool &P, int k, uint32_t waveId, unsigned long long v, bool overwrite = false) {
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off);
    if (__lane_id() == 0) {
        unsigned long long *p = P.stat + (size_t) k * P.nWaves + waveId;
        if (overwrite) *p = v; else if (v) (void) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

