/*
 * phip.hip -- MI355X (gfx950) wavefront path tracer behind the C ABI of include/phip.h.
 *
 * Replaces the reference's per-block CPU loop (SamplingIntegrator::renderBlock ->
 * MIPathTracer::Li, src/librender/integrator.cpp:140-188, src/integrators/path/path.cpp:119-300)
 * with a slot-stable wavefront: a pool of path slots lives in HBM as SoA arrays; every iteration
 * runs   shade -> shadow -> trace   over the pool.
 *
 *   k_shade   one lane per slot.  Consumes the closest-hit record of the slot's current ray:
 *             emitter-hit MIS term, Russian roulette, then at the new vertex emission, NEE
 *             sample (emits a self-contained shadow-queue entry), BSDF sample -> next ray.
 *             When a path ends the SAME lane immediately regenerates a new camera path from a
 *             global sample counter (wave-aggregated atomic), so the pool stays full until the
 *             image runs out of samples.  Radiance is accumulated per sample in a sample buffer
 *             L[sampleId] (float4), so a finished slot has nothing to flush.
 *   k_shadow  any-hit traversal over the compacted shadow queue; unoccluded entries add their
 *             contribution to L[sampleId].
 *   k_trace   closest-hit traversal for every live slot -> hit record.
 *   k_film    per-pixel gather of the filtered samples (ImageBlock::put semantics,
 *             include/mitsuba/render/imageblock.h:124-204) -- no float atomics, deterministic.
 *
 * Traversal: BVH2 with 64-byte nodes (bvh.h), per-lane stack in LDS (interleaved so that lane i
 * owns bank i), Wald triangle test with the reference's arithmetic.  Not MFMA work: irregular,
 * latency/HBM bound.
 *
 * This file is the product; it never includes, links or calls anything under oracle/.
 */
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <mutex>

#include "../../include/phip.h"
#include "dv_scene.h"
#include "bvh.h"

using namespace pt;

/* ======================================================================================
 *  error handling
 * ====================================================================================== */
static thread_local std::string g_err;
static int setErr(int code, const std::string &msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess)                                                                    \
            throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e__));         \
    } while (0)

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
    DEPTH_MASK = 0xFFFFu
};

struct PathPool {
    float4 *rayO;     /* o.xyz, mint */
    float4 *rayD;     /* d.xyz, maxt */
    float4 *hit;      /* t, u, v, bits(prim) */
    float4 *thr;      /* throughput rgb, eta */
    float2 *mis;      /* bsdfPdf of the sampled direction, dot(direction, refN): all the emitter-hit MIS term needs (8 B instead of refN + pdf = 16 B) */
    uint4 *info;      /* sampleId, pixel, sampleIndex, - : written when the slot starts a sample, read-only afterwards */
    uint32_t *state;  /* depth | flags: the only per-iteration slot header (4 B instead of rewriting 16 B) */
    float4 *shadow;   /* 3 float4 per entry: (o.xyz,maxt) (d.xyz,bits(sampleId)) (contrib.rgb,0);
                         block b's entries are compacted at [b*BLOCK, b*BLOCK + shadowCount[b]) */
    uint32_t *shadowCount;            /* per block of BLOCK slots */
    uint32_t *blockDead;              /* per block: every slot is F_DEAD and nothing is queued any more -- the drain phase of a pass skips these blocks */
    unsigned long long *stat;         /* ST_COUNT arrays of nWaves entries */
    uint32_t *spill;                  /* traversal-stack overflow: SPILL_DEPTH entries per lane */
    uint2 *spill8;                    /* group kernels: SPILL8 entries per ray group (8 groups per wave) */
    uint32_t capacity, nWaves;
};

/* Work counters are kept per wave (one owner, plain read-modify-write, no atomics: a single
 * contended word saturates at ~88 atomics/us on MI355X) in SoA arrays stat[k][waveId] and summed
 * by k_reduce_stats when the host wants them. */
enum { ST_CLOSEST_RAYS = 0, ST_NODE, ST_TRI, ST_SHADOW_RAYS, ST_SH_NODE, ST_SH_TRI, ST_VERTICES, ST_SAMPLES, ST_ALIVE, ST_COUNT };

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
    const uint32_t *tileOrigin;       /* per local tile: x | y << 16 (crop-relative) */
    uint32_t countAlive;              /* this iteration records the number of live slots */
    unsigned long long staticIds;     /* ids [0, staticIds) follow the static slot schedule, the rest is handed out dynamically */
    unsigned long long shardIds;      /* dynamic ids per counter shard */
    unsigned long long *dynCounter;   /* DYN_SHARDS counters, one 128-byte line each */
    uint32_t *blockShard;             /* per block: the counter shard it currently draws from */
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

/* per-wave statistics slot: wave-reduce v, lane 0 accumulates into stat[k][waveId] (unique owner) */
__device__ __forceinline__ void waveStat(const PathPool &P, int k, uint32_t waveId, unsigned long long v, bool overwrite = false) {
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off);
    if (__lane_id() == 0) {
        unsigned long long *p = P.stat + (size_t) k * P.nWaves + waveId;
        if (overwrite) *p = v; else if (v) *p += v;
    }
}

/* ======================================================================================
 *  BVH traversal (closest / any hit)
 * ====================================================================================== */
struct TravResult { float t, u, v; uint32_t prim; };

/* scene-box clip + adaptive epsilon, src/librender/skdtree.cpp:112-142 (closest) / :207-226 (shadow) */
template <bool SHADOW>
__device__ __forceinline__ bool clipToScene(const DevScene &S, const V3 &o, const V3 &d, float rayMint, float rayMaxt,
                                            float &mint, float &maxt) {
    float nearT = -INFINITY, farT = INFINITY;
    const float oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float origin = oo[i], minVal = S.sceneMin[i], maxVal = S.sceneMax[i];
        if (dd[i] == 0) {
            if (origin < minVal || origin > maxVal) return false;
        } else {
            const float rcp = 1.0f / dd[i];
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
        if (!SHADOW) m = smax(m, PT_EPSILON);
        rayMinT *= m;
    }
    if (rayMinT > mint) mint = rayMinT;
    if (rayMaxt < maxt) maxt = rayMaxt;
    return maxt > mint;
}

/* Per-lane traversal stack: the first `depth` entries live in LDS (interleaved: entry e of lane l at
 * lds[e * BLOCK + l], so lane i always hits bank i), deeper entries spill to a per-lane HBM array.
 * The same dynamic LDS segment also stages the top of the tree: the first S.nodeCache BVH4 nodes (they
 * are stored in breadth-first order, so these are the levels every ray visits) and, for small scenes,
 * all triangle records.  Cached nodes use a 144-byte stride so that lanes reading different nodes hit
 * different banks with ds_read_b128. */
#define NODE_LDS_STRIDE 9               /* float4 per cached node (8 + 1 pad) */
/* LDS pointers carry their address space in the type: through a generic pointer the compiler emits flat_load for the cached
   nodes/records, which goes through the texture addresser (16 clk per 16-byte wave instruction, shared by the CU's four SIMDs)
   instead of the LDS pipe (ds_read_b128) -- on the Cornell box, where everything is cached, that was the bottleneck. */
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef float f4v __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) f4v lds_cf4;
__device__ __forceinline__ float4 ldsLoad4(lds_cf4 *p) { const f4v v = *p; return make_float4(v.x, v.y, v.z, v.w); }
struct TravStack {
    lds_u32 *lds;           /* lds base + threadIdx.x */
    uint32_t *spill;        /* global: SPILL_DEPTH entries per lane */
    lds_cf4 *nodes;         /* LDS copy of nodes [0, nodeCache) */
    lds_cf4 *tris;          /* LDS copy of triangle records [0, triCache) */
    uint32_t nodeCache, triCache;
    int depth, sp;
    __device__ __forceinline__ void push(uint32_t v) {
        if (sp < depth) lds[sp * BLOCK] = v; else spill[sp - depth] = v;
        ++sp;
    }
    __device__ __forceinline__ uint32_t pop() {
        --sp;
        return sp < depth ? lds[sp * BLOCK] : spill[sp - depth];
    }
};

/* carve the block's dynamic LDS and stage the cached geometry (all threads of the block must call) */
__device__ __forceinline__ void setupTraversal(const DevScene &S, unsigned char *smem, uint32_t *spill, TravStack &stk) {
    uint32_t *stack = (uint32_t *) smem;
    float4 *ln = (float4 *) (smem + (size_t) S.stackDepth * BLOCK * sizeof(uint32_t));
    float4 *lt = ln + (size_t) S.nodeCache * NODE_LDS_STRIDE;
    for (uint32_t i = threadIdx.x; i < S.nodeCache * 8u; i += BLOCK)
        ln[(i >> 3) * NODE_LDS_STRIDE + (i & 7u)] = S.nodes[i];
    for (uint32_t i = threadIdx.x; i < S.triCache * 3u; i += BLOCK)
        lt[i] = S.tris[i];
    __syncthreads();
    stk.lds = (lds_u32 *) (stack + threadIdx.x); stk.spill = spill; stk.nodes = (lds_cf4 *) ln; stk.tris = (lds_cf4 *) lt;
    stk.nodeCache = S.nodeCache; stk.triCache = S.triCache; stk.depth = (int) S.stackDepth; stk.sp = 0;
}

/* TYPED (a constant in the scope of the caller): true = separate LDS (ds_read_b128) and global paths -- right when (almost)
   everything is cached (small scenes); false = one flat_load path with a selected address -- fewer registers and no
   divergence when most lanes read global memory (big scenes; measured 1-3 % faster there, 14 % slower on the Cornell box) */
#define LOAD_NODE(stack, S, cur, mnx, mny, mnz, mxx, mxy, mxz, chf)                                   \
    float4 mnx, mny, mnz, mxx, mxy, mxz, chf;                                                         \
    if (TYPED) {                                                                                      \
        if ((uint32_t) (cur) < (stack).nodeCache) {                                                   \
            lds_cf4 *n_ = (stack).nodes + (uint32_t) (cur) * NODE_LDS_STRIDE;                         \
            mnx = ldsLoad4(n_); mny = ldsLoad4(n_ + 1); mnz = ldsLoad4(n_ + 2); mxx = ldsLoad4(n_ + 3); \
            mxy = ldsLoad4(n_ + 4); mxz = ldsLoad4(n_ + 5); chf = ldsLoad4(n_ + 6);                   \
        } else {                                                                                      \
            const float4 *n_ = (S).nodes + 8 * (size_t) (cur);                                        \
            mnx = n_[0]; mny = n_[1]; mnz = n_[2]; mxx = n_[3]; mxy = n_[4]; mxz = n_[5]; chf = n_[6]; \
        }                                                                                             \
    } else {                                                                                          \
        const float4 *n_ = (uint32_t) (cur) < (stack).nodeCache                                       \
            ? (const float4 *) ((stack).nodes + (uint32_t) (cur) * NODE_LDS_STRIDE) : (S).nodes + 8 * (size_t) (cur); \
        mnx = n_[0]; mny = n_[1]; mnz = n_[2]; mxx = n_[3]; mxy = n_[4]; mxz = n_[5]; chf = n_[6];    \
    }
#define LOAD_TRI(stack, S, idx, a, b, c)                                                              \
    float4 a, b, c;                                                                                   \
    if (TYPED) {                                                                                      \
        if ((uint32_t) (idx) < (stack).triCache) {                                                    \
            lds_cf4 *t_ = (stack).tris + 3 * (uint32_t) (idx); a = ldsLoad4(t_); b = ldsLoad4(t_ + 1); c = ldsLoad4(t_ + 2); \
        } else {                                                                                      \
            const float4 *t_ = (S).tris + 3 * (size_t) (idx); a = t_[0]; b = t_[1]; c = t_[2];        \
        }                                                                                             \
    } else {                                                                                          \
        const float4 *t_ = (uint32_t) (idx) < (stack).triCache ? (const float4 *) ((stack).tris + 3 * (uint32_t) (idx)) : (S).tris + 3 * (size_t) (idx); \
        a = t_[0]; b = t_[1]; c = t_[2];                                                              \
    }
#define SPILL_DEPTH 96

__device__ __forceinline__ void cswap(float &ka, uint32_t &ra, float &kb, uint32_t &rb) {
    const bool sw = kb < ka;
    const float k0 = sw ? kb : ka, k1 = sw ? ka : kb;
    const uint32_t r0 = sw ? rb : ra, r1 = sw ? ra : rb;
    ka = k0; kb = k1; ra = r0; rb = r1;
}

#ifndef SHADOW_ATOMIC_COMMIT
#define SHADOW_ATOMIC_COMMIT 0      /* measured: bit-identical, +1..3 % on the 250k-triangle scenes but the Cornell shadow kernel doubles (1.3 G 4-byte L2 atomics per frame) */
#endif
#ifndef SHADOW_UNSORTED
#define SHADOW_UNSORTED 1
#endif
#define DONE_REF ((int32_t) 0x80000000)   /* 'no more nodes' marker; as a leaf reference it would need 2^28 triangle records */

/* One BVH4 node step: slab test of the four children, nearest-first order, push the farther hits,
   continue with the nearest (or pop).  Shared by the per-slot and the persistent kernels. */
#define NODE_STEP(stack, S, cur, rcp, ordr, mint, maxt, nodeVisits)                                       \
    {                                                                                                     \
        LOAD_NODE(stack, S, cur, mnx, mny, mnz, mxx, mxy, mxz, chf)                                       \
        ++nodeVisits;                                                                                     \
        float key[4]; uint32_t ref[4];                                                                    \
        SLAB(0, x) SLAB(1, y) SLAB(2, z) SLAB(3, w)                                                       \
        /* nearest child first (also a good any-hit order); misses (INFINITY) sort to the end */         \
        cswap(key[0], ref[0], key[1], ref[1]); cswap(key[2], ref[2], key[3], ref[3]);                     \
        cswap(key[0], ref[0], key[2], ref[2]); cswap(key[1], ref[1], key[3], ref[3]);                     \
        cswap(key[1], ref[1], key[2], ref[2]);                                                            \
        if (key[0] < INFINITY) {                                                                          \
            if (stack.sp + 3 <= stack.depth) {      /* branch-free pushes: hits are a prefix of the sorted keys */ \
                stack.lds[stack.sp * BLOCK] = ref[3]; stack.sp += key[3] < INFINITY ? 1 : 0;              \
                stack.lds[stack.sp * BLOCK] = ref[2]; stack.sp += key[2] < INFINITY ? 1 : 0;              \
                stack.lds[stack.sp * BLOCK] = ref[1]; stack.sp += key[1] < INFINITY ? 1 : 0;              \
            } else {                                                                                      \
                if (key[3] < INFINITY) stack.push(ref[3]);                                                \
                if (key[2] < INFINITY) stack.push(ref[2]);                                                \
                if (key[1] < INFINITY) stack.push(ref[1]);                                                \
            }                                                                                             \
            cur = (int32_t) ref[0];                                                                       \
        } else {                                                                                          \
            cur = stack.sp == 0 ? DONE_REF : (int32_t) stack.pop();                                       \
        }                                                                                                 \
    }
/* Any-hit variant: the visiting order of the children does not matter for an unoccluded ray (all of them are
   visited) -- no sorting network.  Branch-free: every hit child is written at the current stack top, the top only
   advances once a later hit shows that the entry has to be kept; the last hit child becomes the next node. */
#define NODE_STEP_ANY(stack, S, cur, rcp, ordr, mint, maxt, nodeVisits)                                   \
    {                                                                                                     \
        LOAD_NODE(stack, S, cur, mnx, mny, mnz, mxx, mxy, mxz, chf)                                       \
        ++nodeVisits;                                                                                     \
        float key[4]; uint32_t ref[4];                                                                    \
        SLAB(0, x) SLAB(1, y) SLAB(2, z) SLAB(3, w)                                                       \
        const bool h0 = key[0] < INFINITY, h1 = key[1] < INFINITY, h2 = key[2] < INFINITY, h3 = key[3] < INFINITY; \
        if (h0 || h1 || h2 || h3) {                                                                       \
            uint32_t nxt = ref[0]; bool have = h0;                                                        \
            if (stack.sp + 3 <= stack.depth) {                                                            \
                stack.lds[stack.sp * BLOCK] = nxt; stack.sp += (h1 && have) ? 1 : 0; nxt = h1 ? ref[1] : nxt; have = have || h1; \
                stack.lds[stack.sp * BLOCK] = nxt; stack.sp += (h2 && have) ? 1 : 0; nxt = h2 ? ref[2] : nxt; have = have || h2; \
                stack.lds[stack.sp * BLOCK] = nxt; stack.sp += (h3 && have) ? 1 : 0; nxt = h3 ? ref[3] : nxt;                     \
            } else {                                                                                      \
                if (h1) { if (have) stack.push(nxt); nxt = ref[1]; have = true; }                         \
                if (h2) { if (have) stack.push(nxt); nxt = ref[2]; have = true; }                         \
                if (h3) { if (have) stack.push(nxt); nxt = ref[3]; have = true; }                         \
            }                                                                                             \
            cur = (int32_t) nxt;                                                                          \
        } else {                                                                                          \
            cur = stack.sp == 0 ? DONE_REF : (int32_t) stack.pop();                                       \
        }                                                                                                 \
    }
#define SLAB(K, C)                                                                                        \
    {                                                                                                     \
        const float x0 = fmaf(mnx.C, rcp.x, -ordr.x), x1 = fmaf(mxx.C, rcp.x, -ordr.x);                   \
        const float y0 = fmaf(mny.C, rcp.y, -ordr.y), y1 = fmaf(mxy.C, rcp.y, -ordr.y);                   \
        const float z0 = fmaf(mnz.C, rcp.z, -ordr.z), z1 = fmaf(mxz.C, rcp.z, -ordr.z);                   \
        const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), mint));          \
        const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), maxt));          \
        key[K] = (tn <= tf) ? tn : INFINITY;                                                              \
        ref[K] = pm_to_bits(chf.C);                                                                       \
    }

/* Traversal as a per-lane state machine whose loop body is ONE node step and ONE triangle test: a lane
 * inside a leaf tests one Wald record per iteration while its neighbours go on with node
 * steps.  (Looping over the whole leaf inside the body made every lane of the wave wait for up to eight
 * triangle tests per iteration although only ~15 % of the lanes sit in a leaf: measured 2x the issue slots.)
 * The order in which a ray tests its triangles is unchanged, hence so are the results. */
template <bool SHADOW>
__device__ __forceinline__ bool traverse(const DevScene &S, const V3 &o, const V3 &d, float mint, float maxt,
                                         TravStack &stack, TravResult &res,
                                         uint32_t &nodeVisits, uint32_t &triTests) {
    constexpr bool TYPED = true;
    /* reciprocal direction for the slab tests (conservative: boxes are padded); the Wald test uses o,d */
    const V3 rcp(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const V3 ordr(o.x * rcp.x, o.y * rcp.y, o.z * rcp.z);
    stack.sp = 0;
    int32_t cur = S.rootRef;
    bool found = false;
    res.prim = PHIP_NO_HIT; res.t = INFINITY; res.u = res.v = 0;
    while (cur != DONE_REF) {
        if (cur >= 0) {
            if (SHADOW && SHADOW_UNSORTED) NODE_STEP_ANY(stack, S, cur, rcp, ordr, mint, maxt, nodeVisits)
            else NODE_STEP(stack, S, cur, rcp, ordr, mint, maxt, nodeVisits)
        }
        if (cur < 0 && cur != DONE_REF) {
            /* a leaf reference doubles as the lane's progress inside the leaf: ~((next record << 3) | records left - 1) */
            const uint32_t r = ~(uint32_t) cur, idx = r >> 3, left = r & 7u;
            LOAD_TRI(stack, S, idx, a, b, c)
            ++triTests;
            float tu, tv, tt;
            if (waldIntersect(a, b, c, o, d, mint, maxt, tu, tv, tt)) {
                if (SHADOW) return true;
                maxt = tt; res.t = tt; res.u = tu; res.v = tv; res.prim = pm_to_bits(c.z);
                found = true;
            }
            cur = left ? (int32_t) ~(((idx + 1u) << 3) | (left - 1u)) : (stack.sp == 0 ? DONE_REF : (int32_t) stack.pop());
        }
    }
    return found;
}

/* ======================================================================================
 *  Persistent per-lane traversal: a fixed grid of resident waves walks the whole ray pool.
 *  A lane that finishes its ray (or finds its slot dead) is refilled from the wave's own
 *  statically strided share of the pool as soon as REFILL_LANES lanes are idle, so the wave
 *  does not wait for its slowest ray ("while-while" + dynamic fetch, but without any global
 *  atomic: the share of wave w is chunks w, w+W, w+2W, ...).
 * ====================================================================================== */
#ifndef REFILL_LANES
#define REFILL_LANES 16
#endif
#define INVALID_RAY 0xFFFFFFFFu
#define DYN_SHARDS 8                    /* one dynamic-sample counter per XCD-sized group of blocks */
#define DYN_STRIDE 16                   /* unsigned long longs between counters (128 B) */

template <bool SHADOW, bool TYPED, typename Source>
__device__ __forceinline__ void persistentTraverse(const DevScene &S, TravStack &stack, Source &src,
                                                   uint32_t &nodeVisits, uint32_t &triTests, uint32_t &raysTraced) {
    bool active = false;
    uint32_t handle = INVALID_RAY;
    V3 o(0.0f), d(0.0f), rcp(0.0f), ordr(0.0f);
    float mint = 0, maxt = 0;
    int32_t cur = 0;
    TravResult res; res.prim = PHIP_NO_HIT; res.t = INFINITY; res.u = res.v = 0;

    for (;;) {
        const unsigned long long idle = __ballot(!active);
        if (idle && src.more() && (__popcll(idle) >= REFILL_LANES || idle == ~0ull)) {
            const uint32_t h = src.assign(!active, idle);
            if (!active && h != INVALID_RAY) {
                float rmint, rmaxt;
                if (src.load(h, o, d, rmint, rmaxt)) {
                    ++raysTraced;
                    res.prim = PHIP_NO_HIT; res.t = INFINITY; res.u = res.v = 0;
                    if (clipToScene<SHADOW>(S, o, d, rmint, rmaxt, mint, maxt)) {
                        rcp = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                        ordr = V3(o.x * rcp.x, o.y * rcp.y, o.z * rcp.z);
                        cur = S.rootRef; stack.sp = 0; handle = h; active = true;
                    } else {
                        src.commit(h, false, res);
                    }
                }
            }
        }
        if (!__any(active)) { if (!src.more()) break; continue; }
        if (active) {
            /* one node step and one triangle test per iteration (see traverse()) */
            for (;;) {
                if (cur >= 0) {
                    if (SHADOW && SHADOW_UNSORTED) NODE_STEP_ANY(stack, S, cur, rcp, ordr, mint, maxt, nodeVisits)
                    else NODE_STEP(stack, S, cur, rcp, ordr, mint, maxt, nodeVisits)
                }
                bool finished = false;
                if (cur < 0 && cur != DONE_REF) {
                    const uint32_t r = ~(uint32_t) cur, idx = r >> 3, left = r & 7u;
                    LOAD_TRI(stack, S, idx, a, b, c)
                    ++triTests;
                    float tu, tv, tt;
                    if (waldIntersect(a, b, c, o, d, mint, maxt, tu, tv, tt)) {
                        if (SHADOW) { res.prim = 0; finished = true; }
                        else { maxt = tt; res.t = tt; res.u = tu; res.v = tv; res.prim = pm_to_bits(c.z); }
                    }
                    cur = left ? (int32_t) ~(((idx + 1u) << 3) | (left - 1u)) : (stack.sp == 0 ? DONE_REF : (int32_t) stack.pop());
                }
                if (cur == DONE_REF) finished = true;
                if (finished) {
                    src.commit(handle, SHADOW ? (res.prim != PHIP_NO_HIT) : false, res);
                    active = false;
                    break;
                }
                if (src.more() && __popcll(__ballot(1)) <= 64 - REFILL_LANES) break;     /* enough idle lanes: refill */
            }
        }
    }
}

/* closest-hit source: all slots of the pool, chunk-strided over the resident waves */
struct TraceSource {
    const PathPool &P; uint32_t chunk, pos, stride, nChunks;
    __device__ __forceinline__ bool more() const { return chunk < nChunks; }
    __device__ __forceinline__ uint32_t assign(bool want, unsigned long long wantMask) {
        const uint32_t idx = pos + (uint32_t) __popcll(wantMask & ((1ull << __lane_id()) - 1ull));
        const uint32_t h = (want && idx < 64u && chunk * 64u + idx < P.capacity) ? chunk * 64u + idx : INVALID_RAY;
        pos += (uint32_t) __popcll(wantMask);
        if (pos >= 64u) { pos = 0; chunk += stride; }
        return h;
    }
    __device__ __forceinline__ bool load(uint32_t slot, V3 &o, V3 &d, float &mint, float &maxt) const {
        if (!(P.state[slot] & F_ALIVE)) return false;
        const float4 ro = P.rayO[slot], rd = P.rayD[slot];
        o = V3(ro.x, ro.y, ro.z); d = V3(rd.x, rd.y, rd.z); mint = ro.w; maxt = rd.w;
        return true;
    }
    __device__ __forceinline__ void commit(uint32_t slot, bool, const TravResult &r) const {
        P.hit[slot] = make_float4(r.t, r.u, r.v, pm_from_bits(r.prim));
    }
};

/* L[id] += c for an unoccluded NEE entry.  A load-add-store here stalls the whole traversal wave for a random HBM round
 * trip every time one of its lanes finishes a ray; three fire-and-forget hardware float atomics do not.  They are plain IEEE
 * round-to-nearest additions at the L2 (no other lane touches L[id] during this kernel, so there is no ordering question), but
 * the L2 adder flushes denormals: radiance contributions are >= 0, so the sum of a normal-or-zero addend and the accumulator
 * (itself a sum of such addends) is never denormal -- an entry with a denormal component takes the load-add-store path. */
__device__ __forceinline__ void addRadiance(float4 *L, uint32_t id, const float4 &c) {
    const float tiny = 1.17549435e-38f;
    const bool plain = (c.x == 0.0f || c.x >= tiny) && (c.y == 0.0f || c.y >= tiny) && (c.z == 0.0f || c.z >= tiny);
#if SHADOW_ATOMIC_COMMIT
    if (plain) {
        float *p = (float *) (L + id);
        if (c.x != 0.0f) unsafeAtomicAdd(p, c.x);
        if (c.y != 0.0f) unsafeAtomicAdd(p + 1, c.y);
        if (c.z != 0.0f) unsafeAtomicAdd(p + 2, c.z);
        return;
    }
#endif
    float4 l = L[id];
    l.x += c.x; l.y += c.y; l.z += c.z;
    L[id] = l;
}

/* any-hit source: the block-compacted shadow queue; wave w walks blocks w, w+W, ... */
struct ShadowSource {
    const PathPool &P; float4 *L; uint32_t blk, pos, cnt, stride, nBlocks;
    __device__ __forceinline__ void skipEmpty() {
        while (blk < nBlocks) { cnt = P.shadowCount[blk]; if (cnt) break; blk += stride; }
    }
    __device__ __forceinline__ bool more() const { return blk < nBlocks; }
    __device__ __forceinline__ uint32_t assign(bool want, unsigned long long wantMask) {
        const uint32_t idx = pos + (uint32_t) __popcll(wantMask & ((1ull << __lane_id()) - 1ull));
        const uint32_t h = (want && idx < cnt) ? blk * BLOCK + idx : INVALID_RAY;
        pos += (uint32_t) __popcll(wantMask);
        if (pos >= cnt) { pos = 0; blk += stride; skipEmpty(); }
        return h;
    }
    __device__ __forceinline__ bool load(uint32_t e, V3 &o, V3 &d, float &mint, float &maxt) const {
        const float4 e0 = P.shadow[3 * (size_t) e], e1 = P.shadow[3 * (size_t) e + 1];
        o = V3(e0.x, e0.y, e0.z); d = V3(e1.x, e1.y, e1.z); mint = PT_EPSILON; maxt = e0.w;
        return true;
    }
    __device__ __forceinline__ void commit(uint32_t e, bool occluded, const TravResult &) const {
        if (!occluded) {
            const float4 e1 = P.shadow[3 * (size_t) e + 1], e2 = P.shadow[3 * (size_t) e + 2];
            addRadiance(L, pm_to_bits(e1.w), e2);
        }
    }
};

#ifndef TRACE_P_WAVES
#define TRACE_P_WAVES 5
#endif
extern __shared__ __attribute__((aligned(16))) unsigned char g_smem[];

/* ---- closest-hit AND any-hit rays of one iteration in ONE persistent launch ----
 * The two ray kinds of an iteration are independent (k_shade consumes both results in the next iteration), so a wave
 * first drains its share of the shadow queue and then, without a kernel boundary, refills idle lanes from its share
 * of the closest-hit queue: one kernel tail (waves waiting for the slowest in-flight rays) and one launch per
 * iteration instead of two.  The kind of a lane's ray is a per-lane flag; the loop body is shared. */
__device__ __forceinline__ bool clipToSceneRT(const DevScene &S, const V3 &o, const V3 &d, float rayMint, float rayMaxt,
                                              float &mint, float &maxt, bool shadow) {
    float nearT = -INFINITY, farT = INFINITY;
    const float oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float origin = oo[i], minVal = S.sceneMin[i], maxVal = S.sceneMax[i];
        if (dd[i] == 0) {
            if (origin < minVal || origin > maxVal) return false;
        } else {
            const float rcp = 1.0f / dd[i];
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

#ifndef RAYS_SHADOW_UNSORTED
#define RAYS_SHADOW_UNSORTED 0
#endif
enum { WC_RAYS = 0, WC_NODE, WC_TRI, WC_SH_RAYS, WC_SH_NODE, WC_SH_TRI, WC_COUNT };

__device__ __forceinline__ void persistentTraverseMixed(const DevScene &S, TravStack &stack, ShadowSource &ss, TraceSource &ts,
                                                        uint32_t *wc /* LDS: WC_COUNT counters of this wave */) {
    constexpr bool TYPED = false;
    bool active = false, shadow = false;
    uint32_t handle = INVALID_RAY;
    V3 o(0.0f), d(0.0f), rcp(0.0f), ordr(0.0f);
    float mint = 0, maxt = 0;
    int32_t cur = 0;
    uint32_t nodeCur = 0, triCur = 0;
    TravResult res; res.prim = PHIP_NO_HIT; res.t = INFINITY; res.u = res.v = 0;

    for (;;) {
        const unsigned long long idle = __ballot(!active);
        const bool moreS = ss.more(), moreAny = moreS || ts.more();              /* wave-uniform */
        if (idle && moreAny && (__popcll(idle) >= REFILL_LANES || idle == ~0ull)) {
            const uint32_t h = moreS ? ss.assign(!active, idle) : ts.assign(!active, idle);
            if (!active && h != INVALID_RAY) {
                float rmint, rmaxt;
                const bool ok = moreS ? ss.load(h, o, d, rmint, rmaxt) : ts.load(h, o, d, rmint, rmaxt);
                if (ok) {
                    atomicAdd(&wc[moreS ? WC_SH_RAYS : WC_RAYS], 1u);
                    res.prim = PHIP_NO_HIT; res.t = INFINITY; res.u = res.v = 0;
                    if (clipToSceneRT(S, o, d, rmint, rmaxt, mint, maxt, moreS)) {
                        rcp = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                        ordr = V3(o.x * rcp.x, o.y * rcp.y, o.z * rcp.z);
                        cur = S.rootRef; stack.sp = 0; handle = h; active = true; shadow = moreS; nodeCur = triCur = 0;
                    } else if (moreS) {
                        ss.commit(h, false, res);
                    } else {
                        ts.commit(h, false, res);
                    }
                }
            }
        }
        if (!__any(active)) { if (!(ss.more() || ts.more())) break; continue; }
        if (active) {
            for (;;) {
                if (cur >= 0) {
#if RAYS_SHADOW_UNSORTED
                    if (shadow) NODE_STEP_ANY(stack, S, cur, rcp, ordr, mint, maxt, nodeCur)     /* (a wave is all-shadow or all-closest except while it changes phase) */
                    else
#endif
                    NODE_STEP(stack, S, cur, rcp, ordr, mint, maxt, nodeCur)
                }
                bool finished = false;
                if (cur < 0 && cur != DONE_REF) {
                    const uint32_t r = ~(uint32_t) cur, idx = r >> 3, left = r & 7u;
                    LOAD_TRI(stack, S, idx, a, b, c)
                    ++triCur;
                    float tu, tv, tt;
                    if (waldIntersect(a, b, c, o, d, mint, maxt, tu, tv, tt)) {
                        if (shadow) { res.prim = 0; finished = true; }
                        else { maxt = tt; res.t = tt; res.u = tu; res.v = tv; res.prim = pm_to_bits(c.z); }
                    }
                    cur = left ? (int32_t) ~(((idx + 1u) << 3) | (left - 1u)) : (stack.sp == 0 ? DONE_REF : (int32_t) stack.pop());
                }
                if (cur == DONE_REF) finished = true;
                if (finished) {
                    if (shadow) ss.commit(handle, res.prim != PHIP_NO_HIT, res);
                    else ts.commit(handle, false, res);
                    atomicAdd(&wc[shadow ? WC_SH_NODE : WC_NODE], nodeCur);
                    atomicAdd(&wc[shadow ? WC_SH_TRI : WC_TRI], triCur);
                    active = false;
                    break;
                }
                if ((ss.more() || ts.more()) && __popcll(__ballot(1)) <= 64 - REFILL_LANES) break;     /* enough idle lanes: refill */
            }
        }
    }
}

#ifndef RAYS_WAVES
#define RAYS_WAVES TRACE_P_WAVES
#endif
__global__ __launch_bounds__(BLOCK, RAYS_WAVES) void k_rays_p(DevScene S, PathPool P, float4 *L) {
    __shared__ uint32_t wcnt[BLOCK / 64][WC_COUNT];
    const uint32_t wave = threadIdx.x >> 6, waveId = (blockIdx.x * BLOCK + threadIdx.x) >> 6, nWavesGrid = gridDim.x * (BLOCK / 64);
    if (threadIdx.x < (BLOCK / 64) * WC_COUNT) (&wcnt[0][0])[threadIdx.x] = 0;
    TravStack stk; setupTraversal(S, g_smem, P.spill + (size_t) (blockIdx.x * BLOCK + threadIdx.x) * SPILL_DEPTH, stk);   /* (barrier inside) */
    ShadowSource ss{ P, L, waveId, 0u, 0u, nWavesGrid, P.capacity / BLOCK };
    ss.skipEmpty();
    TraceSource ts{ P, waveId, 0u, nWavesGrid, (P.capacity + 63u) / 64u };
    persistentTraverseMixed(S, stk, ss, ts, wcnt[wave]);
    if (__lane_id() == 0) {
        const int rows[WC_COUNT] = { ST_CLOSEST_RAYS, ST_NODE, ST_TRI, ST_SHADOW_RAYS, ST_SH_NODE, ST_SH_TRI };
#pragma unroll
        for (int i = 0; i < WC_COUNT; ++i) {
            const uint32_t v = wcnt[wave][i];
            if (v) P.stat[(size_t) rows[i] * P.nWaves + waveId] += v;
        }
    }
}

template <bool TYPED> __global__ __launch_bounds__(BLOCK, TRACE_P_WAVES) void k_trace_p(DevScene S, PathPool P) {
    const uint32_t waveId = (blockIdx.x * BLOCK + threadIdx.x) >> 6, nWavesGrid = gridDim.x * (BLOCK / 64);
    TravStack stk; setupTraversal(S, g_smem, P.spill + (size_t) (blockIdx.x * BLOCK + threadIdx.x) * SPILL_DEPTH, stk);
    TraceSource src{ P, waveId, 0u, nWavesGrid, (P.capacity + 63u) / 64u };
    uint32_t nodeVisits = 0, triTests = 0, rays = 0;
    persistentTraverse<false, TYPED>(S, stk, src, nodeVisits, triTests, rays);
    waveStat(P, ST_CLOSEST_RAYS, waveId, rays);
    waveStat(P, ST_NODE, waveId, nodeVisits);
    waveStat(P, ST_TRI, waveId, triTests);
}

__global__ __launch_bounds__(BLOCK, TRACE_WAVES) void k_shadow_p(DevScene S, PathPool P, float4 *L) {
    const uint32_t waveId = (blockIdx.x * BLOCK + threadIdx.x) >> 6, nWavesGrid = gridDim.x * (BLOCK / 64);
    TravStack stk; setupTraversal(S, g_smem, P.spill + (size_t) (blockIdx.x * BLOCK + threadIdx.x) * SPILL_DEPTH, stk);
    ShadowSource src{ P, L, waveId, 0u, 0u, nWavesGrid, P.capacity / BLOCK };
    src.skipEmpty();
    uint32_t nodeVisits = 0, triTests = 0, rays = 0;
    persistentTraverse<true, true>(S, stk, src, nodeVisits, triTests, rays);      /* k_shadow_p serves the small scenes (big ones use k_rays_p) */
    waveStat(P, ST_SHADOW_RAYS, waveId, rays);
    waveStat(P, ST_SH_NODE, waveId, nodeVisits);
    waveStat(P, ST_SH_TRI, waveId, triTests);
}

/* ======================================================================================
 *  Lane-cooperative traversal ("group" kernels): 8 lanes work on ONE ray over the 8-wide BVH.
 *  Lane k of a group fetches and slab-tests child k (the group's loads cover one contiguous
 *  256-byte node -> fully coalesced), or Wald-tests triangle k of a leaf.  A wave64 therefore
 *  walks 8 rays at a time; trip-count divergence is 8-way instead of 64-way, the per-ray stack
 *  (ref, tnear) lives in LDS at 1/8 of the per-lane cost, and a group that finishes its ray
 *  immediately pulls the next of the wave's 64 rays (wave-local dynamic fetch, no atomics).
 * ====================================================================================== */
#define STACK8 40                       /* (ref, tnear) entries per ray in LDS; deeper ones spill to HBM */
#define NONE_REF 0x7fffffff

struct Stack8 {
    uint2 *lds;             /* this group's STACK8 entries */
    uint2 *spill;           /* this group's SPILL8 entries in HBM */
    __device__ __forceinline__ void put(int i, uint2 v) { if (i < STACK8) lds[i] = v; else spill[i - STACK8] = v; }
    __device__ __forceinline__ uint2 get(int i) const { return i < STACK8 ? lds[i] : spill[i - STACK8]; }
};
#define SPILL8 64

template <bool SHADOW, typename Fetch, typename Commit>
__device__ __forceinline__ void traverseWave8(const DevScene &S, uint2 *waveStack, uint2 *waveSpill, uint32_t nRays,
                                              Fetch fetch, Commit commit, uint32_t &nodeVisits, uint32_t &triTests, uint32_t &raysTraced) {
    const uint32_t lane = __lane_id(), sub = lane & 7u, grp = lane >> 3, grpBase = lane & ~7u;
    Stack8 stk; stk.lds = waveStack + grp * STACK8; stk.spill = waveSpill + grp * SPILL8;
    uint32_t nextRay = 0;                    /* wave-uniform */
    bool needRay = true, active = false;
    uint32_t ray = 0;
    V3 o(0.0f), d(0.0f), rcp(0.0f), ordr(0.0f);
    float mint = 0, maxt = 0;
    int32_t cur = NONE_REF; int sp = 0;
    float bestT = INFINITY, bestU = 0, bestV = 0; uint32_t bestPrim = PHIP_NO_HIT;
    bool occluded = false;

    for (;;) {
        /* ---- hand out rays to the groups that need one (wave-uniform bookkeeping) ---- */
        const unsigned long long want = __ballot(needRay && sub == 0);
        if (want) {
            if (needRay) {
                ray = nextRay + (uint32_t) __popcll(want & ((1ull << grpBase) - 1ull));
                needRay = false;
                if (ray < nRays) {
                    float rmint, rmaxt;
                    if (fetch(ray, o, d, rmint, rmaxt)) {
                        if (sub == 0) ++raysTraced;
                        bestT = INFINITY; bestU = bestV = 0; bestPrim = PHIP_NO_HIT; occluded = false;
                        if (clipToScene<SHADOW>(S, o, d, rmint, rmaxt, mint, maxt)) {
                            rcp = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                            ordr = V3(o.x * rcp.x, o.y * rcp.y, o.z * rcp.z);
                            cur = S.rootRef8; sp = 0; active = true;
                        } else {
                            if (sub == 0) commit(ray, false, bestT, bestU, bestV, bestPrim);
                            needRay = true;
                        }
                    } else {
                        needRay = true;                      /* dead slot: take the next one */
                    }
                }
            }
            nextRay += (uint32_t) __popcll(want);
        }
        if (!__any(active || needRay)) break;
        if (!active) continue;

        /* ---- one traversal step per group ---- */
        bool done = false;
        if (cur == NONE_REF) {                               /* pop (with distance culling for closest hit) */
            if (sp == 0) done = true;
            else {
                --sp;
                const uint2 e = stk.get(sp);
                if (SHADOW || pm_from_bits(e.y) <= maxt) cur = (int32_t) e.x;
            }
        } else if (cur >= 0) {                               /* inner node: lane `sub` tests child `sub` */
            const float4 *p = S.nodes8 + (size_t) cur * 16 + sub * 2;
            const float4 a = p[0], b = p[1];
            if (sub == 0) ++nodeVisits;
            const float x0 = fmaf(a.x, rcp.x, -ordr.x), x1 = fmaf(a.w, rcp.x, -ordr.x);
            const float y0 = fmaf(a.y, rcp.y, -ordr.y), y1 = fmaf(b.x, rcp.y, -ordr.y);
            const float z0 = fmaf(a.z, rcp.z, -ordr.z), z1 = fmaf(b.y, rcp.z, -ordr.z);
            const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), mint));
            const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), maxt));
            const bool hit = tn <= tf;
            const uint32_t ref = pm_to_bits(b.z);
            const uint32_t hm = (uint32_t) (__ballot(hit) >> grpBase) & 0xffu;
            const int nh = __popc(hm);
            if (nh == 0) {
                cur = NONE_REF;
            } else {
                int rank;
                if (SHADOW) {
                    rank = __popc(hm & ((1u << sub) - 1u));  /* any order will do */
                } else {
                    const float key = hit ? tn : INFINITY;
                    rank = 0;
#pragma unroll
                    for (uint32_t j = 0; j < 8; ++j) {
                        const float kj = __shfl(key, (int) (grpBase + j));
                        rank += (kj < key || (kj == key && j < sub)) ? 1 : 0;
                    }
                }
                if (hit && rank > 0) stk.put(sp + nh - 1 - rank, make_uint2(ref, pm_to_bits(tn)));
                const uint32_t fm = (uint32_t) (__ballot(hit && rank == 0) >> grpBase) & 0xffu;
                cur = (int32_t) __shfl(ref, (int) (grpBase + (uint32_t) (__ffs((int) fm) - 1)));
                sp += nh - 1;
            }
        } else {                                             /* leaf: lane `sub` tests triangle `sub` */
            const uint32_t r = ~(uint32_t) cur;
            const uint32_t first = r >> 3, count = (r & 7u) + 1u;
            bool hit = false; float tu = 0, tv = 0, tt = INFINITY; uint32_t prim = PHIP_NO_HIT;
            if (sub < count) {
                const float4 *tp = S.tris + 3 * (size_t) (first + sub);
                const float4 a = tp[0], b = tp[1], c = tp[2];
                hit = waldIntersect(a, b, c, o, d, mint, maxt, tu, tv, tt);
                prim = pm_to_bits(c.z);
            }
            if (sub == 0) triTests += count;
            const uint32_t hm = (uint32_t) (__ballot(hit) >> grpBase) & 0xffu;
            if (hm) {
                if (SHADOW) { occluded = true; done = true; }
                else {
                    float m = hit ? tt : INFINITY;
                    m = fminf(m, __shfl_xor(m, 1)); m = fminf(m, __shfl_xor(m, 2)); m = fminf(m, __shfl_xor(m, 4));
                    /* ties: the later-tested triangle wins (sahkdtree3.h:286-291 semantics, `t <= maxt`) */
                    const uint32_t wm = (uint32_t) (__ballot(hit && tt == m) >> grpBase) & 0xffu;
                    const int jw = (int) grpBase + (31 - __clz((int) wm));
                    maxt = m; bestT = m;
                    bestU = __shfl(tu, jw); bestV = __shfl(tv, jw); bestPrim = __shfl(prim, jw);
                }
            }
            cur = NONE_REF;
        }
        if (done) {
            if (sub == 0) commit(ray, occluded, bestT, bestU, bestV, bestPrim);
            active = false; needRay = true;
        }
    }
}

__global__ __launch_bounds__(BLOCK, 6) void k_trace8(DevScene S, PathPool P) {
    __shared__ uint2 lds[(BLOCK / 64) * 8 * STACK8];
    const uint32_t wave = threadIdx.x >> 6, waveId = (blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const uint32_t base = waveId * 64;
    uint32_t nodeVisits = 0, triTests = 0, rays = 0;
    const uint32_t n = base < P.capacity ? min(64u, P.capacity - base) : 0u;
    traverseWave8<false>(S, lds + wave * 8 * STACK8, P.spill8 + (size_t) waveId * 8 * SPILL8, n,
        [&](uint32_t r, V3 &o, V3 &d, float &mint, float &maxt) -> bool {
            const uint32_t slot = base + r;
            if (!(P.state[slot] & F_ALIVE)) return false;
            const float4 ro = P.rayO[slot], rd = P.rayD[slot];
            o = V3(ro.x, ro.y, ro.z); d = V3(rd.x, rd.y, rd.z); mint = ro.w; maxt = rd.w;
            return true;
        },
        [&](uint32_t r, bool, float t, float u, float v, uint32_t prim) {
            P.hit[base + r] = make_float4(t, u, v, pm_from_bits(prim));
        }, nodeVisits, triTests, rays);
    waveStat(P, ST_CLOSEST_RAYS, waveId, rays);
    waveStat(P, ST_NODE, waveId, nodeVisits);
    waveStat(P, ST_TRI, waveId, triTests);
}

__global__ __launch_bounds__(BLOCK, 6) void k_shadow8(DevScene S, PathPool P, float4 *L) {
    __shared__ uint2 lds[(BLOCK / 64) * 8 * STACK8];
    const uint32_t wave = threadIdx.x >> 6, waveId = (blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const uint32_t count = P.shadowCount[blockIdx.x];
    const uint32_t first = wave * 64;
    if (first >= count) return;
    const uint32_t n = min(64u, count - first);
    const size_t base = (size_t) blockIdx.x * BLOCK + first;
    uint32_t nodeVisits = 0, triTests = 0, rays = 0;
    traverseWave8<true>(S, lds + wave * 8 * STACK8, P.spill8 + (size_t) waveId * 8 * SPILL8, n,
        [&](uint32_t r, V3 &o, V3 &d, float &mint, float &maxt) -> bool {
            const float4 e0 = P.shadow[3 * (base + r)], e1 = P.shadow[3 * (base + r) + 1];
            o = V3(e0.x, e0.y, e0.z); d = V3(e1.x, e1.y, e1.z); mint = PT_EPSILON; maxt = e0.w;
            return true;
        },
        [&](uint32_t r, bool occluded, float, float, float, uint32_t) {
            if (!occluded) {
                const float4 e1 = P.shadow[3 * (base + r) + 1], e2 = P.shadow[3 * (base + r) + 2];
                addRadiance(L, pm_to_bits(e1.w), e2);
            }
        }, nodeVisits, triTests, rays);
    waveStat(P, ST_SHADOW_RAYS, waveId, rays);
    waveStat(P, ST_SH_NODE, waveId, nodeVisits);
    waveStat(P, ST_SH_TRI, waveId, triTests);
}

/* ======================================================================================
 *  kernels
 * ====================================================================================== */
__global__ __launch_bounds__(BLOCK, TRACE_WAVES) void k_trace(DevScene S, PathPool P) {
    if (P.blockDead[blockIdx.x]) return;
    const uint32_t slot = blockIdx.x * BLOCK + threadIdx.x;
    TravStack stk; setupTraversal(S, g_smem, P.spill + (size_t) slot * SPILL_DEPTH, stk);
    uint32_t nodeVisits = 0, triTests = 0, rays = 0;
    if (slot < P.capacity) {
        if (P.state[slot] & F_ALIVE) {
            const float4 ro = P.rayO[slot], rd = P.rayD[slot];
            const V3 o(ro.x, ro.y, ro.z), d(rd.x, rd.y, rd.z);
            float mint, maxt;
            TravResult r; r.prim = PHIP_NO_HIT; r.t = INFINITY; r.u = r.v = 0;
            rays = 1;
            if (clipToScene<false>(S, o, d, ro.w, rd.w, mint, maxt))
                traverse<false>(S, o, d, mint, maxt, stk, r, nodeVisits, triTests);
            P.hit[slot] = make_float4(r.t, r.u, r.v, pm_from_bits(r.prim));
        }
    }
    const uint32_t waveId = (blockIdx.x * BLOCK + threadIdx.x) >> 6;
    waveStat(P, ST_CLOSEST_RAYS, waveId, rays);
    waveStat(P, ST_NODE, waveId, nodeVisits);
    waveStat(P, ST_TRI, waveId, triTests);
}

__global__ __launch_bounds__(BLOCK, TRACE_WAVES) void k_shadow(DevScene S, PathPool P, float4 *L) {
    if (P.blockDead[blockIdx.x]) return;
    TravStack stk; setupTraversal(S, g_smem, P.spill + (size_t) (blockIdx.x * BLOCK + threadIdx.x) * SPILL_DEPTH, stk);
    const uint32_t n = P.shadowCount[blockIdx.x];            /* entries of this block's slots */
    uint32_t nodeVisits = 0, triTests = 0, rays = 0;
    if (threadIdx.x < n) {
        const size_t idx = (size_t) blockIdx.x * BLOCK + threadIdx.x;
        const float4 e0 = P.shadow[3 * idx], e1 = P.shadow[3 * idx + 1], e2 = P.shadow[3 * idx + 2];
        const V3 o(e0.x, e0.y, e0.z), d(e1.x, e1.y, e1.z);
        float mint, maxt;
        bool occluded = false;
        TravResult r;
        rays = 1;
        if (clipToScene<true>(S, o, d, PT_EPSILON, e0.w, mint, maxt))
            occluded = traverse<true>(S, o, d, mint, maxt, stk, r, nodeVisits, triTests);
        if (!occluded) {
            addRadiance(L, pm_to_bits(e1.w), e2);
        }
    }
    if ((threadIdx.x & ~63u) < n) {                          /* waves without entries have nothing to add */
        const uint32_t waveId = (blockIdx.x * BLOCK + threadIdx.x) >> 6;
        waveStat(P, ST_SHADOW_RAYS, waveId, rays);
        waveStat(P, ST_SH_NODE, waveId, nodeVisits);
        waveStat(P, ST_SH_TRI, waveId, triTests);
    }
}

__device__ __forceinline__ float miWeight(float pdfA, float pdfB) {
    pdfA *= pdfA; pdfB *= pdfB;
    return pdfA / (pdfA + pdfB);
}

#ifndef SHADE_WAVES
#define SHADE_WAVES 4
#endif
#define EMITTER_LDS_FLOATS 1024      /* 4 KB */
#define MATERIAL_LDS_MAX 48          /* 3.75 KB */
#ifndef SHADE_WAVES_LEAN
#define SHADE_WAVES_LEAN 4          /* diffuse-only instantiation */
#endif
/* MM: leaf BSDF models present in the scene; STRICT: strictNormals (a compile-time switch: without it the geometric
   normal is dead after fillIntersection and the diffuse-only instantiation fits 80 VGPRs = 6 waves per SIMD) */
template <int MM, bool STRICT> __global__ __launch_bounds__(BLOCK, MM == 0 ? SHADE_WAVES_LEAN : SHADE_WAVES) void k_shade(DevScene S, PathPool P, RenderConst rc, float4 *L) {
    __shared__ uint32_t waveCnt[BLOCK / 64];
    if (P.blockDead[blockIdx.x]) return;                        /* (block-uniform) */
    /* small scene tables are staged in LDS: the emitter table (selection CDF -> emitter -> area CDF is a chain of
       dependent lookups per NEE sample) and the materials */
    __shared__ __align__(16) float ldsEm[EMITTER_LDS_FLOATS];
    __shared__ DevMaterial ldsMat[MATERIAL_LDS_MAX];
    const bool emInLds = S.emitterTabSize <= EMITTER_LDS_FLOATS, matInLds = S.nMaterials <= MATERIAL_LDS_MAX;
    if (emInLds) for (uint32_t i = threadIdx.x; i < S.emitterTabSize; i += BLOCK) ldsEm[i] = S.emitterTab[i];
    if (matInLds) {
        const uint32_t n4 = S.nMaterials * (uint32_t) (sizeof(DevMaterial) / 16);
        for (uint32_t i = threadIdx.x; i < n4; i += BLOCK) ((float4 *) ldsMat)[i] = ((const float4 *) S.materials)[i];
    }
    EmitterTab T; T.t = emInLds ? ldsEm : S.emitterTab; T.n = S.nEmitters; T.normalization = S.emitterNormalization;
    const DevMaterial *materials = matInLds ? ldsMat : S.materials;
    const uint32_t slot = blockIdx.x * BLOCK + threadIdx.x;
    const bool inRange = slot < P.capacity;
    /* all slot state is fetched up front, before the liveness test, so that the five 16-byte loads are
       in flight together (the kernel is latency bound: 70 % of its wave cycles were s_waitcnt) */
    const uint32_t lslot = inRange ? slot : 0u;
    uint4 info = P.info[lslot];
    info.w = P.state[lslot];
    const float4 hit = P.hit[lslot];
    const float4 rd = P.rayD[lslot];
    float4 thr4 = P.thr[lslot];
    const float2 mis = P.mis[lslot];
    if (!inRange) info = make_uint4(0, 0, 0, 0);
    __syncthreads();                                            /* LDS tables are complete */
    bool alive = inRange && (info.w & F_ALIVE);
    bool needNew = inRange && !alive && !(info.w & F_DEAD);
    unsigned long long vertices = 0, done = 0;
    bool pushShadow = false;
    float4 sh0 = make_float4(0, 0, 0, 0), sh1 = sh0, sh2 = sh0;

    if (alive) {
        const uint32_t prim = pm_to_bits(hit.w);
        const V3 rayD(rd.x, rd.y, rd.z);
        V3 thr(thr4.x, thr4.y, thr4.z);
        float eta = thr4.w;
        uint32_t depth = info.w & DEPTH_MASK;
        uint32_t flags = info.w & ~DEPTH_MASK;
        const uint32_t id = info.x;
        bool terminate = false;
        V3 addL(0.0f); bool haveAdd = false;   /* radiance to add to L[id] (in reference order) */
        float4 l = make_float4(0, 0, 0, 0);

        if (prim == PHIP_NO_HIT) {
            terminate = true;
            if (S.envEmitter >= 0) {            /* environment emitter: path.cpp:136-143 (camera ray) / 233-265 (BSDF-sampled ray) */
                const float *em = emitterRecord(T, (uint32_t) S.envEmitter);
                const V3 value = rgb(em + EM_RADIANCE);
                l = L[id];
                if (flags & F_FIRST) {
                    if (!rc.hideEmitters) { l.x += value.x; l.y += value.y; l.z += value.z; }   /* throughput is 1; alpha stays 0 */
                    haveAdd = true;
                } else {
                    const float4 ro = P.rayO[slot];
                    if (envFillDirectRecord(S, V3(ro.x, ro.y, ro.z), rayD)) {
                        const float lumPdf = (!(flags & F_PREV_DELTA))
                            ? pdfEmitterDirectDot(T, (uint32_t) S.envEmitter, mis.y, (flags & F_REFN_ZERO) != 0, 0.0f, 0.0f) : 0;
                        const V3 c = thr * value * miWeight(mis.x, lumPdf);
                        l.x += c.x; l.y += c.y; l.z += c.z;
                        haveAdd = true;
                    }
                }
                if (haveAdd) L[id] = l;
            }
        } else {
            Isect its;
            fillIntersection(S, rayD, prim, hit.y, hit.z, hit.x, its);
            /* L[id] is zero until the sample's first vertex writes it (the buffer is cleared per pass), and later
               vertices only touch it when they hit an emitter: no unconditional 64-byte-sector read per vertex */
            if (flags & F_FIRST) {
                l.w = 1.0f;                     /* alpha, records.inl:117-144 */
                haveAdd = true;
            } else {
                /* ---- tail of the previous loop iteration, path.cpp:257-286 ---- */
                if (its.emitter >= 0) {
                    l = L[id];
                    const float *em = emitterRecord(T, (uint32_t) its.emitter);
                    V3 value = (dot(its.sh.n, -rayD) <= 0) ? V3(0.0f) : rgb(em + EM_RADIANCE);
                    /* DirectSamplingRecord::setQuery (records.inl:170-178): n = shading normal, d = ray direction, dist = t */
                    const float lumPdf = (!(flags & F_PREV_DELTA))
                        ? pdfEmitterDirectDot(T, (uint32_t) its.emitter, mis.y, (flags & F_REFN_ZERO) != 0, dot(rayD, its.sh.n), its.t) : 0;
                    const V3 c = thr * value * miWeight(mis.x, lumPdf);
                    l.x += c.x; l.y += c.y; l.z += c.z;
                    haveAdd = true;
                }
                flags &= ~F_EMITTED;
                if (depth++ >= (uint32_t) rc.rrDepth) {
                    float q = smin(thr.maxc() * eta * eta, 0.95f);
                    const U4 h = pcg4d(info.y, info.z, 2 + 2 * (depth - 2), rc.seed);
                    if (u32ToFloat(h.x) >= q)
                        terminate = true;
                    else
                        thr = thr / q;
                }
            }
            flags &= ~F_FIRST;

            /* ---- head of the loop for this vertex, path.cpp:135-165 ---- */
            if (!terminate && !(depth <= (uint32_t) rc.maxDepth || rc.maxDepth < 0))
                terminate = true;
            if (!terminate) {
                if (its.emitter >= 0 && (flags & F_EMITTED) && (!rc.hideEmitters || (flags & F_SCATTERED))) {
                    const float *em = emitterRecord(T, (uint32_t) its.emitter);
                    V3 le = (dot(its.sh.n, -rayD) <= 0) ? V3(0.0f) : rgb(em + EM_RADIANCE);
                    const V3 c = thr * le;
                    l.x += c.x; l.y += c.y; l.z += c.z;
                    haveAdd = true;
                }
                if (((int) depth >= rc.maxDepth && rc.maxDepth > 0)
                    || (STRICT && dot(rayD, its.geoN) * cosTheta(its.wi) >= 0))
                    terminate = true;
            }
            V3 shD(0.0f), shC(0.0f); float shMaxt = 0;
            if (!terminate) {
                const U4 h = pcg4d(info.y, info.z, 1 + 2 * (depth - 1), rc.seed);
                /* ---- direct illumination sampling, path.cpp:172-200 ---- */
                DirectRec dRec;
                dRec.ref = its.p;
                dRec.refN = (its.flags & TS_TRANS_OR_BACK) ? V3(0.0f) : its.sh.n;
                dRec.pdf = 0; dRec.emitter = -1;
                const BsdfCtx bctx = bsdfResolve(materials, its);
                if (its.flags & TS_MF_SMOOTH) {
                    V3 value = sampleEmitterDirect(S, T, dRec, V2(u32ToFloat(h.x), u32ToFloat(h.y)));
                    if (dRec.pdf != 0 && !value.isZero()) {
                        const V3 wo = its.sh.toLocal(dRec.d);
                        float bPdf;
                        const V3 bsdfVal = bsdfEvalPdf<MM>(bctx, wo, bPdf);
                        if (!bsdfVal.isZero() && (!STRICT || dot(its.geoN, dRec.d) * cosTheta(wo) > 0)) {
                            const float weight = miWeight(dRec.pdf, bPdf);
                            shC = thr * value * bsdfVal * weight;
                            shD = dRec.d; shMaxt = dRec.dist * (1 - PT_SHADOW_EPSILON);
                            pushShadow = true;
                        }
                    }
                }
                /* ---- BSDF sampling, path.cpp:207-226 ---- */
                BSDFSample bs;
                const V3 bsdfWeight = bsdfSample<MM>(bctx, V2(u32ToFloat(h.z), u32ToFloat(h.w)), bs);
                if (bsdfWeight.isZero()) {
                    terminate = true;
                } else {
                    flags |= F_SCATTERED;
                    const V3 wo = its.sh.toWorld(bs.wo);
                    const float woDotGeoN = dot(its.geoN, wo);
                    if (STRICT && woDotGeoN * cosTheta(bs.wo) <= 0) {
                        terminate = true;
                    } else {
                        P.rayO[slot] = make_float4(its.p.x, its.p.y, its.p.z, PT_EPSILON);
                        P.rayD[slot] = make_float4(wo.x, wo.y, wo.z, INFINITY);
                        thr = thr * bsdfWeight;
                        eta *= bs.eta;
                        P.thr[slot] = make_float4(thr.x, thr.y, thr.z, eta);
                        P.mis[slot] = make_float2(bs.pdf, dot(wo, dRec.refN));
                        flags = bs.delta ? (flags | F_PREV_DELTA) : (flags & ~F_PREV_DELTA);
                        flags = dRec.refN.isZero() ? (flags | F_REFN_ZERO) : (flags & ~F_REFN_ZERO);
                    }
                }
            }
            if (haveAdd) L[id] = l;
            if (pushShadow) {   /* self-contained shadow-queue entry: survives the slot being recycled */
                sh0 = make_float4(its.p.x, its.p.y, its.p.z, shMaxt);
                sh1 = make_float4(shD.x, shD.y, shD.z, pm_from_bits(id));
                sh2 = make_float4(shC.x, shC.y, shC.z, 0.0f);
            }
        }
        if (terminate) {
            vertices = depth; done = 1;
            needNew = true;
        } else {
            info.w = flags | depth;
            P.state[slot] = info.w;
        }
    }

    /* ---- shadow queue: compact this block's entries to the front of its own region (no global atomics) ---- */
    uint32_t shadowTotal = 0;
    {
        const unsigned long long m = __ballot(pushShadow);
        const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
        if (lane == 0) waveCnt[wave] = (uint32_t) __popcll(m);
        __syncthreads();
        uint32_t base = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < BLOCK / 64; ++w) { const uint32_t c = waveCnt[w]; if (w < wave) base += c; total += c; }
        if (pushShadow) {
            const size_t sidx = (size_t) blockIdx.x * BLOCK + base + (uint32_t) __popcll(m & ((1ull << lane) - 1ull));
            P.shadow[3 * sidx] = sh0; P.shadow[3 * sidx + 1] = sh1; P.shadow[3 * sidx + 2] = sh2;
        }
        if (threadIdx.x == 0) P.shadowCount[blockIdx.x] = total;
        shadowTotal = total;
    }

    /* ---- regeneration: the lane starts a new camera path right away (integrator.cpp:157-183).
       Sample ids [0, staticIds) follow a static schedule (slot s renders s, s + capacity, ... -- no global
       counter in steady state).  The last part of the frame is handed out dynamically so that slots whose
       paths happened to be short keep working until the frame is really finished: one atomicAdd per BLOCK
       on one of DYN_SHARDS counters (block-aggregated through LDS; each shard owns a contiguous id range). ---- */
    bool nowAlive = alive && !needNew;
    unsigned long long newId = ~0ull;
    bool wantDyn = false;
    if (needNew) {
        unsigned long long id = (info.w & F_FRESH) ? (unsigned long long) slot      /* first sample of this slot */
                              : ((info.w & F_DYNAMIC) ? ~0ull : (unsigned long long) info.x + P.capacity);
        for (;;) {
            if (id >= rc.staticIds) { wantDyn = true; break; }
            uint32_t px, py, k;
            if (decodeId(rc, S.film, id, px, py, k)) { newId = id; break; }
            id += P.capacity;       /* ids outside the crop window (edge blocks) are skipped */
        }
    }
    bool dynamicId = false;
    {
        const unsigned long long m = __ballot(wantDyn);
        const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
        __syncthreads();                                   /* waveCnt is reused from the shadow compaction */
        if (lane == 0) waveCnt[wave] = (uint32_t) __popcll(m);
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < BLOCK / 64; ++w) { const uint32_t c = waveCnt[w]; if (w < wave) before += c; total += c; }
        if (total) {                                       /* block-uniform */
            __shared__ unsigned long long dynBase;
            __shared__ uint32_t dynShard;
            if (threadIdx.x == 0) {
                uint32_t sh = (blockIdx.x + rc.blockShard[blockIdx.x]) % DYN_SHARDS;    /* blockShard = shards this block has seen run dry */
                uint32_t dry = 0;
                unsigned long long base = ~0ull;
                for (int tries = 0; tries < DYN_SHARDS; ++tries) {
                    const unsigned long long old = atomicAdd(rc.dynCounter + (size_t) sh * DYN_STRIDE, (unsigned long long) total);
                    if (old < rc.shardIds) { base = old; break; }
                    sh = (sh + 1) % DYN_SHARDS; ++dry;     /* this shard is used up: move on for good */
                }
                if (dry) rc.blockShard[blockIdx.x] += dry;
                dynBase = base; dynShard = sh;
            }
            __syncthreads();
            if (wantDyn) {
                bool got = false;
                if (dynBase != ~0ull) {
                    const unsigned long long off = dynBase + before + (uint32_t) __popcll(m & ((1ull << lane) - 1ull));
                    const unsigned long long id = rc.staticIds + (unsigned long long) dynShard * rc.shardIds + off;
                    uint32_t px, py, k;
                    if (off < rc.shardIds && id < rc.totalIds) {
                        got = true;                        /* the id is consumed even if it lies outside the crop window */
                        if (decodeId(rc, S.film, id, px, py, k)) { newId = id; dynamicId = true; }
                    }
                }
                if (!got && dynBase == ~0ull) { info.w = F_DEAD; P.state[slot] = F_DEAD; }   /* all shards empty: slot dies */
                else if (newId == ~0ull) { info.w = F_DYNAMIC; P.state[slot] = F_DYNAMIC; }                        /* try again next iteration */
            }
        }
    }
    if (newId != ~0ull) {
        uint32_t px, py, k;
        decodeId(rc, S.film, newId, px, py, k);
        const uint32_t pixel = py * (uint32_t) S.film.width + px;
        const U4 h = pcg4d(pixel, k, 0, rc.seed);
        const float sx = (float) px + u32ToFloat(h.x), sy = (float) py + u32ToFloat(h.y);
        V3 o, d; float mint, maxt;
        cameraRay(S.cam, sx, sy, o, d, mint, maxt);
        P.rayO[slot] = make_float4(o.x, o.y, o.z, mint);
        P.rayD[slot] = make_float4(d.x, d.y, d.z, maxt);
        P.thr[slot] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        P.mis[slot] = make_float2(0.0f, 0.0f);
        info = make_uint4((uint32_t) newId, pixel, k, 1u | F_ALIVE | F_EMITTED | F_FIRST | (dynamicId ? F_DYNAMIC : 0u));
        P.info[slot] = info;
        P.state[slot] = info.w;
        nowAlive = true;
    }
    const uint32_t waveId = slot >> 6;
    /* a slot still waiting for a dynamic sample id counts as live for the termination test */
    const bool live = nowAlive || (inRange && info.w == F_DYNAMIC);
    /* the block retires once none of its slots will ever work again and its last shadow entries have been consumed
       (this launch queued nothing, so shadowCount is 0): later launches of the pass return at the first line */
    const bool retire = !__syncthreads_or(live ? 1 : 0) && shadowTotal == 0;
    if (retire && threadIdx.x == 0) P.blockDead[blockIdx.x] = 1u;
    if (inRange || (slot & ~63u) < P.capacity) {
        waveStat(P, ST_VERTICES, waveId, vertices);
        waveStat(P, ST_SAMPLES, waveId, done);
        if (rc.countAlive || retire) waveStat(P, ST_ALIVE, waveId, live ? 1ull : 0ull, true);
    }
}

/* sums the per-wave statistics: REDUCE_SPLIT blocks per counter row, rows [firstRow, firstRow + gridDim.x);
   the totals must have been zeroed (one atomicAdd per block: 32 per row) */
#define REDUCE_SPLIT 32
__global__ void k_reduce_stats(PathPool P, Counters *C, int firstRow) {
    __shared__ unsigned long long red[256];
    const int row = firstRow + (int) blockIdx.x;
    const unsigned long long *src = P.stat + (size_t) row * P.nWaves;
    unsigned long long v = 0;
    for (uint32_t i = blockIdx.y * 256 + threadIdx.x; i < P.nWaves; i += 256 * REDUCE_SPLIT) v += src[i];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if ((int) threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off]; __syncthreads(); }
    if (threadIdx.x == 0 && red[0]) atomicAdd(&C->total[row], red[0]);
}

/* Film: one lane per crop pixel gathers every sample whose filter footprint covers it.  Restates
 * ImageBlock::put (imageblock.h:124-204) incl. the block-local coordinate arithmetic: a sample
 * taken in pixel (sx,sy) belongs to the render block whose origin is (sx,sy) rounded down to the
 * block size, and its weights are computed in that block's coordinate system. */
__global__ __launch_bounds__(BLOCK) void k_film(DevScene S, RenderConst rc, const float4 *L, const int32_t *tileSlot,
                                               int tilesX, float *out, int accumulate, unsigned long long *invalidCount) {
    const DevFilm &F = S.film;
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= F.width || y >= F.height) return;
    /* a sample of source pixel s lands at s + jitter - 0.5 in [s-0.5, s+0.5): it can reach x iff
       s > x - radius - 0.5 and s <= x + radius + 0.5 */
    const int sx0 = max((int) floorf((float) x - F.radius - 0.5f) + 1, 0), sx1 = min((int) floorf((float) x + F.radius + 0.5f), F.width - 1);
    const int sy0 = max((int) floorf((float) y - F.radius - 0.5f) + 1, 0), sy1 = min((int) floorf((float) y + F.radius + 0.5f), F.height - 1);
    float acc[5] = { 0, 0, 0, 0, 0 };
    unsigned long long invalid = 0;
    for (int sy = sy0; sy <= sy1; ++sy) {
        for (int sx = sx0; sx <= sx1; ++sx) {
            const int tx = sx >> rc.tileShift, ty = sy >> rc.tileShift;
            const int32_t ts = tileSlot[ty * tilesX + tx];
            if (ts < 0) continue;           /* that block belongs to another shard */
            const int offX = tx << rc.tileShift, offY = ty << rc.tileShift;
            const int bw = min(F.blockSize, F.width - offX) + 2 * F.border, bh = min(F.blockSize, F.height - offY) + 2 * F.border;
            /* destination pixel in the source block's bitmap coordinates */
            const int dx = x - (offX - F.border), dy = y - (offY - F.border);
            if (dx < 0 || dy < 0 || dx >= bw || dy >= bh) continue;
            const uint32_t m = spreadBits((uint32_t) (sx - offX)) | (spreadBits((uint32_t) (sy - offY)) << 1);
            const uint32_t pixel = (uint32_t) sy * (uint32_t) F.width + (uint32_t) sx;
            for (uint32_t k = 0; k < rc.sppPass; ++k) {
                const U4 h = pcg4d(pixel, k + rc.sppFirst, 0, rc.seed);
                const float px = (float) sx + u32ToFloat(h.x), py = (float) sy + u32ToFloat(h.y);
                const float posx = px - 0.5f - (float) (offX - F.border), posy = py - 0.5f - (float) (offY - F.border);
                const int minx = max((int) ceilf(posx - F.radius), 0), maxx = min((int) floorf(posx + F.radius), bw - 1);
                const int miny = max((int) ceilf(posy - F.radius), 0), maxy = min((int) floorf(posy + F.radius), bh - 1);
                if (dx < minx || dx > maxx || dy < miny || dy > maxy) continue;
                const unsigned long long id = (((unsigned long long) ts * rc.sppPass + k) << (2 * rc.tileShift)) | m;
                const float4 v = L[id];
                /* validity check of ImageBlock::put: reject non-finite / negative samples (imageblock.h:148-151) */
                if (!(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w)) || v.x < 0 || v.y < 0 || v.z < 0 || v.w < 0) {
                    if (sx == x && sy == y) ++invalid;
                    continue;
                }
                const float wx = F.table[min((int) fabsf(((float) dx - posx) * F.scaleFactor), PHIP_FILTER_RESOLUTION)];
                const float wy = F.table[min((int) fabsf(((float) dy - posy) * F.scaleFactor), PHIP_FILTER_RESOLUTION)];
                const float w = wx * wy;
                acc[0] += w * v.x; acc[1] += w * v.y; acc[2] += w * v.z; acc[3] += w * v.w; acc[4] += w * 1.0f;
            }
        }
    }
    float *o = out + ((size_t) y * F.width + x) * 5;
    if (accumulate) { for (int i = 0; i < 5; ++i) o[i] += acc[i]; }
    else { for (int i = 0; i < 5; ++i) o[i] = acc[i]; }
    if (invalid) atomicAdd(invalidCount, invalid);
}

/* LDS-tiled film gather (filters with reach <= FILM_MAX_REACH pixels, i.e. every reference default): a block owns
 * 16x16 destination pixels.  Per sample index k the block first STAGES each of the (16+2R)^2 source pixels' sample
 * ONCE in LDS: radiance, the frame coordinates of the first pixel of its filter footprint and the separable filter
 * weights of ImageBlock::put (imageblock.h:124-204: footprint clipped to the bitmap of the render block the sample
 * belongs to, weights from the discretised table) -- weights outside the footprint are stored as 0, which adds
 * nothing.  Then every destination lane accumulates its (2R+1)^2 neighbours: two integer subtractions, two weight
 * reads, one product and five multiply-adds per neighbour.  Same arithmetic per (sample, pixel) pair as k_film;
 * only the order of the float additions differs. */
#define FILM_MAX_REACH 4
#define FILM_TILE 16
template <int RMAX>
__global__ __launch_bounds__(BLOCK) void k_film_tiled(DevScene S, RenderConst rc, const float4 *L, const int32_t *tileSlot,
                                                     int tilesX, float *out, int accumulate, unsigned long long *invalidCount, int R) {
    constexpr int TMAX = FILM_TILE + 2 * RMAX;
    constexpr int NW = 2 * RMAX + 2;           /* weights per axis: floor(p + r) - ceil(p - r) + 1 <= 2r + 1 with r < RMAX + 0.5 */
    __shared__ float4 sVal[TMAX * TMAX];       /* radiance rgb + alpha of the source pixel's k-th sample */
    __shared__ int2 sOrg[TMAX * TMAX];         /* frame coordinates of weight [0] of the sample's footprint */
    __shared__ float sWx[TMAX * TMAX * NW], sWy[TMAX * TMAX * NW];
    __shared__ int4 sGeo[TMAX * TMAX];         /* (offX - border, offY - border, bw, bh) of the source pixel's render block */
    __shared__ uint32_t sBase[TMAX * TMAX];    /* low word of the sample id of k = 0 (0xFFFFFFFF: pixel not rendered here) */
    __shared__ float sTable[PHIP_FILTER_RESOLUTION + 1];
    const DevFilm &F = S.film;
    const int T = FILM_TILE + 2 * R;
    const int x0 = blockIdx.x * FILM_TILE, y0 = blockIdx.y * FILM_TILE;
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int x = x0 + lx, y = y0 + ly;
    if (threadIdx.x <= PHIP_FILTER_RESOLUTION) sTable[threadIdx.x] = F.table[threadIdx.x];
    /* per source pixel constants */
    for (int i = threadIdx.x; i < T * T; i += BLOCK) {
        const int sx = x0 - R + (i % T), sy = y0 - R + (i / T);
        uint32_t base = 0xFFFFFFFFu; int4 g = make_int4(0, 0, 0, 0);
        if (sx >= 0 && sy >= 0 && sx < F.width && sy < F.height) {
            const int tx = sx >> rc.tileShift, ty = sy >> rc.tileShift;
            const int32_t ts = tileSlot[ty * tilesX + tx];
            if (ts >= 0) {
                const int offX = tx << rc.tileShift, offY = ty << rc.tileShift;
                g = make_int4(offX - F.border, offY - F.border, min(F.blockSize, F.width - offX) + 2 * F.border, min(F.blockSize, F.height - offY) + 2 * F.border);
                base = (uint32_t) ts;                       /* id(k) = ((ts * sppPass + k) << 2*tileShift) | morton(pixel in block) */
            }
        }
        sBase[i] = base; sGeo[i] = g;
    }
    __syncthreads();

    float acc[5] = { 0, 0, 0, 0, 0 };
    unsigned long long invalid = 0;
    const bool inside = x < F.width && y < F.height;
    /* the radiance of sample k + 1 is fetched while sample k is being gathered (the k loop is a chain of
       barriers otherwise: global-load latency would be paid sppPass times in a row) */
    constexpr int NSTAGE = (TMAX * TMAX + BLOCK - 1) / BLOCK;
    float4 pre[NSTAGE];
    auto fetch = [&](uint32_t k) {
#pragma unroll
        for (int n = 0; n < NSTAGE; ++n) {
            const int i = (int) threadIdx.x + n * BLOCK;
            pre[n] = make_float4(0, 0, 0, 0);
            if (i < T * T && k < rc.sppPass) {
                const uint32_t base = sBase[i];
                if (base != 0xFFFFFFFFu) {
                    const int sx = x0 - R + (i % T), sy = y0 - R + (i / T);
                    const int4 g = sGeo[i];
                    const uint32_t m = spreadBits((uint32_t) (sx - (g.x + F.border))) | (spreadBits((uint32_t) (sy - (g.y + F.border))) << 1);
                    pre[n] = L[(((unsigned long long) base * rc.sppPass + k) << (2 * rc.tileShift)) | m];
                }
            }
        }
    };
    fetch(0);
    for (uint32_t k = 0; k < rc.sppPass; ++k) {
#pragma unroll
        for (int n = 0; n < NSTAGE; ++n) {
            const int i = (int) threadIdx.x + n * BLOCK;
            if (i >= T * T) break;
            const uint32_t base = sBase[i];
            float4 v = make_float4(0, 0, 0, 0);
            int2 org = make_int2(0, 0);
            float wx[NW], wy[NW];
#pragma unroll
            for (int j = 0; j < NW; ++j) { wx[j] = 0.0f; wy[j] = 0.0f; }
            if (base != 0xFFFFFFFFu) {
                const int sx = x0 - R + (i % T), sy = y0 - R + (i / T);
                const int4 g = sGeo[i];
                const uint32_t pixel = (uint32_t) sy * (uint32_t) F.width + (uint32_t) sx;
                const U4 h = pcg4d(pixel, k + rc.sppFirst, 0, rc.seed);
                const float px = (float) sx + u32ToFloat(h.x), py = (float) sy + u32ToFloat(h.y);
                const float posx = px - 0.5f - (float) g.x, posy = py - 0.5f - (float) g.y;   /* block-bitmap coordinates */
                v = pre[n];
                /* validity check of ImageBlock::put (imageblock.h:148-151) */
                if (!(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w)) || v.x < 0 || v.y < 0 || v.z < 0 || v.w < 0) {
                    /* count each rejected sample once: by the block that owns its pixel */
                    if (sx >= x0 && sx < x0 + FILM_TILE && sy >= y0 && sy < y0 + FILM_TILE) ++invalid;
                    v = make_float4(0, 0, 0, 0);
                } else {
                    /* footprint and weights, imageblock.h:159-180 */
                    const int uminx = (int) ceilf(posx - F.radius), uminy = (int) ceilf(posy - F.radius);
                    const int minx = max(uminx, 0), maxx = min((int) floorf(posx + F.radius), g.z - 1);
                    const int miny = max(uminy, 0), maxy = min((int) floorf(posy + F.radius), g.w - 1);
                    org = make_int2(g.x + uminx, g.y + uminy);
#pragma unroll
                    for (int j = 0; j < NW; ++j) {
                        const int bx = uminx + j, by = uminy + j;
                        if (bx >= minx && bx <= maxx) wx[j] = sTable[min((int) fabsf(((float) bx - posx) * F.scaleFactor), PHIP_FILTER_RESOLUTION)];
                        if (by >= miny && by <= maxy) wy[j] = sTable[min((int) fabsf(((float) by - posy) * F.scaleFactor), PHIP_FILTER_RESOLUTION)];
                    }
                }
            }
            sVal[i] = v; sOrg[i] = org;
#pragma unroll
            for (int j = 0; j < NW; ++j) { sWx[i * NW + j] = wx[j]; sWy[i * NW + j] = wy[j]; }
        }
        __syncthreads();
        fetch(k + 1);
        if (inside) {
            for (int dyy = -R; dyy <= R; ++dyy) {
                for (int dxx = -R; dxx <= R; ++dxx) {
                    const int i = (ly + R + dyy) * T + (lx + R + dxx);
                    const int2 org = sOrg[i];
                    const int jx = x - org.x, jy = y - org.y;
                    if ((unsigned) jx >= (unsigned) NW || (unsigned) jy >= (unsigned) NW) continue;
                    const float w = sWx[i * NW + jx] * sWy[i * NW + jy];
                    if (w == 0.0f) continue;                   /* outside the footprint (or a zero of the filter): adds nothing */
                    const float4 v = sVal[i];
                    acc[0] += w * v.x; acc[1] += w * v.y; acc[2] += w * v.z; acc[3] += w * v.w; acc[4] += w * 1.0f;
                }
            }
        }
        __syncthreads();
    }
    if (inside) {
        float *o = out + ((size_t) y * F.width + x) * 5;
        if (accumulate) { for (int i = 0; i < 5; ++i) o[i] += acc[i]; }
        else { for (int i = 0; i < 5; ++i) o[i] = acc[i]; }
    }
    if (invalid) atomicAdd(invalidCount, invalid);
}

/* copy per-sample radiance out in [y][x][sample] order (tests) */
__global__ void k_export_samples(DevScene S, RenderConst rc, const float4 *L, const int32_t *tileSlot, int tilesX,
                                 float4 *out, uint32_t sppTotal) {
    const DevFilm &F = S.film;
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t) F.width * F.height * rc.sppPass;
    if (i >= n) return;
    const uint32_t k = (uint32_t) (i % rc.sppPass);
    const size_t p = i / rc.sppPass;
    const int x = (int) (p % F.width), y = (int) (p / F.width);
    const int tx = x >> rc.tileShift, ty = y >> rc.tileShift;
    const int32_t ts = tileSlot[ty * tilesX + tx];
    float4 v = make_float4(0, 0, 0, 0);
    if (ts >= 0) {
        const uint32_t m = spreadBits((uint32_t) (x - (tx << rc.tileShift))) | (spreadBits((uint32_t) (y - (ty << rc.tileShift))) << 1);
        v = L[(((unsigned long long) ts * rc.sppPass + k) << (2 * rc.tileShift)) | m];
    }
    out[p * sppTotal + rc.sppFirst + k] = v;
}

/* standalone ray casts for phip_trace */
__global__ __launch_bounds__(BLOCK) void k_raycast(DevScene S, const phip_ray *rays, size_t n, phip_hit *hits, uint8_t *occluded, PathPool P) {
    const size_t i = (size_t) blockIdx.x * BLOCK + threadIdx.x;
    TravStack stk; setupTraversal(S, g_smem, P.spill + i * SPILL_DEPTH, stk);
    uint32_t nodeVisits = 0, triTests = 0, shNodeVisits = 0, shTriTests = 0;
    if (i < n) {
        const phip_ray ry = rays[i];
        const V3 o(ry.o[0], ry.o[1], ry.o[2]), d(ry.d[0], ry.d[1], ry.d[2]);
        float mint, maxt;
        if (hits) {
            TravResult r; r.prim = PHIP_NO_HIT; r.t = INFINITY; r.u = r.v = 0;
            if (clipToScene<false>(S, o, d, ry.mint, ry.maxt, mint, maxt))
                traverse<false>(S, o, d, mint, maxt, stk, r, nodeVisits, triTests);
            phip_hit h; h.t = r.t; h.u = r.u; h.v = r.v; h.prim = r.prim;
            hits[i] = h;
        }
        if (occluded) {
            TravResult r; bool occ = false;
            if (clipToScene<true>(S, o, d, ry.mint, ry.maxt, mint, maxt))
                occ = traverse<true>(S, o, d, mint, maxt, stk, r, shNodeVisits, shTriTests);
            occluded[i] = occ ? 1 : 0;
        }
    }
    const uint32_t waveId = (uint32_t) (i >> 6);
    waveStat(P, ST_NODE, waveId, nodeVisits);
    waveStat(P, ST_TRI, waveId, triTests);
    waveStat(P, ST_SH_NODE, waveId, shNodeVisits);
    waveStat(P, ST_SH_TRI, waveId, shTriTests);
}

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

struct phip_scene {
    int device = 0;
    phip_scene_desc descCopy;        /* scalar fields only */
    HostBVH bvh;
    DevBuf<float4> nodes, nodes8, tris, triShade;
    DevBuf<uint2> spill8;
    int traversal = 2;               /* 2 = persistent per-lane BVH4 traversal with dynamic refill (default), 0 = one launch lane per slot, 1 = 8 lanes per ray over the BVH8
                                        (PHIP_TRAVERSAL=group; measured 2-3x slower: too few rays in flight per CU) */
    DevBuf<DevMaterial> materials;
    DevBuf<float> emitterTab;
    DevScene dev;
    /* render-time buffers (grown on demand, reused between calls) */
    DevBuf<float4> rayO, rayD, hit, thr, shadow, L, sampleOut;
    DevBuf<uint4> info; DevBuf<uint32_t> state; DevBuf<float2> mis;
    DevBuf<Counters> counters;
    DevBuf<uint32_t> tileOrigin, shadowCount, blockDead, spill, blockShard; DevBuf<int32_t> tileSlot;
    DevBuf<unsigned long long> dynCounter;
    DevBuf<unsigned long long> stat;
    DevBuf<float> film; DevBuf<unsigned long long> invalid;
    uint32_t lastSpp = 0, nLocalTiles = 0;
    int tileKey[3] = { -1, -1, -1 };
    bool haveSamples = false;
    bool mergedRays = false;         /* last render used k_rays_p (closest + any hit in one launch) */
    int materialMask = MM_ALL;       /* leaf BSDF models present: selects the k_shade instantiation */
    std::atomic<int> cancel{ 0 };
    std::mutex renderLock;
    hipStream_t stream = nullptr;
};

static std::vector<DevMaterial> convertMaterials(const phip_material *materials, uint32_t nMaterials) {
    if (nMaterials && !materials) throw std::runtime_error("materials is NULL");
    std::vector<DevMaterial> mats(nMaterials);
    for (uint32_t i = 0; i < nMaterials; ++i) {
        const phip_material &m = materials[i];
        DevMaterial &o = mats[i];
        memset(&o, 0, sizeof(o));
        o.type = m.type; o.nested0 = m.nested[0]; o.nested1 = m.nested[1];
        for (int k = 0; k < 3; ++k) { o.refl[k] = m.reflectance[k]; o.trans[k] = m.transmittance[k]; o.eta[k] = m.eta[k]; o.k[k] = m.k[k]; }
        o.distribution = m.distribution; o.sampleVisible = m.sample_visible ? 1 : 0;
        switch (m.type) {
            case PHIP_BSDF_DIFFUSE: {
                const float mx = std::max(m.reflectance[0], std::max(m.reflectance[1], m.reflectance[2]));
                if (mx > 1.0f) throw std::runtime_error("diffuse reflectance > 1 (ensureEnergyConservation, diffuse.cpp:95)");
                if (mx > 0) o.flags |= MF_SMOOTH;            /* component list empty otherwise, diffuse.cpp:97-100 */
            } break;
            case PHIP_BSDF_DIELECTRIC:
                if (!(m.eta[0] > 0)) throw std::runtime_error("dielectric eta must be positive");
                o.flags |= MF_TRANS_OR_BACK; break;
            case PHIP_BSDF_ROUGHCONDUCTOR: {
                if (m.distribution > PHIP_MF_GGX) { g_err = "unsupported microfacet distribution"; throw std::invalid_argument("unsupported microfacet distribution (only beckmann, ggx)"); }
                o.flags |= MF_SMOOTH;
                /* alpha = ConstantFloatTexture.eval().average() (roughconductor.cpp:275-280), clamp microfacet.h:113-114 */
                o.alphaU = std::max(V3(m.alpha_u).average(), 1e-4f);
                o.alphaV = std::max(V3(m.alpha_v).average(), 1e-4f);
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
    if (d.abi_version != PHIP_ABI_VERSION) throw std::runtime_error("phip_scene_desc.abi_version mismatch");
    if (d.n_vertices && !d.positions) throw std::runtime_error("positions is NULL");
    if (d.n_triangles && !d.indices) throw std::runtime_error("indices is NULL");
    if (d.film.crop_width <= 0 || d.film.crop_height <= 0 || d.film.width <= 0 || d.film.height <= 0)
        throw std::runtime_error("invalid film size");
    if (d.film.crop_offset_x < 0 || d.film.crop_offset_y < 0 || d.film.crop_offset_x + d.film.crop_width > d.film.width ||
        d.film.crop_offset_y + d.film.crop_height > d.film.height)
        throw std::runtime_error("invalid crop window");          /* film.cpp:44-48 */
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
        o.material = s.material; o.emitter = s.emitter; o.hasNormals = s.has_normals ? 1 : 0;
        o.firstTri = s.first_triangle; o.nTris = s.n_triangles; o.cdfOffset = 0; o.invSurfaceArea = 0; o.pad = 0;
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

    /* materials */
    std::vector<DevMaterial> mats = convertMaterials(d.materials, d.n_materials);

    /* emitters + selection pdf, scene.cpp:375-381 */
    std::vector<DevEmitter> ems(d.n_emitters);
    int32_t envEmitter = -1;
    std::vector<float> ecdf(1, 0.0f);
    for (uint32_t i = 0; i < d.n_emitters; ++i) {
        const phip_emitter &e = d.emitters[i];
        if (e.type == PHIP_EMITTER_CONSTANT) {
            if (envEmitter >= 0) throw std::runtime_error("The scene may only contain one environment emitter");   /* scene.cpp:510-513 */
            envEmitter = (int32_t) i;
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
    buildBVH(d.positions, d.indices, d.n_triangles, sc->bvh);
    if (3 * sc->bvh.maxDepth + 2 > STACK_DEPTH + SPILL_DEPTH) throw std::runtime_error("BVH too deep for the traversal stack");
    if (sc->bvh.tris.size() / 12 >= (1u << 28)) throw std::runtime_error("too many triangle records for the leaf reference encoding");

    /* upload */
    HIP_TRY(hipSetDevice(sc->device));
    /* shading records (dv_scene.h): the per-triangle constants come from the same __host__ __device__
       functions the kernel would run, so precomputing them does not change a single bit */
    std::vector<float4> ts((size_t) TRISHADE_FLOAT4S * d.n_triangles);
    for (uint32_t i = 0; i < d.n_triangles; ++i) {
        const DevShape &sh = shapes[triShape[i]];
        const DevMaterial &m = mats[sh.material];
        const uint32_t *ix = d.indices + 3 * (size_t) i;
        const V3 p0(d.positions[3 * ix[0]], d.positions[3 * ix[0] + 1], d.positions[3 * ix[0] + 2]);
        const V3 p1(d.positions[3 * ix[1]], d.positions[3 * ix[1] + 1], d.positions[3 * ix[1] + 2]);
        const V3 p2(d.positions[3 * ix[2]], d.positions[3 * ix[2] + 1], d.positions[3 * ix[2] + 2]);
        const bool twosided = m.type == PHIP_BSDF_TWOSIDED;
        const uint32_t front = twosided ? m.nested0 : sh.material, back = twosided ? m.nested1 : sh.material;
        uint32_t flags = (sh.hasNormals ? TS_VERTEX_NORMALS : 0u) | (twosided ? TS_TWOSIDED : 0u)
                       | ((m.flags & MF_SMOOTH) ? TS_MF_SMOOTH : 0u) | ((m.flags & MF_TRANS_OR_BACK) ? TS_TRANS_OR_BACK : 0u);
        V3 a, b, c;
        if (sh.hasNormals) {
            a = V3(d.normals[3 * ix[0]], d.normals[3 * ix[0] + 1], d.normals[3 * ix[0] + 2]);
            b = V3(d.normals[3 * ix[1]], d.normals[3 * ix[1] + 1], d.normals[3 * ix[1] + 2]);
            c = V3(d.normals[3 * ix[2]], d.normals[3 * ix[2] + 1], d.normals[3 * ix[2] + 2]);
        } else {
            const V3 side1(p1 - p0), side2(p2 - p0);
            Frame f; triShadingFrame(triFaceNormal(side1, side2), side1, f);
            a = f.n; b = f.s; c = f.t;
        }
        float4 *r = ts.data() + (size_t) TRISHADE_FLOAT4S * i;
        r[0] = make_float4(p0.x, p0.y, p0.z, pm_from_bits(front));
        r[1] = make_float4(p1.x, p1.y, p1.z, pm_from_bits(back));
        r[2] = make_float4(p2.x, p2.y, p2.z, pm_from_bits((uint32_t) sh.emitter));
        r[3] = make_float4(a.x, a.y, a.z, pm_from_bits(flags));
        r[4] = make_float4(b.x, b.y, b.z, 0.0f);
        r[5] = make_float4(c.x, c.y, c.z, 0.0f);
    }
    if (ts.empty()) sc->triShade.alloc(TRISHADE_FLOAT4S); else sc->triShade.upload(ts.data(), ts.size());
    if (sc->bvh.nodes.empty()) sc->nodes.alloc(8);
    else sc->nodes.upload((const float4 *) sc->bvh.nodes.data(), sc->bvh.nodes.size() / 4);
    if (sc->bvh.nodes8.empty()) sc->nodes8.alloc(16);
    else sc->nodes8.upload((const float4 *) sc->bvh.nodes8.data(), sc->bvh.nodes8.size() / 4);
    sc->tris.upload((const float4 *) sc->bvh.tris.data(), sc->bvh.tris.size() / 4);
    sc->materials.upload(mats.data(), mats.size());
    sc->materialMask = 0;
    for (const DevMaterial &m : mats) {
        if (m.type == PHIP_BSDF_ROUGHCONDUCTOR) sc->materialMask |= MM_ROUGH;
        if (m.type == PHIP_BSDF_DIELECTRIC) sc->materialMask |= MM_DIELECTRIC;
    }
    if (const char *e = getenv("PHIP_SHADE_GENERIC")) if (atoi(e)) sc->materialMask = MM_ALL;
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
        r[EM_TYPE] = pm_from_bits(isArea(i) ? (uint32_t) PHIP_EMITTER_AREA : (uint32_t) PHIP_EMITTER_CONSTANT);
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
            const float *src = (const float *) (ts.data() + (size_t) TRISHADE_FLOAT4S * sh.firstTri);
            tab.insert(tab.end(), src, src + (size_t) sh.nTris * 4 * TRISHADE_FLOAT4S);
        }
    }
    if (tab.size() >= (1ull << 31)) throw std::runtime_error("emitter table too large");
    sc->emitterTab.upload(tab.data(), tab.size());

    DevScene &D = sc->dev;
    memset(&D, 0, sizeof(D));
    D.nodes = sc->nodes.p; D.nodes8 = sc->nodes8.p; D.tris = sc->tris.p; D.triShade = sc->triShade.p;
    D.materials = sc->materials.p; D.nMaterials = (uint32_t) mats.size();
    D.emitterTab = sc->emitterTab.p; D.emitterTabSize = (uint32_t) tab.size();
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
    D.rootRef = sc->bvh.rootRef; D.rootRef8 = sc->bvh.rootRef8; D.nTriangles = d.n_triangles;
    /* LDS staging plan: stack depth from the tree depth (3 pushes per BVH4 level), top-of-tree node cache
       (nodes are in breadth-first order), all triangle records if there are few */
    D.stackDepth = (uint32_t) std::min<int>(STACK_DEPTH, std::max<int>(4, 3 * ((int) sc->bvh.maxDepth - 1) + 1));
    D.nodeCache = std::min<uint32_t>(sc->bvh.nNodes, NODE_CACHE_MAX);
    D.triCache = (sc->bvh.tris.size() / 12 <= TRI_CACHE_MAX) ? (uint32_t) (sc->bvh.tris.size() / 12) : 0u;
    if (const char *e = getenv("PHIP_NODE_CACHE")) D.nodeCache = std::min<uint32_t>(sc->bvh.nNodes, (uint32_t) atoi(e));
    if (D.nodeCache == 0) D.triCache = 0;
    if (const char *e = getenv("PHIP_TRAVERSAL")) sc->traversal = (strcmp(e, "group") == 0) ? 1 : (strcmp(e, "lane") == 0 ? 0 : 2);
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
    sc->counters.alloc(1);
    sc->invalid.alloc(1);
    sc->dynCounter.alloc(DYN_SHARDS * DYN_STRIDE);
}

static size_t traversalLdsBytes(const DevScene &D) {
    return (size_t) D.stackDepth * BLOCK * sizeof(uint32_t) + (size_t) D.nodeCache * NODE_LDS_STRIDE * sizeof(float4) + (size_t) D.triCache * 3 * sizeof(float4);
}

static void algorithmicBytes(const phip_scene *sc, phip_stats &st) {
    /* SURVEY 8(d) with this structure's sizes: 128-byte BVH4 node visits, 48-byte triangle records
       (no separate index array: records are stored in leaf order) */
    const double film = 20.0 * (double) sc->dev.film.width * sc->dev.film.height;
    const double nodeBytes = sc->traversal == 1 ? 256.0 : 128.0;
    st.algorithmic_bytes = nodeBytes * (double) (st.closest_node_visits + st.shadow_node_visits) +
           48.0 * (double) (st.closest_triangle_tests + st.shadow_triangle_tests) +
           (64.0 + 40.0 + 108.0) * (double) st.closest_rays + (64.0 + 4.0) * (double) st.shadow_rays +
           104.0 * (double) st.path_vertices + film;
    /* closest-hit kernel: node + triangle fetches, ray read (32 B), hit record write (16 B) ... counted with
       the SURVEY's read+write convention: ray 64 B, hit 40 B */
    st.trace_kernel_bytes = nodeBytes * (double) st.closest_node_visits + 48.0 * (double) st.closest_triangle_tests +
           (64.0 + 40.0) * (double) st.closest_rays;
    if (sc->mergedRays)    /* k_rays_p also casts the shadow rays: their node + record fetches, entry read, 4-byte result */
        st.trace_kernel_bytes += nodeBytes * (double) st.shadow_node_visits + 48.0 * (double) st.shadow_triangle_tests +
               (64.0 + 4.0) * (double) st.shadow_rays;
}

static int renderImpl(phip_scene *sc, const phip_render_params *p, float *dOut /* device */, phip_stats *stats) {
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    if (p->spp <= 0) throw std::invalid_argument("spp must be > 0");
    if (p->rr_depth <= 0) throw std::invalid_argument("'rrDepth' must be set to a value greater than zero!");                       /* integrator.cpp:219-220 */
    if (p->max_depth <= 0 && p->max_depth != -1) throw std::invalid_argument("'maxDepth' must be set to -1 (infinite) or a value greater than zero!"); /* :222-223 */
    if (p->sampler != PHIP_SAMPLER_CTR) throw std::invalid_argument("unknown sampler kind");
    const int bs = p->block_size > 0 ? p->block_size : 32;
    if (bs < 2 || bs > 128 || (bs & (bs - 1))) throw std::invalid_argument("block_size must be a power of two in [2,128] (mitsuba.cpp:233-239 allows 2..128)");
    const int shardCount = p->shard_count > 0 ? p->shard_count : 1;
    if (p->shard_index < 0 || p->shard_index >= shardCount) throw std::invalid_argument("shard_index out of range");
    if (p->device != sc->device) throw std::invalid_argument("scene was created on a different device");
    HIP_TRY(hipSetDevice(sc->device));
    std::lock_guard<std::mutex> lock(sc->renderLock);
    sc->cancel.store(0);

    DevScene D = sc->dev;
    D.film.blockSize = bs;
    if (bs < D.film.border) throw std::invalid_argument("The block size must be larger than the image reconstruction filter radius!"); /* renderproc.cpp:175-176 */
    const int W = D.film.width, H = D.film.height;
    int tileShift = 0; while ((1 << tileShift) < bs) ++tileShift;

    /* tile -> shard assignment in the reference's spiral order (cached between calls with the same layout) */
    const int tilesX = (W + bs - 1) / bs, tilesY = (H + bs - 1) / bs;
    if (sc->tileKey[0] != bs || sc->tileKey[1] != p->shard_index || sc->tileKey[2] != shardCount) {
        std::vector<std::pair<int, int>> spiral;
        spiralBlocks(W, H, bs, spiral);
        std::vector<int32_t> tileSlot((size_t) tilesX * tilesY, -1);
        std::vector<uint32_t> tileOrigin;
        for (size_t i = 0; i < spiral.size(); ++i) {
            if ((int) (i % (size_t) shardCount) != p->shard_index) continue;
            tileSlot[(size_t) spiral[i].second * tilesX + spiral[i].first] = (int32_t) tileOrigin.size();
            tileOrigin.push_back((uint32_t) (spiral[i].first * bs) | ((uint32_t) (spiral[i].second * bs) << 16));
        }
        sc->nLocalTiles = (uint32_t) tileOrigin.size();
        if (tileOrigin.empty()) sc->tileOrigin.alloc(1);
        else sc->tileOrigin.upload(tileOrigin.data(), tileOrigin.size());
        sc->tileSlot.upload(tileSlot.data(), tileSlot.size());
        sc->tileKey[0] = bs; sc->tileKey[1] = p->shard_index; sc->tileKey[2] = shardCount;
    }
    const uint32_t nLocalTiles = sc->nLocalTiles;

    hipStream_t stream = (hipStream_t) p->stream;
    if (!stream) { if (!sc->stream) HIP_TRY(hipStreamCreate(&sc->stream)); stream = sc->stream; }

    /* passes: bound the per-sample buffer (16 B per sample id) */
    const unsigned long long tilePixels = (unsigned long long) bs * bs;
    const unsigned long long maxIdsPerPass = (1ull << 32) - 1;                 /* sample ids are 32-bit in the slot state */
    unsigned long long budgetIds = (24ull << 30) / 16;                        /* 24 GiB of sample buffer */
    if (const char *e = getenv("PHIP_MAX_PASS_SAMPLES")) budgetIds = std::max(1ull, strtoull(e, nullptr, 10));   /* test hook: force several passes */
    unsigned long long idsPerSpp = (unsigned long long) nLocalTiles * tilePixels;
    uint32_t sppPerPass = (uint32_t) p->spp;
    if (idsPerSpp > 0) {
        unsigned long long cap = std::min(maxIdsPerPass, budgetIds) / idsPerSpp;
        if (cap < 1) cap = 1;
        sppPerPass = (uint32_t) std::min<unsigned long long>(cap, (unsigned long long) p->spp);
    }
    const bool keepSamples = (p->flags & PHIP_FLAG_SAMPLE_BUFFER) != 0;
    if (keepSamples) { sc->sampleOut.alloc((size_t) W * H * (size_t) p->spp); HIP_TRY(hipMemsetAsync(sc->sampleOut.p, 0, sc->sampleOut.n * sizeof(float4), stream)); }
    sc->haveSamples = keepSamples; sc->lastSpp = (uint32_t) p->spp;

    /* path pool */
    const unsigned long long idsFirstPass = idsPerSpp * sppPerPass;
    /* pool size: large enough that per-launch fixed costs vanish, small enough that the tail (slots
       running dry at the end of a pass) stays a small fraction of the pass (measured: 4M / 8M slots) */
    const unsigned long long poolCap = idsFirstPass >= (256ull << 20) ? (1ull << 23) : (1ull << 22);
    uint32_t capacity = (uint32_t) std::min<unsigned long long>(std::max<unsigned long long>(idsFirstPass, BLOCK), poolCap);
    capacity = (capacity + BLOCK - 1) / BLOCK * BLOCK;
    if (const char *e = getenv("PHIP_POOL")) { capacity = (uint32_t) std::max(BLOCK, atoi(e)) / BLOCK * BLOCK; }
    const uint32_t nWaves = capacity / 64, nBlocks = capacity / BLOCK;
    if (sc->rayO.n < capacity) {
        sc->rayO.alloc(capacity); sc->rayD.alloc(capacity); sc->hit.alloc(capacity); sc->thr.alloc(capacity);
        sc->mis.alloc(capacity); sc->info.alloc(capacity); sc->state.alloc(capacity); sc->shadow.alloc(3 * (size_t) capacity);
        sc->shadowCount.alloc(nBlocks); sc->blockDead.alloc(nBlocks); sc->blockShard.alloc(nBlocks); sc->stat.alloc((size_t) ST_COUNT * nWaves); sc->spill.alloc((size_t) capacity * SPILL_DEPTH);
        sc->spill8.alloc((size_t) nWaves * 8 * SPILL8);
    }
    PathPool P;
    P.rayO = sc->rayO.p; P.rayD = sc->rayD.p; P.hit = sc->hit.p; P.thr = sc->thr.p; P.mis = sc->mis.p; P.info = sc->info.p; P.state = sc->state.p;
    P.shadow = sc->shadow.p; P.shadowCount = sc->shadowCount.p; P.blockDead = sc->blockDead.p; P.stat = sc->stat.p; P.spill = sc->spill.p; P.spill8 = sc->spill8.p; P.capacity = capacity; P.nWaves = nWaves;
    if (sc->L.n < idsFirstPass) sc->L.alloc((size_t) idsFirstPass);

    phip_stats st; memset(&st, 0, sizeof(st));
    const bool timing = (p->flags & PHIP_FLAG_KERNEL_TIMING) != 0;
    std::vector<hipEvent_t> evTrace, evShadow, evShade, evFilm;
    auto newEvent = [&](std::vector<hipEvent_t> &v) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); v.push_back(e); return e; };

    HIP_TRY(hipMemsetAsync(sc->invalid.p, 0, sizeof(unsigned long long), stream));
    const dim3 grid((capacity + BLOCK - 1) / BLOCK), block(BLOCK);
    const size_t ldsBytes = traversalLdsBytes(D);
    /* persistent kernels: exactly the resident set (TRACE_WAVES waves per SIMD = TRACE_WAVES blocks of 256 per CU) */
    int nCU = 256; { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, sc->device) == hipSuccess) nCU = prop.multiProcessorCount; }
    /* ... but never more blocks per CU than their LDS (stack + node/record cache) allows: a persistent grid larger than
       the resident set would serialise */
    const int ldsFit = (int) std::max<size_t>(1, (size_t) (160 * 1024) / std::max<size_t>(ldsBytes + 64, 1));
    auto persistentGrid = [&](int blocksPerCU) {
        return dim3((unsigned) std::max(1, std::min<int>(nCU * std::min(blocksPerCU, ldsFit), (int) ((capacity + BLOCK - 1) / BLOCK))));
    };
    const dim3 pgrid = persistentGrid(TRACE_WAVES), pgridTrace = persistentGrid(TRACE_P_WAVES), pgridRays = persistentGrid(RAYS_WAVES);
    Counters hc;
    bool cancelled = false;
    const bool forcePersist = getenv("PHIP_TRACE_PERSIST") != nullptr;   /* experiment hook */
    /* big trees: closest-hit and any-hit rays share one persistent launch (measured +2..4 % on the 250k-triangle scenes;
       on the Cornell box the plain per-slot closest-hit launch wins, so the kernels stay separate there) */
    bool merged = sc->traversal == 2 && sc->bvh.nNodes >= 64;
    if (const char *e = getenv("PHIP_MERGED")) merged = sc->traversal == 2 && atoi(e) != 0;
    sc->mergedRays = merged;

    for (uint32_t sppDone = 0; sppDone < (uint32_t) p->spp && !cancelled; sppDone += sppPerPass) {
        RenderConst rc;
        rc.sppPass = std::min(sppPerPass, (uint32_t) p->spp - sppDone); rc.sppFirst = sppDone;
        rc.sppMagic = (uint32_t) std::min<unsigned long long>((1ull << 32) / rc.sppPass, 0xFFFFFFFFull);
        rc.tilePixels = (uint32_t) tilePixels; rc.tileShift = (uint32_t) tileShift; rc.nLocalTiles = nLocalTiles;
        rc.totalIds = idsPerSpp * rc.sppPass;
        rc.maxDepth = p->max_depth; rc.rrDepth = p->rr_depth; rc.strictNormals = p->strict_normals; rc.hideEmitters = p->hide_emitters;
        rc.seed = p->seed; rc.tileOrigin = sc->tileOrigin.p; rc.countAlive = 0;
        /* static share: the first 3/4 of every slot's samples; the remainder is handed out dynamically */
        {
            const unsigned long long perSlot = rc.totalIds / capacity;
            unsigned long long staticPerSlot = perSlot - perSlot / 4;
            if (const char *e = getenv("PHIP_STATIC_PERCENT")) staticPerSlot = perSlot * (unsigned long long) atoi(e) / 100;
            rc.staticIds = staticPerSlot * capacity;
            const unsigned long long dyn = rc.totalIds - rc.staticIds;
            rc.shardIds = (dyn + DYN_SHARDS - 1) / DYN_SHARDS;
            rc.dynCounter = sc->dynCounter.p; rc.blockShard = sc->blockShard.p;
            HIP_TRY(hipMemsetAsync(sc->blockShard.p, 0, nBlocks * sizeof(uint32_t), stream));
            HIP_TRY(hipMemsetAsync(sc->dynCounter.p, 0, DYN_SHARDS * DYN_STRIDE * sizeof(unsigned long long), stream));
        }

        HIP_TRY(hipMemsetAsync(sc->counters.p, 0, sizeof(Counters), stream));
        HIP_TRY(hipMemsetAsync(sc->stat.p, 0, (size_t) ST_COUNT * nWaves * sizeof(unsigned long long), stream));
        HIP_TRY(hipMemsetAsync(sc->shadowCount.p, 0, (size_t) nBlocks * sizeof(uint32_t), stream));
        HIP_TRY(hipMemsetAsync(sc->blockDead.p, 0, (size_t) nBlocks * sizeof(uint32_t), stream));
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t) sc->state.p, (int) F_FRESH, (size_t) capacity, stream));
        if (rc.totalIds) HIP_TRY(hipMemsetAsync(sc->L.p, 0, (size_t) rc.totalIds * sizeof(float4), stream));

        HIP_TRY(hipStreamSynchronize(stream));
        const auto tLoop0 = clk::now();
        uint32_t iter = 0;
        bool done = rc.totalIds == 0;
        while (!done) {
            const bool check = ((iter + 1) & 7) == 0 || rc.totalIds <= (unsigned long long) capacity * 4;
            rc.countAlive = check ? 1 : 0;
            if (timing) HIP_TRY(hipEventRecord(newEvent(evShade), stream));
            {
                typedef void (*ShadeKernel)(DevScene, PathPool, RenderConst, float4 *);
                static const ShadeKernel table[2][4] = {
                    { k_shade<0, false>, k_shade<MM_ROUGH, false>, k_shade<MM_DIELECTRIC, false>, k_shade<MM_ALL, false> },
                    { k_shade<0, true>, k_shade<MM_ROUGH, true>, k_shade<MM_DIELECTRIC, true>, k_shade<MM_ALL, true> } };
                hipLaunchKernelGGL(table[rc.strictNormals ? 1 : 0][sc->materialMask & MM_ALL], grid, block, 0, stream, D, P, rc, sc->L.p);
            }
            if (timing) HIP_TRY(hipEventRecord(newEvent(evShade), stream));
            if (merged) {
                if (timing) HIP_TRY(hipEventRecord(newEvent(evTrace), stream));
                hipLaunchKernelGGL(k_rays_p, pgridRays, block, ldsBytes, stream, D, P, sc->L.p);
                if (timing) HIP_TRY(hipEventRecord(newEvent(evTrace), stream));
            } else {
                if (timing) HIP_TRY(hipEventRecord(newEvent(evShadow), stream));
                if (sc->traversal == 2) hipLaunchKernelGGL(k_shadow_p, pgrid, block, ldsBytes, stream, D, P, sc->L.p);
                else if (sc->traversal == 1) hipLaunchKernelGGL(k_shadow8, grid, block, 0, stream, D, P, sc->L.p);
                else hipLaunchKernelGGL(k_shadow, grid, block, ldsBytes, stream, D, P, sc->L.p);
                if (timing) HIP_TRY(hipEventRecord(newEvent(evShadow), stream));
                if (timing) HIP_TRY(hipEventRecord(newEvent(evTrace), stream));
                if (sc->traversal == 2 && (sc->bvh.nNodes >= 64 || forcePersist)) { if (sc->bvh.nNodes >= 64) hipLaunchKernelGGL(k_trace_p<false>, pgridTrace, block, ldsBytes, stream, D, P); else hipLaunchKernelGGL(k_trace_p<true>, pgridTrace, block, ldsBytes, stream, D, P); }
                else if (sc->traversal == 2) hipLaunchKernelGGL(k_trace, grid, block, ldsBytes, stream, D, P);   /* tiny trees: the plain per-slot launch wins (measured) */
                else if (sc->traversal == 1) hipLaunchKernelGGL(k_trace8, grid, block, 0, stream, D, P);
                else hipLaunchKernelGGL(k_trace, grid, block, ldsBytes, stream, D, P);
                if (timing) HIP_TRY(hipEventRecord(newEvent(evTrace), stream));
            }
            ++iter;
            if (check) {
                /* termination test: only the live-slot row is summed inside the loop */
                HIP_TRY(hipMemsetAsync(&sc->counters.p->total[ST_ALIVE], 0, sizeof(unsigned long long), stream));
                hipLaunchKernelGGL(k_reduce_stats, dim3(1, REDUCE_SPLIT), dim3(256), 0, stream, P, sc->counters.p, (int) ST_ALIVE);
                HIP_TRY(hipMemcpyAsync(&hc.total[ST_ALIVE], &sc->counters.p->total[ST_ALIVE], sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipStreamSynchronize(stream));
                if (hc.total[ST_ALIVE] == 0) done = true;
                if (sc->cancel.load()) { cancelled = true; done = true; }
            }
        }
        HIP_TRY(hipGetLastError());
        st.iterations += iter;
        HIP_TRY(hipStreamSynchronize(stream));
        if (getenv("PHIP_DEBUG_TIMING")) fprintf(stderr, "[phip] setup %.2f ms, loop %.2f ms (%u iterations)\n", std::chrono::duration<double, std::milli>(tLoop0 - t0).count(), std::chrono::duration<double, std::milli>(clk::now() - tLoop0).count(), iter);
        /* film */
        if (timing) HIP_TRY(hipEventRecord(newEvent(evFilm), stream));
        {
            const dim3 fg((W + 15) / 16, (H + 15) / 16);
            const int reach = (int) std::floor(D.film.radius + 0.5f);
            if (reach <= FILM_MAX_REACH && !getenv("PHIP_FILM_GENERIC"))
                if (reach <= 2)
                    hipLaunchKernelGGL(k_film_tiled<2>, fg, block, 0, stream, D, rc, (const float4 *) sc->L.p, (const int32_t *) sc->tileSlot.p, tilesX, dOut,
                                       sppDone > 0 ? 1 : 0, sc->invalid.p, reach);
                else
                    hipLaunchKernelGGL(k_film_tiled<FILM_MAX_REACH>, fg, block, 0, stream, D, rc, (const float4 *) sc->L.p, (const int32_t *) sc->tileSlot.p, tilesX, dOut,
                                       sppDone > 0 ? 1 : 0, sc->invalid.p, reach);
            else
                hipLaunchKernelGGL(k_film, fg, block, 0, stream, D, rc, (const float4 *) sc->L.p, (const int32_t *) sc->tileSlot.p, tilesX, dOut,
                                   sppDone > 0 ? 1 : 0, sc->invalid.p);
        }
        if (timing) HIP_TRY(hipEventRecord(newEvent(evFilm), stream));
        if (keepSamples && rc.totalIds) {
            const size_t n = (size_t) W * H * rc.sppPass;
            hipLaunchKernelGGL(k_export_samples, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, stream, D, rc, (const float4 *) sc->L.p,
                               (const int32_t *) sc->tileSlot.p, tilesX, sc->sampleOut.p, (uint32_t) p->spp);
        }
        HIP_TRY(hipMemsetAsync(sc->counters.p, 0, sizeof(Counters), stream));
        hipLaunchKernelGGL(k_reduce_stats, dim3(ST_COUNT, REDUCE_SPLIT), dim3(256), 0, stream, P, sc->counters.p, 0);
        HIP_TRY(hipMemcpyAsync(&hc, sc->counters.p, sizeof(Counters), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipGetLastError());
        st.samples += hc.total[ST_SAMPLES]; st.closest_rays += hc.total[ST_CLOSEST_RAYS]; st.shadow_rays += hc.total[ST_SHADOW_RAYS];
        st.path_vertices += hc.total[ST_VERTICES]; st.closest_node_visits += hc.total[ST_NODE]; st.closest_triangle_tests += hc.total[ST_TRI];
        st.shadow_node_visits += hc.total[ST_SH_NODE]; st.shadow_triangle_tests += hc.total[ST_SH_TRI];
    }
    if (nLocalTiles == 0) { HIP_TRY(hipMemsetAsync(dOut, 0, (size_t) W * H * 5 * sizeof(float), stream)); HIP_TRY(hipStreamSynchronize(stream)); }
    unsigned long long inv = 0;
    HIP_TRY(hipMemcpy(&inv, sc->invalid.p, sizeof(inv), hipMemcpyDeviceToHost));
    st.invalid_samples = inv;
    auto sumPairs = [&](std::vector<hipEvent_t> &v) { double ms = 0; for (size_t i = 0; i + 1 < v.size(); i += 2) { float t = 0; (void) hipEventElapsedTime(&t, v[i], v[i + 1]); ms += t; } for (auto e : v) (void) hipEventDestroy(e); return ms; };
    st.trace_kernel_ms = sumPairs(evTrace); st.shadow_kernel_ms = sumPairs(evShadow);
    st.shade_kernel_ms = sumPairs(evShade); st.film_kernel_ms = sumPairs(evFilm);
    st.render_ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
    algorithmicBytes(sc, st);
    if (stats) *stats = st;
    if (cancelled) return setErr(PHIP_ERR_CANCELLED, "rendering was cancelled");
    return PHIP_OK;
}

/* ======================================================================================
 *  C ABI
 * ====================================================================================== */
extern "C" {

const char *phip_last_error(void) { return g_err.c_str(); }
const char *phip_version(void) { return "path_hip 0.1 (gfx950, abi 1)"; }

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
    sc->device = device;
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
    (void) hipSetDevice(scene->device);
    if (scene->stream) (void) hipStreamDestroy(scene->stream);
    delete scene;
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

int phip_render(phip_scene *scene, const phip_render_params *params, float *out_rgbaw, phip_stats *out_stats) {
    if (!scene || !params || !out_rgbaw) return setErr(PHIP_ERR_INVALID, "NULL argument");
    try {
        HIP_TRY(hipSetDevice(scene->device));
        const size_t n = (size_t) scene->dev.film.width * scene->dev.film.height * 5;
        if (scene->film.n < n) scene->film.alloc(n);
        int rc = renderImpl(scene, params, scene->film.p, out_stats);
        if (rc != PHIP_OK) return rc;
        HIP_TRY(hipMemcpy(out_rgbaw, scene->film.p, n * sizeof(float), hipMemcpyDeviceToHost));
        return PHIP_OK;
    } catch (const std::invalid_argument &e) {
        return setErr(PHIP_ERR_INVALID, e.what());
    } catch (const std::exception &e) {
        return setErr(PHIP_ERR_DEVICE, e.what());
    }
}

int phip_get_samples(phip_scene *scene, float *out_rgba, size_t n_samples) {
    if (!scene || !out_rgba) return setErr(PHIP_ERR_INVALID, "NULL argument");
    if (!scene->haveSamples) return setErr(PHIP_ERR_INVALID, "last render did not set PHIP_FLAG_SAMPLE_BUFFER");
    const size_t n = (size_t) scene->dev.film.width * scene->dev.film.height * scene->lastSpp;
    if (n_samples != n) return setErr(PHIP_ERR_INVALID, "n_samples does not match crop_w*crop_h*spp");
    hipError_t e = hipMemcpy(out_rgba, scene->sampleOut.p, n * sizeof(float4), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return setErr(PHIP_ERR_DEVICE, hipGetErrorString(e));
    return PHIP_OK;
}

int phip_trace(phip_scene *scene, const phip_ray *rays, size_t n, phip_hit *hits, uint8_t *occluded, phip_stats *out_stats) {
    if (!scene || (!rays && n)) return setErr(PHIP_ERR_INVALID, "NULL argument");
    try {
        HIP_TRY(hipSetDevice(scene->device));
        std::lock_guard<std::mutex> lock(scene->renderLock);
        DevBuf<phip_ray> dr; DevBuf<phip_hit> dh; DevBuf<uint8_t> dz;
        if (n) {
            dr.upload(rays, n);
            if (hits) dh.alloc(n);
            if (occluded) dz.alloc(n);
            DevBuf<unsigned long long> stat;
            PathPool P; memset(&P, 0, sizeof(P));
            P.nWaves = (uint32_t) ((n + 63) / 64 + BLOCK / 64);
            stat.alloc((size_t) ST_COUNT * P.nWaves);
            DevBuf<uint32_t> spill; spill.alloc(((n + BLOCK - 1) / BLOCK * BLOCK) * (size_t) SPILL_DEPTH); P.spill = spill.p;
            HIP_TRY(hipMemset(stat.p, 0, stat.n * sizeof(unsigned long long)));
            P.stat = stat.p;
            hipEvent_t e0, e1; HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
            HIP_TRY(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_raycast, dim3((unsigned) ((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), traversalLdsBytes(scene->dev), 0, scene->dev, (const phip_ray *) dr.p, n, dh.p, dz.p, P);
            HIP_TRY(hipEventRecord(e1, 0));
            HIP_TRY(hipMemsetAsync(scene->counters.p, 0, sizeof(Counters), 0));
            hipLaunchKernelGGL(k_reduce_stats, dim3(ST_COUNT, REDUCE_SPLIT), dim3(256), 0, 0, P, scene->counters.p, 0);
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipGetLastError());
            float ms = 0; (void) hipEventElapsedTime(&ms, e0, e1); (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
            if (hits) HIP_TRY(hipMemcpy(hits, dh.p, n * sizeof(phip_hit), hipMemcpyDeviceToHost));
            if (occluded) HIP_TRY(hipMemcpy(occluded, dz.p, n, hipMemcpyDeviceToHost));
            if (out_stats) {
                Counters hc; HIP_TRY(hipMemcpy(&hc, scene->counters.p, sizeof(hc), hipMemcpyDeviceToHost));
                memset(out_stats, 0, sizeof(*out_stats));
                out_stats->closest_rays = hits ? n : 0; out_stats->shadow_rays = occluded ? n : 0;
                out_stats->closest_node_visits = hc.total[ST_NODE]; out_stats->closest_triangle_tests = hc.total[ST_TRI];
                out_stats->shadow_node_visits = hc.total[ST_SH_NODE]; out_stats->shadow_triangle_tests = hc.total[ST_SH_TRI];
                out_stats->trace_kernel_ms = ms; out_stats->iterations = 1;
                algorithmicBytes(scene, *out_stats);
            }
        }
        return PHIP_OK;
    } catch (const std::exception &e) {
        return setErr(PHIP_ERR_DEVICE, e.what());
    }
}

void phip_cancel(phip_scene *scene) { if (scene) scene->cancel.store(1); }

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
    out->max_depth = scene->traversal == 1 ? scene->bvh.maxDepth8 : scene->bvh.maxDepth; out->node_bytes = scene->traversal == 1 ? 256 : 128; out->triangle_bytes = 48;
    if (scene->traversal == 1) { out->n_nodes = scene->bvh.nNodes8; out->sah_cost = scene->bvh.sahCost8; }
    out->sah_cost = scene->bvh.sahCost; out->build_ms = scene->bvh.buildMs;
    return PHIP_OK;
}

} // extern "C"

#include "phip_debug.inl"
