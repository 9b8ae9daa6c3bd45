/*
 * k_shade.h -- k_shade: the per-vertex work of MIPathTracer::Li and same-lane path regeneration
 * Included by phip_shade.hip (k_shade) and phip_mega.hip (shadeVertex inside the fused kernel); see the header of phip.hip.
 */

__device__ __forceinline__ float miWeight(float pdfA, float pdfB) {
    pdfA *= pdfA; pdfB *= pdfB;
    return pdfA / (pdfA + pdfB);
}

#ifndef SHADE_SORT
#define SHADE_SORT 1                /* scenes with more than one BSDF model: deal the block's slots to its lanes by the model of the surface hit (k_shade) */
#endif
#ifndef SHADE_WAVES
#define SHADE_WAVES 4
#endif
#ifndef SHADE_STAGE_DUMMY
#define SHADE_STAGE_DUMMY 0         /* A/B: 1 = the FEAT-16 kernels (materials in memory) issue the two look-alike staging loads of the materials as before round 5 */
#endif
#ifndef SHADE_WAVES_PLAIN
#define SHADE_WAVES_PLAIN 5         /* scenes with more than one BSDF model but no environment emitter and no textures (the other instantiations spill
                                       100-200 B per lane at this bound): 95-102 VGPRs without the bound, 96 + 12..28 B of scratch with it.  The kernel waits
                                       on memory two thirds of its time, so the fifth wave counts: glass room k_shade -4 %, atrium -1 % (6 waves: no further gain) */
#endif
#ifndef SHADE_WAVES_LEAN
#define SHADE_WAVES_LEAN 4          /* diffuse-only instantiation */
#endif
/* The common tail of the shading kernels: the block's shadow-queue entries are compacted, slots whose sample ended start
 * the next camera sample in the same lane, blocks without work retire, per-wave statistics are recorded. */
template <bool QMC = false>
__device__ __forceinline__ void shadeEpilogue(const DevScene &S, const PathPool &P, const RenderConst &rc, uint32_t *waveCnt,
                                              const uint32_t slot, const bool inRange, uint4 info, const bool alive, bool needNew,
                                              const bool pushShadow, const float4 sh0, const float4 sh1, const float4 sh2,
                                              const unsigned long long vertices, const unsigned long long done,
                                              float4 *outRo = nullptr, float4 *outRd = nullptr, bool *outAlive = nullptr /* k_shade_trace: the camera ray a regenerated slot starts with, and whether the slot has a ray to trace */) {
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
            float4 e0 = sh0, e1 = sh1;
            if (S.preclip) preclipShadow(S, e0, e1);
            P.shadow[3 * sidx] = e0; P.shadow[3 * sidx + 1] = e1; P.shadow[3 * sidx + 2] = sh2;
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
        const V2 jit = streamJitter<QMC>(rc, pixel, k, (uint32_t) S.film.width);
        if (QMC && rc.jitter) rc.jitter[newId] = make_float2(jit.x, jit.y);      /* for the film pass */
        const float sx = (float) px + jit.x, sy = (float) py + jit.y;
        V3 o, d; float mint, maxt;
        cameraRay(S.cam, sx, sy, o, d, mint, maxt);
        float4 ro = make_float4(o.x, o.y, o.z, mint), rd = make_float4(d.x, d.y, d.z, maxt);
        if (S.preclip) preclipRay(S, ro, rd);
        P.rayO[slot] = ro;
        P.rayD[slot] = rd;
        if (outRo) { *outRo = ro; *outRd = rd; }
        P.thr[slot] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        P.mis[slot] = make_float2(0.0f, 0.0f);
        info = make_uint4((uint32_t) newId, pixel, k, 1u | F_ALIVE | F_EMITTED | F_FIRST | (dynamicId ? F_DYNAMIC : 0u));
        P.info[slot] = info;
        P.state[slot] = info.w;
        nowAlive = true;
    }
    if (outAlive) *outAlive = nowAlive;
    const uint32_t waveId = (blockIdx.x * BLOCK + threadIdx.x) >> 6;      /* the wave's position in the grid (NOT slot >> 6: k_shade may have permuted the block's slots) */
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

/* ======================================================================================
 *  One path vertex of MIPathTracer::Li (path.cpp:119-300) on the register image of a path.
 *  Shared by the wavefront kernel k_shade (slot state streamed through HBM, radiance accumulated in L[id]) and the fused
 *  kernel k_mega (k_mega.h: state and accumulator live in registers for the whole path) -- ONE statement of the
 *  integrator's control flow and arithmetic, so both produce the same bits.
 * ====================================================================================== */
struct PathVertex {
    uint32_t id, pixel, k;      /* sample id of the pass, crop pixel index, sample index */
    uint32_t state;             /* depth | F_* flags */
    float4 hit;                 /* in: (t, u, v, bits(prim)) of the ray that arrived here */
    float4 rayO, rayD;          /* in: rayD = that ray's direction; out (newRay): the next ray (o, mint | d, maxt) */
    float4 thr;                 /* throughput rgb, eta */
    float2 mis;                 /* (BSDF pdf of the sampled direction, dot(direction, refN)) of the previous vertex */
};
struct ShadowEntry { float4 e0, e1, e2; };     /* (o, maxt) (d, 0) (contribution, bits(id)); with DevScene::preclip the queue holds (o, maxt') (d, mint'): k_clip.h */

/* Radiance access policies: LGlobal = the per-sample buffer (read only when an emitter is hit: one random sector),
   LRegister = an accumulator register (k_mega).  rayO() is needed by one rare branch (environment hit by a BSDF ray). */
struct LGlobal {
    float4 *L; const PathPool &P; uint32_t slot;
    __device__ __forceinline__ float4 load(uint32_t id) const { return L[id]; }
    __device__ __forceinline__ void store(uint32_t id, const float4 &l) const { L[id] = l; }
    __device__ __forceinline__ float4 rayO(const PathVertex &) const { return P.rayO[slot]; }
    /* the sample's index in the sequence (sobol / halton / hammersley): re-derived from (sample, pixel) at every request */
    __device__ __forceinline__ uint64_t seqIdx(const RenderConst &rc, const PathVertex &v, uint32_t width) const { return seqIndex(rc, v.k, v.pixel % width, v.pixel / width); }
};
struct LRegister {
    float4 &acc;
    const uint32_t *seq;        /* QMC: the lane's two LDS words holding the sample's sequence index, computed once when the camera sample was prepared (k_mega.h) */
    __device__ __forceinline__ uint64_t seqIdx(const RenderConst &rc, const PathVertex &v, uint32_t width) const {
        return seq ? ((uint64_t) seq[0] | ((uint64_t) seq[64] << 32)) : seqIndex(rc, v.k, v.pixel % width, v.pixel / width);
    }
    __device__ __forceinline__ float4 load(uint32_t) const { return acc; }
    __device__ __forceinline__ void store(uint32_t, const float4 &l) const { acc = l; }
    __device__ __forceinline__ float4 rayO(const PathVertex &v) const { return v.rayO; }
};

/* Returns true when the path ends at this vertex (then `vertices` = its depth).  newRay: v.rayO / rayD / thr / mis hold the
   next ray; v.state is updated whenever the path goes on.  A shadow-queue entry is returned in sh when pushShadow.
   FEAT: bit 0 = the scene has an environment emitter (constant / envmap), bit 1 = it has bitmap textures, bit 2 (k_shade only) = the
   emitter table and the materials are known to fit LDS (the kernel then reads them with ds_read instead of flat loads), bit 4 (k_shade only) =
   the emitter table fits LDS, the materials stay in memory; MM: leaf BSDF models
   present in the scene; STRICT: strictNormals (a compile-time switch: without it the geometric normal is dead after
   fillIntersection and the diffuse-only instantiation fits 80 VGPRs = 6 waves per SIMD) */
template <int MM, bool STRICT, int FEAT, typename LAcc>
__device__ __forceinline__ bool shadeVertex(const DevScene &S, const EmitterTab &T, const DevMaterial *materials, const RenderConst &rc,
                                            PathVertex &v, const LAcc &acc, bool &newRay, bool &pushShadow, ShadowEntry &sh, uint32_t &vertices) {
    constexpr bool ENV = (FEAT & 1) != 0, TEX = (FEAT & 2) != 0, QMC = (FEAT & 8) != 0;
    const uint32_t prim = pm_to_bits(v.hit.w);
    const V3 rayD(v.rayD.x, v.rayD.y, v.rayD.z);
    V3 thr(v.thr.x, v.thr.y, v.thr.z);
    float eta = v.thr.w;
    uint32_t depth = v.state & DEPTH_MASK;
    uint32_t flags = v.state & ~DEPTH_MASK;
    const uint32_t id = v.id;
    bool terminate = false;
    bool haveAdd = false;                  /* radiance to add to the sample's accumulator (in reference order) */
    float4 l = make_float4(0, 0, 0, 0);
    newRay = false; pushShadow = false;
    /* VP: the sibling integrator `volpath_simple` on a scene without media (src/integrators/path/volpath_simple.cpp:88-318; round 5): the same
       loop without multiple importance sampling -- emitted radiance counts at the first vertex and behind a delta bounce only (:186-188, 246-255), the emitter
       sample is weighted by the BSDF value alone (:207-227), Russian roulette is evaluated before the loop looks at the next hit (:282-292).  A uniform branch. */
    const bool VP = rc.volpath != 0;
    /* Russian roulette: the end of a loop iteration (path.cpp:278-286 = volpath_simple.cpp:282-292), evaluated by the vertex the iteration's ray arrived at */
    auto russianRoulette = [&]() {
        if (depth++ >= (uint32_t) rc.rrDepth) {
            float q = smin(thr.maxc() * eta * eta, 0.95f);
            /* the (depth - 1 - rrDepth)-th 1D request of the sample (path.cpp:283) */
            float rr;
            if (rc.sampler == PHIP_SAMPLER_LD && depth - 1u - (uint32_t) rc.rrDepth < LD_DIMENSIONS) {
                float unused; ldPoint(v.pixel, v.k, 2u * (depth - 1u - (uint32_t) rc.rrDepth) + 1u, rc.seed, rc.ldMask, rr, unused);
            } else
                rr = u32ToFloat(pcg4d(v.pixel, v.k, 2 + 2 * (depth - 2), rc.seed).x);
            if (QMC) {
                /* the (depth - 1 - rrDepth)-th 1D request of the sample.  sobol: its dimension is two per 2D request made so far (the camera
                   sample and the kq requests of the vertices behind) plus one per earlier 1D request (SobolSampler::next1D, sobol.cpp:226-236) */
                const uint32_t j = depth - 1u - (uint32_t) rc.rrDepth, kq = 2u * (depth - 1u) - (flags >> NS_SHIFT);
                if (isSequenceSampler(rc.sampler)) {
                    const uint32_t dim = 2u * (1u + kq) + j + 1u;      /* (+ 1: SobolSampler::next2D skips dimension 4, see the vertex's requests below) */
                    if (dim < seqDims(rc)) rr = seqSample(rc, acc.seqIdx(rc, v, (uint32_t) S.film.width), dim);
                } else if (rc.sampler == PHIP_SAMPLER_STRATIFIED && j < ST_DIMENSIONS)
                    rr = stPoint1D(v.pixel, v.k, j, rc.seed, rc.stRes, rr);
            }
            if (rr >= q)
                terminate = true;
            else
                thr = thr / q;
        }
    };

    if (prim == PHIP_NO_HIT) {
        bool live = true;
        if (VP && !(flags & F_FIRST)) { russianRoulette(); live = !terminate; }     /* volpath_simple.cpp:282-292 runs before :172-183 sees the miss (path.cpp:233-248 breaks first) */
        terminate = true;
        if (flags & F_FIRST) haveAdd = true;   /* a camera ray that leaves the scene: the sample is (0,0,0, alpha 0) -- written, so the buffer needs no clear */
        if (ENV && S.envEmitter >= 0 && live) {     /* environment emitter: path.cpp:136-143 (camera ray) / 233-265 (BSDF-sampled ray) */
            const float *em = emitterRecord(T, (uint32_t) S.envEmitter);
            const V3 value = (pm_to_bits(em[EM_TYPE]) == PHIP_EMITTER_ENVMAP) ? envmapEval(S.env, rayD) : rgb(em + EM_RADIANCE);
            if (flags & F_FIRST) {
                if (!rc.hideEmitters) {                                                    /* throughput is 1; alpha stays 0 */
                    V3 bg = value;
                    if (rc.envFiltered) {
                        /* the camera ray is the one ray with differentials: filtered lookup, envmap.cpp:395-407.
                           Its sample position is recomputed from the counter stream (a rare branch) */
                        const uint32_t px = v.pixel % (uint32_t) S.film.width, py = v.pixel / (uint32_t) S.film.width;
                        const V2 hc = filmJitter<QMC>(rc, v.id, v.pixel, v.k, (uint32_t) S.film.width);
                        V3 rx, ry;
                        cameraRayDifferentials(S.cam, (float) px + hc.x, (float) py + hc.y, rx, ry);
                        rx = rayD + (rx - rayD) * rc.diffScaleFactor;
                        ry = rayD + (ry - rayD) * rc.diffScaleFactor;
                        bg = envmapEvalDiff(S.env, rayD, rx, ry);
                    }
                    l.x += bg.x; l.y += bg.y; l.z += bg.z;
                }
            } else if (VP) {
                /* volpath_simple.cpp:172-183: throughput * evalEnvironment(ray), when this ray may see emitters at all */
                if ((flags & F_EMITTED) && (!rc.hideEmitters || (flags & F_SCATTERED))) {
                    const V3 c = thr * value;
                    l = acc.load(id);
                    l.x += c.x; l.y += c.y; l.z += c.z;
                    haveAdd = true;
                }
            } else {
                const float4 ro = acc.rayO(v);
                if (envFillDirectRecord(S, V3(ro.x, ro.y, ro.z), rayD)) {
                    const float lumPdf = (!(flags & F_PREV_DELTA))
                        ? pdfEmitterDirectDot<ENV>(S, T, (uint32_t) S.envEmitter, rayD, v.mis.y, (flags & F_REFN_ZERO) != 0, 0.0f, 0.0f) : 0;
                    const V3 c = thr * value * miWeight(v.mis.x, lumPdf);
                    l = acc.load(id);
                    l.x += c.x; l.y += c.y; l.z += c.z;
                    haveAdd = true;
                }
            }
        }
        if (haveAdd) acc.store(id, l);
    } else {
        Isect its;
        fillIntersection(S, rayD, prim, v.hit.y, v.hit.z, v.hit.x, its);
        /* the accumulator is zero until the sample's first vertex writes it, and later vertices only touch it when they
           hit an emitter: no unconditional 64-byte-sector read per vertex (LGlobal) */
        if (flags & F_FIRST) {
            l.w = 1.0f;                     /* alpha, records.inl:117-144 */
            haveAdd = true;
        } else {
            /* ---- tail of the previous loop iteration, path.cpp:257-286 ---- */
            if (VP) {
                /* (the head below adds the emitter's radiance with weight one when F_EMITTED says so; the accumulator is read there) */
                if (its.emitter >= 0 && (flags & F_EMITTED)) l = acc.load(id);
            } else if (its.emitter >= 0) {
                l = acc.load(id);
                const float *em = emitterRecord(T, (uint32_t) its.emitter);
                V3 value = (dot(its.sh.n, -rayD) <= 0) ? V3(0.0f) : rgb(em + EM_RADIANCE);
                /* DirectSamplingRecord::setQuery (records.inl:170-178): n = shading normal, d = ray direction, dist = t */
                const float lumPdf = (!(flags & F_PREV_DELTA))
                    ? pdfEmitterDirectDot<ENV>(S, T, (uint32_t) its.emitter, rayD, v.mis.y, (flags & F_REFN_ZERO) != 0, dot(rayD, its.sh.n), its.t) : 0;
                const V3 c = thr * value * miWeight(v.mis.x, lumPdf);
                l.x += c.x; l.y += c.y; l.z += c.z;
                haveAdd = true;
            }
            if (!VP) flags &= ~F_EMITTED;           /* rRec.type = ERadianceNoEmission (volpath_simple: the previous vertex decided, see below) */
            russianRoulette();
        }
        const bool firstVertex = (flags & F_FIRST) != 0;
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
            if (VP) {
                if (STRICT && dot(rayD, its.geoN) * cosTheta(its.wi) > 0)      /* volpath_simple.cpp:194-198: wiDotGeoN * wiDotShN < 0 with wiDotGeoN = -dot(n, d) */
                    terminate = true;
            } else if (((int) depth >= rc.maxDepth && rc.maxDepth > 0)
                || (STRICT && dot(rayD, its.geoN) * cosTheta(its.wi) >= 0))
                terminate = true;
        }
        V3 shD(0.0f), shC(0.0f); float shMaxt = 0;
        if (!terminate) {
            /* the parity stream is consumed in CALL ORDER, like a Sampler (HISTORY.md 3.5): the k-th 2D request after the pixel jitter is
               pair k & 1 (.xy / .zw) of block 1 + 2 (k >> 1).  A vertex with a smooth BSDF makes two requests (emitter sample, BSDF
               sample), a vertex without one; k0 = 2 (depth - 1) - ns is this vertex's first, ns = the non-smooth vertices so far
               (bits 26..31 of the state word, modulo 64).  All-smooth paths: k0 is even and both pairs come from one block. */
            /* volpath_simple: at depth == maxDepth the query is EEmittedRadiance alone (:239-261 on the vertex before, :103-104 for maxDepth 1): no emitter sample is
               drawn there, the vertex makes the BSDF request only */
            const bool directHere = !(VP && rc.maxDepth > 0 && (int) depth >= rc.maxDepth);
            const bool smoothVertex = (its.flags & TS_MF_SMOOTH) != 0 && directHere;
            const uint32_t k0 = 2u * (depth - 1u) - (flags >> NS_SHIFT);
            U4 h = pcg4d(v.pixel, v.k, 1 + 2 * (k0 >> 1), rc.seed);
            if (k0 & 1u) {
                if (smoothVertex) { const U4 h2 = pcg4d(v.pixel, v.k, 3 + 2 * (k0 >> 1), rc.seed); h = U4{ h.z, h.w, h2.x, h2.y }; }
                else h = U4{ h.z, h.w, h.z, h.w };
            }
            if (!smoothVertex) { h.z = h.x; h.w = h.y; flags += 1u << NS_SHIFT; }      /* its one request is the BSDF sample */
            V2 smpEmitter(u32ToFloat(h.x), u32ToFloat(h.y)), smpBSDF(u32ToFloat(h.z), u32ToFloat(h.w));
            if (rc.sampler == PHIP_SAMPLER_LD) {
                /* 2D requests k0 + 1 (and k0 + 2 at a smooth vertex) of the sample, the pixel jitter being request 0: the first
                   LD_DIMENSIONS come from the scrambled (0,2)-sequences (ldsampler.cpp:218-224) */
                const uint32_t q = k0 + 1u;
                if (smoothVertex) {
                    if (q < LD_DIMENSIONS) ldPoint(v.pixel, v.k, 2u * q, rc.seed, rc.ldMask, smpEmitter.x, smpEmitter.y);
                    if (q + 1u < LD_DIMENSIONS) ldPoint(v.pixel, v.k, 2u * (q + 1u), rc.seed, rc.ldMask, smpBSDF.x, smpBSDF.y);
                } else if (q < LD_DIMENSIONS)
                    ldPoint(v.pixel, v.k, 2u * q, rc.seed, rc.ldMask, smpBSDF.x, smpBSDF.y);
            }
            if (QMC && isSequenceSampler(rc.sampler)) {
                /* SobolSampler::next2D (sobol.cpp:238-257): the requests of this vertex start at dimension 2 (1 + k0) + the 1D requests made so far
                   (one Russian-roulette request behind every vertex from rrDepth on: max(0, depth - rrDepth)) */
                const uint64_t idx = acc.seqIdx(rc, v, (uint32_t) S.film.width);
                const uint32_t nDims = seqDims(rc);
                /* ... and the sampler never hands out dimension 4 to a 2D request (sobol.cpp:241-242: the test for the dimensions reserved to sample
                   arrays, [5, 5) when none is requested, fires for m_dimension == 4): the sample's third 2D request starts there -- no 1D request
                   can come earlier with rrDepth >= 2 -- so it and every later request is shifted by one */
                uint32_t dim = 2u * (1u + k0) + (depth > (uint32_t) rc.rrDepth ? depth - (uint32_t) rc.rrDepth : 0u) + (k0 >= 1u ? 1u : 0u);
                if (rc.sampler == PHIP_SAMPLER_SOBOL) {
                    /* both requests of the vertex in one pass over the index bits (dv_math.h: sobolSample2x2); a request beyond the table keeps the counter stream's numbers */
                    const uint32_t dimE = dim, dimB = smoothVertex ? dim + 2u + (k0 == 0u ? 1u : 0u) : dim;
                    const bool okE = smoothVertex && dimE + 1u < nDims, okB = dimB + 1u < nDims;
                    float q[4];
                    sobolSample2x2(rc.sobol, idx, okE ? dimE : 0u, okB ? dimB : 0u, q);
                    if (okE) smpEmitter = V2(q[0], q[1]);
                    if (okB) smpBSDF = V2(q[2], q[3]);
                } else {
                    if (smoothVertex) {
                        if (dim + 1u < nDims) smpEmitter = seqSample2(rc, idx, dim);
                        dim += 2u + (k0 == 0u ? 1u : 0u);
                    }
                    if (dim + 1u < nDims) smpBSDF = seqSample2(rc, idx, dim);
                }
            } else if (QMC && rc.sampler == PHIP_SAMPLER_STRATIFIED) {
                /* 2D requests k0 + 1 (and k0 + 2 at a smooth vertex) of the sample: the first ST_DIMENSIONS are stratified, jittered by the counter stream's own numbers */
                const uint32_t q = k0 + 1u;
                if (smoothVertex) {
                    if (q < ST_DIMENSIONS) stPoint2D(v.pixel, v.k, q, rc.seed, rc.stRes, smpEmitter.x, smpEmitter.y, smpEmitter.x, smpEmitter.y);
                    if (q + 1u < ST_DIMENSIONS) stPoint2D(v.pixel, v.k, q + 1u, rc.seed, rc.stRes, smpBSDF.x, smpBSDF.y, smpBSDF.x, smpBSDF.y);
                } else if (q < ST_DIMENSIONS)
                    stPoint2D(v.pixel, v.k, q, rc.seed, rc.stRes, smpBSDF.x, smpBSDF.y, smpBSDF.x, smpBSDF.y);
            }
            /* ---- direct illumination sampling, path.cpp:172-200 ---- */
            DirectRec dRec;
            dRec.ref = its.p;
            dRec.refN = (its.flags & TS_TRANS_OR_BACK) ? V3(0.0f) : its.sh.n;
            dRec.pdf = 0; dRec.emitter = -1;
            BsdfCtx bctx = bsdfResolve(materials, its);
            if (TEX && bctx.textured) {
                /* texture->eval(its) of the BSDF's bitmap children: unfiltered level-0 lookup, except at the first vertex, whose UV partials come
                   from the camera-ray differentials (Intersection::getBSDF(ray) -> computePartials, records.inl:69-75) */
                float dudx = 0, dudy = 0, dvdx = 0, dvdy = 0;
                if (firstVertex) {
                    const uint32_t px = v.pixel % (uint32_t) S.film.width, py = v.pixel / (uint32_t) S.film.width;
                    const V2 hc = filmJitter<QMC>(rc, v.id, v.pixel, v.k, (uint32_t) S.film.width);
                    V3 rx, ry;
                    cameraRayDifferentials(S.cam, (float) px + hc.x, (float) py + hc.y, rx, ry);
                    rx = rayD + (rx - rayD) * rc.diffScaleFactor;
                    ry = rayD + (ry - rayD) * rc.diffScaleFactor;
                    const float *cw = S.cam.c2w;
                    computePartials(its, V3(cw[3], cw[7], cw[11]), rx, ry, dudx, dudy, dvdx, dvdy);
                }
                bsdfTextures(S, bctx, its.uv, firstVertex, dudx, dudy, dvdx, dvdy);
            }
            if (smoothVertex) {
                V3 value = sampleEmitterDirect<ENV>(S, T, dRec, smpEmitter);
                if (dRec.pdf != 0 && !value.isZero()) {
                    const V3 wo = its.sh.toLocal(dRec.d);
                    float bPdf;
                    const V3 bsdfVal = bsdfEvalPdf<MM>(bctx, wo, bPdf);
                    if (!bsdfVal.isZero() && (!STRICT || dot(its.geoN, dRec.d) * cosTheta(wo) > 0)) {
                        if (VP) shC = thr * value * bsdfVal;        /* volpath_simple.cpp:223-225: no MIS */
                        else { const float weight = miWeight(dRec.pdf, bPdf); shC = thr * value * bsdfVal * weight; }
                        shD = dRec.d; shMaxt = dRec.dist * (1 - PT_SHADOW_EPSILON);
                        pushShadow = true;
                    }
                }
            }
            /* ---- BSDF sampling, path.cpp:207-226 ---- */
            BSDFSample bs;
            const V3 bsdfWeight = bsdfSample<MM>(bctx, smpBSDF, bs);
            if (bsdfWeight.isZero()) {
                terminate = true;
            } else {
                bool goOn = true;
                if (VP) {
                    /* volpath_simple.cpp:239-261: indirect illumination while depth + 1 < maxDepth; emitted radiance behind a delta bounce while depth < maxDepth
                       and this vertex was asked for direct illumination; nothing of either: the path ends before the ray is traced */
                    const bool indirect = (int) depth + 1 < rc.maxDepth || rc.maxDepth < 0;
                    const bool emittedNext = ((int) depth < rc.maxDepth || rc.maxDepth < 0) && directHere && bs.delta;
                    goOn = indirect || emittedNext;
                    flags = emittedNext ? (flags | F_EMITTED) : (flags & ~F_EMITTED);
                }
                flags |= F_SCATTERED;
                const V3 wo = its.sh.toWorld(bs.wo);
                const float woDotGeoN = dot(its.geoN, wo);
                if (!goOn || (STRICT && woDotGeoN * cosTheta(bs.wo) <= 0)) {
                    terminate = true;
                } else {
                    v.rayO = make_float4(its.p.x, its.p.y, its.p.z, PT_EPSILON);
                    v.rayD = make_float4(wo.x, wo.y, wo.z, INFINITY);
                    thr = thr * bsdfWeight;
                    eta *= bs.eta;
                    v.thr = make_float4(thr.x, thr.y, thr.z, eta);
                    v.mis = make_float2(bs.pdf, dot(wo, dRec.refN));
                    newRay = true;
                    flags = bs.delta ? (flags | F_PREV_DELTA) : (flags & ~F_PREV_DELTA);
                    flags = dRec.refN.isZero() ? (flags | F_REFN_ZERO) : (flags & ~F_REFN_ZERO);
                }
            }
        }
        if (haveAdd) acc.store(id, l);
        if (pushShadow) {   /* self-contained shadow-queue entry: survives the slot being recycled */
            sh.e0 = make_float4(its.p.x, its.p.y, its.p.z, shMaxt);
            sh.e1 = make_float4(shD.x, shD.y, shD.z, 0.0f);
            sh.e2 = make_float4(shC.x, shC.y, shC.z, pm_from_bits(id));
        }
    }
    if (terminate) vertices = depth;
    else v.state = flags | depth;
    return terminate;
}

/* small scene tables are staged in LDS: the emitter table (selection CDF -> emitter -> area CDF is a chain of
   dependent lookups per NEE sample) and the materials (all threads of the block must call; no barrier inside) */
struct ShadeTables { EmitterTab T; const DevMaterial *materials; };
/* MATS = false: the caller knows that the materials stay in memory (k_shade with FEAT bit 4: scenes of many materials, the atrium) -- the two look-alike loads
   that keep the staging branch-free are then two vector-memory instructions per lane for nothing, in a kernel that is bound by their number */
template <bool MATS = true>
__device__ __forceinline__ ShadeTables stageShadeTables(const DevScene &S, float *ldsEm, DevMaterial *ldsMat) {
    const bool emInLds = S.emitterTabSize <= EMITTER_LDS_FLOATS, matInLds = MATS && S.nMaterials <= MATERIAL_LDS_MAX;
    /* Both tables are requested before either is stored: ONE memory round trip at the head of a block (the kernels that call this are
       latency bound), not one per table and per loop iteration.  A thread covers the whole staged range with one float4 of the emitter
       table (the host pads it to whole float4s; EMITTER_LDS_FLOATS / 4 = BLOCK) and MAT_F4 of the materials. */
    constexpr uint32_t MAT_F4 = (MATERIAL_LDS_MAX * (uint32_t) (sizeof(DevMaterial) / 16) + BLOCK - 1) / BLOCK;
    static_assert(EMITTER_LDS_FLOATS / 4 <= BLOCK, "one float4 of the emitter table per thread");
    const uint32_t nE4 = emInLds ? S.emitterTabSize / 4u : 0u, nM4 = matInLds ? S.nMaterials * (uint32_t) (sizeof(DevMaterial) / 16) : 0u;
    /* branch-free: every thread loads (a clamped index, a valid address even for an empty table) and only the stores are predicated --
       predicated loads compile to one skipped block and one wait each */
    const float4 *srcE = nE4 ? (const float4 *) S.emitterTab : (const float4 *) S.triShade;
    const float4 *srcM = nM4 ? (const float4 *) S.materials : (const float4 *) S.triShade;
    const uint32_t lastE = nE4 ? nE4 - 1u : 0u, lastM = nM4 ? nM4 - 1u : 0u;
    float4 e = srcE[threadIdx.x < lastE ? threadIdx.x : lastE];
    static_assert(MAT_F4 == 2, "two float4s of the materials per thread (m0, m1)");
    float4 m0 = make_float4(0, 0, 0, 0), m1 = m0;
    if (MATS) { m0 = srcM[threadIdx.x < lastM ? threadIdx.x : lastM]; m1 = srcM[threadIdx.x + BLOCK < lastM ? threadIdx.x + BLOCK : lastM]; }
    /* the values are "used" HERE, all at once: without this the compiler sinks every load into the predicated block of its store (load,
       wait, store -- one round trip per table) */
    asm volatile("" : "+v"(e.x), "+v"(e.y), "+v"(e.z), "+v"(e.w), "+v"(m0.x), "+v"(m0.y), "+v"(m0.z), "+v"(m0.w), "+v"(m1.x), "+v"(m1.y), "+v"(m1.z), "+v"(m1.w));
    if (threadIdx.x < nE4) ((float4 *) ldsEm)[threadIdx.x] = e;
    if (threadIdx.x < nM4) ((float4 *) ldsMat)[threadIdx.x] = m0;
    if (threadIdx.x + BLOCK < nM4) ((float4 *) ldsMat)[threadIdx.x + BLOCK] = m1;
    ShadeTables t;
    t.T.t = emInLds ? ldsEm : S.emitterTab; t.T.n = S.nEmitters; t.T.normalization = S.emitterNormalization;
    t.materials = matInLds ? ldsMat : S.materials;
    return t;
}

/* Scenes with more than one BSDF model: on the atrium 8 % of the vertices lie on copper, so nearly every wave ran the
   microfacet code -- the longest branch of the vertex by far -- for its two or three conductor lanes (round 2: lane
   utilisation 0.27).  Which LANE shades which of the block's 256 slots is free (all state is addressed by slot), so the
   slots are dealt to the lanes by the class the ray kernel left in the hit record (k_pool.h): diffuse, rough conductor,
   dielectric, then the slots without a live path (they regenerate).  A wave runs the microfacet code only if it got
   conductor vertices (VALU instructions per launch -40 %, lane utilisation 0.29 -> 0.53).  Every lane fetches the state of
   ITS slot (coalesced, one round trip, as without the deal) and hands it to the lane that shades it through LDS.  Results
   cannot change: every slot is shaded by exactly one lane with the same code.  (All threads of the block call; two barriers.) */
#define SHADE_DEAL_BYTES (BLOCK * (16 + 16 + 16 + 16 + 8 + 4))
__device__ __forceinline__ void dealSlotsByClass(unsigned char *xbuf /* SHADE_DEAL_BYTES of LDS, 16-byte aligned */, uint32_t (*clsCnt)[BLOCK / 64] /* LDS: [4][BLOCK / 64] */,
                                                 const PathPool &P, uint4 &info, PathVertex &v, uint32_t &slot, bool &inRange) {
    uint4 *xInfo = (uint4 *) xbuf;
    float4 *xHit = (float4 *) (xInfo + BLOCK), *xRayD = xHit + BLOCK, *xThr = xRayD + BLOCK;
    float2 *xMis = (float2 *) (xThr + BLOCK);
    uint32_t *xSlot = (uint32_t *) (xMis + BLOCK);
    uint32_t cls = 3u;
    if (inRange && (info.w & F_ALIVE)) {
        const uint32_t w = pm_to_bits(v.hit.w);
        if (w != PHIP_NO_HIT) cls = w >> HIT_CLASS_SHIFT;       /* (a path that left the scene ends here: with the idle slots) */
    }
    const uint32_t wave = threadIdx.x >> 6, lane = __lane_id();
    uint32_t rank = 0;
#pragma unroll
    for (uint32_t c = 0; c < 4; ++c) {
        const unsigned long long m = __ballot(cls == c);
        if (cls == c) rank = (uint32_t) __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) clsCnt[c][wave] = (uint32_t) __popcll(m);
    }
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (uint32_t c = 0; c < 4; ++c)
#pragma unroll
        for (uint32_t w = 0; w < BLOCK / 64; ++w) {
            const uint32_t n = clsCnt[c][w];
            if (c < cls || (c == cls && w < wave)) base += n;
        }
    const uint32_t dst = base + rank;
    xInfo[dst] = info; xHit[dst] = v.hit; xRayD[dst] = v.rayD; xThr[dst] = v.thr; xMis[dst] = v.mis; xSlot[dst] = slot;
    __syncthreads();
    info = xInfo[threadIdx.x]; v.hit = xHit[threadIdx.x]; v.rayD = xRayD[threadIdx.x]; v.thr = xThr[threadIdx.x]; v.mis = xMis[threadIdx.x];
    slot = xSlot[threadIdx.x];
    inRange = slot < P.capacity;
}

template <int MM, bool STRICT, int FEAT> __global__ __launch_bounds__(BLOCK, MM == 0 ? SHADE_WAVES_LEAN : ((FEAT & 3) == 0 ? SHADE_WAVES_PLAIN : SHADE_WAVES)) void k_shade(DevScene S, PathPool P, RenderConst rc, float4 *L) {
    __shared__ uint32_t waveCnt[BLOCK / 64];
    __shared__ __align__(16) float ldsEm[EMITTER_LDS_FLOATS];
    __shared__ DevMaterial ldsMat[MATERIAL_LDS_MAX];
    /* The kernel is latency bound (two thirds of its wave cycles are s_waitcnt), so the head of a block is ONE round trip: the slot
       state (five 16-byte loads and a word) and the tables staged in LDS are all requested before anything waits; the block's retired
       flag is looked at only while the pass drains (before that no block can have retired).  (Round 3 found the order flag -> branch -> table loop -> wait -> remainder loop -> wait -> materials -> wait -> state: five
       dependent round trips before the first useful instruction.) */
    if (rc.draining && P.blockDead[blockIdx.x]) return;         /* (block-uniform; the flag costs a scalar round trip, paid only while the pass drains: rc.draining is a kernel argument) */
    uint32_t slot = blockIdx.x * BLOCK + threadIdx.x;
    bool inRange = slot < P.capacity;
    uint32_t lslot = inRange ? slot : 0u;
    uint4 info = P.info[lslot];
    info.w = P.state[lslot];
    PathVertex v;
    v.hit = P.hit[lslot];
    v.rayD = P.rayD[lslot];
    v.thr = P.thr[lslot];
    v.mis = P.mis[lslot];
    ShadeTables tab = stageShadeTables<(FEAT & 16) == 0 || SHADE_STAGE_DUMMY>(S, ldsEm, ldsMat);
    if (FEAT & 4) { tab.T.t = ldsEm; tab.materials = ldsMat; }     /* the host checked that both tables fit: LDS addressing (ds_read), no flat loads */
    else if (FEAT & 16) {
        /* the emitter table fits, the materials do not (the atrium: 252 materials = 24 KB): the emitter look-ups -- two dozen per NEE sample --
           are addressed as LDS, the one material record of the vertex comes from memory.  Without this the whole vertex went through
           generic pointers: k_shade of the atrium issued 52 vector-memory instructions per wave at 53 clk each on a texture-data path
           that was 87 % busy (profiles/r03b_tcp_atrium_*.json), half of them flat loads of LDS addresses */
        tab.T.t = ldsEm; tab.materials = S.materials;
    }
    if (MM != 0 && SHADE_SORT && S.shadeSort) {                 /* (block-uniform) */
        __shared__ __align__(16) unsigned char xbuf[SHADE_DEAL_BYTES];
        __shared__ uint32_t clsCnt[4][BLOCK / 64];
        dealSlotsByClass(xbuf, clsCnt, P, info, v, slot, inRange);
    }
    v.hit.w = pm_from_bits(hitPrim(pm_to_bits(v.hit.w)));       /* (the class bits have served: k_pool.h) */
    if (!inRange) info = make_uint4(0, 0, 0, 0);
    __syncthreads();                                            /* LDS tables are complete */
    bool alive = inRange && (info.w & F_ALIVE);
    bool needNew = inRange && !alive && !(info.w & F_DEAD);
    unsigned long long vertices = 0, done = 0;
    bool pushShadow = false;
    ShadowEntry sh; sh.e0 = make_float4(0, 0, 0, 0); sh.e1 = sh.e0; sh.e2 = sh.e0;

    if (alive) {
        v.id = info.x; v.pixel = info.y; v.k = info.z; v.state = info.w;
        bool newRay; uint32_t nv = 0;
        const LGlobal acc{ L, P, slot };
        if (shadeVertex<MM, STRICT, FEAT>(S, tab.T, tab.materials, rc, v, acc, newRay, pushShadow, sh, nv)) {
            vertices = nv; done = 1;
            needNew = true;
        } else {
            info.w = v.state;
            P.state[slot] = info.w;
        }
        if (newRay) {
            if (S.preclip) preclipRay(S, v.rayO, v.rayD);
            P.rayO[slot] = v.rayO; P.rayD[slot] = v.rayD; P.thr[slot] = v.thr; P.mis[slot] = v.mis;
        }
    }

    shadeEpilogue<(FEAT & 8) != 0>(S, P, rc, waveCnt, slot, inRange, info, alive, needNew, pushShadow, sh.e0, sh.e1, sh.e2, vertices, done);
}
