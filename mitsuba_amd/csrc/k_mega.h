/*
 * k_mega.h -- k_mega: the whole path of MIPathTracer::Li in ONE persistent kernel, for scenes whose acceleration structure,
 * Wald records, shading records, emitter table and materials all fit in LDS (the Cornell box of BASELINE.json configs[1]; since round 5 also with glass and
 * copper blocks: every leaf BSDF model, on the packed leaf table of at most 64 Wald records -- phip_mega.hip).
 *
 * The wavefront design (k_shade -> k_shadow_p -> k_trace, state streamed through HBM between three kernels per iteration)
 * exists to keep traversal kernels small when every node fetch is an HBM/L2 round trip.  When the geometry is a few KB in
 * LDS there is no latency to hide and the round trips ARE the cost: on the Cornell box k_shade moved 0.8 GB of pool state per
 * launch (2.5 TB/s) and the shadow kernel paid a random read-modify-write of L[id] per unoccluded ray, for a scene of 2.4 KB.
 * Here a lane owns one path from the camera sample to its end:
 *
 *   loop:  lanes without a path draw the next sample id of the wave's chunk (same-lane regeneration, integrator.cpp:157-183)
 *          closest hit   (k_traverse.h: trees of <= 32 records -- the Cornell box -- as a flat table of leaf boxes tested in one uniform pass, the Wald
 *                         tests of the wave's rays dealt over its lanes (traverseFlat2W); larger trees: the per-lane BVH4 state machine, all from LDS)
 *          shadeVertex   (k_shade.h: the same statement of path.cpp:119-300 the wavefront kernel runs)
 *          shadow ray    (traverse<true>); an unoccluded entry adds its contribution to the lane's accumulator REGISTER
 *          a finished path stores its (R,G,B,alpha) once: L[id] = acc
 *
 * Ray, hit, throughput, MIS record and radiance never leave registers; HBM sees 16 B per sample (the L store the film
 * kernel reads) instead of ~600 B.  Radiance is added in the reference's order (emitter hit of vertex n, NEE of vertex n,
 * emitter hit of vertex n + 1, ...), exactly as the wavefront path does through L[id], so both are bit-identical.
 *
 * Sample ids are handed out in chunks from one global counter: a wave takes `chunk` consecutive ids with one atomicAdd by
 * lane 0 (guided: the chunk shrinks towards the end of the pass so that the waves finish together) and deals them to its
 * lanes in order -- 64 consecutive ids are one 8x8 pixel patch of one sample index (ids are block-major, then sample, then
 * Morton index), so the camera rays of a wave are coherent.
 */

#ifndef MEGA_PROFILE
#define MEGA_PROFILE 0               /* measurement build: wave clock and active lanes per phase of the loop (reported in the work-counter rows, which it falsifies) */
#endif
#ifndef MEGA_REGEN_QUEUE
#define MEGA_REGEN_QUEUE 1           /* camera samples are prepared 64 at a time by the whole wave into an LDS queue (0: by the lanes whose paths ended, in every pass) */
#endif
#ifndef MEGA_CLIP_SEL
#define MEGA_CLIP_SEL 1              /* the scene-box clip without control flow (k_clip.h: clipToSceneSel) */
#endif
#ifndef MEGA_BALANCE
#define MEGA_BALANCE 1               /* FLAT == 2: the Wald tests of a traversal are dealt over the lanes of the wave (k_traverse.h: traverseFlat2W) instead of looping per lane */
#endif
#ifndef MEGA_CLASS_DEAL
#define MEGA_CLASS_DEAL 1            /* MM != 0 (scenes with glass / copper: k_mega<MM_ALL>): before the vertex phase the paths of the BLOCK are dealt to its lanes by the BSDF
                                        model of the surface they hit (an exchange of the path state through LDS), so that a wave runs the microfacet code only if it got
                                        copper vertices.  Without it nearly every wave of the mixed Cornell box ran all three models for its few copper and glass lanes:
                                        lane utilisation 0.41 against 0.77 on the all-diffuse box, 2.25 x the VALU instructions for 1.17 x the vertices
                                        (profiles/r05_valu_cornell_mixed_*) */
#endif
#define MEGA_CHUNK_MAX 4096u
#define MEGA_CHUNK_MIN 64u

enum { MC_SAMPLES = 0, MC_VERTICES, MC_RAYS, MC_NODE, MC_TRI, MC_SH_RAYS, MC_SH_NODE, MC_SH_TRI, MC_COUNT };

template <int MM, bool STRICT, int FLAT /* 0: BVH4 walk, 1: flat leaf table (traverseFlat), 2: packed table + record masks (traverseFlat2), 3: the same with 33..64 records (two-word masks; MEGA_BALANCE only) */,
          bool QMC /* the reference's sobol / halton / hammersley / stratified streams (FEAT bit 3 of shadeVertex) */> __global__ __launch_bounds__(BLOCK, MEGA_WAVES) void k_mega(DevScene S, MegaParams M, RenderConst rc, float4 *L) {
    __shared__ uint32_t ldsClsCnt[4][BLOCK / 64];                 /* MEGA_CLASS_DEAL: lanes per BSDF model and wave */
    __shared__ uint32_t ldsCount[MC_COUNT][BLOCK];              /* work counters: one LDS word per lane and counter instead of eight VGPRs.  (As ds_add_u32 -- no read,
                                                                   no wait -- and the three that count lanes as ballots in SGPRs: 66.9 vs 66.5 ms per C2 frame and
                                                                   125 instead of 116 VGPRs; the seven read-modify-writes per pass overlap with the rest as they are) */
#define MEGA_COUNT(row, amount) ldsCount[row][threadIdx.x] += (uint32_t) (amount)
#if MEGA_REGEN_QUEUE
    constexpr int RQ_ROWS = QMC ? 10 : 8;
    __shared__ uint32_t ldsRegen[BLOCK / 64][RQ_ROWS][64];      /* per wave: 64 prepared camera samples (mint | d, maxt | id, pixel, k [| the sample's sequence index: QMC]), one word per entry and row;
                                                                   the origin of a pinhole camera's rays is the same for every sample (camO below) */
    __shared__ uint32_t ldsSeq[QMC ? BLOCK / 64 : 1][2][64];    /* QMC: per lane, the sequence index of the path's sample -- shadeVertex draws every number of the path from it; deriving it anew at every
                                                                   request (sobol::look_up: ~30 table rows) was a third of the QMC kernel's sampling cost */
#endif
    /* dynamic LDS: [traversal stack | all nodes | all Wald records] (setupTraversal) [shading records | emitter table | materials],
       sized for THIS scene (megaLdsBytes) so that as many blocks as the registers allow fit a CU */
    float4 *ldsTriShade = (float4 *) (g_smem + traversalLdsBytesOf(S));
    float *ldsEm = (float *) (ldsTriShade + (size_t) S.nTriangles * TRISHADE_FLOAT4S);
    DevMaterial *ldsMat = (DevMaterial *) (ldsEm + ((S.emitterTabSize + 3u) & ~3u));
    ShadeTables tab = stageShadeTables(S, ldsEm, ldsMat);
    /* the host chose this kernel because every table fits (phip.hip: fitsLds): no run-time choice between the LDS copy and HBM, so that
       the compiler can address the tables as LDS (ds_read) instead of through flat loads, which occupy the texture addresser */
    tab.T.t = ldsEm; tab.materials = ldsMat;
    float4 *ldsFlat = (float4 *) (((uintptr_t) (ldsMat + S.nMaterials) + 15u) & ~(uintptr_t) 15u);      /* FLAT: the table of leaf boxes (traverseFlat) */
    if (FLAT) for (uint32_t i = threadIdx.x; i < 2u * S.nFlatLeaves; i += BLOCK) ldsFlat[i] = S.flatLeaves[i];
    lds_cf4 *flat = (lds_cf4 *) ldsFlat;
    for (uint32_t i = threadIdx.x; i < S.nTriangles * TRISHADE_FLOAT4S; i += BLOCK) ldsTriShade[i] = S.triShade[i];
    S.triShade = ldsTriShade;                                   /* (generic pointer into LDS: six loads per vertex, not the inner loop) */
    TravStack stk; setupTraversal(S, g_smem, nullptr, stk);     /* stack + all nodes + all Wald records in LDS (barrier inside); the host checked that nothing can spill */

    const uint32_t waveId = blockIdx.x * (BLOCK / 64) + (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6)), lane = __lane_id();
    unsigned long long next = 0, end = 0;                       /* the wave's chunk of sample ids (wave-uniform) */
    const uint32_t waveInBlock = (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const WaveBalance wb = waveBalanceAt(g_smem, waveInBlock);  /* FLAT >= 2 && MEGA_BALANCE: over the traversal stack, which the flat table does not use (phip.hip sizes it) */
#if MEGA_REGEN_QUEUE
    uint32_t qHead = 0, qCount = 0;                             /* the wave's queue of prepared camera samples (wave-uniform) */
    V3 camO;                                                    /* the origin cameraRay returns for every sample (dv_scene.h: the camera-to-world translation, by its own expression) */
    { V3 d_; float mn_, mx_; cameraRay(S.cam, 0.5f, 0.5f, camO, d_, mn_, mx_); }
#endif
    bool exhausted = rc.totalIds == 0;

    bool alive = false;
    PathVertex v; v.id = v.pixel = v.k = v.state = 0;
    v.hit = v.rayO = v.rayD = v.thr = make_float4(0, 0, 0, 0); v.mis = make_float2(0, 0);
    float4 accum = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MC_COUNT; ++i) ldsCount[i][threadIdx.x] = 0;

#if MEGA_PROFILE
    unsigned long long pfT[4] = { 0, 0, 0, 0 }, pfL[4] = { 0, 0, 0, 0 }, pfIter = 0;      /* regeneration, closest hit, vertex, shadow ray */
#define PF_BEGIN unsigned long long pf0_ = clock64();
#define PF_END(i, lanes) { __builtin_amdgcn_s_waitcnt(0); pfT[i] += clock64() - pf0_; pfL[i] += (unsigned long long) __popcll(lanes); }
#else
#define PF_BEGIN
#define PF_END(i, lanes)
#endif
    for (;;) {
#if MEGA_PROFILE
        ++pfIter;
#endif
        { PF_BEGIN
#if MEGA_PROFILE
        const unsigned long long pfWant_ = __ballot(!alive);
#endif
        /* ---- regeneration: lanes without a path start the next camera sample (integrator.cpp:157-183) ---- */
#if MEGA_REGEN_QUEUE
        /* Camera samples are prepared 64 at a time by ALL lanes of the wave (id decode, pixel jitter, camera ray: ~670 instructions) into a
           per-wave LDS queue; a lane whose path ended pops the next entry (eleven LDS words).  Before, that code ran in every pass of the
           loop for the ~18 lanes (28 %) whose paths had just ended: 14 % of the kernel's time at a quarter of the lanes. */
        for (;;) {
            const unsigned long long want = __ballot(!alive);
            if (!want) break;
            if (qCount == 0u) {
                if (exhausted) break;
                if (next >= end) {                              /* draw a chunk (wave-uniform branch) */
                    unsigned long long base = 0; uint32_t chunk = 0;
                    if (lane == 0) {
                        const unsigned long long seen = __hip_atomic_load(M.nextId, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const bool stop = __hip_atomic_load(M.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
                        if (seen < rc.totalIds && !stop) {
                            /* guided self-scheduling: half of what would remain per wave, within [64, 4096] ids */
                            const unsigned long long share = (rc.totalIds - seen) / (2ull * M.nWaves);
                            chunk = (uint32_t) (share > MEGA_CHUNK_MAX ? MEGA_CHUNK_MAX : (share < MEGA_CHUNK_MIN ? MEGA_CHUNK_MIN : share));
                            chunk &= ~63u;
                            base = atomicAdd(M.nextId, (unsigned long long) chunk);
                        } else {
                            base = rc.totalIds;
                        }
                    }
                    const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t) base), bhi = __builtin_amdgcn_readfirstlane((uint32_t) (base >> 32));
                    chunk = __builtin_amdgcn_readfirstlane(chunk);
                    next = ((unsigned long long) bhi << 32) | blo;
                    end = next + chunk; if (end > rc.totalIds) end = rc.totalIds;
                    if (next >= end) { exhausted = true; break; }
                }
                /* the next 64 ids of the chunk, one per lane (ids outside the crop window -- edge blocks -- are consumed and skipped) */
                const unsigned long long id = next + lane;
                uint32_t px = 0, py = 0, k = 0;
                const bool valid = id < end && decodeId(rc, S.film, id, px, py, k);
                const unsigned long long vmask = __ballot(valid);
                if (valid) {
                    const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t) (vmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) vmask, 0u));
                    const uint32_t pixel = py * (uint32_t) S.film.width + px;
                    const V2 jit = streamJitter<QMC>(rc, pixel, k, (uint32_t) S.film.width);
                    if (QMC && rc.jitter) rc.jitter[id] = make_float2(jit.x, jit.y);      /* for the film pass (64 consecutive ids: one 512-byte store per wave) */
                    const float sx = (float) px + jit.x, sy = (float) py + jit.y;
                    V3 o, d; float mint, maxt;
                    cameraRay(S.cam, sx, sy, o, d, mint, maxt);
                    uint32_t *q = &ldsRegen[waveInBlock][0][pos];
                    q[0 * 64] = pm_to_bits(mint);
                    q[1 * 64] = pm_to_bits(d.x); q[2 * 64] = pm_to_bits(d.y); q[3 * 64] = pm_to_bits(d.z); q[4 * 64] = pm_to_bits(maxt);
                    q[5 * 64] = (uint32_t) id; q[6 * 64] = pixel; q[7 * 64] = k;
                    if (QMC) {
                        const uint64_t sidx = isSequenceSampler(rc.sampler) ? seqIndex(rc, k, px, py) : 0ull;
                        q[8 * 64] = (uint32_t) sidx; q[9 * 64] = (uint32_t) (sidx >> 32);
                    }
                }
                qHead = 0u; qCount = (uint32_t) __popcll(vmask);
                next = (end - next < 64ull) ? end : next + 64ull;
                if (qCount == 0u) continue;
            }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (want >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) want, 0u));
            if (!alive && rank < qCount) {
                const uint32_t *q = &ldsRegen[waveInBlock][0][qHead + rank];
                v.rayO = make_float4(camO.x, camO.y, camO.z, pm_from_bits(q[0 * 64]));
                v.rayD = make_float4(pm_from_bits(q[1 * 64]), pm_from_bits(q[2 * 64]), pm_from_bits(q[3 * 64]), pm_from_bits(q[4 * 64]));
                v.id = q[5 * 64]; v.pixel = q[6 * 64]; v.k = q[7 * 64];
                if (QMC) { ldsSeq[waveInBlock][0][lane] = q[8 * 64]; ldsSeq[waveInBlock][1][lane] = q[9 * 64]; }
                v.thr = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
                v.mis = make_float2(0.0f, 0.0f);
                v.state = 1u | F_ALIVE | F_EMITTED | F_FIRST;
                accum = make_float4(0, 0, 0, 0);
                alive = true;
            }
            const uint32_t wanted = (uint32_t) __popcll(want), took = wanted < qCount ? wanted : qCount;
            qHead += took; qCount -= took;
        }
#else
        for (;;) {
            const unsigned long long want = __ballot(!alive);
            if (!want || exhausted) break;
            if (next >= end) {                                  /* draw a chunk (wave-uniform branch) */
                unsigned long long base = 0; uint32_t chunk = 0;
                if (lane == 0) {
                    const unsigned long long seen = __hip_atomic_load(M.nextId, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const bool stop = __hip_atomic_load(M.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
                    if (seen < rc.totalIds && !stop) {
                        /* guided self-scheduling: half of what would remain per wave, within [64, 4096] ids */
                        const unsigned long long share = (rc.totalIds - seen) / (2ull * M.nWaves);
                        chunk = (uint32_t) (share > MEGA_CHUNK_MAX ? MEGA_CHUNK_MAX : (share < MEGA_CHUNK_MIN ? MEGA_CHUNK_MIN : share));
                        chunk &= ~63u;
                        base = atomicAdd(M.nextId, (unsigned long long) chunk);
                    } else {
                        base = rc.totalIds;
                    }
                }
                const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t) base), bhi = __builtin_amdgcn_readfirstlane((uint32_t) (base >> 32));
                chunk = __builtin_amdgcn_readfirstlane(chunk);
                next = ((unsigned long long) bhi << 32) | blo;
                end = next + chunk; if (end > rc.totalIds) end = rc.totalIds;
                if (next >= end) { exhausted = true; break; }
            }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (want >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) want, 0u));
            const unsigned long long id = next + rank;
            if (!alive && id < end) {
                uint32_t px, py, k;
                if (decodeId(rc, S.film, id, px, py, k)) {      /* ids outside the crop window (edge blocks) are consumed and skipped */
                    const uint32_t pixel = py * (uint32_t) S.film.width + px;
                    const V2 jit = streamJitter<QMC>(rc, pixel, k, (uint32_t) S.film.width);
                    if (QMC && rc.jitter) rc.jitter[id] = make_float2(jit.x, jit.y);
                    const float sx = (float) px + jit.x, sy = (float) py + jit.y;
                    V3 o, d; float mint, maxt;
                    cameraRay(S.cam, sx, sy, o, d, mint, maxt);
                    v.rayO = make_float4(o.x, o.y, o.z, mint);
                    v.rayD = make_float4(d.x, d.y, d.z, maxt);
                    v.thr = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
                    v.mis = make_float2(0.0f, 0.0f);
                    v.id = (uint32_t) id; v.pixel = pixel; v.k = k;
                    v.state = 1u | F_ALIVE | F_EMITTED | F_FIRST;
                    accum = make_float4(0, 0, 0, 0);
                    alive = true;
                }
            }
            const unsigned long long used = (unsigned long long) __popcll(want);
            next = (end - next < used) ? end : next + used;
        }
#endif
        PF_END(0, pfWant_) }
        if (MM != 0 && MEGA_CLASS_DEAL && FLAT >= 2 && MEGA_BALANCE) {
            if (!__syncthreads_or(alive ? 1 : 0)) break;        /* (the waves of a block meet at barriers below: they leave the loop together) */
        } else if (!__any(alive)) break;

        /* ---- closest hit ---- */
        uint32_t hitCls = 0;                                    /* shade class of the record hit (the Wald record's 12th word: 0 diffuse, 1 rough conductor, 2 dielectric) */
        { PF_BEGIN
        if (FLAT >= 2 && MEGA_BALANCE) {                        /* every lane takes part: the tests of the wave's rays are dealt over its lanes */
            const V3 o(v.rayO.x, v.rayO.y, v.rayO.z), d(v.rayD.x, v.rayD.y, v.rayD.z);
            float mint, maxt;
            TravResult r;
            uint32_t nNode = 0, nTri = 0;
            V3 rcp;
            const bool go = alive & clipToSceneSel<false>(S, o, d, v.rayO.w, v.rayD.w, mint, maxt, rcp);
            traverseFlat2W<false, FLAT == 3>(flat, S.nFlatLeaves, stk.tris, wb, lane, go, o, d, rcp, mint, maxt, r, nNode, nTri);
            if (alive) {
                v.hit = make_float4(r.t, r.u, r.v, pm_from_bits(r.prim));
                hitCls = r.cls;
                MEGA_COUNT(MC_RAYS, 1); MEGA_COUNT(MC_NODE, nNode); MEGA_COUNT(MC_TRI, nTri);
            }
        } else
        if (alive) {
            const V3 o(v.rayO.x, v.rayO.y, v.rayO.z), d(v.rayD.x, v.rayD.y, v.rayD.z);
            float mint, maxt;
            TravResult r; r.prim = PHIP_NO_HIT; r.t = INFINITY; r.u = r.v = 0;
            uint32_t nNode = 0, nTri = 0;
            V3 rcp;
            if (MEGA_CLIP_SEL ? clipToSceneSel<false>(S, o, d, v.rayO.w, v.rayD.w, mint, maxt, rcp) : clipToScene<false>(S, o, d, v.rayO.w, v.rayD.w, mint, maxt, rcp)) {
                if (FLAT == 2) traverseFlat2<false>(flat, S.nFlatLeaves, stk.tris, o, d, rcp, mint, maxt, r, nNode, nTri);
                else if (FLAT) traverseFlat<false>(S, flat, S.nFlatLeaves, o, d, rcp, mint, maxt, stk, r, nNode, nTri);
                else traverse<false, true>(S, o, d, rcp, mint, maxt, stk, r, nNode, nTri);
            }
            v.hit = make_float4(r.t, r.u, r.v, pm_from_bits(r.prim));
            MEGA_COUNT(MC_RAYS, 1); MEGA_COUNT(MC_NODE, nNode); MEGA_COUNT(MC_TRI, nTri);
        }

        PF_END(1, __ballot(alive)) }
        /* ---- the paths of the block dealt to its lanes by BSDF model (MEGA_CLASS_DEAL above) ---- */
        if (MM != 0 && MEGA_CLASS_DEAL && FLAT >= 2 && MEGA_BALANCE) {
            /* order: rough conductors, dielectrics, diffuse surfaces (and rays that left the scene), lanes without a path -- the expensive models end up in the
               first wave(s), the idle lanes in the last (which then prepares its camera samples 64 at a time) */
            const uint32_t key = !alive ? 3u : ((pm_to_bits(v.hit.w) == PHIP_NO_HIT || hitCls == 0u) ? 2u : (hitCls == 1u ? 0u : 1u));
            uint32_t rank = 0;
#pragma unroll
            for (uint32_t c = 0; c < 4; ++c) {
                const unsigned long long m = __ballot(key == c);
                if (key == c) rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u));
                if (lane == 0) ldsClsCnt[c][waveInBlock] = (uint32_t) __popcll(m);
            }
            __syncthreads();
            uint32_t base = 0, special = 0;
#pragma unroll
            for (uint32_t c = 0; c < 4; ++c)
#pragma unroll
                for (uint32_t w = 0; w < BLOCK / 64; ++w) {
                    const uint32_t n = ldsClsCnt[c][w];
                    if (c < key || (c == key && w < waveInBlock)) base += n;
                    if (c < 2u) special += n;
                }
            /* block-uniform: nothing to separate in a pass whose vertices are all diffuse.  (Skipping the exchange also when copper and glass already lie in as
               few waves as they fill changes nothing: profiles/r05_gpu_call_p_*) */
            if (special) {
                const uint32_t dst = base + rank;
                uint32_t *x = (uint32_t *) g_smem;               /* [MEGA_DEAL_DWORDS][BLOCK], over the traversal stack / work lists (unused between traversals) */
#define XPUT(j, val) x[(j) * BLOCK + dst] = (val)
#define XGET(j) x[(j) * BLOCK + threadIdx.x]
                XPUT(0, pm_to_bits(v.hit.x)); XPUT(1, pm_to_bits(v.hit.y)); XPUT(2, pm_to_bits(v.hit.z)); XPUT(3, pm_to_bits(v.hit.w));
                XPUT(4, pm_to_bits(v.rayD.x)); XPUT(5, pm_to_bits(v.rayD.y)); XPUT(6, pm_to_bits(v.rayD.z));
                XPUT(7, pm_to_bits(v.thr.x)); XPUT(8, pm_to_bits(v.thr.y)); XPUT(9, pm_to_bits(v.thr.z)); XPUT(10, pm_to_bits(v.thr.w));
                __syncthreads();
                v.hit = make_float4(pm_from_bits(XGET(0)), pm_from_bits(XGET(1)), pm_from_bits(XGET(2)), pm_from_bits(XGET(3)));
                v.rayD = make_float4(pm_from_bits(XGET(4)), pm_from_bits(XGET(5)), pm_from_bits(XGET(6)), v.rayD.w);
                v.thr = make_float4(pm_from_bits(XGET(7)), pm_from_bits(XGET(8)), pm_from_bits(XGET(9)), pm_from_bits(XGET(10)));
                __syncthreads();
                XPUT(0, pm_to_bits(v.mis.x)); XPUT(1, pm_to_bits(v.mis.y)); XPUT(2, v.id); XPUT(3, v.pixel); XPUT(4, v.k | (alive ? 0x80000000u : 0u)); XPUT(5, v.state);
                XPUT(6, pm_to_bits(accum.x)); XPUT(7, pm_to_bits(accum.y)); XPUT(8, pm_to_bits(accum.z)); XPUT(9, pm_to_bits(accum.w));
#if MEGA_REGEN_QUEUE
                if (QMC) { XPUT(10, ldsSeq[QMC ? waveInBlock : 0][0][lane]); XPUT(11, ldsSeq[QMC ? waveInBlock : 0][1][lane]); }
#endif
                __syncthreads();
                v.mis = make_float2(pm_from_bits(XGET(0)), pm_from_bits(XGET(1))); v.id = XGET(2); v.pixel = XGET(3);
                { const uint32_t ka = XGET(4); v.k = ka & 0x7FFFFFFFu; alive = (ka >> 31) != 0u; }
                v.state = XGET(5);
                accum = make_float4(pm_from_bits(XGET(6)), pm_from_bits(XGET(7)), pm_from_bits(XGET(8)), pm_from_bits(XGET(9)));
#if MEGA_REGEN_QUEUE
                if (QMC) { ldsSeq[QMC ? waveInBlock : 0][0][lane] = XGET(10); ldsSeq[QMC ? waveInBlock : 0][1][lane] = XGET(11); }
#endif
#undef XPUT
#undef XGET
                __syncthreads();                                /* the region goes back to the traversals' work lists */
            }
        }

        /* ---- the vertex: emitter hit / Russian roulette / emission / NEE sample / BSDF sample ---- */
        bool pushShadow = false, ended = false;
        ShadowEntry sh;
        if (FLAT >= 2 && MEGA_BALANCE) sh.e0 = sh.e1 = make_float4(0, 0, 0, 0);   /* every lane clips "its" entry (a lane without one takes no part in the result) */
        { PF_BEGIN
        if (alive) {
            uint32_t nv = 0;
            bool newRay;
#if MEGA_REGEN_QUEUE
            const LRegister acc{ accum, (QMC && isSequenceSampler(rc.sampler)) ? &ldsSeq[QMC ? waveInBlock : 0][0][lane] : nullptr };
#else
            const LRegister acc{ accum, nullptr };
#endif
            ended = shadeVertex<MM, STRICT, QMC ? 8 : 0>(S, tab.T, tab.materials, rc, v, acc, newRay, pushShadow, sh, nv);
            if (ended) MEGA_COUNT(MC_VERTICES, nv);
        }

        PF_END(2, __ballot(alive)) }
        /* ---- shadow ray of the NEE sample; unoccluded: the contribution joins the accumulator (path.cpp:187-199) ---- */
        { PF_BEGIN
        if (FLAT >= 2 && MEGA_BALANCE) {
            const V3 o(sh.e0.x, sh.e0.y, sh.e0.z), d(sh.e1.x, sh.e1.y, sh.e1.z);
            float mint, maxt;
            TravResult r;
            uint32_t nNode = 0, nTri = 0;
            V3 rcp;
            const bool go = pushShadow & clipToSceneSel<true>(S, o, d, PT_EPSILON, sh.e0.w, mint, maxt, rcp);
            const bool occluded = traverseFlat2W<true, FLAT == 3>(flat, S.nFlatLeaves, stk.tris, wb, lane, go, o, d, rcp, mint, maxt, r, nNode, nTri);
            if (pushShadow) {
                MEGA_COUNT(MC_SH_RAYS, 1); MEGA_COUNT(MC_SH_NODE, nNode); MEGA_COUNT(MC_SH_TRI, nTri);
                if (!occluded) { accum.x += sh.e2.x; accum.y += sh.e2.y; accum.z += sh.e2.z; }
            }
        } else
        if (pushShadow) {
            const V3 o(sh.e0.x, sh.e0.y, sh.e0.z), d(sh.e1.x, sh.e1.y, sh.e1.z);
            float mint, maxt;
            bool occluded = false;
            TravResult r;
            uint32_t nNode = 0, nTri = 0;
            V3 rcp;
            if (MEGA_CLIP_SEL ? clipToSceneSel<true>(S, o, d, PT_EPSILON, sh.e0.w, mint, maxt, rcp) : clipToScene<true>(S, o, d, PT_EPSILON, sh.e0.w, mint, maxt, rcp))
                occluded = FLAT == 2 ? traverseFlat2<true>(flat, S.nFlatLeaves, stk.tris, o, d, rcp, mint, maxt, r, nNode, nTri)
                         : FLAT ? traverseFlat<true>(S, flat, S.nFlatLeaves, o, d, rcp, mint, maxt, stk, r, nNode, nTri)
                                : traverse<true, true>(S, o, d, rcp, mint, maxt, stk, r, nNode, nTri);
            MEGA_COUNT(MC_SH_RAYS, 1); MEGA_COUNT(MC_SH_NODE, nNode); MEGA_COUNT(MC_SH_TRI, nTri);
            if (!occluded) { accum.x += sh.e2.x; accum.y += sh.e2.y; accum.z += sh.e2.z; }
        }

        PF_END(3, __ballot(pushShadow)) }
        if (ended) {
            L[v.id] = accum;
            MEGA_COUNT(MC_SAMPLES, 1);
            alive = false;
        }
    }
#if MEGA_PROFILE
    if (lane == 0) {      /* rows: closest rays / nodes / tris / shadow rays = ticks of the four phases; shadow nodes / tris / vertices = lanes x 1 of phases 1..3; samples stay */
        ldsCount[MC_RAYS][threadIdx.x] = (uint32_t) (pfT[0] >> 8); ldsCount[MC_NODE][threadIdx.x] = (uint32_t) (pfT[1] >> 8); ldsCount[MC_TRI][threadIdx.x] = (uint32_t) (pfT[2] >> 8);
        ldsCount[MC_SH_RAYS][threadIdx.x] = (uint32_t) (pfT[3] >> 8);
        ldsCount[MC_SH_NODE][threadIdx.x] = (uint32_t) pfL[1]; ldsCount[MC_SH_TRI][threadIdx.x] = (uint32_t) pfL[2]; ldsCount[MC_VERTICES][threadIdx.x] = (uint32_t) pfL[3];
        ldsCount[MC_SAMPLES][threadIdx.x] = (uint32_t) pfIter;
    } else {
#pragma unroll
        for (int i = 0; i < MC_COUNT; ++i) ldsCount[i][threadIdx.x] = 0;
    }
#endif

    /* per-wave statistics (one owner per entry, no atomics) */
    PathPool P; P.stat = M.stat; P.nWaves = M.nWaves;
    const int rows[MC_COUNT] = { ST_SAMPLES, ST_VERTICES, ST_CLOSEST_RAYS, ST_NODE, ST_TRI, ST_SHADOW_RAYS, ST_SH_NODE, ST_SH_TRI };
#pragma unroll
    for (int i = 0; i < MC_COUNT; ++i) waveStat(P, rows[i], waveId, ldsCount[i][threadIdx.x]);
}

