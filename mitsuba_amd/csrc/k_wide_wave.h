/*
 * k_wide_wave.h -- traceWideW: the rays of a wave's 64 paths traversed TO COMPLETION on the compressed 8-wide tree (k_wide_node.h), inside the fused
 * kernel (k_mega<.., FLAT = 4 / 5, ..>: scenes whose tree does not fit LDS -- round 6).
 *
 * k_rays_w (k_wide.h) is a ray SERVER: persistent waves fetch rays from the pool in HBM, refill their idle lanes, and write hits back -- the wavefront's
 * ~350 B of path state per vertex and three launches per iteration are what a scene of 1 000 triangles pays 3.2 x for against the LDS-resident boxes
 * (profiles/r05_gpu_call_w_*), not its traversal (2.1 - 2.6 node steps per ray on a tree that lives in L2).  Here the rays never leave the registers of the lane
 * that owns the path: every lane brings up to two rays -- the shadow ray of the vertex it has just shaded and the next ray of its path (sahkdtree3.h:178-308
 * answers both) -- and the wave runs k_rays_w's flat loop until all of them are decided:
 *   - one node step per iteration for the lanes that have one to take (WIDE_NODE_STEP: five 16-byte loads, the top of the tree from the block's LDS cache);
 *   - the Wald tests the wave has pending dealt over its lanes through a work list in LDS (the same round as persistentTraverseWide: smallest t, at equal
 *     t the highest triangle index -- winsTie -- whatever the order; any-hit: the first hit in group order, so the work counters are those of the
 *     sequential loop);
 *   - a lane whose shadow ray is decided starts its second ray in the next iteration WITHOUT waiting for the wave: the "refill" of k_rays_w, from the lane's
 *     own registers.  One traversal phase per path vertex instead of two: the wave waits for its slowest lane once, and a lane with a short shadow ray and a
 *     long next ray (or the reverse) evens out.
 * Rays arrive clipped to the scene box (k_clip.h: the reference's arithmetic, IEEE divisions); the slab reciprocal is v_rcp_f32 (slabRcpFast: boxes are
 * conservative, the hit is decided by the Wald test on (o, d, mint', maxt')).  Every lane of the wave must call, converged.
 */
#pragma once

/* bytes of dynamic LDS of a block of k_mega<.., FLAT >= 4, ..>: [group stack | top-of-tree node cache | the four waves' round buffers | S-box (mailbox builds) |
   emitter table | materials (FLAT 4)] -- phip.hip sizes the launch with it, k_mega carves it */
__host__ __device__ __forceinline__ size_t megaWideDealOffset(uint32_t nodeCache) { return wideLdsBytes(nodeCache, BLOCK); }
__host__ __device__ __forceinline__ size_t megaWideLdsBytesOf(const DevScene &S, uint32_t nodeCache, bool mailbox, bool matsInLds, size_t sboxBytes) {
    return megaWideDealOffset(nodeCache) + (size_t) (BLOCK / 64u) * WD_WAVE_BYTES + (mailbox ? sboxBytes : 0)
         + (size_t) ((S.emitterTabSize + 3u) & ~3u) * sizeof(float) + (matsInLds ? (size_t) S.nMaterials * sizeof(DevMaterial) : 0) + 16;
}

template <bool HAVE_S, bool HAVE_C>
__device__ __forceinline__ void traceWideW(const DevScene &S, WideStackT<BLOCK> &stack, unsigned char *dealLds /* WD_WAVE_BYTES of this wave */, const uint32_t lane,
                                           bool goS, const V3 &oS, const V3 &dS, const float mintS, const float maxtS,      /* the any-hit ray, clipped */
                                           bool goC, const V3 &oC, const V3 &dC, const float mintC, const float maxtC,      /* the closest-hit ray, clipped */
                                           bool &occluded, TravResult &res,
                                           unsigned long long *wcS, unsigned long long *wcC /* LDS, per wave: node steps | triangle tests << 32 of its any-hit / closest-hit rays (as k_rays_w) */) {
    lds_w64 *slot = (lds_w64 *) dealLds; lds_u2 *uvs = (lds_u2 *) (dealLds + 64u * 8u); lds_w16 *list = (lds_w16 *) (dealLds + 2u * 64u * 8u);
    bool active = false, shadow = false;
    uint32_t steps = 0;
    WideRay ray; ray.o = ray.d = ray.rcp = V3(0.0f); ray.mint = ray.maxt = 0; ray.octinv4 = 0;
    uint2 ng = make_uint2(0u, 0u), tg = make_uint2(0u, 0u);
    occluded = false;
    res.prim = PHIP_NO_HIT; res.t = INFINITY; res.u = res.v = 0; res.cls = 0;
    if (!HAVE_S) goS = false;
    if (!HAVE_C) goC = false;
    for (;;) {
        /* ---- a lane without a ray in flight starts the next one it brought: the shadow ray first ---- */
        if (!active && (goS || goC)) {
            const bool s = HAVE_S && goS;
            const V3 o = (HAVE_S && HAVE_C) ? (s ? oS : oC) : (HAVE_S ? oS : oC), d = (HAVE_S && HAVE_C) ? (s ? dS : dC) : (HAVE_S ? dS : dC);
            const float mint = (HAVE_S && HAVE_C) ? (s ? mintS : mintC) : (HAVE_S ? mintS : mintC), maxt = (HAVE_S && HAVE_C) ? (s ? maxtS : maxtC) : (HAVE_S ? maxtS : maxtC);
            wideRaySetup(ray, o, d, V3(slabRcpFast(d.x), slabRcpFast(d.y), slabRcpFast(d.z)), mint, maxt);
            ng = wideRootGroup(); tg = make_uint2(0u, 0u);
            stack.sp = 0; shadow = s; active = true; steps = 0;
            slot[lane] = ~0ull;
            if (s) goS = false; else goC = false;
        }
        if (!__ballot(active)) break;                            /* (wave-uniform: a lane that has nothing in flight here has nothing left to start) */
        { constexpr bool wideCullOn = false; (void) wideCullOn;
          if (active && tg.y == 0u && (ng.y & 0xff000000u)) WIDE_NODE_STEP(stack, S, ray, ng, tg, steps) }

        /* ---- the triangle round: every lane takes part (k_wide.h: persistentTraverseWide -- the same statements) ---- */
        const uint32_t pending = active ? tg.y : 0u;
        const uint32_t pc = (uint32_t) __popc(pending);
        bool mine = false;                                       /* this lane's group is decided in this round */
        if (__ballot(pc != 0u)) {                                /* (wave-uniform) */
        uint32_t incl = pc;
        incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
        incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x112 /* row_shr:2 */, 0xf, 0xf, true);
        incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x114 /* row_shr:4 */, 0xf, 0xf, true);
        incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x118 /* row_shr:8 */, 0xf, 0xf, true);
        incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
        incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
        const uint32_t nPairs = (uint32_t) __builtin_amdgcn_readlane((int) incl, 63);
        /* a round costs its instructions whatever the number of pairs: it is held back while few pairs are pending and enough other lanes have node steps to take */
        const bool roundNow = WD_THRESHOLD == 0u || nPairs >= WD_THRESHOLD || 2u * (uint32_t) __popcll(__ballot(pc != 0u)) >= (uint32_t) __popcll(__ballot(active));
        if (nPairs && roundNow) {                                /* (wave-uniform) */
            const bool fits = incl <= WD_CAP;                    /* a prefix of the lanes, never empty: a group has at most 24 records */
            const uint32_t nFit = (uint32_t) __popcll(__ballot(fits));
            const uint32_t total = (uint32_t) __builtin_amdgcn_readlane((int) incl, (int) (nFit - 1u));
            mine = pc != 0u && fits;
            if (mine) {
                const uint32_t tag = (shadow ? 0x800u : 0u) | (lane << 5);
                lds_w16 *w = list + (incl - pc);
                uint32_t m = pending;
                do { *w++ = (uint16_t) (tag | (uint32_t) __builtin_ctz(m)); m &= m - 1u; } while (m);
            }
            WD_SYNC()
            for (uint32_t base = 0; base < total; base += 64u) {
                const uint32_t i = base + lane;
                const uint32_t item = list[i];                   /* (behind `total`: stale entries, fetched -- every lane must be active in a ds_bpermute -- and not tested) */
                const uint32_t owner = (item >> 5) & 63u, bit = item & 31u;
                const bool anyHit = (item & 0x800u) != 0u;
                const int src = (int) (owner << 2);
#define WW_FETCH(x) pm_from_bits((uint32_t) __builtin_amdgcn_ds_bpermute(src, (int) pm_to_bits(x)))
                const V3 po(WW_FETCH(ray.o.x), WW_FETCH(ray.o.y), WW_FETCH(ray.o.z)), pd(WW_FETCH(ray.d.x), WW_FETCH(ray.d.y), WW_FETCH(ray.d.z));
                const float pmint = WW_FETCH(ray.mint), pmaxt = WW_FETCH(ray.maxt);
#undef WW_FETCH
                const uint32_t tbase = (uint32_t) __builtin_amdgcn_ds_bpermute(src, (int) tg.x);
                if (i < total) {                                 /* (only the last step of a round is partial) */
                    WIDE_LOAD_TRI(S, tbase + bit, a, b, c)
                    float tu, tv, tt;
                    if (waldIntersectSel(a, b, c, po, pd, pmint, pmaxt, tu, tv, tt)) {
                        const unsigned long long key = anyHit ? (unsigned long long) bit
                            : (((unsigned long long) pm_to_bits(tt) << 32) | (unsigned long long) (((HIT_PRIM_MASK - pm_to_bits(c.z)) << 2) | (pm_to_bits(c.w) & 3u)));
                        __hip_atomic_fetch_min(slot + owner, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (!anyHit && slot[owner] == key) { u2v q; q.x = pm_to_bits(tu); q.y = pm_to_bits(tv); uvs[owner] = q; }
                    }
                }
            }
            WD_SYNC()
        }
        }
        if (active) {
            bool finished = false;
            if (mine) {
                const unsigned long long best = slot[lane];
                if (shadow) {
                    const bool occ = best != ~0ull;
                    steps += (uint32_t) __popc(occ ? (tg.y & ((2u << ((uint32_t) best & 31u)) - 1u)) : tg.y) << 16;
                    finished = occ;
                } else {
                    steps += pc << 16;
                    if (best != ~0ull) ray.maxt = pm_from_bits((uint32_t) (best >> 32));
                }
                tg.y = 0u;
            }
            if (!finished && tg.y == 0u && !(ng.y & 0xff000000u)) {
                if (stack.sp == 0) finished = true;
                else {
                    const uint2 e = stack.pop();
                    if (e.y & 0xff000000u) ng = e; else { tg = e; ng = make_uint2(0u, 0u); }
                }
            }
            if (finished) {
                const unsigned long long best = slot[lane];
                if (shadow) occluded = best != ~0ull;
                else {
                    if (best != ~0ull) {
                        const u2v q = uvs[lane];
                        const uint32_t lo = (uint32_t) best;
                        res.t = pm_from_bits((uint32_t) (best >> 32)); res.u = pm_from_bits(q.x); res.v = pm_from_bits(q.y);
                        res.prim = HIT_PRIM_MASK - (lo >> 2); res.cls = lo & 3u;
                    }
                }
                /* node steps (low word) and triangle tests (high word) of the ray in ONE 64-bit LDS add */
                atomicAdd(shadow ? wcS : wcC, (unsigned long long) (steps & 0xFFFFu) | ((unsigned long long) (steps >> 16) << 32));
                active = false;
            }
        }
    }
}

/* ======================================================================================
 *  traceWidePool: the wave's rays traversed through ONE SHARED TASK STACK (round 6)
 * ======================================================================================
 * traceWideW keeps k_rays_w's per-lane state machine: a lane walks ITS ray, and the wave leaves the loop when its slowest lane does.  Measured on the Cornell box
 * with two 500-triangle spheres (profiles/r06_valu_*): 2.1 node visits per ray on average, ~7 loop iterations per traversal phase -- the node step, 210 VALU
 * instructions, runs at a third of its lanes; lane utilisation of the whole kernel 0.43.  (k_rays_w hides that behind refills from a pool of rays in HBM; a lane that
 * owns a path has no other ray to take.)  Here the unit of work is not a ray but a NODE VISIT, and any lane takes any ray's:
 *   - the wave keeps one stack of tasks in LDS (8 bytes: node index | ray, or -- on overflow of the pair list -- a group of leaf triangles | ray);
 *   - an iteration pops the top 64 tasks, one per lane; a lane fetches the task's ray (closest-hit rays from their owners' registers through ds_bpermute, any-hit
 *     rays from a table in LDS), tests the node's eight quantised child boxes (wideNodeHits: the same statement), pushes one task per inner child it hit -- far
 *     children first, so that the nearest is on top -- and hands the leaf triangles it hit to the iteration's triangle round (the dealt Wald tests of traceWideW:
 *     LDS min of (t, highest triangle index) on the ray's result slot);
 *   - the result slot IS the ray's interval: it starts as (bits(maxt), no hit), every closer hit lowers its high word, and every later task of the ray is tested
 *     against it (tasks pushed before are rejected by their own box test); an any-hit ray that was occluded gets an empty interval -- its remaining tasks die at the fetch.
 * A lane brings up to TWO rays (MEGA_JOINT): the shadow ray of the vertex it shaded in the previous pass and the next ray of its path -- 64 + ~34 rays keep the 64
 * lanes busy where the ~34 shadow rays of a pass alone left every iteration half empty.
 * Results do not depend on the order of the visits (boxes are conservative; the hit is the minimum over all candidates by winsTie), so they are the bits of the
 * sequential walk; the work counters are those of THIS order (a node visited before a closer hit was known counts), not the sequential walk's.
 * The stack is as deep as the wave's pending visits (a few hundred): WP_CAP entries in LDS, the rest in the wave's slice of the spill buffer; an iteration pops
 * fewer tasks when the stack could not take their children.
 */
#ifndef WP_CAP
#define WP_CAP 256u                      /* task-stack entries per wave in LDS (8 B each): 8 KB per block -- with 512 the block's LDS (slots, ray table, pair list, node cache, camera-sample queue) costs the fourth block of a CU */
#endif
#ifndef WP_NEAR_FIRST
#define WP_NEAR_FIRST 1                  /* the nearest child of every lane on top of the stack (0: every lane's children together, far to near) */
#endif
#define WP_PAIRS 256u                    /* (ray, triangle) pairs per round and wave (4 B each: ray << 25 | record) */
#define WP_TRI_MAX (1u << 25)            /* records the pair list can address (phip.hip checks) */
#define WP_WAVE_BYTES (128u * 8u + 64u * 8u + 64u * 32u + WP_PAIRS * 4u)      /* per wave: result slots of 128 rays, (u, v) of the closest hits, the any-hit rays, pair list */
__host__ __device__ __forceinline__ size_t widePoolNodesOffset() { return (size_t) (BLOCK / 64u) * WP_CAP * sizeof(uint2); }
__host__ __device__ __forceinline__ size_t widePoolDealOffset(uint32_t nodeCache) { return widePoolNodesOffset() + (size_t) nodeCache * 5 * sizeof(uint4); }
__host__ __device__ __forceinline__ size_t megaWidePoolLdsBytesOf(const DevScene &S, uint32_t nodeCache, bool mailbox, bool matsInLds, size_t sboxBytes) {
    return widePoolDealOffset(nodeCache) + (size_t) (BLOCK / 64u) * WP_WAVE_BYTES + (mailbox ? sboxBytes : 0)
         + (size_t) ((S.emitterTabSize + 3u) & ~3u) * sizeof(float) + (matsInLds ? (size_t) S.nMaterials * sizeof(DevMaterial) : 0) + 16;
}

typedef __attribute__((address_space(3))) f4v lds_f4;
struct WidePool {
    lds_u2 *pool;            /* WP_CAP entries of this wave */
    lds_w64 *slot;           /* 128 rays (closest-hit ray of lane l: l, its any-hit ray: 64 + l): bits(interval end) << 32 | (max - prim) << 2 | class, low word ~0: no hit */
    lds_u2 *uvs;
    lds_f4 *srays;           /* the any-hit rays: (o, mint) (d, -) of lane l at [2 l], [2 l + 1] */
    lds_w32 *pairs;
    unsigned long long *spill;   /* global: this wave's slice of the spill buffer, spillCap entries */
    uint32_t spillCap;
    lds_cu4 *nodes;          /* LDS copy of wide nodes [0, nodeCache) (the block's) */
    uint32_t nodeCache;
};

/* carve the block's dynamic LDS and stage the top of the tree (all threads of the block must call) */
__device__ __forceinline__ void setupWidePool(const DevScene &S, uint32_t nodeCache, unsigned char *smem, uint32_t *spillBlock /* SPILL_DEPTH words per lane of this block */, uint32_t waveInBlock, WidePool &wp) {
    uint4 *ln = (uint4 *) (smem + widePoolNodesOffset());
    for (uint32_t i = threadIdx.x; i < nodeCache * 5u; i += BLOCK) ln[i] = S.wnodes[(i / 5u) * WIDE_NODE_STRIDE + i % 5u];
    __syncthreads();
    unsigned char *w = smem + widePoolDealOffset(nodeCache) + waveInBlock * WP_WAVE_BYTES;
    wp.pool = (lds_u2 *) (smem + (size_t) waveInBlock * WP_CAP * sizeof(uint2));
    wp.slot = (lds_w64 *) w; wp.uvs = (lds_u2 *) (w + 128u * 8u); wp.srays = (lds_f4 *) (w + 128u * 8u + 64u * 8u); wp.pairs = (lds_w32 *) (w + 128u * 8u + 64u * 8u + 64u * 32u);
    wp.spill = (unsigned long long *) (spillBlock + (size_t) waveInBlock * 64u * SPILL_DEPTH); wp.spillCap = 64u * SPILL_DEPTH / 2u;
    wp.nodes = (lds_cu4 *) ln; wp.nodeCache = nodeCache;
}

#define WP_OCCLUDED 0xBF80000000000000ull    /* an any-hit ray that found a hit: interval end -1 */
/* Every lane of the wave must call, converged; goS / goC = this lane has an any-hit / a closest-hit ray (clipped: (o, d, mint', maxt')).  `overflow` is set when the
   task stack ran out of LDS + spill (the caller refuses the frame).  wc: node visits | triangle tests << 32 of the wave's any-hit / closest-hit rays (LDS, lane 0 adds). */
template <bool HAVE_S, bool HAVE_C>
__device__ __forceinline__ void traceWidePool(const DevScene &S, const WidePool &wp, const uint32_t lane,
                                              bool goS, const V3 &oS, const V3 &dS, const float mintS, const float maxtS,
                                              bool goC, const V3 &o, const V3 &d, const float mint, const float maxt,
                                              bool &occluded, TravResult &res, unsigned long long *wcS, unsigned long long *wcC, bool &overflow) {
    auto poolWrite = [&](uint32_t i, const uint2 v) {
        if (i < WP_CAP) { u2v t; t.x = v.x; t.y = v.y; wp.pool[i] = t; }
        else if (i - WP_CAP < wp.spillCap) __hip_atomic_store(wp.spill + (i - WP_CAP), ((unsigned long long) v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else overflow = true;
    };
    auto poolRead = [&](uint32_t i) -> uint2 {
        if (i < WP_CAP) { const u2v t = wp.pool[i]; return make_uint2(t.x, t.y); }
        const unsigned long long v = __hip_atomic_load(wp.spill + (i - WP_CAP), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      /* (written by another lane of this wave: past the L1) */
        return make_uint2((uint32_t) v, (uint32_t) (v >> 32));
    };
    if (!HAVE_S) goS = false;
    if (!HAVE_C) goC = false;
    /* the rays' slots: the interval's end, no hit; the any-hit rays' table */
    if (HAVE_C) wp.slot[lane] = ((unsigned long long) pm_to_bits(goC ? maxt : -1.0f) << 32) | 0xFFFFFFFFull;
    if (HAVE_S) {
        wp.slot[64u + lane] = ((unsigned long long) pm_to_bits(goS ? maxtS : -1.0f) << 32) | 0xFFFFFFFFull;
        f4v r0, r1; r0.x = oS.x; r0.y = oS.y; r0.z = oS.z; r0.w = mintS; r1.x = dS.x; r1.y = dS.y; r1.z = dS.z; r1.w = 0.0f;
        wp.srays[2u * lane] = r0; wp.srays[2u * lane + 1u] = r1;
    }
    /* the roots, once per ray: any-hit rays deepest (the closest-hit rays of the wave start first; the mix fills the iterations either way) */
    const unsigned long long gs = __ballot(goS), gc = __ballot(goC);
    const uint32_t nS0 = (uint32_t) __popcll(gs);
    uint32_t count = nS0 + (uint32_t) __popcll(gc);             /* tasks on the stack (wave-uniform) */
    if (goS) poolWrite(__builtin_amdgcn_mbcnt_hi((uint32_t) (gs >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) gs, 0u)), make_uint2(0u, 0x80000000u | 64u | lane));
    if (goC) poolWrite(nS0 + __builtin_amdgcn_mbcnt_hi((uint32_t) (gc >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) gc, 0u)), make_uint2(0u, 0x80000000u | lane));
    uint32_t nNodeS = 0, nNodeC = 0, nTriS = 0, nTriC = 0;      /* (wave-uniform) */
    const uint32_t room = WP_CAP + wp.spillCap;
    uint32_t nQ = 0;                                            /* (ray, record) pairs that wait in the queue (wave-uniform) */
    WD_SYNC()
#define WP_BPERM(srcLane4, x) pm_from_bits((uint32_t) __builtin_amdgcn_ds_bpermute(srcLane4, (int) pm_to_bits(x)))
    /* the ray of a task (every lane must execute: ds_bpermute): origin, direction, interval */
#define WP_FETCH_RAY(ray, po, pd, pmint, pmaxt)                                                                                   \
        V3 po, pd; float pmint, pmaxt;                                                                                            \
        {                                                                                                                         \
            const int src_ = (int) (((ray) & 63u) << 2);                                                                          \
            if (HAVE_C) { po = V3(WP_BPERM(src_, o.x), WP_BPERM(src_, o.y), WP_BPERM(src_, o.z)); pd = V3(WP_BPERM(src_, d.x), WP_BPERM(src_, d.y), WP_BPERM(src_, d.z)); pmint = WP_BPERM(src_, mint); } \
            if (HAVE_S) {                                                                                                         \
                const f4v r0_ = wp.srays[2u * ((ray) & 63u)], r1_ = wp.srays[2u * ((ray) & 63u) + 1u];                             \
                const bool s_ = !HAVE_C || ((ray) & 64u) != 0u;                                                                   \
                po = s_ ? V3(r0_.x, r0_.y, r0_.z) : po; pd = s_ ? V3(r1_.x, r1_.y, r1_.z) : pd; pmint = s_ ? r0_.w : pmint;       \
            }                                                                                                                     \
            pmaxt = pm_from_bits((uint32_t) (wp.slot[(ray)] >> 32));                                                              \
        }
    while (count | nQ) {
        /* ---- pop: the top n tasks, one per lane (fewer when the stack could not take eight children and a triangle group of each) ---- */
        uint32_t n = count < 64u ? count : 64u;
        if (count + 9u * n > room) { const uint32_t fit = room > count ? (room - count) / 9u : 0u; n = fit < 1u ? 1u : (fit < n ? fit : n); }
        const bool have = lane < n;
        uint2 e = make_uint2(0u, 0u);
        if (have) e = poolRead(count - 1u - lane);
        count -= n;
        const bool isNode = have && (e.y >> 31) != 0u;
        const uint32_t ray = isNode ? (e.y & 127u) : (have ? ((e.y >> 24) & 127u) : lane);
        WP_FETCH_RAY(ray, po, pd, pmint, pmaxt)
        const bool live = have && pmaxt >= pmint;
        /* ---- the node visit ---- */
        uint32_t inner = 0u, imask = 0u, childBase = 0u, octant = 0u;     /* inner children hit (bits 24..31, traversal order) */
        uint32_t pending = 0u, tbase = 0u;                       /* leaf triangles to test in this iteration's round */
        if (isNode && live) {
            WIDE_LOAD_NODE(wp, S, e.x, n0, n1, n2, n3, n4)
            WideRay r; wideRaySetup(r, po, pd, V3(slabRcpFast(pd.x), slabRcpFast(pd.y), slabRcpFast(pd.z)), pmint, pmaxt);
            const uint32_t hits = wideNodeHits(n0, n1, n2, n3, n4, r);
            inner = hits & 0xff000000u; imask = n0.w >> 24; childBase = n1.x; octant = r.octinv4 & 7u;
            pending = hits & 0x00ffffffu; tbase = n1.y;
        } else if (live) { pending = e.y & 0x00ffffffu; tbase = e.x; }       /* a triangle group that did not fit an earlier round */
        {
            const unsigned long long nv = __ballot(isNode && live), sv = __ballot((ray & 64u) != 0u);
            nNodeS += (uint32_t) __popcll(nv & sv); nNodeC += (uint32_t) __popcll(nv & ~sv);
        }
        /* ---- the leaf triangles hit join the wave's queue of (ray, record) pairs ---- */
        const uint32_t pc = (uint32_t) __popc(pending);
        bool keep = false;                                       /* this lane's group goes back on the stack (the queue is full) */
        if (__ballot(pc != 0u)) {                                /* (wave-uniform) */
            uint32_t incl = pc;
            incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
            incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x112 /* row_shr:2 */, 0xf, 0xf, true);
            incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x114 /* row_shr:4 */, 0xf, 0xf, true);
            incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x118 /* row_shr:8 */, 0xf, 0xf, true);
            incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
            incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
            const bool fits = nQ + incl <= WP_PAIRS;             /* a prefix of the lanes (fewer than 64 pairs wait from the last iteration, a group has at most 24 records: never empty) */
            const uint32_t nFit = (uint32_t) __popcll(__ballot(fits));
            keep = pc != 0u && !fits;
            if (pc != 0u && fits) {
                const uint32_t tag = ray << 25;
                lds_w32 *w = wp.pairs + nQ + (incl - pc);
                uint32_t m = pending;
                do { *w++ = tag | (tbase + (uint32_t) __builtin_ctz(m)); m &= m - 1u; } while (m);
            }
            nQ += (uint32_t) __builtin_amdgcn_readlane((int) incl, (int) (nFit - 1u));
        }
        /* ---- push: the inner children hit.  The NEAREST child of every lane goes to the top segment of the stack, its other children (far to near) and a triangle group the
                queue had no room for below the top segments of all lanes: the next iteration then pops nearest children of MANY rays rather than all children of a few --
                depth first, front to back per ray, which is what lets a hit found in the near child reject the far ones (WP_NEAR_FIRST: node visits per closest ray of the
                atrium 11.3 -> 10.9, of the glass room 10.9 -> 10.4; the per-lane walk makes 10.0) ---- */
        const uint32_t nInner = (uint32_t) __popc(inner);
        const bool hasTop = WP_NEAR_FIRST && nInner != 0u;
        const uint32_t k = nInner - (hasTop ? 1u : 0u) + (keep ? 1u : 0u);
        const unsigned long long topMask = __ballot(hasTop);
        if (__ballot(k != 0u) | topMask) {                       /* (wave-uniform) */
            uint32_t incl = k;
            incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x111, 0xf, 0xf, true);
            incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x112, 0xf, 0xf, true);
            incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x114, 0xf, 0xf, true);
            incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x118, 0xf, 0xf, true);
            incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x142, 0xa, 0xf, false);
            incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x143, 0xc, 0xf, false);
            const uint32_t nRest = (uint32_t) __builtin_amdgcn_readlane((int) incl, 63);
            uint32_t pos = count + incl - k;
            uint32_t m = hasTop ? (inner & ~(0x80000000u >> __clz((int) inner))) : inner;      /* (without the highest bit: the first child in traversal order) */
            while (m) {                                          /* lowest bit = last in traversal order: it goes deepest */
                const uint32_t bit = (uint32_t) __builtin_ctz(m); m &= m - 1u;
                const uint32_t slotIdx = (bit - 24u) ^ octant;
                poolWrite(pos++, make_uint2(childBase + (uint32_t) __popc(imask & ((1u << slotIdx) - 1u)), 0x80000000u | ray));
            }
            if (keep) poolWrite(pos, make_uint2(tbase, pending | (ray << 24)));
            if (hasTop) {
                const uint32_t bit = 31u - (uint32_t) __clz((int) inner);
                const uint32_t slotIdx = (bit - 24u) ^ octant;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (topMask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) topMask, 0u));
                poolWrite(count + nRest + rank, make_uint2(childBase + (uint32_t) __popc(imask & ((1u << slotIdx) - 1u)), 0x80000000u | ray));
            }
            count += nRest + (uint32_t) __popcll(topMask);
        }
        WD_SYNC()
        /* ---- the triangle steps: one pair per lane; only FULL steps while node visits are left (a step costs its ~100 instructions whatever the number of its pairs:
                the remainder waits for the pairs of the next iteration), everything when the stack is empty ---- */
        const uint32_t nTest = count ? (nQ & ~63u) : nQ;
        if (nTest) {                                             /* (wave-uniform) */
            for (uint32_t base = 0; base < nTest; base += 64u) {
                const uint32_t i = base + lane;
                const uint32_t item = wp.pairs[i];               /* (behind nQ: stale entries, fetched -- every lane must be active in a ds_bpermute -- and not tested) */
                const uint32_t tray = item >> 25;
                WP_FETCH_RAY(tray, to, td, tmint, tmaxt)
                const bool test = i < nTest;                     /* (only the last step of the traversal is partial) */
                if (HAVE_S && HAVE_C) nTriS += (uint32_t) __popcll(__ballot(test && (tray & 64u) != 0u));
                if (test) {
                    WIDE_LOAD_TRI(S, item & (WP_TRI_MAX - 1u), a, b, c)
                    float tu, tv, tt;
                    if (waldIntersectSel(a, b, c, to, td, tmint, tmaxt, tu, tv, tt)) {
                        if (HAVE_S && (!HAVE_C || (tray & 64u))) wp.slot[tray] = WP_OCCLUDED;
                        else {
                            const unsigned long long key = ((unsigned long long) pm_to_bits(tt) << 32) | (unsigned long long) (((HIT_PRIM_MASK - pm_to_bits(c.z)) << 2) | (pm_to_bits(c.w) & 3u));
                            __hip_atomic_fetch_min(wp.slot + tray, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (wp.slot[tray] == key) { u2v q; q.x = pm_to_bits(tu); q.y = pm_to_bits(tv); wp.uvs[tray] = q; }
                        }
                    }
                }
            }
            if (HAVE_S && HAVE_C) nTriC += nTest; else if (HAVE_S) nTriS += nTest; else nTriC += nTest;     /* (joint: nTriC counts both kinds, corrected below) */
            /* the pairs that wait move to the front of the queue */
            const uint32_t rest = nQ - nTest;
            const uint32_t moved = lane < rest ? wp.pairs[nTest + lane] : 0u;
            WD_SYNC()
            if (lane < rest) wp.pairs[lane] = moved;
            nQ = rest;
            WD_SYNC()
        }
    }
#undef WP_FETCH_RAY
#undef WP_BPERM
    if (HAVE_S && HAVE_C) nTriC -= nTriS;
    if (lane == 0u) {
        if (HAVE_S) *wcS += (unsigned long long) nNodeS | ((unsigned long long) nTriS << 32);
        if (HAVE_C) *wcC += (unsigned long long) nNodeC | ((unsigned long long) nTriC << 32);
    }
    occluded = HAVE_S && wp.slot[64u + lane] == WP_OCCLUDED;
    res.prim = PHIP_NO_HIT; res.t = INFINITY; res.u = res.v = 0; res.cls = 0;
    if (HAVE_C) {
        const unsigned long long best = wp.slot[lane];
        const uint32_t lo = (uint32_t) best;
        if (lo != 0xFFFFFFFFu) {
            const u2v q = wp.uvs[lane];
            res.t = pm_from_bits((uint32_t) (best >> 32)); res.u = pm_from_bits(q.x); res.v = pm_from_bits(q.y);
            res.prim = HIT_PRIM_MASK - (lo >> 2); res.cls = lo & 3u;
        }
    }
}
