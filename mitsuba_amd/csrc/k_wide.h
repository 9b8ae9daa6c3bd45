/*
 * k_wide.h -- traversal of the compressed 8-wide BVH (bvh.h: buildWide) for the big scenes: k_rays_w (persistent waves with
 * refill; closest-hit and any-hit rays of an iteration in one launch) and k_raycast_w (phip_trace).  Included by phip.hip
 * after k_rays.h, whose ray sources (ShadowSource / TraceSource) it shares.
 *
 * Why a second structure: the BVH4 ray kernel of the 250k-triangle scenes was measured (round 2, SQ counters) at 34 % VALU
 * issue with 64 % of its wave cycles waiting -- every lane of a wave fetches its own node, i.e. 64 different cache lines per
 * load instruction and seven instructions per 128-byte node, so the CU's vector-memory path is the bound, not the ALUs.  An
 * 80-byte node with eight quantised child boxes (Ylitie, Karras, Laine: "Efficient Incoherent Ray Traversal on GPUs Through
 * Compressed Wide BVHs", HPG 2017) costs five load instructions and replaces ~2.3 BVH4 nodes: fewer bytes AND fewer
 * dependent round trips per ray, paid for with ALU work there is room for.
 *
 * Per-lane state machine, as in k_traverse.h: one node step per loop iteration; the triangle tests the lanes of a wave have pending are dealt
 * over the whole wave (round 4: persistentTraverseWide under WIDE_DEAL, below; k_raycast_w keeps one test per lane and iteration).  The traversal stack
 * holds GROUPS, 8 bytes each: a node group (childBase, hit bits 24..31 | imask) or a triangle group (triBase, hit bits 0..23),
 * so a node pushes at most one entry however many of its children are hit and the stack is as deep as the tree (LDS:
 * WIDE_STACK_LDS entries per lane, the rest spills to HBM).  Children are visited in the order slot ^ rayOctant, which the
 * builder's slot assignment makes approximately front to back -- no sort.  Hits are decided by the same Wald test on the same
 * records (waldIntersect, dv_scene.h); boxes are conservative, so results do not depend on the structure.
 */

#include "k_wide_node.h"


/* ---- work distribution of k_rays_w: chunks of 64 slots (closest-hit rays) and blocks of the shadow queue are DRAWN from sharded
 *      counters instead of being dealt statically (wave w: chunks w, w + W, ...).  Rays differ in cost by an order of magnitude, so
 *      the static deal left the waves finishing far apart: 3.2 of 4 waves per SIMD resident on average over a launch (round-2 SQ
 *      counters).  RAY_SHARDS counters, one 128-byte line each, every shard owns a contiguous range; a wave draws from the shard
 *      of its block and moves on to the next shard when that one is empty.  One atomic per 64 rays (per 256-entry shadow block):
 *      ~20 per wave and launch -- not the per-wave-per-iteration atomics on five shared words that round 1 banned. ---- */
#define RAY_SHARDS 32
#define RAY_SHARD_STRIDE 32              /* uint32 between counters (128 B) */
#ifndef RAY_STATIC_PERCENT
#define RAY_STATIC_PERCENT 60            /* share of a launch's work units dealt statically (wave w: units w, w + W, ...) before the waves draw from the counters */
#endif
/* The source state of a wave (which chunk, how far into it, the draw counters' bookkeeping) is wave-uniform, but derives from
   threadIdx, returning atomics and vector loads of uniform addresses -- values the compiler must assume divergent, so it kept a dozen
   of them in VGPRs across the traversal loop.  readfirstlane pins them to SGPRs. */
#define WIDE_UNIFORM(x) ((uint32_t) __builtin_amdgcn_readfirstlane((int) (x)))
struct DrawCounter {
    unsigned int *ctr;                   /* RAY_SHARDS counters (zeroed before the launch) */
    uint32_t shard0, tried, perShard, total;
    /* A returning atomic on a loaded chip is a round trip of more than a microsecond in which the whole wave stands still (round 3:
       7 % of the ray kernel's wave time went into these draws).  So the first RAY_STATIC_PERCENT of the units are dealt statically
       -- no memory access at all -- and only the rest is drawn: enough to level the waves out (the static deal alone left a fifth
       of the wave slots empty towards the end of a launch), a third of the atomics. */
    uint32_t sNext, sStride, nStatic;    /* static part: units sNext, sNext + sStride, ... < nStatic; dynamic part: units [nStatic, total) */
    __device__ __forceinline__ void init(unsigned int *c, uint32_t shard, uint32_t totalUnits, uint32_t waveId, uint32_t nWavesGrid) {
        ctr = c; shard0 = shard; tried = 0; total = totalUnits;
        nStatic = (uint32_t) ((unsigned long long) totalUnits * RAY_STATIC_PERCENT / 100u) / nWavesGrid * nWavesGrid;
        sNext = WIDE_UNIFORM(waveId); sStride = nWavesGrid;      /* (waveId comes from threadIdx: the compiler cannot know it is wave-uniform) */
        perShard = (total - nStatic + RAY_SHARDS - 1) / RAY_SHARDS;
    }
    /* next unit of this wave (wave-uniform), or 0xFFFFFFFF when there is none left */
    __device__ __forceinline__ uint32_t draw() {
        if (sNext < nStatic) { const uint32_t u = sNext; sNext += sStride; return u; }
        while (tried < RAY_SHARDS) {
            const uint32_t s = (shard0 + tried) % RAY_SHARDS;
            uint32_t c = 0;
            if (__lane_id() == 0) c = atomicAdd(ctr + (size_t) s * RAY_SHARD_STRIDE, 1u);
            c = WIDE_UNIFORM(c);
            const uint32_t first = nStatic + s * perShard, n = first >= total ? 0u : (total - first < perShard ? total - first : perShard);
            if (c < n) return first + c;
            ++tried;                     /* this shard is used up: for good */
        }
        return 0xFFFFFFFFu;
    }
};

/* closest-hit rays of the pool: chunks of 64 slots drawn from the sharded counters.  The rays are PRE-CLIPPED by the shading kernels
   (DevScene::preclip, k_clip.h): (o, mint' | d, maxt'), maxt' < mint' = the ray misses the scene box */
struct TraceSourceDyn {
    const PathPool &P; DrawCounter q; uint32_t chunk, pos;
    __device__ __forceinline__ void start() { chunk = WIDE_UNIFORM(q.draw()); pos = 0; }
    __device__ __forceinline__ bool more() const { return chunk != 0xFFFFFFFFu; }
    __device__ __forceinline__ uint32_t assign(bool want, unsigned long long wantMask) {
        const uint32_t idx = pos + (uint32_t) __popcll(wantMask & ((1ull << __lane_id()) - 1ull));
        const uint32_t h = (want && idx < 64u && chunk * 64u + idx < P.capacity) ? chunk * 64u + idx : INVALID_RAY;
        pos = WIDE_UNIFORM(pos + (uint32_t) __popcll(wantMask));
        if (pos >= 64u) start();
        return h;
    }
    __device__ __forceinline__ bool load(uint32_t slot, V3 &o, V3 &d, float &mint, float &maxt) const {
        /* state and ray in flight together: one memory round trip per refill instead of two (nearly every slot is alive outside the drain phase) */
        const uint32_t st = P.state[slot];
        const float4 ro = P.rayO[slot], rd = P.rayD[slot];
        o = V3(ro.x, ro.y, ro.z); d = V3(rd.x, rd.y, rd.z); mint = ro.w; maxt = rd.w;
        return (st & F_TRACE_MASK) == F_ALIVE;
    }
    __device__ __forceinline__ void commit(uint32_t slot, bool, const TravResult &r) const {
        P.hit[slot] = make_float4(r.t, r.u, r.v, pm_from_bits(r.prim));      /* r.prim = the packed hit word: bits(prim) | shade class << 30 (k_pool.h) */
    }
};

/* the block-compacted shadow queue: blocks of up to 256 entries drawn from the sharded counters; entries pre-clipped: (o, maxt' | d, mint' | c, id) */
struct ShadowSourceDyn {
    const PathPool &P; float4 *L; DrawCounter q; uint32_t blk, pos, cnt;
    __device__ __forceinline__ void start() {
        for (;;) {
            blk = WIDE_UNIFORM(q.draw()); pos = 0; cnt = 0;
            if (blk == 0xFFFFFFFFu) return;
            cnt = WIDE_UNIFORM(P.shadowCount[blk]);      /* (a vector load of a uniform address: the value is uniform, the register is not) */
            if (cnt) return;
        }
    }
    __device__ __forceinline__ bool more() const { return blk != 0xFFFFFFFFu; }
    __device__ __forceinline__ uint32_t assign(bool want, unsigned long long wantMask) {
        const uint32_t idx = pos + (uint32_t) __popcll(wantMask & ((1ull << __lane_id()) - 1ull));
        const uint32_t h = (want && idx < cnt) ? blk * BLOCK + idx : INVALID_RAY;
        pos = WIDE_UNIFORM(pos + (uint32_t) __popcll(wantMask));
        if (pos >= cnt) start();
        return h;
    }
    __device__ __forceinline__ bool load(uint32_t e, V3 &o, V3 &d, float &mint, float &maxt) const {
        const float4 e0 = P.shadow[3 * (size_t) e], e1 = P.shadow[3 * (size_t) e + 1];
        o = V3(e0.x, e0.y, e0.z); d = V3(e1.x, e1.y, e1.z); mint = e1.w; maxt = e0.w;
        return true;
    }
    __device__ __forceinline__ void commit(uint32_t e, bool occluded, const TravResult &) const {
        if (!occluded) {
            const float4 e2 = P.shadow[3 * (size_t) e + 2];
            addRadiance(L, pm_to_bits(e2.w), e2);      /* (three fire-and-forget float atomics instead: C3 158.4 vs 159.3 ms, C4 356 vs 325 ms -- worse; removed, round 3) */
        }
    }
};


/* ---- persistent waves with refill: closest-hit AND any-hit rays of one iteration in ONE launch (as k_rays_p) ----
 * Per-lane state is kept small (the kernel's speed follows its resident waves: it is bound by memory latency and by the CU's
 * vector-memory path, WIDE_WAVES): `steps` = node steps | triangle tests << 16 of the ray in flight, the hit word carries the shade
 * class, the any-hit flag is a lane mask.  (nested variant below: `meta` = handle | any-hit flag << 31) */
enum { WW_RAYS = 0, WW_STEPS, WW_SH_RAYS, WW_SH_STEPS, WW_COUNT };     /* 64-bit LDS counters of a wave: rays, node steps | triangle tests << 32 */
#define WM_HANDLE 0x0FFFFFFFu
#define WM_SHADOW 0x80000000u
#ifndef WIDE_FLAT
#ifndef WIDE_WALD_SEL
#define WIDE_WALD_SEL 1                  /* the Wald test's axis permutation as selects (k_traverse.h: waldIntersectSel) instead of three divergent branches */
#endif
#define WIDE_FLAT 1                      /* ONE loop (refill test, then one traversal iteration of the live lanes) instead of a traversal loop nested in a refill loop */
#endif
#ifndef WIDE_DEAL
#define WIDE_DEAL 1                      /* the triangle tests of an iteration dealt over the lanes of the wave (0: the flat loop, one record per lane and iteration) */
#endif
#if WIDE_DEAL
/* The flat loop below tests ONE Wald record per lane and iteration: a ray that entered leaves with four triangles stays four iterations
 * before it may take its next node step, while every iteration executes the node block AND the triangle block for whoever needs them
 * (tools/wave_sim.py: 45 of 64 lanes in a node block, 21 in a triangle block).  Here the pending (ray, record) pairs of the whole wave go
 * to a work list in LDS and every lane tests one pair per step, as k_mega does (k_traverse.h: traverseFlat2W): all the triangles a ray has
 * pending are decided in the iteration that found them, the wave needs a fifth fewer iterations for the same rays (simulated), i.e. a
 * fifth fewer executions of both blocks -- on a kernel bound by the number of vector-memory instructions it issues.
 *   - list entry = any-hit flag << 11 | owner lane << 5 | bit of the owner's triangle group; the tester fetches the owner's ray and group
 *     base with ds_bpermute, the record from memory, and runs the same Wald test on the same operands against the owner's CURRENT interval;
 *   - closest hit: LDS min of (bits(t) << 32 | (0x3FFFFFFF - prim) << 2 | class) on the owner's slot -- smallest t, at equal t the highest
 *     triangle index (winsTie), whatever the order; the lane whose key stands in the slot after the step writes (u, v) beside it;
 *     any hit: LDS min of the group bit (the sequential loop stops at the first hit in bit order: the work counter stays what it was);
 *   - the slot is the ray's result: (t, u, v, prim) leave the registers, the owner only pulls its new maxt after a round.
 * The rays' step sequences are unchanged (all records of a group, then the next node), hence so are results and counters. */
__device__ __forceinline__ void persistentTraverseWide(const DevScene &S, WideStackT<WIDE_BLOCK> &stack, ShadowSourceDyn &ss, TraceSourceDyn &ts,
                                                       unsigned long long *wc /* LDS: WC_COUNT counters of this wave */, unsigned char *dealLds /* WD_WAVE_BYTES of this wave */) {
    lds_w64 *slot = (lds_w64 *) dealLds; lds_u2 *uvs = (lds_u2 *) (dealLds + 64u * 8u); lds_w16 *list = (lds_w16 *) (dealLds + 2u * 64u * 8u);
    const uint32_t lane = __lane_id();
    bool active = false, shadow = false;
    uint32_t handle = 0, steps = 0;
    WideRay ray; ray.o = ray.d = ray.rcp = V3(0.0f); ray.mint = ray.maxt = 0; ray.octinv4 = 0;
    uint2 ng = make_uint2(0u, 0u), tg = make_uint2(0u, 0u);
    for (;;) {
        const unsigned long long idle = __ballot(!active);
        const bool moreS = ss.more(), moreAny = moreS || ts.more();              /* wave-uniform */
        if (moreAny) {
            if (__popcll(idle) >= WD_REFILL) {
                const uint32_t h = moreS ? ss.assign(!active, idle) : ts.assign(!active, idle);
                if (!active && h != INVALID_RAY) {
                    V3 o, d; float mint, maxt;                   /* (already clipped to the scene box) */
                    const bool ok = moreS ? ss.load(h, o, d, mint, maxt) : ts.load(h, o, d, mint, maxt);
                    if (ok) {
                        const unsigned long long got = __ballot(1);
                        if (lane == (uint32_t) __ffsll((long long) got) - 1u) wc[moreS ? WW_SH_RAYS : WW_RAYS] += (uint32_t) __popcll(got);
                        TravResult res; res.prim = PHIP_NO_HIT; res.t = INFINITY; res.u = res.v = 0;
                        if (maxt > mint) {
                            wideRaySetup(ray, o, d, V3(slabRcpFast(d.x), slabRcpFast(d.y), slabRcpFast(d.z)), mint, maxt);
                            ng = wideRootGroup(); tg = make_uint2(0u, 0u);
                            stack.sp = 0; handle = h; shadow = moreS; active = true; steps = 0;
                            slot[lane] = ~0ull;
                        } else if (moreS) {
                            ss.commit(h, false, res);
                        } else {
                            ts.commit(h, false, res);
                        }
                    }
                }
            }
        } else if (idle == ~0ull) break;
        if (active && tg.y == 0u && (ng.y & 0xff000000u)) { constexpr bool wideCullOn = true; (void) wideCullOn; WIDE_NODE_STEP(stack, S, ray, ng, tg, steps) }   /* (any-hit rays carry the bound too: their interval never shrinks, so it never culls -- a lane-varying switch costs more) */

        /* ---- the triangle round: every lane takes part ---- */
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
        /* a round costs its instructions whatever the number of pairs: it is held back while few pairs are pending and enough other lanes have node
           steps to take (tools/wave_sim.py, dealt + threshold) */
        const uint32_t nPairs = (uint32_t) __builtin_amdgcn_readlane((int) incl, 63);
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
                const uint32_t item = list[i];                   /* (behind `total`: stale entries, fetched -- every lane must be active in a ds_bpermute, a
                                                                    disabled SOURCE lane reads as zero -- and not tested) */
                const uint32_t owner = (item >> 5) & 63u, bit = item & 31u;
                const bool anyHit = (item & 0x800u) != 0u;
                const int src = (int) (owner << 2);
#define WD_FETCH(x) pm_from_bits((uint32_t) __builtin_amdgcn_ds_bpermute(src, (int) pm_to_bits(x)))
                const V3 po(WD_FETCH(ray.o.x), WD_FETCH(ray.o.y), WD_FETCH(ray.o.z)), pd(WD_FETCH(ray.d.x), WD_FETCH(ray.d.y), WD_FETCH(ray.d.z));
                const float pmint = WD_FETCH(ray.mint), pmaxt = WD_FETCH(ray.maxt);
#undef WD_FETCH
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
                    uint2 e = stack.pop();
                    if (e.y & 0xff000000u) { WIDE_CULL_POP(e, ray) ng = e; } else { tg = e; ng = make_uint2(0u, 0u); }
                }
            }
            if (finished) {
                const unsigned long long best = slot[lane];
                TravResult res; res.prim = PHIP_NO_HIT; res.t = INFINITY; res.u = res.v = 0;
                if (shadow) ss.commit(handle, best != ~0ull, res);
                else {
                    if (best != ~0ull) {
                        const u2v q = uvs[lane];
                        const uint32_t lo = (uint32_t) best;
                        res.t = pm_from_bits((uint32_t) (best >> 32)); res.u = pm_from_bits(q.x); res.v = pm_from_bits(q.y);
                        res.prim = (HIT_PRIM_MASK - (lo >> 2)) | ((lo & 3u) << HIT_CLASS_SHIFT);
                    }
                    ts.commit(handle, false, res);
                }
                /* node steps (low word) and triangle tests (high word) of the ray in ONE 64-bit LDS add */
                atomicAdd(&wc[shadow ? WW_SH_STEPS : WW_STEPS], (unsigned long long) (steps & 0xFFFFu) | ((unsigned long long) (steps >> 16) << 32));
                active = false;
            }
        }
    }
}
#elif !PHIP_EXPERIMENTS
#error "WIDE_DEAL=0 (the flat / nested loops of rounds 2-3) is an experiment build: add -DPHIP_EXPERIMENTS=1"
#elif WIDE_FLAT
/* The loop is flat: every pass tests the refill condition (two scalar instructions on the ballot of the idle lanes) and then runs one
   traversal iteration -- one node step, one triangle test, one pop -- for the lanes that have a ray.  The nested form (an inner loop the
   live lanes stay in until enough of them have finished) made the compiler keep two register images of the lane state, one per loop,
   and copy between them at every entry and exit; flat, the state has one home and the kernel fits 96 VGPRs (5 waves per SIMD)
   without spilling. */
__device__ __forceinline__ void persistentTraverseWide(const DevScene &S, WideStackT<WIDE_BLOCK> &stack, ShadowSourceDyn &ss, TraceSourceDyn &ts,
                                                       unsigned long long *wc /* LDS: WC_COUNT counters of this wave */) {
    bool active = false, shadow = false;
    uint32_t handle = 0, steps = 0;
    WideRay ray; ray.o = ray.d = ray.rcp = V3(0.0f); ray.mint = ray.maxt = 0; ray.octinv4 = 0;
    uint2 ng = make_uint2(0u, 0u), tg = make_uint2(0u, 0u);
    TravResult res; res.prim = PHIP_NO_HIT; res.t = INFINITY; res.u = res.v = 0;
    for (;;) {
        const unsigned long long idle = __ballot(!active);
        const bool moreS = ss.more(), moreAny = moreS || ts.more();              /* wave-uniform */
        if (moreAny) {
            if (__popcll(idle) >= REFILL_LANES) {
                const uint32_t h = moreS ? ss.assign(!active, idle) : ts.assign(!active, idle);
                if (!active && h != INVALID_RAY) {
                    V3 o, d; float mint, maxt;                   /* (already clipped to the scene box) */
                    const bool ok = moreS ? ss.load(h, o, d, mint, maxt) : ts.load(h, o, d, mint, maxt);
                    if (ok) {
                        const unsigned long long got = __ballot(1);
                        if (__lane_id() == (uint32_t) __ffsll((long long) got) - 1u) wc[moreS ? WW_SH_RAYS : WW_RAYS] += (uint32_t) __popcll(got);
                        res.prim = PHIP_NO_HIT; res.t = INFINITY; res.u = res.v = 0;
                        if (maxt > mint) {
                            wideRaySetup(ray, o, d, V3(slabRcpFast(d.x), slabRcpFast(d.y), slabRcpFast(d.z)), mint, maxt);
                            ng = wideRootGroup(); tg = make_uint2(0u, 0u);
                            stack.sp = 0; handle = h; shadow = moreS; active = true; steps = 0;
                        } else if (moreS) {
                            ss.commit(h, false, res);
                        } else {
                            ts.commit(h, false, res);
                        }
                    }
                }
            }
        } else if (idle == ~0ull) break;
        if (active) {
            /* one node step and one triangle test per iteration */
            { constexpr bool wideCullOn = false; (void) wideCullOn;        /* (WIDE_CULL: the dealt loop only -- ADVICE r5: the macro names it in every loop) */
              if (tg.y == 0u && (ng.y & 0xff000000u)) WIDE_NODE_STEP(stack, S, ray, ng, tg, steps) }
            bool finished = false;
            if (tg.y) {
                const uint32_t bit = (uint32_t) __ffs((int) tg.y) - 1u;
                tg.y &= tg.y - 1u;
                WIDE_LOAD_TRI(S, tg.x + bit, a, b, c)
                steps += 0x10000u;
                float tu, tv, tt;
                if (WIDE_WALD_SEL ? waldIntersectSel(a, b, c, ray.o, ray.d, ray.mint, ray.maxt, tu, tv, tt) : waldIntersect(a, b, c, ray.o, ray.d, ray.mint, ray.maxt, tu, tv, tt)) {
                    if (shadow) { res.prim = 0; finished = true; }
                    else if (winsTie(tt, pm_to_bits(c.z), res.t, res.prim & HIT_PRIM_MASK)) {      /* (res.prim is the packed hit word: one register for primitive and class) */
                        ray.maxt = tt; res.t = tt; res.u = tu; res.v = tv; res.prim = pm_to_bits(c.z) | (pm_to_bits(c.w) << HIT_CLASS_SHIFT);
                    }
                }
            }
            if (!finished && tg.y == 0u && !(ng.y & 0xff000000u)) {
                if (stack.sp == 0) finished = true;
                else {
                    const uint2 e = stack.pop();
                    if (e.y & 0xff000000u) ng = e; else { tg = e; ng = make_uint2(0u, 0u); }
                }
            }
            if (finished) {
                if (shadow) ss.commit(handle, res.prim != PHIP_NO_HIT, res);
                else ts.commit(handle, false, res);
                /* node steps (low word) and triangle tests (high word) of the ray in ONE 64-bit LDS add */
                atomicAdd(&wc[shadow ? WW_SH_STEPS : WW_STEPS], (unsigned long long) (steps & 0xFFFFu) | ((unsigned long long) (steps >> 16) << 32));
                active = false;
            }
        }
    }
}
#else
__device__ __forceinline__ void persistentTraverseWide(const DevScene &S, WideStackT<WIDE_BLOCK> &stack, ShadowSourceDyn &ss, TraceSourceDyn &ts,
                                                       unsigned long long *wc /* LDS: WC_COUNT counters of this wave */) {
    bool active = false;
    uint32_t meta = 0, steps = 0;
    WideRay ray; ray.o = ray.d = ray.rcp = V3(0.0f); ray.mint = ray.maxt = 0; ray.octinv4 = 0;
    uint2 ng = make_uint2(0u, 0u), tg = make_uint2(0u, 0u);
    TravResult res; res.prim = PHIP_NO_HIT; res.t = INFINITY; res.u = res.v = 0;

#if WIDE_PROFILE
    /* measurement build: wave clock spent in the refill branch, refills, loop iterations (reported in the rows of the any-hit
       counters, which this build therefore falsifies) */
    unsigned long long pfRefill = 0, pfAssign = 0, pfLoad = 0, pfStart = clock64(); uint32_t pfRefills = 0, pfIters = 0;
#endif
    for (;;) {
        const unsigned long long idle = __ballot(!active);
        const bool moreS = ss.more(), moreAny = moreS || ts.more();              /* wave-uniform */
        if (idle && moreAny && (__popcll(idle) >= REFILL_LANES || idle == ~0ull)) {
#if WIDE_PROFILE
            const unsigned long long pf0 = clock64(); ++pfRefills;
#endif
            const uint32_t h = moreS ? ss.assign(!active, idle) : ts.assign(!active, idle);
#if WIDE_PROFILE
            __builtin_amdgcn_s_waitcnt(0); const unsigned long long pf1 = clock64(); pfAssign += pf1 - pf0;
#endif
            if (!active && h != INVALID_RAY) {
                V3 o, d; float mint, maxt;                   /* (already clipped to the scene box) */
                const bool ok = moreS ? ss.load(h, o, d, mint, maxt) : ts.load(h, o, d, mint, maxt);
#if WIDE_PROFILE
                __builtin_amdgcn_s_waitcnt(0); pfLoad += clock64() - pf1;
#endif
                if (ok) {
                    const unsigned long long got = __ballot(1);
                    if (__lane_id() == (uint32_t) __ffsll((long long) got) - 1u) wc[moreS ? WW_SH_RAYS : WW_RAYS] += (uint32_t) __popcll(got);
                    res.prim = PHIP_NO_HIT; res.t = INFINITY; res.u = res.v = 0;
                    if (maxt > mint) {
                        wideRaySetup(ray, o, d, V3(slabRcpFast(d.x), slabRcpFast(d.y), slabRcpFast(d.z)), mint, maxt);
                        ng = wideRootGroup(); tg = make_uint2(0u, 0u);
                        stack.sp = 0; meta = h | (moreS ? WM_SHADOW : 0u); active = true; steps = 0;
                    } else if (moreS) {
                        ss.commit(h, false, res);
                    } else {
                        ts.commit(h, false, res);
                    }
                }
            }
#if WIDE_PROFILE
            __builtin_amdgcn_s_waitcnt(0); pfRefill += clock64() - pf0;
#endif
        }
        if (!__any(active)) { if (!(ss.more() || ts.more())) break; continue; }
        if (active) {
            for (;;) {
#if WIDE_PROFILE
                if (__builtin_amdgcn_readfirstlane(__lane_id()) == __lane_id()) ++pfIters;
#endif
                /* one node step and one triangle test per iteration */
                { constexpr bool wideCullOn = false; (void) wideCullOn;
                  if (tg.y == 0u && (ng.y & 0xff000000u)) WIDE_NODE_STEP(stack, S, ray, ng, tg, steps) }
                bool finished = false;
                if (tg.y) {
                    const uint32_t bit = (uint32_t) __ffs((int) tg.y) - 1u;
                    tg.y &= tg.y - 1u;
                    WIDE_LOAD_TRI(S, tg.x + bit, a, b, c)
                    steps += 0x10000u;
                    float tu, tv, tt;
                    if (waldIntersect(a, b, c, ray.o, ray.d, ray.mint, ray.maxt, tu, tv, tt)) {
                        if (meta & WM_SHADOW) { res.prim = 0; finished = true; }
                        else if (winsTie(tt, pm_to_bits(c.z), res.t, res.prim & HIT_PRIM_MASK)) {      /* (res.prim is the packed hit word: one register for primitive and class) */
                            ray.maxt = tt; res.t = tt; res.u = tu; res.v = tv; res.prim = pm_to_bits(c.z) | (pm_to_bits(c.w) << HIT_CLASS_SHIFT);
                        }
                    }
                }
                if (!finished && tg.y == 0u && !(ng.y & 0xff000000u)) {
                    if (stack.sp == 0) finished = true;
                    else {
                        const uint2 e = stack.pop();
                        if (e.y & 0xff000000u) ng = e; else { tg = e; ng = make_uint2(0u, 0u); }
                    }
                }
                if (finished) {
                    const bool shadow = (meta & WM_SHADOW) != 0;
                    if (shadow) ss.commit(meta & WM_HANDLE, res.prim != PHIP_NO_HIT, res);
                    else ts.commit(meta & WM_HANDLE, false, res);
                    /* node steps (low word) and triangle tests (high word) of the ray in ONE 64-bit LDS add */
                    atomicAdd(&wc[shadow ? WW_SH_STEPS : WW_STEPS], (unsigned long long) (steps & 0xFFFFu) | ((unsigned long long) (steps >> 16) << 32));
                    active = false;
                    break;
                }
                if ((ss.more() || ts.more()) && __popcll(__ballot(1)) <= 64 - REFILL_LANES) break;     /* enough idle lanes: refill */
            }
        }
    }
#if WIDE_PROFILE
    {
        const unsigned long long tot = clock64() - pfStart;
        uint32_t it = pfIters;
        for (int off = 32; off > 0; off >>= 1) it += __shfl_down(it, off);          /* one lane per iteration counted: the sum is the wave's iterations */
        uint32_t ld = (uint32_t) pfLoad;
        for (int off = 32; off > 0; off >>= 1) { const uint32_t o_ = __shfl_down(ld, off); ld = o_ > ld ? o_ : ld; }      /* the lane that waited longest */
        if (__lane_id() == 0) { wc[WW_SH_STEPS] = (pfRefill & 0xFFFFFFFFull) | (tot << 32); wc[WW_SH_RAYS] = pfRefills; wc[WW_STEPS] = (unsigned long long) (uint32_t) pfAssign | ((unsigned long long) it << 32); wc[WW_RAYS] = ld; }
    }
#endif
}

#endif  /* WIDE_FLAT */

__global__ __launch_bounds__(WIDE_BLOCK, WIDE_WAVES) void k_rays_w(DevScene S, PathPool P, float4 *L, unsigned int *drawCounters /* 2 * RAY_SHARDS lines, zeroed */) {
    __shared__ unsigned long long wcnt[WIDE_BLOCK / 64][WW_COUNT];
    const uint32_t wave = WIDE_UNIFORM(threadIdx.x >> 6), waveId = blockIdx.x * (WIDE_BLOCK / 64) + wave;
    if (threadIdx.x < (WIDE_BLOCK / 64) * WW_COUNT) (&wcnt[0][0])[threadIdx.x] = 0;
    WideStackT<WIDE_BLOCK> stk; setupWide<WIDE_BLOCK>(S, S.wideNodeCache, g_smem, P.spill + (size_t) blockIdx.x * WIDE_BLOCK * SPILL_DEPTH, stk);   /* (barrier inside) */
    const uint32_t nBlk = P.capacity / BLOCK, nChunk = (P.capacity + 63u) / 64u, nWavesGrid = gridDim.x * (WIDE_BLOCK / 64);
    ShadowSourceDyn ss{ P, L, {}, 0u, 0u, 0u };
    TraceSourceDyn ts{ P, {}, 0u, 0u };
    ss.q.init(drawCounters, blockIdx.x % RAY_SHARDS, nBlk, waveId, nWavesGrid);
    ts.q.init(drawCounters + RAY_SHARDS * RAY_SHARD_STRIDE, blockIdx.x % RAY_SHARDS, nChunk, waveId, nWavesGrid);
    ss.start(); ts.start();
#if WIDE_DEAL
    persistentTraverseWide(S, stk, ss, ts, wcnt[wave], g_smem + wideLdsBytes(S.wideNodeCache, WIDE_BLOCK) + wave * WD_WAVE_BYTES);
#else
    persistentTraverseWide(S, stk, ss, ts, wcnt[wave]);
#endif
    if (__lane_id() == 0) {
        const unsigned long long v[6] = { wcnt[wave][WW_RAYS], wcnt[wave][WW_STEPS] & 0xFFFFFFFFull, wcnt[wave][WW_STEPS] >> 32,
                                          wcnt[wave][WW_SH_RAYS], wcnt[wave][WW_SH_STEPS] & 0xFFFFFFFFull, wcnt[wave][WW_SH_STEPS] >> 32 };
        const int rows[6] = { ST_CLOSEST_RAYS, ST_NODE, ST_TRI, ST_SHADOW_RAYS, ST_SH_NODE, ST_SH_TRI };
#pragma unroll
        for (int i = 0; i < 6; ++i)
            if (v[i]) P.stat[(size_t) rows[i] * P.nWaves + waveId] += v[i];
    }
}

/* standalone ray casts for phip_trace on the wide tree */
__global__ __launch_bounds__(BLOCK) void k_raycast_w(DevScene S, const phip_ray *rays, size_t n, phip_hit *hits, uint8_t *occluded, PathPool P) {
    const size_t i = (size_t) blockIdx.x * BLOCK + threadIdx.x;
    WideStack stk; setupWide<BLOCK>(S, wideRaycastCache(S.wideNodeCache), g_smem, P.spill + (size_t) blockIdx.x * BLOCK * SPILL_DEPTH, stk);
    uint32_t nodeVisits = 0, triTests = 0, shNodeVisits = 0, shTriTests = 0;
    if (i < n) {
        const phip_ray ry = rays[i];
        const V3 o(ry.o[0], ry.o[1], ry.o[2]), d(ry.d[0], ry.d[1], ry.d[2]);
        float mint, maxt;
        if (hits) {
            TravResult r; r.prim = PHIP_NO_HIT; r.t = INFINITY; r.u = r.v = 0;
            V3 rcp;
            if (clipToScene<false>(S, o, d, ry.mint, ry.maxt, mint, maxt, rcp))
                traverseWide<false>(S, o, d, rcp, mint, maxt, stk, r, nodeVisits, triTests);
            phip_hit h; h.t = r.t; h.u = r.u; h.v = r.v; h.prim = r.prim;
            hits[i] = h;
        }
        if (occluded) {
            TravResult r; bool occ = false;
            V3 rcp;
            if (clipToScene<true>(S, o, d, ry.mint, ry.maxt, mint, maxt, rcp))
                occ = traverseWide<true>(S, o, d, rcp, mint, maxt, stk, r, shNodeVisits, shTriTests);
            occluded[i] = occ ? 1 : 0;
        }
    }
    const uint32_t waveId = (uint32_t) (i >> 6);
    waveStat(P, ST_NODE, waveId, nodeVisits);
    waveStat(P, ST_TRI, waveId, triTests);
    waveStat(P, ST_SH_NODE, waveId, shNodeVisits);
    waveStat(P, ST_SH_TRI, waveId, shTriTests);
}
