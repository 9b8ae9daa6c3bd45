/*
 * k_rays.h -- the ray kernels of the wavefront path: persistent waves with refill (k_trace_p / k_shadow_p / k_rays_p, which casts
 * the closest-hit and the any-hit rays of an iteration in one launch), one lane per slot (k_trace / k_shadow, small scenes),
 * k_raycast (phip_trace).  Included by phip.hip after k_traverse.h; see the header of phip.hip.
 */

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

template <bool SHADOW, bool TYPED, typename Source>
__device__ __forceinline__ void persistentTraverse(const DevScene &S, TravStack &stack, Source &src,
                                                   uint32_t &nodeVisits, uint32_t &triTests, uint32_t &raysTraced) {
    constexpr bool ALL_LDS = false;
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
                    if (clipToScene<SHADOW>(S, o, d, rmint, rmaxt, mint, maxt, rcp)) {
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
                        else if (winsTie(tt, pm_to_bits(c.z), res.t, res.prim)) { maxt = tt; res.t = tt; res.u = tu; res.v = tv; res.prim = pm_to_bits(c.z); }
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
        if ((P.state[slot] & F_TRACE_MASK) != F_ALIVE) return false;
        const float4 ro = P.rayO[slot], rd = P.rayD[slot];
        o = V3(ro.x, ro.y, ro.z); d = V3(rd.x, rd.y, rd.z); mint = ro.w; maxt = rd.w;
        return true;
    }
    __device__ __forceinline__ void commit(uint32_t slot, bool, const TravResult &r) const {
        P.hit[slot] = make_float4(r.t, r.u, r.v, pm_from_bits(r.prim == PHIP_NO_HIT ? r.prim : (r.prim | (r.cls << HIT_CLASS_SHIFT))));
    }
};

/* L[id] += c for an unoccluded NEE entry (no other lane touches L[id] during the ray kernels).  Measured alternatives,
 * both reverted (HISTORY.md 3.4): three fire-and-forget float atomics (+1..3 % on the big scenes, but the Cornell shadow
 * kernel doubled), a per-slot accumulator flushed once per sample (k_shade then pays for it).  Scenes that fit LDS avoid
 * the read-modify-write altogether: k_mega keeps the accumulator in a register. */
__device__ __forceinline__ void addRadiance(float4 *L, uint32_t id, const float4 &c) {
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
            const float4 e2 = P.shadow[3 * (size_t) e + 2];
            addRadiance(L, pm_to_bits(e2.w), e2);
        }
    }
};

#if PHIP_EXPERIMENTS      /* the BVH4 ray kernels of rounds 1-2 for big trees (the product walks the compressed wide tree: k_wide.h); A/B builds only */
#ifndef TRACE_P_WAVES
#define TRACE_P_WAVES 5
#endif

/* ---- closest-hit AND any-hit rays of one iteration in ONE persistent launch ----
 * The two ray kinds of an iteration are independent (k_shade consumes both results in the next iteration), so a wave
 * first drains its share of the shadow queue and then, without a kernel boundary, refills idle lanes from its share
 * of the closest-hit queue: one kernel tail (waves waiting for the slowest in-flight rays) and one launch per
 * iteration instead of two.  The kind of a lane's ray is a per-lane flag; the loop body is shared. */

enum { WC_RAYS = 0, WC_NODE, WC_TRI, WC_SH_RAYS, WC_SH_NODE, WC_SH_TRI, WC_COUNT };

__device__ __forceinline__ void persistentTraverseMixed(const DevScene &S, TravStack &stack, ShadowSource &ss, TraceSource &ts,
                                                        uint32_t *wc /* LDS: WC_COUNT counters of this wave */) {
    constexpr bool TYPED = false, ALL_LDS = false;
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
                    if (clipToSceneRT(S, o, d, rmint, rmaxt, mint, maxt, moreS, rcp)) {
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
                        else if (winsTie(tt, pm_to_bits(c.z), res.t, res.prim)) { maxt = tt; res.t = tt; res.u = tu; res.v = tv; res.prim = pm_to_bits(c.z); }
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
    TravStack stk; setupTraversal(S, g_smem, spillOf(P, (size_t) blockIdx.x * BLOCK + threadIdx.x), stk);   /* (barrier inside) */
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
    TravStack stk; setupTraversal(S, g_smem, spillOf(P, (size_t) blockIdx.x * BLOCK + threadIdx.x), stk);
    TraceSource src{ P, waveId, 0u, nWavesGrid, (P.capacity + 63u) / 64u };
    uint32_t nodeVisits = 0, triTests = 0, rays = 0;
    persistentTraverse<false, TYPED>(S, stk, src, nodeVisits, triTests, rays);
    waveStat(P, ST_CLOSEST_RAYS, waveId, rays);
    waveStat(P, ST_NODE, waveId, nodeVisits);
    waveStat(P, ST_TRI, waveId, triTests);
}

/* (round 6: k_shadow_p, k_trace and k_raycast below -- the BVH4 ray kernels of the small scenes -- left the product as well: every scene has the compressed wide tree on
   the device, and k_rays_w / k_raycast_w are the wavefront path's and phip_trace's only ray kernels) */

__global__ __launch_bounds__(BLOCK, TRACE_WAVES) void k_shadow_p(DevScene S, PathPool P, float4 *L) {
    const uint32_t waveId = (blockIdx.x * BLOCK + threadIdx.x) >> 6, nWavesGrid = gridDim.x * (BLOCK / 64);
    TravStack stk; setupTraversal(S, g_smem, spillOf(P, (size_t) blockIdx.x * BLOCK + threadIdx.x), stk);
    ShadowSource src{ P, L, waveId, 0u, 0u, nWavesGrid, P.capacity / BLOCK };
    src.skipEmpty();
    uint32_t nodeVisits = 0, triTests = 0, rays = 0;
    persistentTraverse<true, true>(S, stk, src, nodeVisits, triTests, rays);      /* k_shadow_p serves the small scenes (big ones use k_rays_p) */
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
    TravStack stk; setupTraversal(S, g_smem, spillOf(P, slot), stk);
    uint32_t nodeVisits = 0, triTests = 0, rays = 0;
    if (slot < P.capacity) {
        if ((P.state[slot] & F_TRACE_MASK) == F_ALIVE) {
            const float4 ro = P.rayO[slot], rd = P.rayD[slot];
            const V3 o(ro.x, ro.y, ro.z), d(rd.x, rd.y, rd.z);
            float mint, maxt;
            TravResult r; r.prim = PHIP_NO_HIT; r.t = INFINITY; r.u = r.v = 0;
            rays = 1;
            V3 rcp;
            if (clipToScene<false>(S, o, d, ro.w, rd.w, mint, maxt, rcp))
                traverse<false>(S, o, d, rcp, mint, maxt, stk, r, nodeVisits, triTests);
            P.hit[slot] = make_float4(r.t, r.u, r.v, pm_from_bits(r.prim));
        }
    }
    const uint32_t waveId = (blockIdx.x * BLOCK + threadIdx.x) >> 6;
    waveStat(P, ST_CLOSEST_RAYS, waveId, rays);
    waveStat(P, ST_NODE, waveId, nodeVisits);
    waveStat(P, ST_TRI, waveId, triTests);
}

/* one lane per shadow-queue entry (PHIP_TRAVERSAL=lane) */
__global__ __launch_bounds__(BLOCK, TRACE_WAVES) void k_shadow(DevScene S, PathPool P, float4 *L) {
    if (P.blockDead[blockIdx.x]) return;
    TravStack stk; setupTraversal(S, g_smem, spillOf(P, (size_t) blockIdx.x * BLOCK + threadIdx.x), stk);
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
        V3 rcp;
        if (clipToScene<true>(S, o, d, PT_EPSILON, e0.w, mint, maxt, rcp))
            occluded = traverse<true>(S, o, d, rcp, mint, maxt, stk, r, nodeVisits, triTests);
        if (!occluded) {
            addRadiance(L, pm_to_bits(e2.w), e2);
        }
    }
    if ((threadIdx.x & ~63u) < n) {                          /* waves without entries have nothing to add */
        const uint32_t waveId = (blockIdx.x * BLOCK + threadIdx.x) >> 6;
        waveStat(P, ST_SHADOW_RAYS, waveId, rays);
        waveStat(P, ST_SH_NODE, waveId, nodeVisits);
        waveStat(P, ST_SH_TRI, waveId, triTests);
    }
}


/* standalone ray casts for phip_trace */
__global__ __launch_bounds__(BLOCK) void k_raycast(DevScene S, const phip_ray *rays, size_t n, phip_hit *hits, uint8_t *occluded, PathPool P) {
    const size_t i = (size_t) blockIdx.x * BLOCK + threadIdx.x;
    TravStack stk; setupTraversal(S, g_smem, spillOf(P, i), stk);
    uint32_t nodeVisits = 0, triTests = 0, shNodeVisits = 0, shTriTests = 0;
    if (i < n) {
        const phip_ray ry = rays[i];
        const V3 o(ry.o[0], ry.o[1], ry.o[2]), d(ry.d[0], ry.d[1], ry.d[2]);
        float mint, maxt;
        if (hits) {
            TravResult r; r.prim = PHIP_NO_HIT; r.t = INFINITY; r.u = r.v = 0;
            V3 rcp;
            if (clipToScene<false>(S, o, d, ry.mint, ry.maxt, mint, maxt, rcp))
                traverse<false>(S, o, d, rcp, mint, maxt, stk, r, nodeVisits, triTests);
            phip_hit h; h.t = r.t; h.u = r.u; h.v = r.v; h.prim = r.prim;
            hits[i] = h;
        }
        if (occluded) {
            TravResult r; bool occ = false;
            V3 rcp;
            if (clipToScene<true>(S, o, d, ry.mint, ry.maxt, mint, maxt, rcp))
                occ = traverse<true>(S, o, d, rcp, mint, maxt, stk, r, shNodeVisits, shTriTests);
            occluded[i] = occ ? 1 : 0;
        }
    }
    const uint32_t waveId = (uint32_t) (i >> 6);
    waveStat(P, ST_NODE, waveId, nodeVisits);
    waveStat(P, ST_TRI, waveId, triTests);
    waveStat(P, ST_SH_NODE, waveId, shNodeVisits);
    waveStat(P, ST_SH_TRI, waveId, shTriTests);
}
#endif  /* PHIP_EXPERIMENTS */
