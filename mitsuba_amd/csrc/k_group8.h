/*
 * k_group8.h -- 8 lanes per ray over the BVH8 (documented negative result, PHIP_TRAVERSAL=group)
 * Part of the single translation unit phip.hip (included there, in this order: k_pool.h, k_traverse.h,
 * k_group8.h, k_shade.h, k_film.h); see the header of phip.hip for the kernel overview.
 */

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
            if ((P.state[slot] & F_TRACE_MASK) != F_ALIVE) return false;
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

