/*
 * k_traverse.h -- per-lane BVH4 traversal (state machine), persistent kernels with refill, closest-hit / any-hit / merged ray kernels, phip_trace
 * Part of the single translation unit phip.hip (included there, in this order: k_pool.h, k_traverse.h,
 * k_group8.h, k_shade.h, k_film.h); see the header of phip.hip for the kernel overview.
 */

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
        if ((P.state[slot] & F_TRACE_MASK) != F_ALIVE) return false;
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
 *  kernels
 * ====================================================================================== */
__global__ __launch_bounds__(BLOCK, TRACE_WAVES) void k_trace(DevScene S, PathPool P) {
    if (P.blockDead[blockIdx.x]) return;
    const uint32_t slot = blockIdx.x * BLOCK + threadIdx.x;
    TravStack stk; setupTraversal(S, g_smem, P.spill + (size_t) slot * SPILL_DEPTH, stk);
    uint32_t nodeVisits = 0, triTests = 0, rays = 0;
    if (slot < P.capacity) {
        if ((P.state[slot] & F_TRACE_MASK) == F_ALIVE) {
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

