/*
 * k_traverse.h -- per-lane BVH4 traversal as a state machine: scene-box clip, LDS-staged stack and tree cache, node step, traverse<>
 * Included by phip.hip (before k_rays.h, the ray kernels) and phip_mega.hip (traverse<> inside the fused kernel); see the header of phip.hip.
 */

/* ======================================================================================
 *  BVH traversal (closest / any hit)
 * ====================================================================================== */
struct TravResult { float t, u, v; uint32_t prim; uint32_t cls = 0; /* shade class of the record hit (k_rays_w only: k_pool.h) */ };

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
        if (sp < depth) lds[sp * BLOCK] = v;
        else { if (!spill) __builtin_trap(); spill[sp - depth] = v; }     /* (no spill region: the host's depth bound was wrong -- stop, do not write out of bounds) */
        ++sp;
    }
    __device__ __forceinline__ uint32_t pop() {
        --sp;
        return sp < depth ? lds[sp * BLOCK] : spill[sp - depth];
    }
    __device__ __forceinline__ uint32_t popLds() { --sp; return lds[sp * BLOCK]; }    /* the caller knows that nothing spilled */
};

/* k_mega<MM_ALL>: the region [0, 12 KB) of the dynamic LDS, over the traversal stack (phip.hip sizes it) -- MEGA_CLASS_DEAL (QMC build): dwords per lane and exchange round;
   MEGA_MAILBOX: the four waves' work lists (6 KB), then the S-box */
#define MEGA_DEAL_DWORDS 12u
/* bytes of dynamic LDS setupTraversal() uses; k_mega appends its shading tables (megaLdsBytesOf) */
__host__ __device__ __forceinline__ size_t traversalLdsBytesOf(const DevScene &S) {
    return (size_t) S.stackDepth * BLOCK * sizeof(uint32_t) + (size_t) S.nodeCache * NODE_LDS_STRIDE * sizeof(float4) + (size_t) S.triCache * 3 * sizeof(float4);
}
__host__ __device__ __forceinline__ size_t megaLdsBytesOf(const DevScene &S) {
    return traversalLdsBytesOf(S) + (size_t) S.nTriangles * TRISHADE_FLOAT4S * sizeof(float4) + (size_t) ((S.emitterTabSize + 3u) & ~3u) * sizeof(float)
         + (size_t) S.nMaterials * sizeof(DevMaterial) + 16 /* alignment of the next array */ + (size_t) S.nFlatLeaves * 2 * sizeof(float4);
}

/* k_shade_trace (k_shade_trace.h), dynamic LDS of a block: the packed leaf table, the Wald records, the emitter table and the materials -- sized for THIS scene (the static 8.5 KB of k_shade's
   two tables would cost a block of occupancy; the work lists of the dealt traversals live in the static exchange buffer) */
__host__ __device__ __forceinline__ size_t shadeTraceLdsBytes(const DevScene &S) {
    return (size_t) S.nFlatLeaves * 2 * sizeof(float4) + (size_t) S.triCache * 3 * sizeof(float4) + (size_t) ((S.emitterTabSize + 3u) & ~3u) * sizeof(float)
         + (size_t) S.nMaterials * sizeof(DevMaterial);
}

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

/* ALL_LDS (a constant in the scope of the caller): the whole tree, every record and the whole stack are in LDS (k_mega) --
   no residency test, no spill path.
   TYPED (a constant in the scope of the caller): true = separate LDS (ds_read_b128) and global paths -- right when (almost)
   everything is cached (small scenes); false = one flat_load path with a selected address -- fewer registers and no
   divergence when most lanes read global memory (big scenes; measured 1-3 % faster there, 14 % slower on the Cornell box) */
#define LOAD_NODE(stack, S, cur, mnx, mny, mnz, mxx, mxy, mxz, chf)                                   \
    float4 mnx, mny, mnz, mxx, mxy, mxz, chf;                                                         \
    if (TYPED) {                                                                                      \
        if (ALL_LDS || (uint32_t) (cur) < (stack).nodeCache) {                                        \
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
        if (ALL_LDS || (uint32_t) (idx) < (stack).triCache) {                                         \
            lds_cf4 *t_ = (stack).tris + 3 * (uint32_t) (idx); a = ldsLoad4(t_); b = ldsLoad4(t_ + 1); c = ldsLoad4(t_ + 2); \
        } else {                                                                                      \
            const float4 *t_ = (S).tris + 3 * (size_t) (idx); a = t_[0]; b = t_[1]; c = t_[2];        \
        }                                                                                             \
    } else {                                                                                          \
        const float4 *t_ = (uint32_t) (idx) < (stack).triCache ? (const float4 *) ((stack).tris + 3 * (uint32_t) (idx)) : (S).tris + 3 * (size_t) (idx); \
        a = t_[0]; b = t_[1]; c = t_[2];                                                              \
    }
#define SPILL_DEPTH 96
static_assert(SPILL_DEPTH == 96, "k_pool.h: spillOf");


__device__ __forceinline__ void cswap(float &ka, uint32_t &ra, float &kb, uint32_t &rb) {
    const bool sw = kb < ka;
    const float k0 = sw ? kb : ka, k1 = sw ? ka : kb;
    const uint32_t r0 = sw ? rb : ra, r1 = sw ? ra : rb;
    ka = k0; kb = k1; ra = r0; rb = r1;
}

#define SHADOW_UNSORTED 1
#ifndef MEGA_UNSORTED
#define MEGA_UNSORTED 1          /* k_mega (whole tree in LDS, 7 nodes on the Cornell box): closest-hit rays visit the children unsorted too -- the sorting network costs more than the culling it buys: 106.9 -> 103.7 ms per C2 frame (the answer does not depend on the order: winsTie) */
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
            if (ALL_LDS || stack.sp + 3 <= stack.depth) {      /* branch-free pushes: hits are a prefix of the sorted keys */ \
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
            cur = stack.sp == 0 ? DONE_REF : (int32_t) (ALL_LDS ? stack.popLds() : stack.pop());          \
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
            if (ALL_LDS || stack.sp + 3 <= stack.depth) {                                                 \
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
            cur = stack.sp == 0 ? DONE_REF : (int32_t) (ALL_LDS ? stack.popLds() : stack.pop());          \
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
template <bool SHADOW, bool ALL_LDS = false>
__device__ __forceinline__ bool traverse(const DevScene &S, const V3 &o, const V3 &d, const V3 &rcp /* clipToScene's slab reciprocal */, float mint, float maxt,
                                         TravStack &stack, TravResult &res,
                                         uint32_t &nodeVisits, uint32_t &triTests) {
    constexpr bool TYPED = true;
    /* the slab tests use the reciprocal direction (conservative: boxes are padded); the Wald test uses o,d */
    const V3 ordr(o.x * rcp.x, o.y * rcp.y, o.z * rcp.z);
    stack.sp = 0;
    int32_t cur = S.rootRef;
    bool found = false;
    res.prim = PHIP_NO_HIT; res.t = INFINITY; res.u = res.v = 0;
    while (cur != DONE_REF) {
        if (cur >= 0) {
            if ((SHADOW && SHADOW_UNSORTED) || (ALL_LDS && MEGA_UNSORTED)) NODE_STEP_ANY(stack, S, cur, rcp, ordr, mint, maxt, nodeVisits)
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
                if (winsTie(tt, pm_to_bits(c.z), res.t, res.prim)) { maxt = tt; res.t = tt; res.u = tu; res.v = tv; res.prim = pm_to_bits(c.z); }
                found = true;
            }
            cur = left ? (int32_t) ~(((idx + 1u) << 3) | (left - 1u)) : (stack.sp == 0 ? DONE_REF : (int32_t) (ALL_LDS ? stack.popLds() : stack.pop()));
        }
    }
    return found;
}

/* Trees of at most FLAT_LEAVES_MAX leaves in LDS (k_mega; the Cornell box has 17 leaves over 7 nodes): no walk at all.
 *   pass 1, uniform: every lane tests the SAME leaf box per step (the table entry is one LDS broadcast) -> a bit mask of the leaves
 *           its ray enters; 16 VALU per leaf at full lane utilisation instead of 3.3 divergent node steps of ~85;
 *   pass 2, per lane: the Wald records of the leaves in the mask, one test per iteration (the loop body is the triangle block alone,
 *           not node step + triangle block).
 * Leaves are visited in table order, not front to back: with the tie rule (winsTie) the answer does not depend on it; a leaf whose
 * box lies behind a hit found earlier is still tested (its records fail the t <= maxt test). */
template <bool SHADOW>
__device__ __forceinline__ bool traverseFlat(const DevScene &S, lds_cf4 *flat, uint32_t nFlat, const V3 &o, const V3 &d, const V3 &rcp, float mint, float maxt,
                                             TravStack &stack, TravResult &res, uint32_t &nodeVisits, uint32_t &triTests) {
    constexpr bool ALL_LDS = true, TYPED = true;
    const V3 ordr(o.x * rcp.x, o.y * rcp.y, o.z * rcp.z);
    uint32_t mask = 0;
    for (uint32_t c = 0; c < nFlat; ++c) {
        const float4 mn = ldsLoad4(flat + 2 * c), mx = ldsLoad4(flat + 2 * c + 1);      /* (scalar loads from the kernel argument instead: 87.6 vs 80.4 ms per C2 frame) */
        const float x0 = fmaf(mn.x, rcp.x, -ordr.x), x1 = fmaf(mx.x, rcp.x, -ordr.x);
        const float y0 = fmaf(mn.y, rcp.y, -ordr.y), y1 = fmaf(mx.y, rcp.y, -ordr.y);
        const float z0 = fmaf(mn.z, rcp.z, -ordr.z), z1 = fmaf(mx.z, rcp.z, -ordr.z);
        const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), mint));
        const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), maxt));
        mask |= (tn <= tf) ? (1u << c) : 0u;
    }
    ++nodeVisits;
    bool found = false;
    res.prim = PHIP_NO_HIT; res.t = INFINITY; res.u = res.v = 0;
    int32_t cur = DONE_REF;
    for (;;) {
        if (cur == DONE_REF) {
            if (mask == 0) break;
            const uint32_t c = (uint32_t) __ffs((int) mask) - 1u;
            mask &= mask - 1u;
            cur = (int32_t) pm_to_bits(ldsLoad4(flat + 2 * c).w);
        }
        /* a leaf reference doubles as the lane's progress inside the leaf: ~((next record << 3) | records left - 1) */
        const uint32_t r = ~(uint32_t) cur, idx = r >> 3, left = r & 7u;
        LOAD_TRI(stack, S, idx, a, b, c)
        ++triTests;
        float tu, tv, tt;
        if (waldIntersect(a, b, c, o, d, mint, maxt, tu, tv, tt)) {
            if (SHADOW) return true;
            if (winsTie(tt, pm_to_bits(c.z), res.t, res.prim)) { maxt = tt; res.t = tt; res.u = tu; res.v = tv; res.prim = pm_to_bits(c.z); }
            found = true;
        }
        cur = left ? (int32_t) ~(((idx + 1u) << 3) | (left - 1u)) : DONE_REF;
    }
    return found;
}

/* Second form of the flat table, for trees of at most 32 Wald records (the Cornell box: 32 triangles in 17 leaves): entry c =
 *   A = (min.x, max.x, min.y, max.y)   B = (min.z, max.z, bits(mask of the leaf's records), 0)
 * so that (a) the two planes of an axis are one packed multiply-add (v_pk_fma_f32), and (b) pass 1 yields a bit mask of RECORDS: pass 2
 * is `while (mask) test record ffs(mask)` -- no leaf reference to fetch and decode between records, a record referenced by two leaves is
 * tested once.  Pass 1 takes the entries four at a time (eight LDS broadcasts in flight instead of a wait per leaf), the rest one by one.  The Wald test is branch-free here (waldIntersectSel: the axis permutation as twelve
 * selects instead of three divergent branches -- inside traverseFlat the exec-mask bookkeeping was 40 scalar instructions per record).
 * Same arithmetic on the same operands, so (t, u, v, prim) are the same bits; records in index order instead of leaf order: winsTie. */
#ifndef MEGA_WALD_PAIR
#define MEGA_WALD_PAIR 0
#endif
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bool waldIntersectSel(const float4 &a, const float4 &b, const float4 &c, const V3 &o, const V3 &d,
                                                 float mint, float maxt, float &u, float &v, float &t) {
    const uint32_t k = pm_to_bits(a.x);
    const bool k0 = k == 0, k2 = k == 2;                       /* k == 1: the default of the selects; k == 3 (degenerate record) fails below */
    const float o_u = k0 ? o.y : (k2 ? o.x : o.z), o_v = k0 ? o.z : (k2 ? o.y : o.x), o_k = k0 ? o.x : (k2 ? o.z : o.y);
    const float d_u = k0 ? d.y : (k2 ? d.x : d.z), d_v = k0 ? d.z : (k2 ? d.y : d.x), d_k = k0 ? d.x : (k2 ? d.z : d.y);
    const float n_u = a.y, n_v = a.z, n_d = a.w;
    t = (n_d - o_u * n_u - o_v * n_v - o_k) / (d_u * n_u + d_v * n_v + d_k);
    const float hu = o_u + t * d_u - b.x;
    const float hv = o_v + t * d_v - b.y;
    u = hv * b.z + hu * b.w;
    v = hu * c.x + hv * c.y;
    /* waldIntersect: `if (t < mint || t > maxt) return false; ... return u >= 0 && v >= 0 && u + v <= 1` (a NaN t fails through u) */
    return (k < 3u) & !(t < mint) & !(t > maxt) & (u >= 0) & (v >= 0) & (u + v <= 1.0f);
}


/* Pass 1 of the packed flat table (DevScene::flatMode 2): the bit mask of the Wald records whose leaf box the ray enters.  Two forms of the
 * table, a compile-time choice shared with the host code that packs it (phip.hip):
 *   MEGA_FLAT_CH = 0: entry = (min.x, max.x, min.y, max.y) (min.z, max.z, bits(records), 0): an axis is one v_pk_fma_f32, then the pair is
 *      ordered with a min and a max -- 15.3 VALU per box;
 *   MEGA_FLAT_CH = 1: entry = (c.x, c.y, c.z, 0) (h.x, h.y, h.z, bits(records)), centre and half extent: with c' = c * rcp - o * rcp the slab
 *      distances of an axis are c' -+ h * |rcp| -- ALREADY ordered, one v_pk_fma_f32 whose source modifiers negate h for the low half and
 *      replicate h, |rcp| and c' into both halves (inline assembly: the compiler does not form them) -- 11.3 VALU per box.  The two forms
 *      round differently; the test only has to be conservative, and the host pads h for it (phip.hip). */
#ifndef MEGA_FLAT_CH
#define MEGA_FLAT_CH 1
#endif
#ifndef MEGA_BALANCE
#define MEGA_BALANCE 1               /* k_mega, packed flat table: the Wald tests of a traversal are dealt over the lanes of the wave (traverseFlat2W below) instead of looping per lane */
#endif
/* R64 (DevScene::flatMode 3, trees of 33..64 Wald records): the record mask has a second word, kept in the centre's spare word (A.w) */
template <bool R64 = false>
__device__ __forceinline__ uint32_t flat2Pass1(lds_cf4 *flat, uint32_t nFlat, const V3 &o, const V3 &rcp, float mint, float maxt, uint32_t *maskHi = nullptr) {
    uint32_t mask = 0, hi = 0;
#if MEGA_FLAT_CH
    const f2v rxy = { rcp.x, rcp.y }, oxy = { -(o.x * rcp.x), -(o.y * rcp.y) };
    const f2v rz2 = { rcp.z, 0.0f }, oz2 = { -(o.z * rcp.z), 0.0f };
    f2v axy = { fabsf(rcp.x), fabsf(rcp.y) }, az2 = { fabsf(rcp.z), 0.0f };
    asm("" : "+v"(axy), "+v"(az2));                   /* (opaque: otherwise the two v_and are rematerialised inside the loop) */
    /* the orderings in assembly as well: on values that come out of inline assembly fmaxf / fminf first canonicalise every operand (v_max_f32 x, x, x) */
#define FLAT2_BOX(c_)                                                                                                                  \
        {                                                                                                                              \
            const f4v A = flat[2 * (c_)], B = flat[2 * (c_) + 1];                                                                      \
            const f2v cxy = __builtin_elementwise_fma(A.xy, rxy, oxy), cz = __builtin_elementwise_fma(A.zw, rz2, oz2);                 \
            f2v tx, ty, tz;                           /* (near, far) = (c' - h |r|, c' + h |r|) */                                      \
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,0,0] neg_lo:[1,0,0]" : "=v"(tx) : "v"(B.xy), "v"(axy), "v"(cxy));            \
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,1] op_sel_hi:[1,1,1] neg_lo:[1,0,0]" : "=v"(ty) : "v"(B.xy), "v"(axy), "v"(cxy)); \
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,0,0] neg_lo:[1,0,0]" : "=v"(tz) : "v"(B.zw), "v"(az2), "v"(cz));             \
            float tn, tf;                                                                                                              \
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tn) : "v"(tx.x), "v"(ty.x), "v"(mint));                                              \
            asm("v_max_f32 %0, %1, %2" : "=v"(tn) : "v"(tn), "v"(tz.x));                                                               \
            asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tf) : "v"(tx.y), "v"(ty.y), "v"(maxt));                                              \
            asm("v_min_f32 %0, %1, %2" : "=v"(tf) : "v"(tf), "v"(tz.y));                                                               \
            mask |= (tn <= tf) ? pm_to_bits(B.w) : 0u;                                                                                 \
            if (R64) hi |= (tn <= tf) ? pm_to_bits(A.w) : 0u;                                                                          \
        }
#else
    const f2v rx = { rcp.x, rcp.x }, ry = { rcp.y, rcp.y }, rz = { rcp.z, rcp.z };
    const f2v ox = { -(o.x * rcp.x), -(o.x * rcp.x) }, oy = { -(o.y * rcp.y), -(o.y * rcp.y) }, oz = { -(o.z * rcp.z), -(o.z * rcp.z) };
#define FLAT2_BOX(c_)                                                                                                                  \
        {                                                                                                                              \
            const f4v A = flat[2 * (c_)], B = flat[2 * (c_) + 1];                                                                      \
            const f2v x = __builtin_elementwise_fma(A.xy, rx, ox), y = __builtin_elementwise_fma(A.zw, ry, oy), z = __builtin_elementwise_fma(B.xy, rz, oz); \
            const float tn = fmaxf(fmaxf(fminf(x.x, x.y), fminf(y.x, y.y)), fmaxf(fminf(z.x, z.y), mint));                             \
            const float tf = fminf(fminf(fmaxf(x.x, x.y), fmaxf(y.x, y.y)), fminf(fmaxf(z.x, z.y), maxt));                             \
            mask |= (tn <= tf) ? pm_to_bits(B.z) : 0u;                                                                                 \
        }
#endif
    /* groups of four entries (eight LDS broadcasts in flight), then the rest one by one: the Cornell box has 17 leaves -- padded to 20 it paid for three
       boxes no ray can enter, 15 % of a pass that is a third of the traversal */
    const uint32_t nFlat4 = nFlat & ~3u;
    for (uint32_t c4 = 0; c4 < nFlat4; c4 += 4) {
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) FLAT2_BOX(c4 + j)
    }
    for (uint32_t c = nFlat4; c < nFlat; ++c) FLAT2_BOX(c)
#undef FLAT2_BOX
    if (R64) *maskHi = hi;
    return mask;
}

template <bool SHADOW>
__device__ __forceinline__ bool traverseFlat2(lds_cf4 *flat, uint32_t nFlat, lds_cf4 *tris, const V3 &o, const V3 &d, const V3 &rcp,
                                              float mint, float maxt, TravResult &res, uint32_t &nodeVisits, uint32_t &triTests) {
    uint32_t mask = flat2Pass1<false>(flat, nFlat, o, rcp, mint, maxt);
    ++nodeVisits;
    bool found = false;
    res.prim = PHIP_NO_HIT; res.t = INFINITY; res.u = res.v = 0;
#define FLAT2_TEST(a, b, c)                                                                                        \
    {                                                                                                              \
        ++triTests;                                                                                                \
        float tu, tv, tt;                                                                                          \
        const bool hit = waldIntersectSel(a, b, c, o, d, mint, maxt, tu, tv, tt);                                  \
        if (SHADOW) {                                                                                              \
            found = found | hit;                                                                                   \
            mask = hit ? 0u : mask;                                                                                \
        } else {                                                                                                   \
            const bool win = hit & winsTie(tt, pm_to_bits(c.z), res.t, res.prim);                                  \
            maxt = win ? tt : maxt; res.t = win ? tt : res.t; res.u = win ? tu : res.u; res.v = win ? tv : res.v;  \
            res.prim = win ? pm_to_bits(c.z) : res.prim;                                                           \
            found = found | hit;                                                                                   \
        }                                                                                                          \
    }
#if MEGA_WALD_PAIR
    /* two records per pass of the loop: two independent dependency chains (each Wald test is ~35 dependent instructions behind an LDS round
       trip, and four waves per SIMD do not cover that).  The second record is judged after the first, against the interval the first may have
       shortened -- the sequential loop's decisions in the sequential loop's order.  A lane with one record left tests it twice. */
    while (mask) {
        const uint32_t i0 = (uint32_t) __builtin_ctz(mask);
        mask &= mask - 1u;
        const bool two = mask != 0;
        const uint32_t i1 = two ? (uint32_t) __builtin_ctz(mask) : i0;
        mask &= mask - 1u;
        const float4 a0 = ldsLoad4(tris + 3 * i0), b0 = ldsLoad4(tris + 3 * i0 + 1), c0 = ldsLoad4(tris + 3 * i0 + 2);
        const float4 a1 = ldsLoad4(tris + 3 * i1), b1 = ldsLoad4(tris + 3 * i1 + 1), c1 = ldsLoad4(tris + 3 * i1 + 2);
        float u0, v0, t0, u1, v1, t1;
        const bool h0 = waldIntersectSel(a0, b0, c0, o, d, mint, maxt, u0, v0, t0);
        bool h1 = two & waldIntersectSel(a1, b1, c1, o, d, mint, maxt, u1, v1, t1);
        if (SHADOW) {
            triTests += (two & !h0) ? 2u : 1u;                   /* the sequential loop stops at the first hit */
            found = found | h0 | h1;
            mask = (h0 | h1) ? 0u : mask;
        } else {
            triTests += two ? 2u : 1u;
            const bool w0 = h0 & winsTie(t0, pm_to_bits(c0.z), res.t, res.prim);
            maxt = w0 ? t0 : maxt; res.t = w0 ? t0 : res.t; res.u = w0 ? u0 : res.u; res.v = w0 ? v0 : res.v; res.prim = w0 ? pm_to_bits(c0.z) : res.prim;
            h1 = h1 & !(t1 > maxt);
            const bool w1 = h1 & winsTie(t1, pm_to_bits(c1.z), res.t, res.prim);
            maxt = w1 ? t1 : maxt; res.t = w1 ? t1 : res.t; res.u = w1 ? u1 : res.u; res.v = w1 ? v1 : res.v; res.prim = w1 ? pm_to_bits(c1.z) : res.prim;
            found = found | h0 | h1;
        }
    }
#else
    while (mask) {
        const uint32_t idx = (uint32_t) __builtin_ctz(mask);
        mask &= mask - 1u;
        lds_cf4 *t_ = tris + 3 * idx;
        const float4 a = ldsLoad4(t_), b = ldsLoad4(t_ + 1), c = ldsLoad4(t_ + 2);
        FLAT2_TEST(a, b, c)
    }
#endif
#undef FLAT2_TEST
    return found;
}

/* traverseFlat2 with pass 2 DEALT OVER THE WAVE (k_mega, MEGA_BALANCE).  The per-lane loop above runs for the wave's slowest lane -- about
 * eight Wald tests where the average ray needs 3.1, and in the shadow phase 30 of 64 lanes have no ray at all.  Here the (ray, record) pairs
 * of the whole wave are written to a work list in LDS and every lane tests one pair per step, whoever the ray belongs to:
 *   1. a prefix sum of popcount(mask) over the wave (six DPP steps) gives every lane its segment of the list; it writes (lane << 5 | record)
 *      per set bit -- a loop of the slowest lane's length, but of seven instructions, not seventy;
 *   2. ceil(pairs / 64) steps: entry -> the owner's ray through ds_bpermute (eight dwords), the record from LDS, the same Wald test on the
 *      same operands, against the owner's ORIGINAL interval (the sequential loop's shrinking maxt only rejects candidates that lose anyway);
 *      closest hit: LDS min of (bits(t) << 32 | (0x7FFFFFF - prim) << 5 | record) on the owner's slot -- the smallest t, at equal t the
 *      highest triangle index: winsTie, independent of the order; shadow ray: LDS min of the record index (the sequential loop stops at the
 *      first hit in index order: the work counter stays what it was);
 *   3. the owner reads its slot and repeats the winning test with its own registers for (t, u, v): same operands, same bits.
 * A work list of BAL_CAP pairs; a wave with more (never seen on the Cornell box: 64 x 3.1) goes round again with the lanes that did not fit.
 * Every lane of the wave must call, converged; `go` = this lane has a ray.  The buffers lie over the traversal stack (unused by the flat table). */
#define BAL_CAP 512u
#ifndef BAL_ILP
#define BAL_ILP 1                        /* pairs per lane and step of the test loop (BAL_CAP is a multiple of 64 * BAL_ILP).  Measured: 1 / 2 / 4 = 53.0 / 56.0 / 59.8 ms
                                            per C2 frame (profiles/r04_gpu_call_m_*): the list's tail is rounded up to whole steps, and wider steps waste more tests than their
                                            interleaving hides */
#endif
#define BAL_WAVE_BYTES (64u * 8u + BAL_CAP * 2u)                          /* slots, work list */
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
struct WaveBalance { lds_u64 *slot; lds_u16 *list; };
__device__ __forceinline__ WaveBalance waveBalanceAt(unsigned char *smem, uint32_t waveInBlock) {
    unsigned char *p = smem + waveInBlock * BAL_WAVE_BYTES;
    WaveBalance wb; wb.slot = (lds_u64 *) p; wb.list = (lds_u16 *) (p + 64u * 8u);
    return wb;
}
#define BAL_SYNC() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }    /* DS operations of a wave execute in order: this only pins the compiler's order */

/* R64: up to 64 records (a two-word mask; list entry = lane << 6 | record, key = ... (0x3FFFFFF - prim) << 6 | record) -- round 5: a scene of 33..64 records
   (a Cornell box with a third block) used to leave the dealt traversal for the per-lane leaf table */
template <bool R64> struct RecMask;
template <> struct RecMask<false> { typedef uint32_t T; static __device__ __forceinline__ uint32_t popc(T m) { return (uint32_t) __popc(m); }
                                    static __device__ __forceinline__ uint32_t ctz(T m) { return (uint32_t) __builtin_ctz(m); }
                                    static __device__ __forceinline__ T upTo(uint32_t i) { return (2u << i) - 1u; } };
template <> struct RecMask<true>  { typedef unsigned long long T; static __device__ __forceinline__ uint32_t popc(T m) { return (uint32_t) __popcll(m); }
                                    static __device__ __forceinline__ uint32_t ctz(T m) { return (uint32_t) __builtin_ctzll(m); }
                                    static __device__ __forceinline__ T upTo(uint32_t i) { return i >= 63u ? ~0ull : (2ull << i) - 1ull; } };
template <bool SHADOW, bool R64 = false>
__device__ __forceinline__ bool traverseFlat2W(lds_cf4 *flat, uint32_t nFlat, lds_cf4 *tris, const WaveBalance &wb, uint32_t lane, bool go,
                                               const V3 &o, const V3 &d, const V3 &rcp, float mint, float maxt, TravResult &res, uint32_t &nodeVisits, uint32_t &triTests) {
    typedef RecMask<R64> RM; typedef typename RM::T Mask;
    constexpr uint32_t RB = R64 ? 6u : 5u, RMSK = (1u << RB) - 1u;     /* bits of a record index */
    uint32_t maskHi = 0;
    const uint32_t maskLo = flat2Pass1<R64>(flat, nFlat, o, rcp, mint, maxt, &maskHi);
    Mask mask = R64 ? (Mask) (((unsigned long long) maskHi << 32) | maskLo) : (Mask) maskLo;
    mask = go ? mask : (Mask) 0;                                 /* (a lane without a ray ran pass 1 on whatever its registers held) */
    nodeVisits += go ? 1u : 0u;
    const Mask mask0 = mask;

    if (SHADOW) *(lds_u32 *) (wb.slot + lane) = 0xFFFFFFFFu; else wb.slot[lane] = ~0ull;
    while (__ballot(mask != 0)) {
        /* list segments in lane order: an inclusive scan of the pair counts over the wave (every lane is active here: plain DPP, the
           sequence the compiler itself emits for wave-aggregated atomics -- four shifts inside the rows of 16, then two row broadcasts) */
        const uint32_t pc = RM::popc(mask);
        uint32_t incl = pc;
        incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
        incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x112 /* row_shr:2 */, 0xf, 0xf, true);
        incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x114 /* row_shr:4 */, 0xf, 0xf, true);
        incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x118 /* row_shr:8 */, 0xf, 0xf, true);
        incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
        incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
        /* the lanes whose segment ends inside the list (a prefix of the wave, never empty: a lane has at most 64 pairs) write it and are done */
        const bool fits = incl <= BAL_CAP;
        const uint32_t nFit = (uint32_t) __popcll(__ballot(fits));
        const uint32_t total = (uint32_t) __builtin_amdgcn_readlane((int) incl, (int) (nFit - 1u));
        if (pc && fits) {
            const uint32_t tag = lane << RB;
            lds_u16 *w = wb.list + (incl - pc);
            do {
                *w++ = (uint16_t) (tag | RM::ctz(mask));
                mask &= mask - 1u;
            } while (mask);
        }
        BAL_SYNC()
        /* one pair per lane and step (BAL_ILP of them interleaved measured slower, see above) */
        for (uint32_t base = 0; base < total; base += 64u * BAL_ILP) {
            bool hit[BAL_ILP]; uint32_t own[BAL_ILP], lo[BAL_ILP], hi[BAL_ILP];
#pragma unroll
            for (uint32_t j = 0; j < BAL_ILP; ++j) {             /* the tests, free of control flow so that the compiler interleaves them ... */
                const uint32_t i = base + 64u * j + lane;
                const uint32_t item = wb.list[i];                /* (entries behind `total` hold stale pairs: tested, not committed) */
                const uint32_t owner = (item >> RB) & 63u, rec = item & RMSK;
                const int src = (int) (owner << 2);
                const V3 po(pm_from_bits((uint32_t) __builtin_amdgcn_ds_bpermute(src, (int) pm_to_bits(o.x))), pm_from_bits((uint32_t) __builtin_amdgcn_ds_bpermute(src, (int) pm_to_bits(o.y))),
                            pm_from_bits((uint32_t) __builtin_amdgcn_ds_bpermute(src, (int) pm_to_bits(o.z))));
                const V3 pd(pm_from_bits((uint32_t) __builtin_amdgcn_ds_bpermute(src, (int) pm_to_bits(d.x))), pm_from_bits((uint32_t) __builtin_amdgcn_ds_bpermute(src, (int) pm_to_bits(d.y))),
                            pm_from_bits((uint32_t) __builtin_amdgcn_ds_bpermute(src, (int) pm_to_bits(d.z))));
                const float pmint = pm_from_bits((uint32_t) __builtin_amdgcn_ds_bpermute(src, (int) pm_to_bits(mint)));
                const float pmaxt = pm_from_bits((uint32_t) __builtin_amdgcn_ds_bpermute(src, (int) pm_to_bits(maxt)));
                lds_cf4 *t_ = tris + 3 * rec;
                const float4 a = ldsLoad4(t_), b = ldsLoad4(t_ + 1), c = ldsLoad4(t_ + 2);
                float tu, tv, tt;
                hit[j] = (i < total) & waldIntersectSel(a, b, c, po, pd, pmint, pmaxt, tu, tv, tt);
                own[j] = owner; hi[j] = pm_to_bits(tt);
                lo[j] = SHADOW ? rec : ((((0xFFFFFFFFu >> RB) - pm_to_bits(c.z)) << RB) | rec);
            }
#pragma unroll
            for (uint32_t j = 0; j < BAL_ILP; ++j)               /* ... then the commits */
                if (hit[j]) {
                    if (SHADOW) __hip_atomic_fetch_min((lds_u32 *) (wb.slot + own[j]), lo[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else __hip_atomic_fetch_min(wb.slot + own[j], ((unsigned long long) hi[j] << 32) | (unsigned long long) lo[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
        }
        BAL_SYNC()
    }
    res.prim = PHIP_NO_HIT; res.t = INFINITY; res.u = res.v = 0;
    if (SHADOW) {
        const uint32_t first = *(lds_u32 *) (wb.slot + lane);
        const bool found = first != 0xFFFFFFFFu;
        triTests += RM::popc(found ? (Mask) (mask0 & RM::upTo(first & RMSK)) : mask0);
        return found;
    } else {
        const unsigned long long best = wb.slot[lane];
        const bool found = best != ~0ull;
        triTests += RM::popc(mask0);
        lds_cf4 *t_ = tris + 3 * ((uint32_t) best & RMSK);
        const float4 a = ldsLoad4(t_), b = ldsLoad4(t_ + 1), c = ldsLoad4(t_ + 2);
        /* t is the key's high word (the tester's quotient, bit for bit); (u, v) follow from it as in the Wald test -- no division, no o_k / d_k selects */
        const float tt = pm_from_bits((uint32_t) (best >> 32));
        const uint32_t k = pm_to_bits(a.x);
        const bool k0 = k == 0, k2 = k == 2;
        const float o_u = k0 ? o.y : (k2 ? o.x : o.z), o_v = k0 ? o.z : (k2 ? o.y : o.x);
        const float d_u = k0 ? d.y : (k2 ? d.x : d.z), d_v = k0 ? d.z : (k2 ? d.y : d.x);
        const float hu = o_u + tt * d_u - b.x, hv = o_v + tt * d_v - b.y;
        const float tu = hv * b.z + hu * b.w, tv = hu * c.x + hv * c.y;
        res.t = found ? tt : res.t; res.u = found ? tu : res.u; res.v = found ? tv : res.v; res.prim = found ? pm_to_bits(c.z) : res.prim;
        res.cls = found ? (pm_to_bits(c.w) & 3u) : 0u;           /* the record's shade class (k_shade_trace passes it on in the hit word: k_pool.h) */
        return found;
    }
}

/* the block's dynamic LDS: traversal stack + node / record cache (setupTraversal) */
extern __shared__ __attribute__((aligned(16))) unsigned char g_smem[];
