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

#ifndef WIDE_STACK_LDS
#define WIDE_STACK_LDS 6                 /* 8-byte entries per lane in LDS (12 KB per block of 256; nine until the triangle rounds of WIDE_DEAL took 6 KB per block: measured the same, 153.2 ms per C3 frame either way) */
#endif
#ifndef WIDE_BLOCK
#define WIDE_BLOCK 256                   /* threads per block of k_rays_w.  Measured (round 2): ONE block of 1024 per CU, whose LDS then holds a single copy of the
                                            top 800 nodes (the first four levels) instead of four copies of 96, changes nothing -- C3 405.7 vs 406.3 Msamples/s,
                                            C4 436 vs 438, and the same again with the cache cut back to 96, 300 or 585 nodes: the node fetches of the upper
                                            levels are not what the kernel waits for (they hit L2; the Wald records come from the Infinity Cache).  Round 3, at 6 waves
                                            per SIMD: blocks of 512 with a 150-node cache / of 768 with 240 nodes: 159.5 / 158.6 ms vs 158.8 (C3) -- still nothing */
#endif
#ifndef WIDE_NODE_CACHE_MAX
#define WIDE_NODE_CACHE_MAX 48           /* top-of-tree nodes (BFS order) staged in LDS by k_rays_w: 3.75 KB per block */
#endif
#define WIDE_NODE_CACHE_RAYCAST 64       /* ... by k_raycast_w (blocks of 256, several per CU) */
#ifndef WIDE_TYPED
#define WIDE_TYPED 1                     /* cached nodes are read with ds_read_b128 (LDS pipe) instead of flat_load (which sends LDS addresses through the
                                            texture addresser / data path the kernel is bound by: TD busy 95 %, round-2 counters) */
#endif
#ifndef WIDE_NODE_STRIDE
#define WIDE_NODE_STRIDE 5               /* uint4 per node in HBM: 5 = packed 80-byte nodes (half of them straddle two 128-byte lines), 8 = one node per 128-byte line */
#endif
#ifndef WIDE_DUMMY_LOADS
#define WIDE_DUMMY_LOADS 0               /* measurement: extra 16-byte loads of the node's own line per node step (L1 hits): what does one more vector-memory instruction cost? */
#endif
#ifndef WIDE_DUMMY_VALU
#define WIDE_DUMMY_VALU 0                /* measurement: extra VALU instructions per node step (independent v_fma_f32 on a scratch register): what does the ALU work cost? */
#endif
#ifndef WIDE_PROFILE
#define WIDE_PROFILE 0
#endif
#ifndef WIDE_CULL
#define WIDE_CULL 0                      /* experiment (round 5): a node group carries, in the 16 free bits of its hit word, the entry distance (rounded down to
                                            bfloat16) of the child that is visited SECOND; when the group is popped and a hit found meanwhile lies in front of it, that child
                                            is skipped without fetching its node.  (The entry distance of the node that pushed the group cannot cull: every hit found between
                                            push and pop lies inside that node.)  ~30 VALU per node step with two or more inner hits, closest-hit rays only. */
#endif
#ifndef WIDE_WAVES
#define WIDE_WAVES 7                     /* waves per SIMD of k_rays_w = blocks of 256 per CU.  Round 3: 74 VGPRs (flat loop, wave-uniform state in SGPRs, stack
                                            addresses rebuilt from the lane index: see persistentTraverseWide), six waves -- measured C3 / C4 at 128 spp, ray-kernel
                                            ms per frame: nested loop at 4 waves (110 VGPRs) 191.8 / 385.4 -- flat loop at 4 waves 188.7 / 376.6 -- 5 waves
                                            169.1 / 343.0 -- 6 waves 158.9 / 327.5 -- 7 waves (72 VGPRs, 9-entry stack, 48-node cache) 167.0 / 336.9 -- 8 waves
                                            (64 VGPRs + 52 B of scratch) 241.0 / 488.3.  Round 4: with the Wald test's axis permutation as selects
                                            (WIDE_WALD_SEL) the kernel needs 64 VGPRs without scratch, and the seventh wave pays (profiles/r04_gpu_call_e_*):
                                            branches, 6 waves 159.3 / 327.5 -- selects, 6 waves 156.1 / 321.6 -- 7 waves (9-entry stack, 48-node cache)
                                            152.6 / 317.1 -- 8 waves (8-entry stack, 32-node cache) 155.6 / 323.5.  Its 91 SGPRs admit 7 blocks per CU (MI355X
                                            guide: 82..96 SGPRs -> 7), 7 x (18 KB stack + 3.75 KB node cache) fit the CU's 160 KB of LDS.
                                            tests/test_kernel_resources.py pins the register counts. */
#endif

typedef uint32_t u2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u2v lds_u2;
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) u4v lds_cu4;

/* The stack keeps ONE per-lane register, sp.  Its addresses -- LDS entry e of thread t at (e * NB + t) * 8, spill entry at
   spillBlock[t * SPILL_DEPTH / 2 + e] -- are rebuilt from the lane index at every push / pop (v_mbcnt, two instructions): the LDS
   base and the 64-bit spill pointer used to be three VGPRs that lived across the whole persistent loop, in a kernel whose
   occupancy is decided by its VGPR count (k_rays_w: WIDE_WAVES).  The asm is volatile so that the compiler does not hoist the
   lane index back out of the loop. */
template <int NB> struct WideStackT {
    lds_u2 *ldsBlock;       /* LDS: the block's stack region (wave-uniform) */
    uint2 *spillBlock;      /* global: the block's spill region, SPILL_DEPTH / 2 entries per lane (the BVH4 kernels' region, reinterpreted) */
    lds_cu4 *nodes;         /* LDS copy of wide nodes [0, nodeCache) */
    uint32_t nodeCache;
    uint32_t waveBase;      /* first thread of this wave in the block (wave-uniform) */
    int sp;
    __device__ __forceinline__ uint32_t tid() const {
        uint32_t l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return waveBase + l;
    }
    __device__ __forceinline__ void push(uint2 v) {
        if (sp < WIDE_STACK_LDS) { u2v t; t.x = v.x; t.y = v.y; ldsBlock[(uint32_t) sp * NB + tid()] = t; }
        else spillBlock[(size_t) tid() * (SPILL_DEPTH / 2) + (uint32_t) (sp - WIDE_STACK_LDS)] = v;
        ++sp;
    }
    __device__ __forceinline__ uint2 pop() {
        --sp;
        if (sp < WIDE_STACK_LDS) { const u2v t = ldsBlock[(uint32_t) sp * NB + tid()]; return make_uint2(t.x, t.y); }
        return spillBlock[(size_t) tid() * (SPILL_DEPTH / 2) + (uint32_t) (sp - WIDE_STACK_LDS)];
    }
};

typedef WideStackT<BLOCK> WideStack;

__host__ __device__ __forceinline__ size_t wideLdsBytes(uint32_t nodeCache, uint32_t blockThreads) {
    return (size_t) WIDE_STACK_LDS * blockThreads * sizeof(uint2) + (size_t) nodeCache * 5 * sizeof(uint4);
}
#ifndef WIDE_DEAL
#define WIDE_DEAL 1
#endif
__host__ __device__ __forceinline__ size_t wideDealBytes(uint32_t blockThreads) { return WIDE_DEAL ? (size_t) (blockThreads / 64u) * (64u * 8u + 64u * 8u + 256u * 2u) : 0; }   /* k_rays_w: WD_WAVE_BYTES per wave */
__host__ __device__ __forceinline__ uint32_t wideRaycastCache(uint32_t nodeCache) { return nodeCache < WIDE_NODE_CACHE_RAYCAST ? nodeCache : WIDE_NODE_CACHE_RAYCAST; }

/* carve the block's dynamic LDS and stage the top of the tree (all threads of the block must call) */
template <int NB> __device__ __forceinline__ void setupWide(const DevScene &S, uint32_t nodeCache, unsigned char *smem, uint32_t *spillBlock /* of this BLOCK's first thread */, WideStackT<NB> &stk) {
    uint2 *stack = (uint2 *) smem;
    uint4 *ln = (uint4 *) (smem + (size_t) WIDE_STACK_LDS * NB * sizeof(uint2));
    for (uint32_t i = threadIdx.x; i < nodeCache * 5u; i += NB) ln[i] = S.wnodes[(i / 5u) * WIDE_NODE_STRIDE + i % 5u];
    __syncthreads();
    stk.ldsBlock = (lds_u2 *) stack; stk.spillBlock = (uint2 *) spillBlock; stk.nodes = (lds_cu4 *) ln; stk.nodeCache = nodeCache; stk.sp = 0;
    stk.waveBase = (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x & ~63u));
}

struct WideRay {
    V3 o, d, rcp;
    float mint, maxt;
    uint32_t octinv4;       /* (7 - octant) replicated into the four bytes; octant bit a = direction component a is negative */
};

DV void wideRaySetup(WideRay &r, const V3 &o, const V3 &d, const V3 &rcp /* the slab reciprocal (clipToScene) */, float mint, float maxt) {
    r.o = o; r.d = d; r.mint = mint; r.maxt = maxt;
    r.rcp = rcp;
    const uint32_t oct = (r.rcp.x < 0 ? 1u : 0u) | (r.rcp.y < 0 ? 2u : 0u) | (r.rcp.z < 0 ? 4u : 0u);
    r.octinv4 = (7u - oct) * 0x01010101u;
}

DV float ubyte(uint32_t v, int k) { return (float) ((v >> (8 * k)) & 0xffu); }     /* v_cvt_f32_ubyte<k> */

/* One node: slab test of the eight quantised child boxes.  Returns the hit bits: 24..31 inner children in traversal order
   (highest bit = first), 0..23 the leaf triangles of the hit leaves. */
/* WIDE_CULL: *second = (priority << 16 | bfloat16(entry distance, rounded down)) of the hit inner child that is visited SECOND (0: fewer than two) -- the two
   largest keys of the eight children, kept with a max and a median per child */
DV uint32_t wideNodeHits(const uint4 &n0, const uint4 &n1, const uint4 &n2, const uint4 &n3, const uint4 &n4, const WideRay &r, uint32_t *second = nullptr) {
    uint32_t key1 = 0, key2 = 0;
    /* child box plane = p + q * 2^(e-127): t = q * (2^e * rcp) + (p - o) * rcp */
    const float sx = pm_from_bits((n0.w & 0xffu) << 23) * r.rcp.x, sy = pm_from_bits(((n0.w >> 8) & 0xffu) << 23) * r.rcp.y,
                sz = pm_from_bits(((n0.w >> 16) & 0xffu) << 23) * r.rcp.z;
    const float bx = (pm_from_bits(n0.x) - r.o.x) * r.rcp.x, by = (pm_from_bits(n0.y) - r.o.y) * r.rcp.y, bz = (pm_from_bits(n0.z) - r.o.z) * r.rcp.z;
    /* near / far planes by the sign of the direction: swap whole dwords (four children each) */
    const bool nx = r.rcp.x < 0, ny = r.rcp.y < 0, nz = r.rcp.z < 0;
    const uint32_t lox[2] = { n2.x, n2.y }, loy[2] = { n2.z, n2.w }, loz[2] = { n3.x, n3.y }, hix[2] = { n3.z, n3.w }, hiy[2] = { n4.x, n4.y }, hiz[2] = { n4.z, n4.w };
    const uint32_t meta[2] = { n1.z, n1.w };
    uint32_t hits = 0;
#if WIDE_DUMMY_VALU && defined(__HIP_DEVICE_COMPILE__)
    { float dv_ = sx;
#pragma unroll
      for (int i_ = 0; i_ < WIDE_DUMMY_VALU; ++i_) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(dv_) : "v"(sy), "v"(sz));
      asm volatile("" :: "v"(dv_)); }
#endif
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (h) __builtin_amdgcn_sched_barrier(0);        /* four children at a time: the two halves interleaved cost more live registers */
#endif
        const uint32_t qnx = nx ? hix[h] : lox[h], qfx = nx ? lox[h] : hix[h];
        const uint32_t qny = ny ? hiy[h] : loy[h], qfy = ny ? loy[h] : hiy[h];
        const uint32_t qnz = nz ? hiz[h] : loz[h], qfz = nz ? loz[h] : hiz[h];
        /* byte-parallel decode of the four meta bytes (CWBVH): inner children (low 5 bits >= 24) get their slot xor-ed with the
           inverted ray octant, which turns "slot" into "traversal priority"; leaves keep their triangle offset */
        const uint32_t m4 = meta[h];
        const uint32_t isInner4 = (m4 & (m4 << 1)) & 0x10101010u;                    /* bit 4 of a byte: bits 3 and 4 both set <=> low5 >= 24 */
        const uint32_t innerMask4 = (isInner4 >> 4) * 0xffu;                         /* 0xff in the bytes of inner children */
        const uint32_t bitIndex4 = (m4 ^ (r.octinv4 & innerMask4)) & 0x1f1f1f1fu;
        const uint32_t childBits4 = (m4 >> 5) & 0x07070707u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float tnx = fmaf(ubyte(qnx, k), sx, bx), tfx = fmaf(ubyte(qfx, k), sx, bx);
            const float tny = fmaf(ubyte(qny, k), sy, by), tfy = fmaf(ubyte(qfy, k), sy, by);
            const float tnz = fmaf(ubyte(qnz, k), sz, bz), tfz = fmaf(ubyte(qfz, k), sz, bz);
            const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, r.mint));
            const float tf = fminf(fminf(tfx, tfy), fminf(tfz, r.maxt));
            const uint32_t bits = (childBits4 >> (8 * k)) & 0xffu, idx = (bitIndex4 >> (8 * k)) & 0xffu;
            hits |= (tn <= tf) ? (bits << idx) : 0u;
#if WIDE_CULL
            if (second) {
                const uint32_t key = (tn <= tf && idx >= 24u) ? ((idx << 16) | (pm_to_bits(tn) >> 16)) : 0u;     /* (tn >= mint >= 0: truncation rounds down) */
                const uint32_t lo = key1 < key ? key1 : key, hi2 = key2 > lo ? key2 : lo;
                key2 = hi2; key1 = key1 > key ? key1 : key;
            }
#endif
        }
    }
#if WIDE_CULL
    if (second) *second = key2;
#endif
    (void) key1; (void) key2;
    return hits;
}

__device__ __forceinline__ uint4 ldsLoadU4(lds_cu4 *p) { const u4v v = *p; return make_uint4(v.x, v.y, v.z, v.w); }
#define WIDE_LOAD_NODE(stack, S, idx, n0, n1, n2, n3, n4)                                             \
    uint4 n0, n1, n2, n3, n4;                                                                         \
    if (WIDE_TYPED && (idx) < (stack).nodeCache) {                                                    \
        lds_cu4 *l_ = (stack).nodes + 5u * (idx);                                                     \
        n0 = ldsLoadU4(l_); n1 = ldsLoadU4(l_ + 1); n2 = ldsLoadU4(l_ + 2); n3 = ldsLoadU4(l_ + 3); n4 = ldsLoadU4(l_ + 4); \
    } else {                                                                                          \
        const uint4 *g_ = (!WIDE_TYPED && (idx) < (stack).nodeCache) ? (const uint4 *) ((stack).nodes + 5u * (idx)) : (S).wnodes + WIDE_NODE_STRIDE * (size_t) (idx); \
        n0 = g_[0]; n1 = g_[1]; n2 = g_[2]; n3 = g_[3]; n4 = g_[4];                                   \
        for (int dl_ = 0; dl_ < WIDE_DUMMY_LOADS; ++dl_) {                                            \
            f4v dv_; asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(dv_) : "v"(g_) : "memory"); asm volatile("" :: "v"(dv_)); \
        }                                                                                             \
    }

/* a Wald record = three 16-byte loads.  Written as inline assembly: the compiler narrows the loads to the eleven dwords in use
   and re-splits them into FOUR instructions (12 + 16 + 16 + 4 bytes), and this kernel is bound by the number of vector-memory
   instructions it issues (texture-data path 95 % busy), not by bytes. */
#define WIDE_LOAD_TRI(S, idx, a, b, c)                                                                \
    float4 a, b, c;                                                                                   \
    {                                                                                                 \
        const float4 *t_ = (S).tris + 3 * (size_t) (idx);                                             \
        f4v va_, vb_, vc_;                                                                            \
        asm volatile("global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %3, off offset:16\n\tglobal_load_dwordx4 %2, %3, off offset:32\n\ts_waitcnt vmcnt(0)" \
                     : "=&v"(va_), "=&v"(vb_), "=&v"(vc_) : "v"(t_) : "memory");                       \
        a = make_float4(va_.x, va_.y, va_.z, va_.w); b = make_float4(vb_.x, vb_.y, vb_.z, vb_.w); c = make_float4(vc_.x, vc_.y, vc_.z, vc_.w); \
    }

/* One node step of a lane whose node group `ng` has inner hits: take the first child in traversal order, push the rest of the
   group, intersect the child node -> new node group and triangle group.  A pending triangle group must be empty. */
#define WIDE_NODE_STEP(stack, S, ray, ng, tg, nodeVisits)                                             \
    {                                                                                                 \
        const uint32_t bit_ = 31u - (uint32_t) __clz((int) (ng).y);                                   \
        (ng).y &= ~(1u << bit_);                                                                      \
        if ((ng).y & 0xff000000u) (stack).push(ng);                                                   \
        const uint32_t slot_ = (bit_ - 24u) ^ ((ray).octinv4 & 7u);                                   \
        const uint32_t idx_ = (ng).x + (uint32_t) __popc((ng).y & ((1u << slot_) - 1u) & 0xffu);      \
        WIDE_LOAD_NODE(stack, S, idx_, n0, n1, n2, n3, n4)                                            \
        ++nodeVisits;                                                                                 \
        uint32_t second_ = 0u;                                                                        \
        const uint32_t hits_ = wideNodeHits(n0, n1, n2, n3, n4, ray, (WIDE_CULL && wideCullOn) ? &second_ : nullptr); \
        (ng) = make_uint2(n1.x, (hits_ & 0xff000000u) | (n0.w >> 24) | ((second_ & 0xffffu) << 8));   \
        (tg) = make_uint2(n1.y, hits_ & 0x00ffffffu);                                                 \
    }
#if WIDE_CULL
/* a popped node group: skip its next child when that child's entry lies behind the closest hit so far; the bound has then served */
#define WIDE_CULL_POP(e, ray)                                                                         \
        {                                                                                             \
            if (pm_from_bits(((e).y & 0x00ffff00u) << 8) > (ray).maxt) (e).y &= ~(0x80000000u >> __clz((int) (e).y)); \
            (e).y &= 0xff0000ffu;                                                                     \
        }
#else
#define WIDE_CULL_POP(e, ray)
#endif

/* the root: node 0 is entered as the only child of a virtual group (child base 0, no inner slots below it: rank 0) */
__device__ __forceinline__ uint2 wideRootGroup() { return make_uint2(0u, 0x80000000u); }

/* per-lane traversal to completion (k_raycast_w) */
template <bool SHADOW>
__device__ __forceinline__ bool traverseWide(const DevScene &S, const V3 &o, const V3 &d, const V3 &rcp, float mint, float maxt,
                                             WideStack &stack, TravResult &res, uint32_t &nodeVisits, uint32_t &triTests) {
    WideRay ray; wideRaySetup(ray, o, d, rcp, mint, maxt);
    stack.sp = 0;
    uint2 ng = wideRootGroup(), tg = make_uint2(0u, 0u);
    bool found = false;
    constexpr bool wideCullOn = false; (void) wideCullOn;        /* (WIDE_CULL: the persistent kernel only) */
    res.prim = PHIP_NO_HIT; res.t = INFINITY; res.u = res.v = 0;
    for (;;) {
        if (tg.y == 0u && (ng.y & 0xff000000u)) WIDE_NODE_STEP(stack, S, ray, ng, tg, nodeVisits)
        if (tg.y) {
            const uint32_t bit = (uint32_t) __ffs((int) tg.y) - 1u;
            tg.y &= tg.y - 1u;
            WIDE_LOAD_TRI(S, tg.x + bit, a, b, c)
            ++triTests;
            float tu, tv, tt;
            if (waldIntersect(a, b, c, o, d, ray.mint, ray.maxt, tu, tv, tt)) {
                if (SHADOW) return true;
                if (winsTie(tt, pm_to_bits(c.z), res.t, res.prim)) { ray.maxt = tt; res.t = tt; res.u = tu; res.v = tv; res.prim = pm_to_bits(c.z); }
                found = true;
            }
        }
        if (tg.y == 0u && !(ng.y & 0xff000000u)) {
            if (stack.sp == 0) break;
            const uint2 e = stack.pop();
            if (e.y & 0xff000000u) ng = e; else { tg = e; ng = make_uint2(0u, 0u); }
        }
    }
    return found;
}

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

/* the slab reciprocal of a pre-clipped ray: v_rcp_f32 (1 ulp) instead of the IEEE division the clip made in the shading kernel --
   the slab tests only have to be conservative (boxes are quantised outwards and padded by 2e-6 of the scene extent, twenty times the
   error this adds to a plane distance), the hit itself is decided by the Wald test on (o, d, mint', maxt') */
__device__ __forceinline__ float slabRcpFast(float d) { return slabRcpFrom(d, __builtin_amdgcn_rcpf(d)); }

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
#ifndef WD_THRESHOLD
#define WD_THRESHOLD 32u                 /* pairs that must be pending before a round runs (0: every iteration that has any).  Ray kernel, C3 at 64 spp / C4 at 128 spp:
                                            flat loop 153.4 / 318.5 ms -- dealt, threshold 0: 148.3 / 306.7 -- 24: 143.6 / 293.4 -- 40: 143.7 / 292.9 -- 56: 149.0 / 302.2 */
#endif
#ifndef WD_REFILL
#define WD_REFILL 8                      /* idle lanes at which the wave fetches new rays (the flat loop: REFILL_LANES = 16; here 8 / 16 / 24 measured 140.5 / 142.0 / 149.0 ms per C3 frame) */
#endif
#define WD_CAP 256u                      /* list entries per wave (a multiple of 64); lanes whose pairs do not fit wait for the next iteration */
#define WD_WAVE_BYTES (64u * 8u + 64u * 8u + WD_CAP * 2u)
typedef __attribute__((address_space(3))) uint16_t lds_w16;
typedef __attribute__((address_space(3))) unsigned long long lds_w64;
typedef __attribute__((address_space(3))) uint32_t lds_w32;
#define WD_SYNC() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
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
            if (tg.y == 0u && (ng.y & 0xff000000u)) WIDE_NODE_STEP(stack, S, ray, ng, tg, steps)
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
                if (tg.y == 0u && (ng.y & 0xff000000u)) WIDE_NODE_STEP(stack, S, ray, ng, tg, steps)
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
