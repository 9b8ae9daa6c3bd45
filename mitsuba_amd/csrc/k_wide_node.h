/*
 * k_wide_node.h -- the compressed 8-wide BVH (bvh.h: buildWide) as the kernels see it: the group stack, one node step, one Wald record, the per-lane
 * traversal loop, and the constants of the dealt triangle rounds.  Shared by k_wide.h (k_rays_w / k_raycast_w: phip.hip) and k_wide_wave.h (the fused
 * kernel's traversal of a tree that lives in L2: phip_mega.hip); split from k_wide.h in round 6 so that both units compile the same statements.
 */
#pragma once

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
        const float4 *t_ = (S).wtris + 3 * (size_t) (idx);                                            \
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

/* ---- the dealt triangle rounds (k_wide.h: persistentTraverseWide; k_wide_wave.h: traceWideW): thresholds and the per-wave LDS buffers ---- */
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

/* the slab reciprocal of a pre-clipped ray: v_rcp_f32 (1 ulp) instead of the IEEE division the clip made in the shading kernel --
   the slab tests only have to be conservative (boxes are quantised outwards and padded by 2e-6 of the scene extent, twenty times the
   error this adds to a plane distance), the hit itself is decided by the Wald test on (o, d, mint', maxt') */
__device__ __forceinline__ float slabRcpFast(float d) { return slabRcpFrom(d, __builtin_amdgcn_rcpf(d)); }
