/*
 * k_shade_trace.h -- k_shade_trace: one path vertex AND its two rays in one kernel, for scenes whose tree is the fused kernel's packed leaf
 * table (at most 64 Wald records in at most 64 leaves: DevScene::flatMode 2 / 3) but whose materials / emitters / textures keep them off
 * k_mega (which exists for diffuse scenes only: with the microfacet code next to the traversal it needs more than 256 VGPRs).
 *
 * Round 5: such a scene -- a Cornell box with a glass and a copper block -- ran k_shade -> k_shadow_p -> k_trace per
 * iteration: three launches that hand ~350 B per vertex to each other through HBM (ray out / ray in, shadow entry out / in, L[id]
 * read-modify-write, hit out / in) around a BVH4 walk of a tree that is 2 KB of LDS; measured on the all-diffuse box: 1720 Msamples/s
 * against the fused kernel's 4745.  Here a lane shades its slot's vertex (shadeVertex: the same statement), traces the vertex's shadow ray,
 * starts the next camera sample if the path ended, and traces the next ray -- both traversals with the Wald tests dealt over the wave
 * (k_traverse.h: traverseFlat2W) on tables staged in LDS.  Per vertex HBM sees the slot state once in and once out (~170 B) and the sample's
 * accumulator when it changes; the ray kernels and the shadow queue are not used at all.  Every slot is still shaded by the same code in the
 * same order, the shadow ray's contribution joins L[id] after the vertex's own terms as k_shadow_p's read-modify-write did: same bits.
 */
#ifndef SHADE_TRACE_WAVES
#define SHADE_TRACE_WAVES 5              /* 96 VGPRs (4: 99..110).  Mixed Cornell box, 4 M slots: 1643 -> 1759 Msamples/s (profiles/r05_gpu_call_e_*) */
#endif

/* Radiance policy: the vertex's additions are held back until its shadow ray is decided, then written once (LGlobal wrote L[id] in k_shade and k_shadow_p
   re-read it; a load after a store of the same thread in one kernel would also have to get past the L1's write-through) */
struct LPending {
    float4 *L; const PathPool &P; uint32_t slot; float4 &pend; bool &have;
    __device__ __forceinline__ float4 load(uint32_t id) const { return have ? pend : L[id]; }
    __device__ __forceinline__ void store(uint32_t, const float4 &l) const { pend = l; have = true; }
    __device__ __forceinline__ float4 rayO(const PathVertex &) const { return P.rayO[slot]; }
    __device__ __forceinline__ uint64_t seqIdx(const RenderConst &rc, const PathVertex &v, uint32_t width) const { return seqIndex(rc, v.k, v.pixel % width, v.pixel / width); }
};

template <int MM, bool STRICT, int FEAT> __global__ __launch_bounds__(BLOCK, SHADE_TRACE_WAVES) void k_shade_trace(DevScene S, PathPool P, RenderConst rc, float4 *L) {
    __shared__ uint32_t waveCnt[BLOCK / 64];
    /* one buffer, two uses: the class deal's slot exchange at the head of the block (scenes with more than one BSDF model), then -- behind a barrier -- the
       slots and work lists of the dealt traversals */
    __shared__ __align__(16) unsigned char xbuf[MM != 0 ? SHADE_DEAL_BYTES : (BLOCK / 64) * BAL_WAVE_BYTES];
    static_assert(SHADE_DEAL_BYTES >= (BLOCK / 64) * BAL_WAVE_BYTES, "the traversal buffers lie over the exchange buffer");
    __shared__ uint32_t clsCnt[4][BLOCK / 64];
    if (rc.draining && P.blockDead[blockIdx.x]) return;         /* (block-uniform) */
    uint32_t slot = blockIdx.x * BLOCK + threadIdx.x;            /* (the pool's capacity is a multiple of BLOCK: every lane has a slot) */
    bool inRange = slot < P.capacity;
    const uint32_t lslot = inRange ? slot : 0u;
    uint4 info = P.info[lslot];
    info.w = P.state[lslot];
    PathVertex v;
    v.hit = P.hit[lslot];
    v.rayD = P.rayD[lslot];
    v.thr = P.thr[lslot];
    v.mis = P.mis[lslot];
    /* dynamic LDS: [the packed leaf table][the Wald records][emitter table][materials] (the host checked that both tables fit: phip.hip, flatTrace) */
    float4 *ldsFlat = (float4 *) g_smem, *ldsTris = ldsFlat + 2u * S.nFlatLeaves;
    float *ldsEm = (float *) (ldsTris + 3u * S.triCache);
    DevMaterial *ldsMat = (DevMaterial *) (ldsEm + ((S.emitterTabSize + 3u) & ~3u));
    ShadeTables tab = stageShadeTables(S, ldsEm, ldsMat);
    tab.T.t = ldsEm; tab.materials = ldsMat;
    for (uint32_t i = threadIdx.x; i < 2u * S.nFlatLeaves; i += BLOCK) ldsFlat[i] = S.flatLeaves[i];
    for (uint32_t i = threadIdx.x; i < 3u * S.triCache; i += BLOCK) ldsTris[i] = S.tris[i];
    lds_cf4 *flat = (lds_cf4 *) ldsFlat, *tris = (lds_cf4 *) ldsTris;
    const uint32_t lane = __lane_id();
    const WaveBalance wb = waveBalanceAt(xbuf, (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6)));
    /* the lane deal by BSDF model (k_shade.h: dealSlotsByClass): the class is what THIS kernel left in the hit word when it traced the ray */
    if (MM != 0 && SHADE_SORT && S.shadeSort) dealSlotsByClass(xbuf, clsCnt, P, info, v, slot, inRange);
    v.hit.w = pm_from_bits(hitPrim(pm_to_bits(v.hit.w)));
    if (!inRange) info = make_uint4(0, 0, 0, 0);
    __syncthreads();                                            /* LDS tables are complete */
    const bool alive = inRange && (info.w & F_ALIVE);
    bool needNew = inRange && !alive && !(info.w & F_DEAD);
    unsigned long long vertices = 0, done = 0;
    bool pushShadow = false, newRay = false;
    ShadowEntry sh; sh.e0 = make_float4(0, 0, 0, 0); sh.e1 = sh.e0; sh.e2 = sh.e0;
    float4 pend = make_float4(0, 0, 0, 0); bool havePend = false;

    if (alive) {
        v.id = info.x; v.pixel = info.y; v.k = info.z; v.state = info.w;
        uint32_t nv = 0;
        const LPending acc{ L, P, slot, pend, havePend };
        if (shadeVertex<MM, STRICT, FEAT>(S, tab.T, tab.materials, rc, v, acc, newRay, pushShadow, sh, nv)) {
            vertices = nv; done = 1;
            needNew = true;
        } else {
            info.w = v.state;
            P.state[slot] = info.w;
        }
        if (newRay) { P.rayO[slot] = v.rayO; P.rayD[slot] = v.rayD; P.thr[slot] = v.thr; P.mis[slot] = v.mis; }
    }

    /* ---- the vertex's shadow ray (path.cpp:187-199): every lane takes part in the dealt traversal ---- */
    uint32_t shNode = 0, shTri = 0;
    {
        const V3 o(sh.e0.x, sh.e0.y, sh.e0.z), d(sh.e1.x, sh.e1.y, sh.e1.z);
        float mint, maxt; V3 rcp; TravResult r;
        const bool go = pushShadow & clipToSceneSel<true>(S, o, d, PT_EPSILON, sh.e0.w, mint, maxt, rcp);
        const bool occluded = traverseFlat2W<true, true>(flat, S.nFlatLeaves, tris, wb, lane, go, o, d, rcp, mint, maxt, r, shNode, shTri);
        if (pushShadow && !occluded) {
            if (!havePend) pend = L[pm_to_bits(sh.e2.w)];
            pend.x += sh.e2.x; pend.y += sh.e2.y; pend.z += sh.e2.z;
            havePend = true;
        }
    }
    if (havePend) L[v.id] = pend;

    /* ---- regeneration (shadeEpilogue: static schedule + dynamic tail), then the next ray of every live slot ---- */
    float4 ro = v.rayO, rd = v.rayD;
    bool nowAlive = false;
    shadeEpilogue<(FEAT & 8) != 0>(S, P, rc, waveCnt, slot, inRange, info, alive, needNew, false, sh.e0, sh.e1, sh.e2, vertices, done, &ro, &rd, &nowAlive);
    uint32_t nNode = 0, nTri = 0;
    {
        const V3 o(ro.x, ro.y, ro.z), d(rd.x, rd.y, rd.z);
        float mint, maxt; V3 rcp; TravResult r;
        const bool go = nowAlive & clipToSceneSel<false>(S, o, d, ro.w, rd.w, mint, maxt, rcp);
        traverseFlat2W<false, true>(flat, S.nFlatLeaves, tris, wb, lane, go, o, d, rcp, mint, maxt, r, nNode, nTri);
        if (nowAlive) P.hit[slot] = make_float4(r.t, r.u, r.v, pm_from_bits(r.prim == PHIP_NO_HIT ? r.prim : (r.prim | (r.cls << HIT_CLASS_SHIFT))));
    }
    const uint32_t waveId = (blockIdx.x * BLOCK + threadIdx.x) >> 6;
    waveStat(P, ST_CLOSEST_RAYS, waveId, nowAlive ? 1ull : 0ull);
    waveStat(P, ST_NODE, waveId, nNode);
    waveStat(P, ST_TRI, waveId, nTri);
    waveStat(P, ST_SHADOW_RAYS, waveId, pushShadow ? 1ull : 0ull);
    waveStat(P, ST_SH_NODE, waveId, shNode);
    waveStat(P, ST_SH_TRI, waveId, shTri);
}
