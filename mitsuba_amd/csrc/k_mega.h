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
#ifndef MEGA_MB_DIAG
#define MEGA_MB_DIAG 0
#endif
#ifndef MEGA_MB_FAULT
#define MEGA_MB_FAULT 0              /* fault injection (tests/test_gpu_dropin.py builds it): the first wave of the grid reports that it gave up -- the host must then re-render the pass on the
                                        wavefront kernels and deliver the same frame (phip.hip) */
#endif
#ifndef MEGA_MAILBOX_QMC
#define MEGA_MAILBOX_QMC 1           /* the mailboxes in the QMC builds of k_mega<MM_ALL> as well (0: they keep the class deal, as in round 5) */
#endif
#ifndef MEGA_POOL
#define MEGA_POOL 1                  /* FLAT >= 4 (the tree in memory): the wave's rays are traversed through ONE shared stack of node visits, any lane takes any ray's (k_wide_wave.h:
                                        traceWidePool); 0: every lane walks its own ray (traceWideW, with MEGA_JOINT) */
#endif
#ifndef MEGA_JOINT
#define MEGA_JOINT 1                 /* FLAT >= 4 (the tree in memory): the shadow ray of a vertex and the next ray of its path share ONE traversal phase (k_wide_wave.h) */
#endif
#ifndef MEGA_MAILBOX
#define MEGA_MAILBOX 1               /* MM != 0, counter stream (round 5's last step; the QMC build keeps MEGA_CLASS_DEAL: its static LDS leaves no room): wave 0 of the block SERVES the
                                        rough-conductor vertices.  The other waves (clients) hand a path that hit copper to a 64-entry mailbox in LDS (S-box: each client owns a
                                        third of it) and start another path; the server takes the vertices out when MEGA_MB_THRESH of them wait (or some have waited for
                                        MEGA_MB_PATIENCE of its passes), shades them, traces their shadow rays and hands the continued paths back through a second mailbox (R-box),
                                        from which the clients fill their free lanes before they take camera samples.  Between batches the server starts camera samples 64 at a
                                        time and hands their continuations over as well: it holds no path between passes.  No block barrier: the microfacet code runs for
                                        near-full waves, once per batch instead of once per pass, and nobody waits for the wave that runs it.  Mixed Cornell box 2427 -> 3070
                                        Msamples/s, copper block only 2694 -> 3484, glass block only 3360 -> 3479, every sample bit-identical (profiles/r05_gpu_call_mb_*).
                                        Glass vertices stay with the clients (MEGA_MB_CLASSES bit 1): batching them too is slower than not batching at all (2378) */
#endif
#ifndef MEGA_MB_CLASSES
#define MEGA_MB_CLASSES 1            /* shade classes the clients hand over: bit 0 rough conductor, bit 1 dielectric */
#endif
#ifndef MEGA_MB_THRESH
#define MEGA_MB_THRESH 40            /* 24 .. 48 and patience 3 .. 10 measure the same (81.6 - 82.9 ms per frame) */
#endif
#ifndef MEGA_MB_PATIENCE
#define MEGA_MB_PATIENCE 6
#endif
/* MB_NS = 64 entries of the S-box (dynamic LDS, behind the work lists; k_pool.h) and MB_NR of the R-box (static: what four blocks per CU leave) */
#define MB_NR 48u
/* MB_DW = 24 (k_pool.h): dwords per mailbox entry (S-box: hit 4, direction 3, throughput 4, MIS 2, id, pixel, k, state, accumulator 4 = 21; R-box: origin + mint 4,
                                        direction + maxt 4 instead of hit and direction = 22) */
static_assert(MEGA_DEAL_DWORDS * sizeof(uint32_t) == WIDE_STACK_LDS * sizeof(uint2), "FLAT >= 4: the class deal's exchange buffer lies over the group stack");
static_assert(MEGA_DEAL_DWORDS * BLOCK * sizeof(uint32_t) <= (BLOCK / 64u) * WP_WAVE_BYTES, "FLAT >= 4, MEGA_POOL: ... over the waves' round buffers (slots, ray table, pair list: free between traversals)");
static_assert((BLOCK / 64u) * BAL_WAVE_BYTES + MB_DW * MB_NS * sizeof(uint32_t) <= MEGA_DEAL_DWORDS * BLOCK * sizeof(uint32_t), "the S-box lies behind the work lists in the region phip.hip sizes with MEGA_DEAL_DWORDS");
#define MEGA_CHUNK_MAX 4096u
#define MEGA_CHUNK_MIN 64u

enum { MC_SAMPLES = 0, MC_VERTICES, MC_RAYS, MC_NODE, MC_TRI, MC_SH_RAYS, MC_SH_NODE, MC_SH_TRI, MC_COUNT };

template <int MM, bool STRICT, int FLAT /* 0: BVH4 walk, 1: flat leaf table (traverseFlat), 2: packed table + record masks (traverseFlat2), 3: the same with 33..64 records (two-word masks; MEGA_BALANCE only);
                                           round 6 -- 4: the compressed 8-wide tree in L2 / HBM (k_wide_wave.h: traceWideW), emitter table and materials in LDS, 5: the same with the materials in memory */,
          bool QMC /* the reference's sobol / halton / hammersley / stratified streams (FEAT bit 3 of shadeVertex) */,
          bool DIRECT = false /* round 6: MIDirectIntegrator::Li (k_shade_direct.h: directVertex) instead of the path tracer's vertex -- a lane owns a CAMERA SAMPLE through its
                                 emitter and BSDF sampling rounds; the loop, the traversals and the camera-sample queue are the same */> __global__ __launch_bounds__(BLOCK, MEGA_WAVES) void k_mega(DevScene S, MegaParams M, RenderConst rc, float4 *L) {
    constexpr bool WIDE = FLAT >= 4;                            /* the tree, its Wald records and the shading records stay in memory: a lane still owns its path from the camera sample to its last vertex */
    /* (the QMC build's static LDS leaves no room for the R-box at four blocks per CU; nor does the LDS of the tree-in-memory builds: there the mailboxes' 10 KB cost the fourth
       block, and the class deal at four blocks measures 3 % faster than the mailboxes at three -- profiles/r06_gpu_call_i_*) */
    constexpr bool MAILBOX = MM != 0 && FLAT >= 2 && !WIDE && !DIRECT && MEGA_BALANCE && MEGA_MAILBOX && MEGA_REGEN_QUEUE && (!QMC || MEGA_MAILBOX_QMC);      /* (round 6: the QMC builds too -- their work counters became per-wave ones (WCNT: 8 KB of static LDS), which is the room the R-box needed; the sample's sequence index travels with the path) */
    /* (WIDE: neither -- the deal's five block barriers per pass cost more than the divergence they remove once a pass is dominated by a traversal whose length differs from
       wave to wave: glass + copper spheres 1554 -> 1803, glass room 510 -> 602, atrium 444 -> 457 Msamples/s without it, profiles/r06_gpu_call_n_*) */
    constexpr bool DEAL = MM != 0 && FLAT >= 2 && !WIDE && !DIRECT && MEGA_BALANCE && MEGA_CLASS_DEAL && !MAILBOX;      /* (`direct`: the camera vertex's rounds carry the camera hit along -- no exchange) */
    __shared__ uint32_t ldsClsCnt[4][BLOCK / 64];                 /* MEGA_CLASS_DEAL: lanes per BSDF model and wave */
    __shared__ uint32_t mbR[MAILBOX ? MB_DW * MB_NR : 1u];        /* (MB_DW = 24: the last two words of an entry carry the sample's sequence index in the QMC builds) */
    __shared__ uint32_t mbR_unused_doc[1];        /* MEGA_MAILBOX: the R-box, [MB_DW][MB_NR]; the S-box lies in the dynamic LDS behind the traversals' work lists (phip.hip sizes the region) */
    __shared__ uint32_t mbState[MAILBOX ? MB_NS + MB_NR : 1u];   /* entry states, S-box then R-box: 0 empty, 2 full, 3 being read (R-box: three consumers claim by compare-and-swap) */
    __shared__ int mbLive;                                        /* sample ids drawn by the block's waves that have not ended as a sample yet (queued camera samples and paths, wherever they are) */
    /* WIDE: per-wave counters instead (WCNT: lanes are counted as ballots, node steps and triangle tests by the traversal as k_rays_w does) -- the 8 KB are a fifth
       of what a block may take at four blocks per CU once stack, node cache and round buffers are in */
    constexpr bool WCNT = (WIDE || (MAILBOX && QMC)) && !MEGA_PROFILE && !MEGA_MB_DIAG;
    __shared__ uint32_t ldsCount[WCNT ? 1 : MC_COUNT][WCNT ? 1 : BLOCK];   /* work counters: one LDS word per lane and counter instead of eight VGPRs.  (As ds_add_u32 -- no read,
                                                                   no wait -- and the three that count lanes as ballots in SGPRs: 66.9 vs 66.5 ms per C2 frame and
                                                                   125 instead of 116 VGPRs; the seven read-modify-writes per pass overlap with the rest as they are) */
#define MEGA_COUNT(row, amount) ldsCount[row][threadIdx.x] += (uint32_t) (amount)
    enum { WC_SAMPLES = 0, WC_VERTICES, WC_RAYS, WC_STEPS, WC_SH_RAYS, WC_SH_STEPS, WC_COUNT };      /* WC_STEPS: node steps | triangle tests << 32 (k_wide_wave.h) */
    __shared__ unsigned long long wcnt[WCNT || WIDE ? BLOCK / 64 : 1][WC_COUNT];
    /* JOINT (WIDE): the shadow ray of a vertex is traced TOGETHER with the next ray of its path, in the traversal phase of the next pass (traceWideW: a lane brings two rays) --
       one wait for the wave's slowest lane per vertex instead of two.  A path that ended with its shadow ray pending parks its accumulator here and frees the lane */
    constexpr bool POOL = WIDE && MEGA_POOL;
    constexpr bool JOINT = WIDE && MEGA_JOINT;
#if MEGA_REGEN_QUEUE
    constexpr int RQ_ROWS = QMC ? 10 : 8;
    __shared__ uint32_t ldsRegen[BLOCK / 64][RQ_ROWS][64];      /* per wave: 64 prepared camera samples (mint | d, maxt | id, pixel, k [| the sample's sequence index: QMC]), one word per entry and row;
                                                                   the origin of a pinhole camera's rays is the same for every sample (camO below) */
    __shared__ uint32_t ldsSeq[QMC ? BLOCK / 64 : 1][2][64];    /* QMC: per lane, the sequence index of the path's sample -- shadeVertex draws every number of the path from it; deriving it anew at every
                                                                   request (sobol::look_up: ~30 table rows) was a third of the QMC kernel's sampling cost */
#endif
    /* dynamic LDS: [traversal stack | all nodes | all Wald records] (setupTraversal) [shading records | emitter table | materials],
       sized for THIS scene (megaLdsBytes) so that as many blocks as the registers allow fit a CU */
    /* WIDE: [group stack | top-of-tree node cache] (setupWide) [the four waves' round buffers | S-box] [emitter table | materials (FLAT 4)] (k_wide_wave.h: megaWideLdsBytesOf) */
    constexpr uint32_t WAVE_BYTES = POOL ? WP_WAVE_BYTES : WD_WAVE_BYTES;      /* the waves' round buffers (their first 512 bytes: the result slots, which serve mbAssign between traversals) */
    unsigned char *wideDeal = g_smem + (POOL ? widePoolDealOffset(M.nodeCache) : megaWideDealOffset(M.nodeCache));
    float4 *ldsTriShade = (float4 *) (g_smem + traversalLdsBytesOf(S));
    float *ldsEm = WIDE ? (float *) (wideDeal + (BLOCK / 64u) * WAVE_BYTES + (MAILBOX ? MB_DW * MB_NS * sizeof(uint32_t) : 0u))
                        : (float *) (ldsTriShade + (size_t) S.nTriangles * TRISHADE_FLOAT4S);
    DevMaterial *ldsMat = (DevMaterial *) (ldsEm + ((S.emitterTabSize + 3u) & ~3u));
    ShadeTables tab = stageShadeTables<FLAT != 5>(S, ldsEm, ldsMat);
    /* the host chose this kernel because every table fits (phip.hip: fitsLds): no run-time choice between the LDS copy and HBM, so that
       the compiler can address the tables as LDS (ds_read) instead of through flat loads, which occupy the texture addresser */
    tab.T.t = ldsEm; tab.materials = FLAT == 5 ? S.materials : ldsMat;
    float4 *ldsFlat = (float4 *) (((uintptr_t) (ldsMat + S.nMaterials) + 15u) & ~(uintptr_t) 15u);      /* FLAT: the table of leaf boxes (traverseFlat) */
    if (FLAT && !WIDE) for (uint32_t i = threadIdx.x; i < 2u * S.nFlatLeaves; i += BLOCK) ldsFlat[i] = S.flatLeaves[i];
    lds_cf4 *flat = (lds_cf4 *) ldsFlat; (void) flat;
    if (!WIDE) {
        for (uint32_t i = threadIdx.x; i < S.nTriangles * TRISHADE_FLOAT4S; i += BLOCK) ldsTriShade[i] = S.triShade[i];
        S.triShade = ldsTriShade;                               /* (generic pointer into LDS: six loads per vertex, not the inner loop) */
    }
    if (MAILBOX) { if (threadIdx.x < MB_NS + MB_NR) mbState[threadIdx.x] = 0u; if (threadIdx.x == 0u) mbLive = 0; }
    TravStack stk; stk.tris = nullptr;
    WideStackT<BLOCK> wstk; WidePool wpool;
    if (POOL) setupWidePool(S, M.nodeCache, g_smem, M.spill + (size_t) blockIdx.x * BLOCK * SPILL_DEPTH, (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6)), wpool);   /* the waves' task stacks (LDS, HBM spill behind them) + the top of the tree (barrier inside) */
    else if (WIDE) setupWide<BLOCK>(S, M.nodeCache, g_smem, M.spill + (size_t) blockIdx.x * BLOCK * SPILL_DEPTH, wstk);     /* group stack (LDS, HBM spill behind it) + the top of the tree (barrier inside) */
    else setupTraversal(S, g_smem, nullptr, stk);               /* stack + all nodes + all Wald records in LDS (barrier inside); the host checked that nothing can spill */

    const uint32_t waveId = blockIdx.x * (BLOCK / 64) + (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6)), lane = __lane_id();
    unsigned long long next = 0, end = 0;                       /* the wave's chunk of sample ids (wave-uniform) */
    const uint32_t waveInBlock = (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    /* FLAT 2 / 3 && MEGA_BALANCE: over the traversal stack, which the flat table does not use (phip.hip sizes it); WIDE: the wave's round buffers (the slots serve mbAssign between traversals) */
    unsigned char *const waveDeal = wideDeal + waveInBlock * WAVE_BYTES;
    WaveBalance wb = waveBalanceAt(g_smem, waveInBlock);
    if (WIDE) { wb.slot = (lds_u64 *) waveDeal; wb.list = (lds_u16 *) (waveDeal + 2u * 64u * 8u); }
    bool poolOverflow = false;                                  /* POOL: a task stack ran out of LDS + spill (the host refuses the frame, as for a mailbox time-out) */
#if MEGA_REGEN_QUEUE
    uint32_t qHead = 0, qCount = 0;                             /* the wave's queue of prepared camera samples (wave-uniform) */
    V3 camO;                                                    /* the origin cameraRay returns for every sample (dv_scene.h: the camera-to-world translation, by its own expression) */
    { V3 d_; float mn_, mx_; cameraRay(S.cam, 0.5f, 0.5f, camO, d_, mn_, mx_); }
#endif
    bool exhausted = rc.totalIds == 0;
    /* MEGA_MAILBOX */
    uint32_t *mbS = WIDE ? (uint32_t *) (wideDeal + (BLOCK / 64u) * WAVE_BYTES)
                         : (uint32_t *) (g_smem + (BLOCK / 64u) * BAL_WAVE_BYTES);      /* the S-box, [MB_DW][64], behind the four waves' work lists */
    const bool server = MAILBOX && waveInBlock == 0u;
    bool haveHit = false;                                       /* server: the lane's path came out of the S-box with its hit */
    uint32_t idleSpins = 0, patience = 0, idleSig = 0;
    bool mbTimedOut = false;
#if MEGA_MB_DIAG
    uint32_t dgDeposit = 0, dgLocal = 0, dgWithdrawn = 0, dgKept = 0, dgServerPass = 0, dgClientPass = 0, dgRefill = 0, dgServerRegen = 0;
#endif
    /* the k-th lane that `wants` gets the index of the k-th entry that is `avail` (lane l looks at entry l); through the wave's slot array, free between traversals */
    auto mbAssign = [&](bool wants, bool avail, uint32_t &e) -> bool {
        const unsigned long long am = __ballot(avail), wm = __ballot(wants);
        lds_u32 *map = (lds_u32 *) wb.slot;
        if (avail) map[__builtin_amdgcn_mbcnt_hi((uint32_t) (am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) am, 0u))] = lane;
        BAL_SYNC();
        const uint32_t rw = __builtin_amdgcn_mbcnt_hi((uint32_t) (wm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) wm, 0u));
        const bool got = wants && rw < (uint32_t) __popcll(am);
        e = got ? map[rw] : 0u;
        BAL_SYNC();
        return got;
    };

    bool alive = false;
    PathVertex v; v.id = v.pixel = v.k = v.state = 0;
    v.hit = v.rayO = v.rayD = v.thr = make_float4(0, 0, 0, 0); v.mis = make_float2(0, 0);
    float4 accum = make_float4(0, 0, 0, 0);
    float4 camHit = make_float4(0, 0, 0, 0);                    /* DIRECT: the camera ray's hit record, kept through the sample's rounds (the wavefront path's PathPool::camHit) */
    if (!WCNT) {
#pragma unroll
        for (int i = 0; i < MC_COUNT; ++i) ldsCount[WCNT ? 0 : i][WCNT ? 0 : threadIdx.x] = 0;
    }
    unsigned long long *const wc = wcnt[(WCNT || WIDE) ? waveInBlock : 0u];
    if ((WCNT || WIDE) && lane < (uint32_t) WC_COUNT) wc[lane] = 0ull;      /* (every wave its own row: no barrier needed) */
    /* WCNT outside the tree-in-memory builds: a per-lane count summed over the wave (six DPP steps), lane 63 adds it to the wave's counter */
    auto wcAdd = [&](int row, uint32_t perLane, bool shiftHigh) {
        uint32_t t = perLane;
        t += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) t, 0x111, 0xf, 0xf, true);
        t += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) t, 0x112, 0xf, 0xf, true);
        t += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) t, 0x114, 0xf, 0xf, true);
        t += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) t, 0x118, 0xf, 0xf, true);
        t += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) t, 0x142, 0xa, 0xf, false);
        t += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) t, 0x143, 0xc, 0xf, false);
        if (lane == 63u) wc[row] += shiftHigh ? ((unsigned long long) t << 32) : (unsigned long long) t;
    };
    bool cPush = false, cPend = false;                          /* JOINT: a shadow ray waits for the next traversal phase; its path ended at that vertex (accumulator parked) */
    float4 cPark = make_float4(0, 0, 0, 0);                     /* (registers: 4 KB of LDS per block would cost the fourth block of a CU) */
    ShadowEntry cSh; cSh.e0 = cSh.e1 = cSh.e2 = make_float4(0, 0, 0, 0);

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
        bool skipRegen = false;
        if (MAILBOX) {
            haveHit = false;
            if (__any(!alive)) {
                if (server) {
                    /* the server's free lanes take the copper vertices out of the S-box (the only consumer: no claim needed) */
                    const uint32_t stt = __hip_atomic_load(&mbState[lane], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const uint32_t nFull = (uint32_t) __popcll(__ballot(stt == 2u));
                    bool waiting = nFull != 0u;
                    if (waiting && nFull < (uint32_t) MEGA_MB_THRESH && !exhausted && ++patience <= (uint32_t) MEGA_MB_PATIENCE) waiting = false;      /* not a full wave yet: camera samples meanwhile */
                    if (waiting) patience = 0u;
                    skipRegen = waiting;
                    uint32_t e;
                    if (waiting && mbAssign(!alive, stt == 2u, e)) {
                        const uint32_t *x = mbS + e;
                        v.hit = make_float4(pm_from_bits(x[0 * MB_NS]), pm_from_bits(x[1 * MB_NS]), pm_from_bits(x[2 * MB_NS]), pm_from_bits(x[3 * MB_NS]));
                        v.rayD = make_float4(pm_from_bits(x[4 * MB_NS]), pm_from_bits(x[5 * MB_NS]), pm_from_bits(x[6 * MB_NS]), 0.0f);
                        v.thr = make_float4(pm_from_bits(x[7 * MB_NS]), pm_from_bits(x[8 * MB_NS]), pm_from_bits(x[9 * MB_NS]), pm_from_bits(x[10 * MB_NS]));
                        v.mis = make_float2(pm_from_bits(x[11 * MB_NS]), pm_from_bits(x[12 * MB_NS]));
                        v.id = x[13 * MB_NS]; v.pixel = x[14 * MB_NS]; v.k = x[15 * MB_NS]; v.state = x[16 * MB_NS];
                        accum = make_float4(pm_from_bits(x[17 * MB_NS]), pm_from_bits(x[18 * MB_NS]), pm_from_bits(x[19 * MB_NS]), pm_from_bits(x[20 * MB_NS]));
#if MEGA_REGEN_QUEUE
                        if (QMC) { ldsSeq[QMC ? waveInBlock : 0][0][lane] = x[22 * MB_NS]; ldsSeq[QMC ? waveInBlock : 0][1][lane] = x[23 * MB_NS]; }
#endif
                        alive = true; haveHit = true;
#if MEGA_MB_DIAG
                        ++dgWithdrawn;
#endif
                        __hip_atomic_store(&mbState[e], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                } else {
                    /* a client's free lanes take continued paths out of the R-box (three consumers: claim by compare-and-swap) before they take camera samples */
                    const uint32_t stt = lane < MB_NR ? __hip_atomic_load(&mbState[MB_NS + lane], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) : ~0u;
                    uint32_t e;
                    if (__any(stt == 2u)) {
                        bool got = mbAssign(!alive, stt == 2u, e);
                        if (got) {      /* (an ACQUIRE claim -- ADVICE r5: an entry refilled between the load above and the claim must not have its payload read before the server's release store) */
                            uint32_t expect = 2u;
                            got = __hip_atomic_compare_exchange_strong(&mbState[MB_NS + e], &expect, 3u, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        if (got) {
                            const uint32_t *x = mbR + e;
                            v.rayO = make_float4(pm_from_bits(x[0 * MB_NR]), pm_from_bits(x[1 * MB_NR]), pm_from_bits(x[2 * MB_NR]), pm_from_bits(x[3 * MB_NR]));
                            v.rayD = make_float4(pm_from_bits(x[4 * MB_NR]), pm_from_bits(x[5 * MB_NR]), pm_from_bits(x[6 * MB_NR]), pm_from_bits(x[7 * MB_NR]));
                            v.thr = make_float4(pm_from_bits(x[8 * MB_NR]), pm_from_bits(x[9 * MB_NR]), pm_from_bits(x[10 * MB_NR]), pm_from_bits(x[11 * MB_NR]));
                            v.mis = make_float2(pm_from_bits(x[12 * MB_NR]), pm_from_bits(x[13 * MB_NR]));
                            v.id = x[14 * MB_NR]; v.pixel = x[15 * MB_NR]; v.k = x[16 * MB_NR]; v.state = x[17 * MB_NR];
                            accum = make_float4(pm_from_bits(x[18 * MB_NR]), pm_from_bits(x[19 * MB_NR]), pm_from_bits(x[20 * MB_NR]), pm_from_bits(x[21 * MB_NR]));
#if MEGA_REGEN_QUEUE
                            if (QMC) { ldsSeq[QMC ? waveInBlock : 0][0][lane] = x[22 * MB_NR]; ldsSeq[QMC ? waveInBlock : 0][1][lane] = x[23 * MB_NR]; }
#endif
                            alive = true;
#if MEGA_MB_DIAG
                            ++dgRefill;
#endif
                            __hip_atomic_store(&mbState[MB_NS + e], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                }
            }
        }
        /* ---- regeneration: lanes without a path start the next camera sample (integrator.cpp:157-183) ---- */
#if MEGA_REGEN_QUEUE
        /* Camera samples are prepared 64 at a time by ALL lanes of the wave (id decode, pixel jitter, camera ray: ~670 instructions) into a
           per-wave LDS queue; a lane whose path ended pops the next entry (eleven LDS words).  Before, that code ran in every pass of the
           loop for the ~18 lanes (28 %) whose paths had just ended: 14 % of the kernel's time at a quarter of the lanes. */
        for (;;) {
            const unsigned long long want = __ballot(!alive);
            if (!want || (MAILBOX && skipRegen)) break;
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
                            if (MAILBOX) {
                                /* the ids are counted as the block's BEFORE they are drawn (and what the draw does not grant is taken back below): a wave that finds
                                   the supply exhausted by this very draw must not read mbLive without it */
                                atomicAdd(&mbLive, (int) chunk);
                                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                            }
                            base = atomicAdd(M.nextId, (unsigned long long) chunk);
                        } else {
                            base = rc.totalIds;
                        }
                    }
                    const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t) base), bhi = __builtin_amdgcn_readfirstlane((uint32_t) (base >> 32));
                    chunk = __builtin_amdgcn_readfirstlane(chunk);
                    next = ((unsigned long long) bhi << 32) | blo;
                    end = next + chunk; if (end > rc.totalIds) end = rc.totalIds;
                    if (MAILBOX && lane == 0u && chunk) {       /* every granted id ends as a sample or is skipped below */
                        const unsigned long long granted = end > next ? end - next : 0ull;
                        if (granted < chunk) atomicSub(&mbLive, (int) (chunk - granted));
                    }
                    if (next >= end) { exhausted = true; break; }
                }
                /* the next 64 ids of the chunk, one per lane (ids outside the crop window -- edge blocks -- are consumed and skipped) */
                const unsigned long long id = next + lane;
                uint32_t px = 0, py = 0, k = 0;
                const bool valid = id < end && decodeId(rc, S.film, id, px, py, k);
                const unsigned long long vmask = __ballot(valid);
                if (MAILBOX) { const int skipped = __popcll(__ballot(id < end && !valid)); if (skipped && lane == 0u) atomicSub(&mbLive, skipped); }
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
        if (DEAL) {
            if (!__syncthreads_or((alive || (JOINT && cPush)) ? 1 : 0)) break;        /* (the waves of a block meet at barriers below: they leave the loop together) */
        } else if (MAILBOX) {
            if (!__any(alive || (JOINT && cPush))) {
                /* nothing in this wave's lanes: done when no id is left anywhere AND every id the block's waves drew has ended as a sample (a path may sit in a
                   mailbox or in another wave and come here yet); until then look into the mailboxes again.  The wait is bounded: no bug may hang the device */
                const int live_ = __hip_atomic_load(&mbLive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (exhausted && qCount == 0u && live_ <= 0) break;
                /* the bound is on waiting WITHOUT PROGRESS (a wall-clock bound alone could trip under counter passes, a debugger, or one very long last path): the
                   count of the block's live ids and the pattern of full mailbox entries are the protocol's state -- while they move, somebody is working */
                {
                    const uint32_t stS_ = __hip_atomic_load(&mbState[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const uint32_t stR_ = lane < MB_NR ? __hip_atomic_load(&mbState[MB_NS + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
                    const unsigned long long fs_ = __ballot(stS_ == 2u), fr_ = __ballot(stR_ == 2u);
                    const uint32_t sig_ = (uint32_t) live_ * 0x9E3779B9u ^ (uint32_t) fs_ ^ (uint32_t) (fs_ >> 32) * 3u ^ (uint32_t) fr_ * 5u ^ (uint32_t) (fr_ >> 32) * 7u;
                    if (sig_ != idleSig) { idleSig = sig_; idleSpins = 0u; }
                }
                if (++idleSpins > (1u << 22)) { mbTimedOut = true; break; }      /* (the host re-renders the pass on the wavefront kernels: phip.hip) */
                __builtin_amdgcn_s_sleep(8);
                continue;
            }
            idleSpins = 0;
        } else if (!__any(alive || (JOINT && cPush))) break;

        /* ---- closest hit ---- */
        uint32_t hitCls = 0;                                    /* shade class of the record hit (the Wald record's 12th word: 0 diffuse, 1 rough conductor, 2 dielectric) */
        { PF_BEGIN
        if (WIDE) {                                             /* every lane takes part (k_wide_wave.h) */
            const V3 o(v.rayO.x, v.rayO.y, v.rayO.z), d(v.rayD.x, v.rayD.y, v.rayD.z);
            float mint, maxt;
            TravResult r;
            V3 rcp;
            const bool trace = alive && !(MAILBOX && haveHit) && !(DIRECT && (v.state & F_NOTRACE));      /* (DIRECT: a round without a BSDF sample has no closest-hit query) */
            const bool go = trace & clipToSceneSel<false>(S, o, d, v.rayO.w, v.rayD.w, mint, maxt, rcp);
            /* JOINT: ... and the shadow ray of the vertex this lane shaded in the previous pass (its own path's, or that of the path that ended there) */
            const V3 so(cSh.e0.x, cSh.e0.y, cSh.e0.z), sd(cSh.e1.x, cSh.e1.y, cSh.e1.z);
            float smint = 0.0f, smaxt = 0.0f; bool goS = false;
            if (JOINT) { V3 srcp; goS = cPush & clipToSceneSel<true>(S, so, sd, PT_EPSILON, cSh.e0.w, smint, smaxt, srcp); }
            bool occluded = false;
            if (POOL) traceWidePool<JOINT, true>(S, wpool, lane, goS, so, sd, smint, smaxt, go, o, d, mint, maxt, occluded, r, wc + WC_SH_STEPS, wc + WC_STEPS, poolOverflow);
            else traceWideW<JOINT, true>(S, wstk, waveDeal, lane, goS, so, sd, smint, smaxt, go, o, d, mint, maxt, occluded, r, wc + WC_SH_STEPS, wc + WC_STEPS);
            if (trace) {
                v.hit = make_float4(r.t, r.u, r.v, pm_from_bits(r.prim));
                hitCls = r.cls;
            }
            { const uint32_t n_ = (uint32_t) __popcll(__ballot(trace)); if (lane == 0u) wc[WC_RAYS] += n_; }
            if (JOINT) {
                /* the shadow ray's verdict (path.cpp:187-199): the contribution joins the path's accumulator -- the register if the path goes on in this lane, the parked one
                   if it ended at that vertex, which is then the sample's value */
                if (cPush && !occluded) {
                    if (cPend) { cPark.x += cSh.e2.x; cPark.y += cSh.e2.y; cPark.z += cSh.e2.z; }
                    else { accum.x += cSh.e2.x; accum.y += cSh.e2.y; accum.z += cSh.e2.z; }
                }
                if (cPend) L[pm_to_bits(cSh.e2.w)] = cPark;
                const uint32_t nS_ = (uint32_t) __popcll(__ballot(cPush)), nE_ = (uint32_t) __popcll(__ballot(cPend));
                if (lane == 0u) { wc[WC_SH_RAYS] += nS_; wc[WC_SAMPLES] += nE_; }
                cPush = false; cPend = false;
            }
        } else
        if (FLAT >= 2 && MEGA_BALANCE) {                        /* every lane takes part: the tests of the wave's rays are dealt over its lanes */
            const V3 o(v.rayO.x, v.rayO.y, v.rayO.z), d(v.rayD.x, v.rayD.y, v.rayD.z);
            float mint, maxt;
            TravResult r;
            uint32_t nNode = 0, nTri = 0;
            V3 rcp;
            const bool trace = alive && !(MAILBOX && haveHit) && !(DIRECT && (v.state & F_NOTRACE));
            const bool go = trace & clipToSceneSel<false>(S, o, d, v.rayO.w, v.rayD.w, mint, maxt, rcp);
            traverseFlat2W<false, FLAT == 3>(flat, S.nFlatLeaves, stk.tris, wb, lane, go, o, d, rcp, mint, maxt, r, nNode, nTri);
            if (trace) {
                v.hit = make_float4(r.t, r.u, r.v, pm_from_bits(r.prim));
                hitCls = r.cls;
                if (!WCNT) { MEGA_COUNT(MC_RAYS, 1); MEGA_COUNT(MC_NODE, nNode); MEGA_COUNT(MC_TRI, nTri); }
            }
            if (WCNT) { wcAdd(WC_RAYS, trace ? 1u : 0u, false); wcAdd(WC_STEPS, trace ? nNode : 0u, false); wcAdd(WC_STEPS, trace ? nTri : 0u, true); }
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
        /* ---- MEGA_MAILBOX: a client hands the paths that hit copper to the server (if its third of the S-box has room: otherwise it shades them itself) ---- */
        if (MAILBOX && !server) {
            const bool special = alive && pm_to_bits(v.hit.w) != PHIP_NO_HIT && hitCls != 0u && ((MEGA_MB_CLASSES >> (hitCls - 1u)) & 1u);
            if (__any(special)) {
                /* the wave's own third of the S-box: clients that run in step would pick the same free entries otherwise, and all but one lose the claim */
                const uint32_t stt = __hip_atomic_load(&mbState[lane], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                uint32_t e;
                const bool got = mbAssign(special, stt == 0u && (lane * 3u) / MB_NS == waveInBlock - 1u, e);
                if (got) {
                    uint32_t *x = mbS + e;
                    x[0 * MB_NS] = pm_to_bits(v.hit.x); x[1 * MB_NS] = pm_to_bits(v.hit.y); x[2 * MB_NS] = pm_to_bits(v.hit.z); x[3 * MB_NS] = pm_to_bits(v.hit.w);
                    x[4 * MB_NS] = pm_to_bits(v.rayD.x); x[5 * MB_NS] = pm_to_bits(v.rayD.y); x[6 * MB_NS] = pm_to_bits(v.rayD.z);
                    x[7 * MB_NS] = pm_to_bits(v.thr.x); x[8 * MB_NS] = pm_to_bits(v.thr.y); x[9 * MB_NS] = pm_to_bits(v.thr.z); x[10 * MB_NS] = pm_to_bits(v.thr.w);
                    x[11 * MB_NS] = pm_to_bits(v.mis.x); x[12 * MB_NS] = pm_to_bits(v.mis.y);
                    x[13 * MB_NS] = v.id; x[14 * MB_NS] = v.pixel; x[15 * MB_NS] = v.k; x[16 * MB_NS] = v.state;
                    x[17 * MB_NS] = pm_to_bits(accum.x); x[18 * MB_NS] = pm_to_bits(accum.y); x[19 * MB_NS] = pm_to_bits(accum.z); x[20 * MB_NS] = pm_to_bits(accum.w);
#if MEGA_REGEN_QUEUE
                    if (QMC) { x[22 * MB_NS] = ldsSeq[QMC ? waveInBlock : 0][0][lane]; x[23 * MB_NS] = ldsSeq[QMC ? waveInBlock : 0][1][lane]; }
#endif
                    __hip_atomic_store(&mbState[e], 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    alive = false;
#if MEGA_MB_DIAG
                    ++dgDeposit;
#endif
                }
#if MEGA_MB_DIAG
                if (special && alive) ++dgLocal;
#endif
            }
#if MEGA_MB_DIAG
            if (lane == 0u) ++dgClientPass;
#endif
        }
#if MEGA_MB_DIAG
        if (MAILBOX && server && lane == 0u) ++dgServerPass;
        if (MAILBOX && server && alive && !haveHit && (v.state & F_FIRST)) ++dgServerRegen;
#endif
        /* ---- the paths of the block dealt to its lanes by BSDF model (MEGA_CLASS_DEAL above) ---- */
        if (DEAL) {
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
                uint32_t *x = POOL ? (uint32_t *) wideDeal : (uint32_t *) g_smem;               /* [MEGA_DEAL_DWORDS][BLOCK], over the traversal stack / work lists (POOL: the waves' round buffers), unused between traversals */
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
        if ((FLAT >= 2 && MEGA_BALANCE) || WIDE) sh.e0 = sh.e1 = make_float4(0, 0, 0, 0);   /* every lane clips "its" entry (a lane without one takes no part in the result) */
        { PF_BEGIN
        if (alive) {
            uint32_t nv = 0;
            bool newRay;
#if MEGA_REGEN_QUEUE
            const LRegister acc{ accum, (QMC && isSequenceSampler(rc.sampler)) ? &ldsSeq[QMC ? waveInBlock : 0][0][lane] : nullptr };
#else
            const LRegister acc{ accum, nullptr };
#endif
            if (DIRECT) ended = directVertex<MM, QMC ? 8 : 0>(S, tab.T, tab.materials, rc, v, camHit, acc, newRay, pushShadow, sh, nv);
            else ended = shadeVertex<MM, STRICT, QMC ? 8 : 0>(S, tab.T, tab.materials, rc, v, acc, newRay, pushShadow, sh, nv);
            if (ended) { if (WCNT) atomicAdd(&wc[WC_VERTICES], (unsigned long long) nv); else MEGA_COUNT(MC_VERTICES, nv); }
        }

        PF_END(2, __ballot(alive)) }
        /* ---- shadow ray of the NEE sample; unoccluded: the contribution joins the accumulator (path.cpp:187-199) ---- */
        { PF_BEGIN
        if (JOINT && !server) {
            /* the shadow ray waits for the traversal phase of the next pass.  A path that ended here frees the lane now: without a shadow ray its accumulator is the sample;
               with one the accumulator is parked until the ray is decided */
            cPush = pushShadow; cSh = sh;
            if (ended) {
                if (pushShadow) { cPark = accum; cPend = true; }
                else L[v.id] = accum;
                alive = false;
            }
            const uint32_t n_ = (uint32_t) __popcll(__ballot(ended && !pushShadow));
            if (lane == 0u) wc[WC_SAMPLES] += n_;
        } else
        if (WIDE) {
            const V3 o(sh.e0.x, sh.e0.y, sh.e0.z), d(sh.e1.x, sh.e1.y, sh.e1.z);
            float mint, maxt;
            TravResult r;
            V3 rcp;
            const bool go = pushShadow & clipToSceneSel<true>(S, o, d, PT_EPSILON, sh.e0.w, mint, maxt, rcp);
            bool occluded;
            if (POOL) traceWidePool<true, false>(S, wpool, lane, go, o, d, mint, maxt, false, o, d, 0.0f, 0.0f, occluded, r, wc + WC_SH_STEPS, wc + WC_STEPS, poolOverflow);
            else traceWideW<true, false>(S, wstk, waveDeal, lane, go, o, d, mint, maxt, false, o, d, 0.0f, 0.0f, occluded, r, wc + WC_SH_STEPS, wc + WC_STEPS);
            if (pushShadow && !occluded) { accum.x += sh.e2.x; accum.y += sh.e2.y; accum.z += sh.e2.z; }
            { const uint32_t n_ = (uint32_t) __popcll(__ballot(pushShadow)); if (lane == 0u) wc[WC_SH_RAYS] += n_; }
        } else
        if (FLAT >= 2 && MEGA_BALANCE) {
            const V3 o(sh.e0.x, sh.e0.y, sh.e0.z), d(sh.e1.x, sh.e1.y, sh.e1.z);
            float mint, maxt;
            TravResult r;
            uint32_t nNode = 0, nTri = 0;
            V3 rcp;
            const bool go = pushShadow & clipToSceneSel<true>(S, o, d, PT_EPSILON, sh.e0.w, mint, maxt, rcp);
            const bool occluded = traverseFlat2W<true, FLAT == 3>(flat, S.nFlatLeaves, stk.tris, wb, lane, go, o, d, rcp, mint, maxt, r, nNode, nTri);
            if (pushShadow) {
                if (!WCNT) { MEGA_COUNT(MC_SH_RAYS, 1); MEGA_COUNT(MC_SH_NODE, nNode); MEGA_COUNT(MC_SH_TRI, nTri); }
                if (!occluded) { accum.x += sh.e2.x; accum.y += sh.e2.y; accum.z += sh.e2.z; }
            }
            if (WCNT) { wcAdd(WC_SH_RAYS, pushShadow ? 1u : 0u, false); wcAdd(WC_SH_STEPS, pushShadow ? nNode : 0u, false); wcAdd(WC_SH_STEPS, pushShadow ? nTri : 0u, true); }
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
        if (ended && !(JOINT && !server)) {                  /* (JOINT: stored or parked above) */
            L[v.id] = accum;
            if (!WCNT) MEGA_COUNT(MC_SAMPLES, 1);
            alive = false;
        }
        if (WCNT && !(JOINT && !server)) { const uint32_t n_ = (uint32_t) __popcll(__ballot(ended)); if (lane == 0u) wc[WC_SAMPLES] += n_; }
        if (MAILBOX) {
            const int nEnded = __popcll(__ballot(ended));
            if (nEnded && lane == 0u) atomicSub(&mbLive, nEnded);
            if (server && __any(alive)) {
                /* the server hands every continued path back (the only producer of the R-box: no claim); what finds no room stays and is traced here in the next pass */
                const uint32_t stt = lane < MB_NR ? __hip_atomic_load(&mbState[MB_NS + lane], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) : ~0u;
                uint32_t e;
                if (mbAssign(alive, stt == 0u, e)) {
                    uint32_t *x = mbR + e;
                    x[0 * MB_NR] = pm_to_bits(v.rayO.x); x[1 * MB_NR] = pm_to_bits(v.rayO.y); x[2 * MB_NR] = pm_to_bits(v.rayO.z); x[3 * MB_NR] = pm_to_bits(v.rayO.w);
                    x[4 * MB_NR] = pm_to_bits(v.rayD.x); x[5 * MB_NR] = pm_to_bits(v.rayD.y); x[6 * MB_NR] = pm_to_bits(v.rayD.z); x[7 * MB_NR] = pm_to_bits(v.rayD.w);
                    x[8 * MB_NR] = pm_to_bits(v.thr.x); x[9 * MB_NR] = pm_to_bits(v.thr.y); x[10 * MB_NR] = pm_to_bits(v.thr.z); x[11 * MB_NR] = pm_to_bits(v.thr.w);
                    x[12 * MB_NR] = pm_to_bits(v.mis.x); x[13 * MB_NR] = pm_to_bits(v.mis.y);
                    x[14 * MB_NR] = v.id; x[15 * MB_NR] = v.pixel; x[16 * MB_NR] = v.k; x[17 * MB_NR] = v.state;
                    x[18 * MB_NR] = pm_to_bits(accum.x); x[19 * MB_NR] = pm_to_bits(accum.y); x[20 * MB_NR] = pm_to_bits(accum.z); x[21 * MB_NR] = pm_to_bits(accum.w);
#if MEGA_REGEN_QUEUE
                    if (QMC) { x[22 * MB_NR] = ldsSeq[QMC ? waveInBlock : 0][0][lane]; x[23 * MB_NR] = ldsSeq[QMC ? waveInBlock : 0][1][lane]; }
#endif
                    __hip_atomic_store(&mbState[MB_NS + e], 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    alive = false;
                }
#if MEGA_MB_DIAG
                if (alive) ++dgKept;
#endif
            }
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

#if MEGA_MB_DIAG
    ldsCount[MC_NODE][threadIdx.x] = dgDeposit; ldsCount[MC_TRI][threadIdx.x] = dgLocal; ldsCount[MC_SH_NODE][threadIdx.x] = dgWithdrawn; ldsCount[MC_SH_TRI][threadIdx.x] = dgKept;
    ldsCount[MC_RAYS][threadIdx.x] = dgServerPass; ldsCount[MC_SH_RAYS][threadIdx.x] = dgClientPass; ldsCount[MC_VERTICES][threadIdx.x] = dgRefill + (dgServerRegen << 0) * 0u; ldsCount[MC_SAMPLES][threadIdx.x] = dgServerRegen;
#endif
    /* per-wave statistics (one owner per entry, no atomics) */
    PathPool P; P.stat = M.stat; P.nWaves = M.nWaves;
    const int rows[MC_COUNT] = { ST_SAMPLES, ST_VERTICES, ST_CLOSEST_RAYS, ST_NODE, ST_TRI, ST_SHADOW_RAYS, ST_SH_NODE, ST_SH_TRI };
#pragma unroll
    for (int i = 0; i < MC_COUNT; ++i) {
        unsigned long long val = 0ull;
        if (WCNT) {
            if (lane == 0u)
                val = i == MC_SAMPLES ? wc[WC_SAMPLES] : i == MC_VERTICES ? wc[WC_VERTICES] : i == MC_RAYS ? wc[WC_RAYS] : i == MC_NODE ? (wc[WC_STEPS] & 0xFFFFFFFFull)
                    : i == MC_TRI ? (wc[WC_STEPS] >> 32) : i == MC_SH_RAYS ? wc[WC_SH_RAYS] : i == MC_SH_NODE ? (wc[WC_SH_STEPS] & 0xFFFFFFFFull) : (wc[WC_SH_STEPS] >> 32);
        } else val = ldsCount[WCNT ? 0 : i][WCNT ? 0 : threadIdx.x];
        const bool poison = (MAILBOX && mbTimedOut) || (POOL && __any(poolOverflow)) || (MEGA_MB_FAULT && waveId == 0u);
        waveStat(P, rows[i], waveId, val + ((i == MC_SAMPLES && poison && lane == 0u) ? (1ull << 62) : 0ull));
    }
}

