/*
 * phip_mega.hip -- k_mega<materials, strictNormals, traversal form, QMC> (k_mega.h): the fused single-kernel path.  Compiled twice (phip_common.h):
 *   -DMEGA_PART=0  scenes that fit LDS: BVH4 walk / leaf tables (FLAT 0 .. 3)
 *   -DMEGA_PART=1  round 6: scenes whose tree stays in memory -- the compressed 8-wide tree walked from L2 (FLAT 4 / 5: k_wide_wave.h)
 *   -DMEGA_PART=2  round 6: the `direct` integrator in the same kernel (k_mega<.., DIRECT = true>), packed leaf tables and the tree in memory
 */
#include "phip_common.h"
#include "k_traverse.h"
#include "k_wide_node.h"
#include "k_wide_wave.h"
#include "k_shade.h"
#include "k_shade_direct.h"
#include "k_mega.h"

#ifndef MEGA_PART
#define MEGA_PART 0
#endif

typedef void (*MegaKernel)(DevScene, MegaParams, RenderConst, float4 *);

#if MEGA_PART == 0
/* Leaf BSDF models: diffuse only (every traversal form), or all three (round 5) on the packed leaf table -- the Cornell box with a glass and a copper block.
   (Round 2 measured "more than 256 VGPRs" for the microfacet / dielectric code next to the traversal; since then the traversal became the dealt table pass,
   the work counters moved to LDS and the camera samples to a queue: k_mega<MM_ALL, false, 2, false> builds at 128 VGPRs with 16 B of scratch, four waves.)
   A scene with glass but no copper runs the build that also knows copper: one more material set, not three. */
template <bool QMC> static MegaKernel megaKernelOf(int materialMask, bool strictNormals, int flat) {
    if (materialMask & MM_ALL) {
        if (flat == 3) return MEGA_BALANCE ? (strictNormals ? k_mega<MM_ALL, true, 3, QMC> : k_mega<MM_ALL, false, 3, QMC>) : nullptr;
        if (flat == 2) return strictNormals ? k_mega<MM_ALL, true, 2, QMC> : k_mega<MM_ALL, false, 2, QMC>;
        return nullptr;
    }
    if (flat == 3) return MEGA_BALANCE ? (strictNormals ? k_mega<0, true, 3, QMC> : k_mega<0, false, 3, QMC>) : nullptr;
    if (flat == 2) return strictNormals ? k_mega<0, true, 2, QMC> : k_mega<0, false, 2, QMC>;
#if PHIP_EXPERIMENTS      /* the BVH4 walk and the per-lane leaf table in LDS (rounds 2-3): since round 6 every scene past the packed leaf table's 64 records is on the 8-wide tree (FLAT 4 / 5) */
    if (flat) return strictNormals ? k_mega<0, true, 1, QMC> : k_mega<0, false, 1, QMC>;
    return strictNormals ? k_mega<0, true, 0, QMC> : k_mega<0, false, 0, QMC>;
#else
    return nullptr;
#endif
}
#define MEGA_ENTRY(name) name
#elif MEGA_PART == 2
/* round 6: `direct` in the fused kernel (k_shade_direct.h: directVertex) -- the packed leaf tables of the LDS-resident scenes and the tree in memory; strictNormals is a run-time
   switch of that integrator (direct.cpp:177-190) */
template <bool QMC> static MegaKernel megaKernelOf(int materialMask, bool, int flat) {
    const bool all = (materialMask & MM_ALL) != 0;
    switch (flat) {
        case 2: return all ? k_mega<MM_ALL, false, 2, QMC, true> : k_mega<0, false, 2, QMC, true>;
        case 3: return all ? k_mega<MM_ALL, false, 3, QMC, true> : k_mega<0, false, 3, QMC, true>;
        case 4: return all ? k_mega<MM_ALL, false, 4, QMC, true> : k_mega<0, false, 4, QMC, true>;
        case 5: return all ? k_mega<MM_ALL, false, 5, QMC, true> : k_mega<0, false, 5, QMC, true>;
        default: return nullptr;
    }
}
#else
/* the tree in memory: 4 = emitter table and materials in LDS, 5 = the materials stay in memory (the atrium's 252) */
template <bool QMC> static MegaKernel megaKernelOf(int materialMask, bool strictNormals, int flat) {
    if (flat != 4 && flat != 5) return nullptr;
    if (materialMask & MM_ALL) {
        if (flat == 4) return strictNormals ? k_mega<MM_ALL, true, 4, QMC> : k_mega<MM_ALL, false, 4, QMC>;
        return strictNormals ? k_mega<MM_ALL, true, 5, QMC> : k_mega<MM_ALL, false, 5, QMC>;
    }
    if (flat == 4) return strictNormals ? k_mega<0, true, 4, QMC> : k_mega<0, false, 4, QMC>;
    return strictNormals ? k_mega<0, true, 5, QMC> : k_mega<0, false, 5, QMC>;
}
#define MEGA_ENTRY(name) name##Wide
#endif
#if MEGA_PART == 2
#undef MEGA_ENTRY
#define MEGA_ENTRY(name) name##Direct
#endif
static MegaKernel megaKernel(int materialMask, bool strictNormals, int flat, bool qmc) {
    return qmc ? megaKernelOf<true>(materialMask, strictNormals, flat) : megaKernelOf<false>(materialMask, strictNormals, flat);
}

int MEGA_ENTRY(phipMegaBlocksPerCU)(int materialMask, bool strictNormals, int flat, bool qmc, size_t ldsBytes) {
    int n = 0;
    if (!megaKernel(materialMask, strictNormals, flat, qmc)) return 0;
    if (ldsBytes > 48 * 1024 && hipFuncSetAttribute((const void *) megaKernel(materialMask, strictNormals, flat, qmc), hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsBytes) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *) megaKernel(materialMask, strictNormals, flat, qmc), BLOCK, ldsBytes) != hipSuccess) return 0;
    return n;
}

void MEGA_ENTRY(phipLaunchMega)(int materialMask, bool strictNormals, int flat, bool qmc, dim3 grid, size_t ldsBytes, hipStream_t stream,
                                const DevScene &S, const MegaParams &M, const RenderConst &rc, float4 *L) {
    hipLaunchKernelGGL(megaKernel(materialMask, strictNormals, flat, qmc), grid, dim3(BLOCK), ldsBytes, stream, S, M, rc, L);
}
