/*
 * phip_mega.hip -- k_mega<materials, strictNormals, table form, QMC> (k_mega.h): the fused single-kernel path for scenes that fit LDS.
 * One of libphip.so's three translation units (phip_common.h).
 */
#include "phip_common.h"
#include "k_traverse.h"
#include "k_shade.h"
#include "k_mega.h"

typedef void (*MegaKernel)(DevScene, MegaParams, RenderConst, float4 *);

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
    if (flat) return strictNormals ? k_mega<0, true, 1, QMC> : k_mega<0, false, 1, QMC>;
    return strictNormals ? k_mega<0, true, 0, QMC> : k_mega<0, false, 0, QMC>;
}
static MegaKernel megaKernel(int materialMask, bool strictNormals, int flat, bool qmc) {
    return qmc ? megaKernelOf<true>(materialMask, strictNormals, flat) : megaKernelOf<false>(materialMask, strictNormals, flat);
}

int phipMegaBlocksPerCU(int materialMask, bool strictNormals, int flat, bool qmc, size_t ldsBytes) {
    int n = 0;
    if (!megaKernel(materialMask, strictNormals, flat, qmc)) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *) megaKernel(materialMask, strictNormals, flat, qmc), BLOCK, ldsBytes) != hipSuccess) return 0;
    return n;
}

void phipLaunchMega(int materialMask, bool strictNormals, bool qmc, dim3 grid, size_t ldsBytes, hipStream_t stream,
                    const DevScene &S, const MegaParams &M, const RenderConst &rc, float4 *L) {
    hipLaunchKernelGGL(megaKernel(materialMask, strictNormals, S.nFlatLeaves ? (int) S.flatMode : 0, qmc), grid, dim3(BLOCK), ldsBytes, stream, S, M, rc, L);
}
