/*
 * phip_mega.hip -- k_mega<materials, strictNormals, table form, QMC> (k_mega.h): the fused single-kernel path for scenes that fit LDS.
 * One of libphip.so's three translation units (phip_common.h).
 */
#include "phip_common.h"
#include "k_traverse.h"
#include "k_shade.h"
#include "k_mega.h"

typedef void (*MegaKernel)(DevScene, MegaParams, RenderConst, float4 *);

/* Only the diffuse instantiation exists: with the microfacet / dielectric code inlined next to the traversal the kernel needs
   more than 256 VGPRs (measured: 256 + scratch at 2 waves per SIMD), and the scenes of that kind that fit LDS are test
   scenes, not workloads -- they keep the wavefront kernels. */
template <bool QMC> static MegaKernel megaKernelOf(bool strictNormals, int flat) {
    if (flat == 3) return MEGA_BALANCE ? (strictNormals ? k_mega<0, true, 3, QMC> : k_mega<0, false, 3, QMC>) : nullptr;
    if (flat == 2) return strictNormals ? k_mega<0, true, 2, QMC> : k_mega<0, false, 2, QMC>;
    if (flat) return strictNormals ? k_mega<0, true, 1, QMC> : k_mega<0, false, 1, QMC>;
    return strictNormals ? k_mega<0, true, 0, QMC> : k_mega<0, false, 0, QMC>;
}
static MegaKernel megaKernel(int materialMask, bool strictNormals, int flat, bool qmc) {
    if (materialMask & MM_ALL) return nullptr;
    return qmc ? megaKernelOf<true>(strictNormals, flat) : megaKernelOf<false>(strictNormals, flat);
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
