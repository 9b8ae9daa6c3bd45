/*
 * phip_mega.hip -- k_mega<materials, strictNormals> (k_mega.h): the fused single-kernel path for scenes that fit LDS.
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
static MegaKernel megaKernel(int materialMask, bool strictNormals, int flat) {
    if (materialMask & MM_ALL) return nullptr;
    if (flat == 2) return strictNormals ? k_mega<0, true, 2> : k_mega<0, false, 2>;
    if (flat) return strictNormals ? k_mega<0, true, 1> : k_mega<0, false, 1>;
    return strictNormals ? k_mega<0, true, 0> : k_mega<0, false, 0>;
}

int phipMegaBlocksPerCU(int materialMask, bool strictNormals, int flat, size_t ldsBytes) {
    int n = 0;
    if (!megaKernel(materialMask, strictNormals, flat)) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *) megaKernel(materialMask, strictNormals, flat), BLOCK, ldsBytes) != hipSuccess) return 0;
    return n;
}

void phipLaunchMega(int materialMask, bool strictNormals, dim3 grid, size_t ldsBytes, hipStream_t stream,
                    const DevScene &S, const MegaParams &M, const RenderConst &rc, float4 *L) {
    hipLaunchKernelGGL(megaKernel(materialMask, strictNormals, S.nFlatLeaves ? (int) S.flatMode : 0), grid, dim3(BLOCK), ldsBytes, stream, S, M, rc, L);
}
