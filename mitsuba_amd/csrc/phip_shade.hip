/*
 * phip_shade.hip -- the shading kernels of the wavefront path: k_shade<materials, strictNormals, features> (k_shade.h) and
 * k_shade_direct<materials, features> (k_shade_direct.h), 40 instantiations.  Compiled once per feature set
 * (-DSHADE_FEAT=0..3: bit 0 = environment emitter, bit 1 = bitmap textures) so that the four objects build in parallel;
 * phip.hip dispatches on the scene's feature set (phipLaunchShade / phipLaunchShadeDirect).  See phip_common.h.
 */
#include "phip_common.h"
#include "k_shade.h"
#include "k_shade_direct.h"

#ifndef SHADE_FEAT
#error "compile with -DSHADE_FEAT=0..3"
#endif
#define SHADE_CAT2(a, b) a##b
#define SHADE_CAT(a, b) SHADE_CAT2(a, b)

typedef void (*ShadeKernel)(DevScene, PathPool, RenderConst, float4 *);

void SHADE_CAT(phipLaunchShadeF, SHADE_FEAT)(bool strictNormals, int materialMask, dim3 grid, hipStream_t stream,
                                             const DevScene &S, const PathPool &P, const RenderConst &rc, float4 *L) {
#define SHADE_ROW(S_) { k_shade<0, S_, SHADE_FEAT>, k_shade<MM_ROUGH, S_, SHADE_FEAT>, k_shade<MM_DIELECTRIC, S_, SHADE_FEAT>, k_shade<MM_ALL, S_, SHADE_FEAT> }
    static const ShadeKernel table[2][4] = { SHADE_ROW(false), SHADE_ROW(true) };
#undef SHADE_ROW
    hipLaunchKernelGGL(table[strictNormals ? 1 : 0][materialMask & MM_ALL], grid, dim3(BLOCK), 0, stream, S, P, rc, L);
}

void SHADE_CAT(phipLaunchShadeDirectF, SHADE_FEAT)(int materialMask, dim3 grid, hipStream_t stream,
                                                   const DevScene &S, const PathPool &P, const RenderConst &rc, float4 *L) {
    /* `direct`: leaf BSDF models = diffuse only / all, strictNormals at run time */
    static const ShadeKernel table[2] = { k_shade_direct<0, SHADE_FEAT>, k_shade_direct<MM_ALL, SHADE_FEAT> };
    hipLaunchKernelGGL(table[(materialMask & MM_ALL) ? 1 : 0], grid, dim3(BLOCK), 0, stream, S, P, rc, L);
}
