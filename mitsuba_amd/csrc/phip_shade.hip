/*
 * phip_shade.hip -- the shading kernels of the wavefront path: k_shade<materials, strictNormals, features> (k_shade.h) and
 * k_shade_direct<materials, features> (k_shade_direct.h), 44 instantiations.  Compiled once per feature set
 * (-DSHADE_FEAT=0..3: bit 0 = environment emitter, bit 1 = bitmap textures; 8 and 11: bit 3 = the QMC samplers, without / with both other features)
 * so that the objects build in parallel;
 * phip.hip dispatches on the scene's feature set (phipLaunchShade / phipLaunchShadeDirect).  See phip_common.h.
 */
#include "phip_common.h"
#include "k_traverse.h"
#include "k_shade.h"
#include "k_shade_direct.h"
#include "k_shade_trace.h"

#ifndef SHADE_FEAT
#error "compile with -DSHADE_FEAT=0..3, 8 or 11"
#endif
#define SHADE_CAT2(a, b) a##b
#define SHADE_CAT(a, b) SHADE_CAT2(a, b)

typedef void (*ShadeKernel)(DevScene, PathPool, RenderConst, float4 *);

void SHADE_CAT(phipLaunchShadeF, SHADE_FEAT)(bool strictNormals, int materialMask, dim3 grid, hipStream_t stream,
                                             const DevScene &S, const PathPool &P, const RenderConst &rc, float4 *L) {
#define SHADE_ROW(S_, F_) { k_shade<0, S_, F_>, k_shade<MM_ROUGH, S_, F_>, k_shade<MM_DIELECTRIC, S_, F_>, k_shade<MM_ALL, S_, F_> }
    static const ShadeKernel table[2][4] = { SHADE_ROW(false, SHADE_FEAT), SHADE_ROW(true, SHADE_FEAT) };
    const ShadeKernel *row = table[strictNormals ? 1 : 0];
#if SHADE_FEAT == 0
    /* no environment emitter, no textures (the configurations the metric is quoted on): a second set of kernels for scenes whose emitter
       table and materials fit LDS -- nearly all -- which addresses them as LDS */
    static const ShadeKernel tableLds[2][4] = { SHADE_ROW(false, 4), SHADE_ROW(true, 4) };
    static const ShadeKernel tableEmLds[2][4] = { SHADE_ROW(false, 16), SHADE_ROW(true, 16) };      /* ... only the emitter table fits (scenes of many materials) */
    if (S.emitterTabSize <= EMITTER_LDS_FLOATS && !getenv("PHIP_SHADE_FLAT_TABLES"))
        row = (S.nMaterials <= MATERIAL_LDS_MAX ? tableLds : tableEmLds)[strictNormals ? 1 : 0];
#endif
#undef SHADE_ROW
    hipLaunchKernelGGL(row[materialMask & MM_ALL], grid, dim3(BLOCK), 0, stream, S, P, rc, L);
}

void SHADE_CAT(phipLaunchShadeDirectF, SHADE_FEAT)(int materialMask, dim3 grid, hipStream_t stream,
                                                   const DevScene &S, const PathPool &P, const RenderConst &rc, float4 *L) {
    /* `direct`: leaf BSDF models = diffuse only / all, strictNormals at run time */
    static const ShadeKernel table[2] = { k_shade_direct<0, SHADE_FEAT>, k_shade_direct<MM_ALL, SHADE_FEAT> };
    hipLaunchKernelGGL(table[(materialMask & MM_ALL) ? 1 : 0], grid, dim3(BLOCK), 0, stream, S, P, rc, L);
}

/* k_shade_trace (k_shade_trace.h): scenes on the packed leaf table (<= 64 Wald records) that k_mega does not serve; `path` only */
void SHADE_CAT(phipLaunchShadeTraceF, SHADE_FEAT)(bool strictNormals, int materialMask, dim3 grid, size_t ldsBytes, hipStream_t stream,
                                                  const DevScene &S, const PathPool &P, const RenderConst &rc, float4 *L) {
#if SHADE_FEAT == 0
    /* (phip.hip takes this path only when emitter table and materials fit LDS: the LDS-addressed tables, FEAT bit 2) */
#define TRACE_ROW(S_) { k_shade_trace<0, S_, 4>, k_shade_trace<MM_ROUGH, S_, 4>, k_shade_trace<MM_DIELECTRIC, S_, 4>, k_shade_trace<MM_ALL, S_, 4> }
#else
#define TRACE_ROW(S_) { k_shade_trace<0, S_, SHADE_FEAT>, k_shade_trace<MM_ROUGH, S_, SHADE_FEAT>, k_shade_trace<MM_DIELECTRIC, S_, SHADE_FEAT>, k_shade_trace<MM_ALL, S_, SHADE_FEAT> }
#endif
    static const ShadeKernel table[2][4] = { TRACE_ROW(false), TRACE_ROW(true) };
#undef TRACE_ROW
    hipLaunchKernelGGL(table[strictNormals ? 1 : 0][materialMask & MM_ALL], grid, dim3(BLOCK), ldsBytes, stream, S, P, rc, L);
}
