/*
 * phip_shade.hip -- the shading kernels of the wavefront path: k_shade<materials, strictNormals, features> (k_shade.h), k_shade_direct<materials, features>
 * (k_shade_direct.h) and k_shade_trace<materials, strictNormals, features> (k_shade_trace.h).  Compiled once per feature set (-DSHADE_FEAT=0..3: bit 0 =
 * environment emitter, bit 1 = bitmap textures; 8 and 11: bit 3 = the QMC samplers, without / with both other features) AND per part (-DSHADE_PART: 0 / 1 =
 * k_shade without / with strictNormals, 2 = k_shade_direct + the launchers phip.hip calls, 3 = k_shade_trace), 24 objects that build in parallel: the
 * texture code (MIP / EWA look-ups) is inlined into every kernel that can meet a textured leaf, and a unit with all of them took 3.5 minutes on its own.
 * phip.hip dispatches on the scene's feature set (phipLaunchShade / phipLaunchShadeDirect / phipLaunchShadeTrace).  See phip_common.h.
 */
#include "phip_common.h"
#include "k_traverse.h"
#include "k_shade.h"
#include "k_shade_direct.h"
#include "k_shade_trace.h"

#if !defined(SHADE_FEAT) || !defined(SHADE_PART)
#error "compile with -DSHADE_FEAT=0..3, 8 or 11 and -DSHADE_PART=0..3"
#endif
#define SHADE_CAT2(a, b) a##b
#define SHADE_CAT(a, b) SHADE_CAT2(a, b)

typedef void (*ShadeKernel)(DevScene, PathPool, RenderConst, float4 *);
/* the kernels of the other parts of this feature set: (leaf BSDF models present, which table set: 0 = generic pointers, 1 = emitter table and materials
   addressed as LDS, 2 = the emitter table only -- FEAT 0 only) -> kernel */
ShadeKernel SHADE_CAT(phipShadeKernelS0F, SHADE_FEAT)(int materialMask, int tables);
ShadeKernel SHADE_CAT(phipShadeKernelS1F, SHADE_FEAT)(int materialMask, int tables);
ShadeKernel SHADE_CAT(phipShadeTraceKernelF, SHADE_FEAT)(bool strictNormals, int materialMask);

#if SHADE_PART == 0 || SHADE_PART == 1
#define SHADE_STRICT (SHADE_PART == 1)
ShadeKernel SHADE_CAT(SHADE_CAT(SHADE_CAT(phipShadeKernelS, SHADE_PART), F), SHADE_FEAT)(int materialMask, int tables) {
#define SHADE_ROW(F_) { k_shade<0, SHADE_STRICT, F_>, k_shade<MM_ROUGH, SHADE_STRICT, F_>, k_shade<MM_DIELECTRIC, SHADE_STRICT, F_>, k_shade<MM_ALL, SHADE_STRICT, F_> }
    static const ShadeKernel table[4] = SHADE_ROW(SHADE_FEAT);
#if SHADE_FEAT == 0
    /* no environment emitter, no textures (the configurations the metric is quoted on): a second set of kernels for scenes whose emitter
       table and materials fit LDS -- nearly all -- which addresses them as LDS; a third where only the emitter table fits (scenes of many materials) */
    static const ShadeKernel tableLds[4] = SHADE_ROW(4), tableEmLds[4] = SHADE_ROW(16);
    if (tables == 1) return tableLds[materialMask & MM_ALL];
    if (tables == 2) return tableEmLds[materialMask & MM_ALL];
#endif
#undef SHADE_ROW
    (void) tables;
    return table[materialMask & MM_ALL];
}

#elif SHADE_PART == 2
void SHADE_CAT(phipLaunchShadeF, SHADE_FEAT)(bool strictNormals, int materialMask, dim3 grid, hipStream_t stream,
                                             const DevScene &S, const PathPool &P, const RenderConst &rc, float4 *L) {
    int tables = 0;
#if SHADE_FEAT == 0
    if (S.emitterTabSize <= EMITTER_LDS_FLOATS && !(PHIP_EXPERIMENTS && getenv("PHIP_SHADE_FLAT_TABLES"))) tables = S.nMaterials <= MATERIAL_LDS_MAX ? 1 : 2;
#endif
    const ShadeKernel k = strictNormals ? SHADE_CAT(phipShadeKernelS1F, SHADE_FEAT)(materialMask, tables) : SHADE_CAT(phipShadeKernelS0F, SHADE_FEAT)(materialMask, tables);
    hipLaunchKernelGGL(k, grid, dim3(BLOCK), 0, stream, S, P, rc, L);
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
    hipLaunchKernelGGL(SHADE_CAT(phipShadeTraceKernelF, SHADE_FEAT)(strictNormals, materialMask), grid, dim3(BLOCK), ldsBytes, stream, S, P, rc, L);
}

#elif SHADE_PART == 3
ShadeKernel SHADE_CAT(phipShadeTraceKernelF, SHADE_FEAT)(bool strictNormals, int materialMask) {
    /* leaf BSDF models = diffuse only / all (a scene with glass but no copper runs the kernel that also knows copper: two builds per feature set, not four);
       FEAT 0: phip.hip takes this path only when emitter table and materials fit LDS -- the kernel stages them itself, whatever FEAT says */
    static const ShadeKernel table[2][2] = { { k_shade_trace<0, false, SHADE_FEAT>, k_shade_trace<MM_ALL, false, SHADE_FEAT> },
                                             { k_shade_trace<0, true, SHADE_FEAT>, k_shade_trace<MM_ALL, true, SHADE_FEAT> } };
    return table[strictNormals ? 1 : 0][(materialMask & MM_ALL) ? 1 : 0];
}
#endif
