/*
 * phip_common.h -- what every translation unit of libphip.so starts with: HIP, the C ABI, the device scene layout and
 * shading functions (dv_scene.h), the path-pool layout (k_pool.h), the error macro.
 *
 * libphip.so is built from three sources (26 objects) so that they compile in parallel:
 *   phip.hip        host side (scene build, render loop, multi-device orchestration, C ABI) + traversal and film kernels
 *   phip_shade.hip  k_shade / k_shade_direct / k_shade_trace instantiations behind phipLaunchShade*F<n> -- compiled per feature set (-DSHADE_FEAT=0..3, 8 and 11:
 *                   environment emitter, bitmap textures, the QMC samplers) and per part (-DSHADE_PART=0..3), 24 objects: see its header
 *   phip_mega.hip   k_mega instantiations behind phipLaunchMega (-DMEGA_PART=0: scenes in LDS) / phipLaunchMegaWide (-DMEGA_PART=1: the 8-wide tree in memory)
 * No device function is called across units (everything on the device side is inline in headers), so no -fgpu-rdc.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <thread>

#include "../../include/phip.h"
#include "dv_scene.h"

using namespace pt;

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess)                                                                    \
            throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e__));         \
    } while (0)

#include "k_pool.h"
#include "k_clip.h"

/* ---- launchers defined in the other translation units ---- */
/* k_shade<materials, strictNormals, FEAT> / k_shade_direct<materials, FEAT> over the pool: phip_shade.hip compiled with -DSHADE_FEAT=n */
#define PHIP_DECLARE_SHADE(n)                                                                                              \
    void phipLaunchShadeF##n(bool strictNormals, int materialMask, dim3 grid, hipStream_t stream,                          \
                             const DevScene &S, const PathPool &P, const RenderConst &rc, float4 *L);                      \
    void phipLaunchShadeDirectF##n(int materialMask, dim3 grid, hipStream_t stream,                                        \
                                   const DevScene &S, const PathPool &P, const RenderConst &rc, float4 *L);                \
    void phipLaunchShadeTraceF##n(bool strictNormals, int materialMask, dim3 grid, size_t ldsBytes, hipStream_t stream,    \
                                  const DevScene &S, const PathPool &P, const RenderConst &rc, float4 *L);
PHIP_DECLARE_SHADE(0) PHIP_DECLARE_SHADE(1) PHIP_DECLARE_SHADE(2) PHIP_DECLARE_SHADE(3) PHIP_DECLARE_SHADE(8) PHIP_DECLARE_SHADE(11)
#undef PHIP_DECLARE_SHADE
/* k_mega<materials, strictNormals, traversal form> (phip_mega.hip): blocks of BLOCK threads that fit one CU with ldsBytes of dynamic LDS.
   flat 0 .. 3 (0 / DevScene::flatMode: scenes that fit LDS) live in the object compiled with -DMEGA_PART=0, flat 4 / 5 (the 8-wide tree in memory) in -DMEGA_PART=1 */
int  phipMegaBlocksPerCU(int materialMask, bool strictNormals, int flat, bool qmc, size_t ldsBytes);
void phipLaunchMega(int materialMask, bool strictNormals, int flat, bool qmc, dim3 grid, size_t ldsBytes, hipStream_t stream,
                    const DevScene &S, const MegaParams &M, const RenderConst &rc, float4 *L);
int  phipMegaBlocksPerCUDirect(int materialMask, bool strictNormals, int flat, bool qmc, size_t ldsBytes);      /* -DMEGA_PART=2: `direct`, flat 2 .. 5 */
void phipLaunchMegaDirect(int materialMask, bool strictNormals, int flat, bool qmc, dim3 grid, size_t ldsBytes, hipStream_t stream,
                          const DevScene &S, const MegaParams &M, const RenderConst &rc, float4 *L);
int  phipMegaBlocksPerCUWide(int materialMask, bool strictNormals, int flat, bool qmc, size_t ldsBytes);
void phipLaunchMegaWide(int materialMask, bool strictNormals, int flat, bool qmc, dim3 grid, size_t ldsBytes, hipStream_t stream,
                        const DevScene &S, const MegaParams &M, const RenderConst &rc, float4 *L);
