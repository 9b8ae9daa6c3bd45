/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.
 * hw_glue.cpp: src/libhw/basicshader.cpp (which holds ConstantSpectrumTexture & co., the textures every BSDF plugin
 * instantiates for constant parameters) refers to two members of the OpenGL preview renderer (src/libhw/renderer.cpp,
 * needs GL headers).  Nothing on the CPU rendering path calls them.
 */
#include <mitsuba/hw/renderer.h>
#include <cstdio>
#include <cstdlib>

MTS_NAMESPACE_BEGIN
Shader *Renderer::registerShaderForResource(const HWResource *) { fprintf(stderr, "hw_glue: no hardware renderer in the oracle build\n"); abort(); }
void Renderer::unregisterShaderForResource(const HWResource *) { fprintf(stderr, "hw_glue: no hardware renderer in the oracle build\n"); abort(); }
MTS_NAMESPACE_END
