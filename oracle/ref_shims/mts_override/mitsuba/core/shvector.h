/* ORACLE -- TEST INFRASTRUCTURE ONLY.
 * Stand-in for include/mitsuba/core/shvector.h when compiling the reference's command-line front end (src/mitsuba/mitsuba.cpp), which
 * only calls SHVector::staticInitialization / staticShutdown: the real class stores its coefficients in Eigen matrices (Eigen is not in
 * this image and src/libcore/shvector.cpp is not built, oracle/Makefile.ref); nothing on the `path` hot path uses spherical harmonics. */
#pragma once
#include <mitsuba/mitsuba.h>
MTS_NAMESPACE_BEGIN
struct SHVector {
    static void staticInitialization() { }
    static void staticShutdown() { }
};
MTS_NAMESPACE_END
