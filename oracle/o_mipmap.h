/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see o_math.h).
 *
 * o_mipmap.h: lookups into a MIP pyramid, restated from include/mitsuba/render/mipmap.h: evalTexel with the five
 * boundary conditions (:503-571), evalBox (:566-569), evalBilinear (:575-596), eval = nearest / bilinear / trilinear /
 * EWA with the anisotropy clamp (:629-712), evalEWA (:780-833), the 64-entry Gaussian weight table (:296-301).
 * The pyramid itself (resampling filter, half-precision storage) is DATA handed over by the caller.
 * Used by the `envmap` emitter (o_envmap.h) and the `bitmap` texture (o_scene.h).
 */
#pragma once
#include "o_math.h"
#include "../include/phip.h"
#include <vector>
#include <stdexcept>

namespace orc {

struct MipMap {
    std::vector<std::vector<Spectrum>> levels;
    std::vector<int> lw, lh;
    int nLevels = 0;
    uint32_t bcu = PHIP_WRAP_REPEAT, bcv = PHIP_WRAP_REPEAT, filterType = PHIP_FILTER_EWA;
    Float maxAnisotropy = 20.0f;
    Float weightLut[64];

    /* level sizes: mipmap.h:182-192; data[l] = RGB floats of level l (data[0] must be given) */
    void load(uint32_t width, uint32_t height, uint32_t n_levels, const float *const *data) {
        const int w = (int) width, h = (int) height;
        if (w <= 0 || h <= 0 || !data || !data[0]) throw std::runtime_error("oracle: MIP map without level 0");
        lw.assign(1, w); lh.assign(1, h);
        if (n_levels > 1) {
            int sx = w, sy = h;
            while (sx > 1 || sy > 1) { sx = std::max(1, (sx + 1) / 2); sy = std::max(1, (sy + 1) / 2); lw.push_back(sx); lh.push_back(sy); }
            if ((uint32_t) lw.size() != n_levels) throw std::runtime_error("oracle: a MIP map needs 1 level or the complete pyramid");
        }
        nLevels = (int) lw.size();
        levels.resize(nLevels);
        for (int l = 0; l < nLevels; ++l) {
            if (!data[l]) throw std::runtime_error("oracle: MIP level pointer is NULL");
            levels[l].resize((size_t) lw[l] * lh[l]);
            for (size_t i = 0; i < levels[l].size(); ++i) levels[l][i] = Spectrum(data[l] + 3 * i);
        }
        for (int i = 0; i < 64; ++i) {                    /* mipmap.h:296-301 */
            Float r2 = (Float) i / (Float) 63;
            weightLut[i] = om::fastexp(-2.0f * r2) - om::fastexp(-2.0f);
        }
    }

    static int modulo(int a, int b) { int r = a % b; return (r < 0) ? r + b : r; }   /* math.h:67-70 */
    /* one coordinate under a boundary condition, mipmap.h:506-565; false: the texel is the constant `outside` */
    static bool wrap(int &x, int size, uint32_t bc, Float &outside) {
        if (x < 0 || x >= size) {
            switch (bc) {
                case PHIP_WRAP_REPEAT: x = modulo(x, size); break;
                case PHIP_WRAP_CLAMP: x = std::min(std::max(x, 0), size - 1); break;
                case PHIP_WRAP_MIRROR: x = modulo(x, 2 * size); if (x >= size) x = 2 * size - x - 1; break;
                case PHIP_WRAP_ZERO: outside = 0.0f; return false;
                case PHIP_WRAP_ONE: outside = 1.0f; return false;
            }
        }
        return true;
    }
    /* mipmap.h:503-571 */
    Spectrum evalTexel(int level, int x, int y) const {
        const int sw = lw[level], sh = lh[level];
        Float outside = 0;
        if (!wrap(x, sw, bcu, outside)) return Spectrum(outside);
        if (!wrap(y, sh, bcv, outside)) return Spectrum(outside);
        return levels[level][(size_t) y * sw + x];
    }
    /* mipmap.h:566-569 */
    Spectrum evalBox(int level, const Vec2 &uv) const {
        return evalTexel(level, (int) std::floor(uv.x * lw[level]), (int) std::floor(uv.y * lh[level]));
    }
    /* mipmap.h:575-596 */
    Spectrum evalBilinear(int level, const Vec2 &uv) const {
        if (!std::isfinite(uv.x) || !std::isfinite(uv.y)) return Spectrum(0.0f);
        if (level >= nLevels) return evalBox(nLevels - 1, uv);
        Float u = uv.x * lw[level] - 0.5f, v = uv.y * lh[level] - 0.5f;
        int xPos = (int) std::floor(u), yPos = (int) std::floor(v);
        Float dx1 = u - xPos, dx2 = 1.0f - dx1, dy1 = v - yPos, dy2 = 1.0f - dy1;
        return evalTexel(level, xPos, yPos) * dx2 * dy2
             + evalTexel(level, xPos, yPos + 1) * dx2 * dy1
             + evalTexel(level, xPos + 1, yPos) * dx1 * dy2
             + evalTexel(level, xPos + 1, yPos + 1) * dx1 * dy1;
    }
    static Float log2f_(Float value) {                    /* math.cpp:103-106 */
        const Float invLn2 = 1.0f / om::log(2.0f);
        return om::fastlog(value) * invLn2;
    }
    /* mipmap.h:780-833 */
    Spectrum evalEWA(int level, const Vec2 &uv, Float A, Float B, Float C) const {
        if (!std::isfinite(A + B + C + uv.x + uv.y)) return Spectrum(0.0f);
        if (level >= nLevels) return evalBox(nLevels - 1, uv);
        Float u = uv.x * lw[level] - 0.5f;
        Float v = uv.y * lh[level] - 0.5f;
        const Float ratioX = (Float) lw[level] / (Float) lw[0], ratioY = (Float) lh[level] / (Float) lh[0];   /* m_sizeRatio, mipmap.h:273-275 */
        A /= ratioX * ratioX;
        B /= ratioX * ratioY;
        C /= ratioY * ratioY;
        Float invDet = 1.0f / (-B * B + 4.0f * A * C),
              deltaU = 2.0f * std::sqrt(C * invDet),
              deltaV = 2.0f * std::sqrt(A * invDet);
        int u0 = (int) std::ceil(u - deltaU), u1 = (int) std::floor(u + deltaU);
        int v0 = (int) std::ceil(v - deltaV), v1 = (int) std::floor(v + deltaV);
        if ((long) u1 - u0 > 4096 || (long) v1 - v0 > 4096) return evalBilinear(level, uv);     /* same guard as the product (not in the reference) */
        Float As = A * 64, Bs = B * 64, Cs = C * 64;
        Spectrum result(0.0f);
        Float denominator = 0.0f;
        Float ddq = 2 * As, uu0 = (Float) u0 - u;
        for (int vt = v0; vt <= v1; ++vt) {
            const Float vv = (Float) vt - v;
            Float q = As * uu0 * uu0 + (Bs * uu0 + Cs * vv) * vv;
            Float dq = As * (2 * uu0 + 1) + Bs * vv;
            for (int ut = u0; ut <= u1; ++ut) {
                if (q < (Float) 64) {
                    uint32_t qi = (uint32_t) q;
                    if (qi < 64) {
                        const Float weight = weightLut[(int) q];
                        result += evalTexel(level, ut, vt) * weight;
                        denominator += weight;
                    }
                }
                q += dq;
                dq += ddq;
            }
        }
        if (denominator == 0)
            return evalBilinear(level, uv);
        return result / denominator;
    }
    /* mipmap.h:629-712 */
    Spectrum eval(const Vec2 &uv, const Vec2 &d0, const Vec2 &d1) const {
        if (filterType == PHIP_FILTER_NEAREST) return evalBox(0, uv);
        else if (filterType == PHIP_FILTER_BILINEAR) return evalBilinear(0, uv);
        Float du0 = d0.x * lw[0], dv0 = d0.y * lh[0], du1 = d1.x * lw[0], dv1 = d1.y * lh[0];
        Float A = dv0 * dv0 + dv1 * dv1,
              B = -2.0f * (du0 * dv0 + du1 * dv1),
              C = du0 * du0 + du1 * du1,
              F = A * C - B * B * 0.25f;
        Float root = hypot2(A - C, B),
              Aprime = 0.5f * (A + C - root),
              Cprime = 0.5f * (A + C + root),
              majorRadius = Aprime != 0 ? std::sqrt(F / Aprime) : 0,
              minorRadius = Cprime != 0 ? std::sqrt(F / Cprime) : 0;
        if (filterType == PHIP_FILTER_TRILINEAR || !(minorRadius > 0) || !(majorRadius > 0) || F < 0) {
            Float level = log2f_(std::max(majorRadius, ORC_EPSILON));
            int ilevel = (int) std::floor(level);
            if (ilevel < 0) {
                return evalBilinear(0, uv);
            } else {
                Float a = level - ilevel;
                return evalBilinear(ilevel, uv) * (1.0f - a) + evalBilinear(ilevel + 1, uv) * a;
            }
        } else {
            if (minorRadius * maxAnisotropy < majorRadius) {
                minorRadius = majorRadius / maxAnisotropy;
                Float theta = 0.5f * om::atan(B / (A - C)), sinTheta, cosTheta;
                om::sincos(theta, &sinTheta, &cosTheta);
                Float a2 = majorRadius * majorRadius, b2 = minorRadius * minorRadius,
                      sinTheta2 = sinTheta * sinTheta, cosTheta2 = cosTheta * cosTheta,
                      sin2Theta = 2 * sinTheta * cosTheta;
                A = a2 * cosTheta2 + b2 * sinTheta2;
                B = (a2 - b2) * sin2Theta;
                C = a2 * sinTheta2 + b2 * cosTheta2;
                F = a2 * b2;
            }
            Float scl = 1.0f / F;
            A *= scl; B *= scl; C *= scl;
            Float level = std::max((Float) 0.0f, log2f_(minorRadius));
            int ilevel = (int) level;
            Float a = level - ilevel;
            if (majorRadius < 1 || !(A > 0 && C > 0))
                return evalBilinear(ilevel, uv);
            else
                return evalEWA(ilevel, uv, A, B, C) * (1.0f - a) + evalEWA(ilevel + 1, uv, A, B, C) * a;
        }
    }

};

} // namespace orc
