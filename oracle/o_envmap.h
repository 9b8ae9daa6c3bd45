/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see o_math.h).
 *
 * o_envmap.h: the illumination side of the `envmap` emitter, restated from src/emitters/envmap.cpp:
 *   configure()                 :262-328   marginal / conditional CDFs over luminance * sin(theta)
 *   evalEnvironment (no ray differentials) :380-394 + MIPMap::evalBilinear / evalTexel (mipmap.h:503-596)
 *   sampleDirect / pdfDirect    :516-556
 *   internalSampleDirection / internalPdfDirection :567-632,  sampleReuse :657-662
 * plus the filtered lookup of directly visible pixels (camera rays carry differentials): envmap.cpp:395-407 and
 * MIPMap::eval / evalEWA / evalBox (include/mitsuba/render/mipmap.h:566-569, 629-712, 780-833; EWA with
 * maxAnisotropy 10, 64-entry Gaussian LUT :296-301).  The pyramid itself (Lanczos resampling, mipmap.h:262-280) is
 * DATA: the levels arrive through the ABI as the reference built and stored them (half precision, handed over as
 * floats); with a single level the filtered lookup is unavailable.  Wrap modes: ERepeat in u, EClamp in v
 * (envmap.cpp:176-178).
 */
#pragma once
#include "o_mipmap.h"

namespace orc {

inline Float rgbLuminance(const Spectrum &s) { return s[0] * 0.212671f + s[1] * 0.715160f + s[2] * 0.072169f; }   /* spectrum.h:724-727 */

inline Float intervalToTent(Float sample) {   /* warp.cpp:143-155 */
    Float sign;
    if (sample < 0.5f) { sign = 1; sample *= 2; }
    else { sign = -1; sample = 2 * (sample - 0.5f); }
    return sign * (1 - std::sqrt(sample));
}

struct EnvMap {
    int w = 0, h = 0;
    MipMap mip;                                           /* ERepeat x EClamp, EWA, maxAnisotropy 10 (envmap.cpp:138-139,176-178) */
    int nLevels = 1;
    Float scale = 1.0f;
    Mat4 toWorld, toLocal;
    std::vector<float> cdfRows, cdfCols;
    std::vector<Float> rowWeights;
    Float normalization = 0;
    Float pixelSizeX = 0, pixelSizeY = 0;

    bool valid() const { return w > 0; }

    static Vec3 xformVec(const Mat4 &m, const Vec3 &v) {   /* transform.h:175-183 */
        Float x = m.m[0][0] * v.x + m.m[0][1] * v.y + m.m[0][2] * v.z;
        Float y = m.m[1][0] * v.x + m.m[1][1] * v.y + m.m[1][2] * v.z;
        Float z = m.m[2][0] * v.x + m.m[2][1] * v.y + m.m[2][2] * v.z;
        return Vec3(x, y, z);
    }

    void load(const phip_envmap &e) {
        if (!e.texels || e.width == 0 || e.height == 0) throw std::runtime_error("oracle: envmap emitter without texels");
        if (std::max(e.width, e.height) > 0xFFFF) throw std::runtime_error("Environment maps images must be smaller than 65536 pixels in width and height");
        w = (int) e.width; h = (int) e.height; scale = e.scale;
        const float *data[PHIP_ENVMAP_MAX_LEVELS];
        data[0] = e.texels;
        for (uint32_t l = 1; l < e.n_levels && l < PHIP_ENVMAP_MAX_LEVELS; ++l) data[l] = e.levels[l];
        mip.bcu = PHIP_WRAP_REPEAT; mip.bcv = PHIP_WRAP_CLAMP; mip.filterType = PHIP_FILTER_EWA; mip.maxAnisotropy = 10.0f;
        mip.load(e.width, e.height, e.n_levels > 1 ? e.n_levels : 1, data);
        nLevels = mip.nLevels;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) toWorld.m[i][j] = e.to_world[4 * i + j];
        if (!toWorld.invert(toLocal)) throw std::runtime_error("oracle: envmap toWorld is singular");
        configure();
    }

    /* envmap.cpp:262-328 */
    void configure() {
        size_t nEntries = (size_t) (w + 1) * (size_t) h;
        cdfCols.assign(nEntries, 0.0f); cdfRows.assign(h + 1, 0.0f); rowWeights.assign(h, 0.0f);
        size_t colPos = 0, rowPos = 0;
        Float rowSum = 0.0f;
        cdfRows[rowPos++] = 0;
        for (int y = 0; y < h; ++y) {
            Float colSum = 0;
            cdfCols[colPos++] = 0;
            for (int x = 0; x < w; ++x) {
                Spectrum value(mip.levels[0][(size_t) y * w + x]);
                colSum += rgbLuminance(value);
                cdfCols[colPos++] = (float) colSum;
            }
            float norm = 1.0f / (float) colSum;
            for (int x = 1; x < w; ++x)
                cdfCols[colPos - x - 1] *= norm;
            cdfCols[colPos - 1] = 1.0f;
            Float s, c; om::sincos((y + 0.5f) * ORC_PI / h, &s, &c);
            Float weight = s;
            rowWeights[y] = weight;
            rowSum += colSum * weight;
            cdfRows[rowPos++] = (float) rowSum;
        }
        float norm = 1.0f / (float) rowSum;
        for (int y = 1; y < h; ++y)
            cdfRows[rowPos - y - 1] *= norm;
        cdfRows[rowPos - 1] = 1.0f;
        if (rowSum == 0) throw std::runtime_error("The environment map is completely black -- this is not allowed.");
        if (!std::isfinite(rowSum)) throw std::runtime_error("The environment map contains an invalid floating point value (nan/inf) -- giving up.");
        normalization = 1.0f / (rowSum * (2 * ORC_PI / w) * (ORC_PI / h));
        pixelSizeX = 2 * ORC_PI / w; pixelSizeY = ORC_PI / h;
    }

    Spectrum evalTexel(int x, int y) const { return mip.evalTexel(0, x, y); }
    Spectrum evalBilinear(const Vec2 &uv) const { return mip.evalBilinear(0, uv); }

    /* envmap.cpp:380-394,408-409: a ray WITHOUT differentials (every ray Li spawns; path.cpp:229 assigns a plain Ray) */
    Spectrum evalEnvironment(const Vec3 &rayD) const {
        Vec3 v = xformVec(toLocal, rayD);
        Vec2 uv(om::atan2(v.x, -v.z) * ORC_INV_TWOPI, om::acos(std::min(1.0f, std::max(-1.0f, v.y))) * ORC_INV_PI);
        return evalBilinear(uv) * scale;
    }
    /* envmap.cpp:380-409 for a camera ray: rx / ry = the (scaled) differential directions */
    Spectrum evalEnvironment(const Vec3 &rayD, const Vec3 &rxDirection, const Vec3 &ryDirection) const {
        Vec3 v = xformVec(toLocal, rayD);
        Vec2 uv(om::atan2(v.x, -v.z) * ORC_INV_TWOPI, om::acos(std::min(1.0f, std::max(-1.0f, v.y))) * ORC_INV_PI);
        Vec3 dvdx = xformVec(toLocal, rxDirection) - v, dvdy = xformVec(toLocal, ryDirection) - v;
        Float t1 = ORC_INV_TWOPI / (v.x * v.x + v.z * v.z),
              t2 = -ORC_INV_PI / std::max(om::safe_sqrt(1.0f - v.y * v.y), ORC_EPSILON);
        Vec2 dudx(t1 * (dvdx.z * v.x - dvdx.x * v.z), t2 * dvdx.y),
             dudy(t1 * (dvdy.z * v.x - dvdy.x * v.z), t2 * dvdy.y);
        return mip.eval(uv, dudx, dudy) * scale;
    }

    /* envmap.cpp:657-662 */
    static uint32_t sampleReuse(const float *cdf, uint32_t size, Float &sample) {
        const float *entry = std::lower_bound(cdf, cdf + size + 1, (float) sample);
        uint32_t index = std::min((uint32_t) std::max((ptrdiff_t) 0, entry - cdf - 1), size - 1);
        sample = (sample - (Float) cdf[index]) / (Float) (cdf[index + 1] - cdf[index]);
        return index;
    }

    /* envmap.cpp:567-600 */
    void internalSampleDirection(Vec2 sample, Vec3 &d, Spectrum &value, Float &pdf) const {
        uint32_t row = sampleReuse(cdfRows.data(), (uint32_t) h, sample.y),
                 col = sampleReuse(cdfCols.data() + (size_t) row * (w + 1), (uint32_t) w, sample.x);
        Vec2 pos = Vec2((Float) col + intervalToTent(sample.x), (Float) row + intervalToTent(sample.y));
        int xPos = (int) std::floor(pos.x), yPos = (int) std::floor(pos.y);
        Float dx1 = pos.x - xPos, dx2 = 1.0f - dx1, dy1 = pos.y - yPos, dy2 = 1.0f - dy1;
        Spectrum value1 = evalTexel(xPos, yPos) * dx2 * dy2 + evalTexel(xPos + 1, yPos) * dx1 * dy2;
        Spectrum value2 = evalTexel(xPos, yPos + 1) * dx2 * dy1 + evalTexel(xPos + 1, yPos + 1) * dx1 * dy1;
        value = (value1 + value2) * scale;
        pdf = (rgbLuminance(value1) * rowWeights[std::min(std::max(yPos, 0), h - 1)] +
               rgbLuminance(value2) * rowWeights[std::min(std::max(yPos + 1, 0), h - 1)]) * normalization;
        Float sinPhi, cosPhi, sinTheta, cosTheta;
        om::sincos(pixelSizeX * (pos.x + 0.5f), &sinPhi, &cosPhi);
        om::sincos(pixelSizeY * (pos.y + 0.5f), &sinTheta, &cosTheta);
        d = Vec3(sinPhi * sinTheta, cosTheta, -cosPhi * sinTheta);
        pdf /= std::max(std::abs(sinTheta), ORC_EPSILON);
    }

    /* envmap.cpp:603-632 */
    Float internalPdfDirection(const Vec3 &d) const {
        Vec2 uv(om::atan2(d.x, -d.z) * ORC_INV_TWOPI, om::acos(std::min(1.0f, std::max(-1.0f, d.y))) * ORC_INV_PI);
        if (!std::isfinite(uv.x) || !std::isfinite(uv.y)) return 0.0f;
        Float u = uv.x * w - 0.5f, v = uv.y * h - 0.5f;
        int xPos = (int) std::floor(u), yPos = (int) std::floor(v);
        Float dx1 = u - xPos, dx2 = 1.0f - dx1, dy1 = v - yPos, dy2 = 1.0f - dy1;
        Spectrum value1 = evalTexel(xPos, yPos) * dx2 * dy2 + evalTexel(xPos + 1, yPos) * dx1 * dy2;
        Spectrum value2 = evalTexel(xPos, yPos + 1) * dx2 * dy1 + evalTexel(xPos + 1, yPos + 1) * dx1 * dy1;
        Float sinTheta = om::safe_sqrt(1 - d.y * d.y);
        return (rgbLuminance(value1) * rowWeights[std::min(std::max(yPos, 0), h - 1)] +
                rgbLuminance(value2) * rowWeights[std::min(std::max(yPos + 1, 0), h - 1)])
            * normalization / std::max(std::abs(sinTheta), ORC_EPSILON);
    }
};

} // namespace orc
