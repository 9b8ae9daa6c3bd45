/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see o_math.h).
 *
 * o_bsdf.h: the BSDF models on the path, restating (file:line under /root/reference):
 *   src/bsdfs/diffuse.cpp:110-150            SmoothDiffuse eval / pdf / sample
 *   src/bsdfs/dielectric.cpp:217-226,277-333 SmoothDielectric reflect / refract / sample
 *   src/bsdfs/roughconductor.cpp:253-415     RoughConductor eval / pdf / sample
 *   src/bsdfs/microfacet.h:191-234           MicrofacetDistribution::eval
 *   src/bsdfs/microfacet.h:240-249,287-395   sample / sampleAll
 *   src/bsdfs/microfacet.h:421-522           sampleVisible, pdfVisible, smithG1, G
 *   src/bsdfs/microfacet.h:541-551,573-697   projectRoughness, sampleVisible11 (Beckmann + GGX)
 *   src/bsdfs/twosided.cpp:108-183           TwoSidedBRDF eval / pdf / sample
 * Only ERadiance transport and (typeMask = EAll, component = -1) queries occur on the path.
 */
#pragma once
#include "o_scene.h"

namespace orc {

struct BSDFSamplingRecord {
    Vec3 wi, wo;
    Float eta;
    bool sampledDelta;   /* sampledType & EDelta */
};

struct MicrofacetDistribution {
    int type; Float alphaU, alphaV; bool sampleVisibleFlag;
    MicrofacetDistribution(int type_, Float aU, Float aV, bool sv) : type(type_), alphaU(aU), alphaV(aV), sampleVisibleFlag(sv) {
        alphaU = std::max(alphaU, (Float) 1e-4f);
        alphaV = std::max(alphaV, (Float) 1e-4f);
    }
    bool isIsotropic() const { return alphaU == alphaV; }

    Float eval(const Vec3 &m) const { /* microfacet.h:191-234 */
        if (Frame::cosTheta(m) <= 0)
            return 0.0f;
        Float cosTheta2 = Frame::cosTheta2(m);
        Float beckmannExponent = ((m.x * m.x) / (alphaU * alphaU) + (m.y * m.y) / (alphaV * alphaV)) / cosTheta2;
        Float result;
        if (type == PHIP_MF_BECKMANN) {
            result = om::fastexp(-beckmannExponent) / (ORC_PI * alphaU * alphaV * cosTheta2 * cosTheta2);
        } else {
            Float root = ((Float) 1 + beckmannExponent) * cosTheta2;
            result = (Float) 1 / (ORC_PI * alphaU * alphaV * root * root);
        }
        if (result * Frame::cosTheta(m) < 1e-20f)
            result = 0;
        return result;
    }

    Float projectRoughness(const Vec3 &v) const { /* microfacet.h:541-551 */
        Float invSinTheta2 = 1 / Frame::sinTheta2(v);
        if (isIsotropic() || invSinTheta2 <= 0)
            return alphaU;
        Float cosPhi2 = v.x * v.x * invSinTheta2;
        Float sinPhi2 = v.y * v.y * invSinTheta2;
        return std::sqrt(cosPhi2 * alphaU * alphaU + sinPhi2 * alphaV * alphaV);
    }

    Float smithG1(const Vec3 &v, const Vec3 &m) const { /* microfacet.h:477-517 */
        if (dot(v, m) * Frame::cosTheta(v) <= 0)
            return 0.0f;
        Float tanTheta = std::abs(Frame::tanTheta(v));
        if (tanTheta == 0.0f)
            return 1.0f;
        Float alpha = projectRoughness(v);
        if (type == PHIP_MF_BECKMANN) {
            Float a = 1.0f / (alpha * tanTheta);
            if (a >= 1.6f)
                return 1.0f;
            Float aSqr = a * a;
            return (3.535f * a + 2.181f * aSqr) / (1.0f + 2.276f * a + 2.577f * aSqr);
        } else {
            Float root = alpha * tanTheta;
            return 2.0f / (1.0f + hypot2((Float) 1.0f, root));
        }
    }
    Float G(const Vec3 &wi, const Vec3 &wo, const Vec3 &m) const { return smithG1(wi, m) * smithG1(wo, m); }

    Float pdfVisible(const Vec3 &wi, const Vec3 &m) const { /* microfacet.h:462-466 */
        if (Frame::cosTheta(wi) == 0)
            return 0.0f;
        return smithG1(wi, m) * absDot(wi, m) * eval(m) / std::abs(Frame::cosTheta(wi));
    }
    Float pdfAll(const Vec3 &m) const { return eval(m) * Frame::cosTheta(m); }
    Float pdf(const Vec3 &wi, const Vec3 &m) const { return sampleVisibleFlag ? pdfVisible(wi, m) : pdfAll(m); }

    Vec3 sampleAll(const Vec2 &sample, Float &pdf) const { /* microfacet.h:287-395 */
        Float cosThetaM = 0.0f;
        Float sinPhiM, cosPhiM;
        Float alphaSqr;
        if (isIsotropic()) {
            om::sincos((2.0f * ORC_PI) * sample.y, &sinPhiM, &cosPhiM);
            alphaSqr = alphaU * alphaU;
        } else {
            Float phiM = om::atan(alphaV / alphaU * om::tan(ORC_PI + 2 * ORC_PI * sample.y)) + ORC_PI * std::floor(2 * sample.y + 0.5f);
            om::sincos(phiM, &sinPhiM, &cosPhiM);
            Float cosSc = cosPhiM / alphaU, sinSc = sinPhiM / alphaV;
            alphaSqr = 1.0f / (cosSc * cosSc + sinSc * sinSc);
        }
        if (type == PHIP_MF_BECKMANN) {
            Float tanThetaMSqr = alphaSqr * -om::fastlog(1.0f - sample.x);
            cosThetaM = 1.0f / std::sqrt(1.0f + tanThetaMSqr);
            pdf = (1.0f - sample.x) / (ORC_PI * alphaU * alphaV * cosThetaM * cosThetaM * cosThetaM);
        } else {
            Float tanThetaMSqr = alphaSqr * sample.x / (1.0f - sample.x);
            cosThetaM = 1.0f / std::sqrt(1.0f + tanThetaMSqr);
            Float temp = 1 + tanThetaMSqr / alphaSqr;
            pdf = ORC_INV_PI / (alphaU * alphaV * cosThetaM * cosThetaM * cosThetaM * temp * temp);
        }
        if (pdf < 1e-20f)
            pdf = 0;
        Float sinThetaM = std::sqrt(std::max((Float) 0, 1 - cosThetaM * cosThetaM));
        return Vec3(sinThetaM * cosPhiM, sinThetaM * sinPhiM, cosThetaM);
    }

    Vec2 sampleVisible11(Float thetaI, Vec2 sample) const { /* microfacet.h:573-697 */
        const Float SQRT_PI_INV = 1 / std::sqrt(ORC_PI);
        Vec2 slope;
        if (type == PHIP_MF_BECKMANN) {
            if (thetaI < 1e-4f) {
                Float sinPhi, cosPhi;
                Float r = std::sqrt(-om::fastlog(1.0f - sample.x));
                om::sincos(2 * ORC_PI * sample.y, &sinPhi, &cosPhi);
                return Vec2(r * cosPhi, r * sinPhi);
            }
            Float tanThetaI = om::tan(thetaI);
            Float cotThetaI = 1 / tanThetaI;
            Float a = -1, c = mts_erf(cotThetaI);
            Float sample_x = std::max(sample.x, (Float) 1e-6f);
            Float fit = 1 + thetaI * (-0.876f + thetaI * (0.4265f - 0.0594f * thetaI));
            Float b = c - (1 + c) * om::pow(1 - sample_x, fit);
            Float normalization = 1 / (1 + c + SQRT_PI_INV * tanThetaI * om::exp(-cotThetaI * cotThetaI));
            int it = 0;
            while (++it < 10) {
                if (!(b >= a && b <= c))
                    b = 0.5f * (a + c);
                Float invErf = mts_erfinv(b);
                Float value = normalization * (1 + b + SQRT_PI_INV * tanThetaI * om::exp(-invErf * invErf)) - sample_x;
                Float derivative = normalization * (1 - invErf * tanThetaI);
                if (std::abs(value) < 1e-5f)
                    break;
                if (value > 0)
                    c = b;
                else
                    a = b;
                b -= value / derivative;
            }
            slope.x = mts_erfinv(b);
            slope.y = mts_erfinv(2.0f * std::max(sample.y, (Float) 1e-6f) - 1.0f);
        } else {
            if (thetaI < 1e-4f) {
                Float sinPhi, cosPhi;
                Float r = om::safe_sqrt(sample.x / (1 - sample.x));
                om::sincos(2 * ORC_PI * sample.y, &sinPhi, &cosPhi);
                return Vec2(r * cosPhi, r * sinPhi);
            }
            Float tanThetaI = om::tan(thetaI);
            Float a = 1 / tanThetaI;
            Float G1 = 2.0f / (1.0f + om::safe_sqrt(1.0f + 1.0f / (a * a)));
            Float A = 2.0f * sample.x / G1 - 1.0f;
            if (std::abs(A) == 1)
                A -= om::signum(A) * ORC_EPSILON;
            Float tmp = 1.0f / (A * A - 1.0f);
            Float B = tanThetaI;
            Float D = om::safe_sqrt(B * B * tmp * tmp - (A * A - B * B) * tmp);
            Float slope_x_1 = B * tmp - D;
            Float slope_x_2 = B * tmp + D;
            slope.x = (A < 0.0f || slope_x_2 > 1.0f / tanThetaI) ? slope_x_1 : slope_x_2;
            Float S;
            if (sample.y > 0.5f) { S = 1.0f; sample.y = 2.0f * (sample.y - 0.5f); }
            else { S = -1.0f; sample.y = 2.0f * (0.5f - sample.y); }
            Float z = (sample.y * (sample.y * (sample.y * (-(Float) 0.365728915865723) + (Float) 0.790235037209296) -
                        (Float) 0.424965825137544) + (Float) 0.000152998850436920) /
                      (sample.y * (sample.y * (sample.y * (sample.y * (Float) 0.169507819808272 - (Float) 0.397203533833404) -
                        (Float) 0.232500544458471) + (Float) 1) - (Float) 0.539825872510702);
            slope.y = S * z * std::sqrt(1.0f + slope.x * slope.x);
        }
        return slope;
    }

    Vec3 sampleVisible(const Vec3 &_wi, const Vec2 &sample) const { /* microfacet.h:421-459 */
        Vec3 wi = normalize(Vec3(alphaU * _wi.x, alphaV * _wi.y, _wi.z));
        Float theta = 0, phi = 0;
        if (wi.z < (Float) 0.99999) {
            theta = om::acos(wi.z);
            phi = om::atan2(wi.y, wi.x);
        }
        Float sinPhi, cosPhi;
        om::sincos(phi, &sinPhi, &cosPhi);
        Vec2 slope = sampleVisible11(theta, sample);
        slope = Vec2(cosPhi * slope.x - sinPhi * slope.y, sinPhi * slope.x + cosPhi * slope.y);
        slope.x *= alphaU;
        slope.y *= alphaV;
        Float normalization = (Float) 1 / std::sqrt(slope.x * slope.x + slope.y * slope.y + (Float) 1.0);
        return Vec3(-slope.x * normalization, -slope.y * normalization, normalization);
    }

    Vec3 sample(const Vec3 &wi, const Vec2 &sample, Float &pdf) const { /* microfacet.h:240-249 */
        Vec3 m;
        if (sampleVisibleFlag) {
            m = sampleVisible(wi, sample);
            pdf = pdfVisible(wi, m);
        } else {
            m = sampleAll(sample, pdf);
        }
        return m;
    }
};

class BSDF {
public:
    const Scene &scene;
    const Intersection *its = nullptr;     /* the vertex being shaded: textures are evaluated there (bRec.its in the reference) */
    explicit BSDF(const Scene &s) : scene(s) {}

    /* m_reflectance / m_specularReflectance ->eval(bRec.its): ConstantSpectrumTexture or a `bitmap` texture (texture.cpp:112-121) */
    Spectrum diffuseReflectance(const Material &M) const {
        if (M.m.reflectance_texture != 0 && its)
            return scene.textures[M.m.reflectance_texture - 1].eval(*its);
        return Spectrum(M.m.reflectance);
    }

    /* m_specularTransmittance->eval(bRec.its), dielectric.cpp:307,363 */
    Spectrum specularTransmittance(const Material &M) const {
        if (M.m.transmittance_texture != 0 && its)
            return scene.textures[M.m.transmittance_texture - 1].eval(*its);
        return Spectrum(M.m.transmittance);
    }
    /* m_alphaU / m_alphaV ->eval(bRec.its).average(), roughconductor.cpp:275-280 (the MicrofacetDistribution constructor clamps) */
    Float alphaU(const Material &M) const {
        if (M.m.alpha_u_texture != 0 && its) return scene.textures[M.m.alpha_u_texture - 1].eval(*its).average();
        return M.alphaU;
    }
    Float alphaV(const Material &M) const {
        if (M.m.alpha_v_texture != 0 && its) return scene.textures[M.m.alpha_v_texture - 1].eval(*its).average();
        return M.alphaV;
    }

    /* ---- diffuse.cpp:110-150 ---- */
    Spectrum diffuseEval(const Material &M, const Vec3 &wi, const Vec3 &wo) const {
        if (Frame::cosTheta(wi) <= 0 || Frame::cosTheta(wo) <= 0)
            return Spectrum(0.0f);
        return diffuseReflectance(M) * (ORC_INV_PI * Frame::cosTheta(wo));
    }
    static Float diffusePdf(const Material &, const Vec3 &wi, const Vec3 &wo) {
        if (Frame::cosTheta(wi) <= 0 || Frame::cosTheta(wo) <= 0)
            return 0.0f;
        return squareToCosineHemispherePdf(wo);
    }
    Spectrum diffuseSample(const Material &M, BSDFSamplingRecord &bRec, Float &pdf, const Vec2 &sample) const {
        if (Frame::cosTheta(bRec.wi) <= 0)
            return Spectrum(0.0f);
        bRec.wo = squareToCosineHemisphere(sample);
        bRec.eta = 1.0f;
        bRec.sampledDelta = false;
        pdf = squareToCosineHemispherePdf(bRec.wo);
        return diffuseReflectance(M);
    }

    /* ---- dielectric.cpp:217-226,277-333 ---- */
    Spectrum dielectricSample(const Material &M, BSDFSamplingRecord &bRec, Float &pdf, const Vec2 &sample) const {
        const Float eta = M.m.eta[0], invEta = 1 / eta;
        Float cosThetaT;
        Float F = fresnelDielectricExt(Frame::cosTheta(bRec.wi), cosThetaT, eta);
        bRec.sampledDelta = true;
        if (sample.x <= F) {
            bRec.wo = Vec3(-bRec.wi.x, -bRec.wi.y, bRec.wi.z);
            bRec.eta = 1.0f;
            pdf = F;
            return diffuseReflectance(M);          /* m_specularReflectance->eval(bRec.its), dielectric.cpp:300 */
        } else {
            Float scale = -(cosThetaT < 0 ? invEta : eta);
            bRec.wo = Vec3(scale * bRec.wi.x, scale * bRec.wi.y, cosThetaT);
            bRec.eta = cosThetaT < 0 ? eta : invEta;
            pdf = 1 - F;
            Float factor = cosThetaT < 0 ? invEta : eta;   /* ERadiance */
            return specularTransmittance(M) * (factor * factor);
        }
    }

    /* ---- roughconductor.cpp:253-415 ---- */
    static Vec3 reflect(const Vec3 &wi, const Vec3 &m) { return 2 * dot(wi, m) * m - wi; }

    Spectrum roughEval(const Material &M, const Vec3 &wi, const Vec3 &wo) const {
        if (Frame::cosTheta(wi) <= 0 || Frame::cosTheta(wo) <= 0)
            return Spectrum(0.0f);
        Vec3 H = normalize(wo + wi);
        MicrofacetDistribution distr((int) M.m.distribution, alphaU(M), alphaV(M), M.m.sample_visible != 0);
        const Float D = distr.eval(H);
        if (D == 0)
            return Spectrum(0.0f);
        const Spectrum F = fresnelConductorExact(dot(wi, H), Spectrum(M.m.eta), Spectrum(M.m.k)) * diffuseReflectance(M);
        const Float G = distr.G(wi, wo, H);
        Float model = D * G / (4.0f * Frame::cosTheta(wi));
        return F * model;
    }
    Float roughPdf(const Material &M, const Vec3 &wi, const Vec3 &wo) const {
        if (Frame::cosTheta(wi) <= 0 || Frame::cosTheta(wo) <= 0)
            return 0.0f;
        Vec3 H = normalize(wo + wi);
        MicrofacetDistribution distr((int) M.m.distribution, alphaU(M), alphaV(M), M.m.sample_visible != 0);
        if (M.m.sample_visible)
            return distr.eval(H) * distr.smithG1(wi, H) / (4.0f * Frame::cosTheta(wi));
        else
            return distr.pdf(wi, H) / (4 * absDot(wo, H));
    }
    Spectrum roughSample(const Material &M, BSDFSamplingRecord &bRec, Float &pdf, const Vec2 &sample) const {
        if (Frame::cosTheta(bRec.wi) < 0)
            return Spectrum(0.0f);
        MicrofacetDistribution distr((int) M.m.distribution, alphaU(M), alphaV(M), M.m.sample_visible != 0);
        Vec3 m = distr.sample(bRec.wi, sample, pdf);
        if (pdf == 0)
            return Spectrum(0.0f);
        bRec.wo = reflect(bRec.wi, m);
        bRec.eta = 1.0f;
        bRec.sampledDelta = false;
        if (Frame::cosTheta(bRec.wo) <= 0)
            return Spectrum(0.0f);
        Spectrum F = fresnelConductorExact(dot(bRec.wi, m), Spectrum(M.m.eta), Spectrum(M.m.k)) * diffuseReflectance(M);
        Float weight;
        if (M.m.sample_visible)
            weight = distr.smithG1(bRec.wo, m);
        else
            weight = distr.eval(m) * distr.G(bRec.wi, bRec.wo, m) * dot(bRec.wi, m) / (pdf * Frame::cosTheta(bRec.wi));
        pdf /= 4.0f * dot(bRec.wo, m);
        return F * weight;
    }

    /* ---- dispatch incl. twosided.cpp:108-183 ---- */
    Spectrum eval(const Material &M, const Vec3 &wi, const Vec3 &wo) const {
        switch (M.m.type) {
            case PHIP_BSDF_DIFFUSE: return diffuseEval(M, wi, wo);
            case PHIP_BSDF_ROUGHCONDUCTOR: return roughEval(M, wi, wo);
            case PHIP_BSDF_DIELECTRIC: return Spectrum(0.0f);   /* measure == ESolidAngle on a delta BSDF */
            case PHIP_BSDF_TWOSIDED: {
                if (Frame::cosTheta(wi) > 0)
                    return eval(scene.materials[M.m.nested[0]], wi, wo);
                Vec3 fwi = wi, fwo = wo; fwi.z *= -1; fwo.z *= -1;
                return eval(scene.materials[M.m.nested[1]], fwi, fwo);
            }
        }
        return Spectrum(0.0f);
    }
    Float pdf(const Material &M, const Vec3 &wi, const Vec3 &wo) const {
        switch (M.m.type) {
            case PHIP_BSDF_DIFFUSE: return diffusePdf(M, wi, wo);
            case PHIP_BSDF_ROUGHCONDUCTOR: return roughPdf(M, wi, wo);
            case PHIP_BSDF_DIELECTRIC: return 0.0f;
            case PHIP_BSDF_TWOSIDED: {
                if (wi.z > 0)
                    return pdf(scene.materials[M.m.nested[0]], wi, wo);
                Vec3 fwi = wi, fwo = wo; fwi.z *= -1; fwo.z *= -1;
                return pdf(scene.materials[M.m.nested[1]], fwi, fwo);
            }
        }
        return 0.0f;
    }
    Spectrum sample(const Material &M, BSDFSamplingRecord &bRec, Float &pdf, const Vec2 &smp) const {
        switch (M.m.type) {
            case PHIP_BSDF_DIFFUSE: return diffuseSample(M, bRec, pdf, smp);
            case PHIP_BSDF_ROUGHCONDUCTOR: return roughSample(M, bRec, pdf, smp);
            case PHIP_BSDF_DIELECTRIC: return dielectricSample(M, bRec, pdf, smp);
            case PHIP_BSDF_TWOSIDED: {
                bool flipped = false;
                if (Frame::cosTheta(bRec.wi) < 0) { bRec.wi.z *= -1; flipped = true; }
                Spectrum result = sample(scene.materials[M.m.nested[flipped ? 1 : 0]], bRec, pdf, smp);
                if (flipped) {
                    bRec.wi.z *= -1;
                    if (!result.isZero() && pdf != 0)
                        bRec.wo.z *= -1;
                }
                return result;
            }
        }
        return Spectrum(0.0f);
    }
};

} // namespace orc
