// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.hpp).
// orc_reflection.hpp: src/reflection/{mod,bsdf,microfacet}.rs, src/fresnel.rs, src/material/*.rs
#pragma once
#include "orc_shapes.hpp"
#include "../include/fountain_hip.h"

namespace orc {

// BxDFType bitflags: src/reflection/mod.rs:14-22
enum : uint8_t { BSDF_REFLECTION = 1, BSDF_TRANSMISSION = 2, BSDF_DIFFUSE = 4, BSDF_GLOSSY = 8, BSDF_SPECULAR = 16, BSDF_ALL = 31 };

// :24-73
inline Float cos_theta(Vec3 w) { return w.z; }
inline Float cos2_theta(Vec3 w) { return w.z * w.z; }
inline Float abs_cos_theta(Vec3 w) { return fabsf(w.z); }
inline Float sin2_theta(Vec3 w) { return fmax_(0.0f, 1.0f - cos2_theta(w)); }
inline Float sin_theta(Vec3 w) { return sqrtf(sin2_theta(w)); }
inline Float tan_theta(Vec3 w) { return sin_theta(w) / cos_theta(w); }
inline Float tan2_theta(Vec3 w) { return sin2_theta(w) / cos2_theta(w); }
inline Float cos_phi(Vec3 w) { Float s = sin_theta(w); return s == 0.0f ? 1.0f : clampf(w.x / s, -1.0f, 1.0f); }
inline Float sin_phi(Vec3 w) { Float s = sin_theta(w); return s == 0.0f ? 0.0f : clampf(w.y / s, -1.0f, 1.0f); }
inline Float cos2_phi(Vec3 w) { return cos_phi(w) * cos_phi(w); }
inline Float sin2_phi(Vec3 w) { return sin_phi(w) * sin_phi(w); }
// refract: :75-83
inline bool refract(Vec3 wi, Vec3 n, Float eta, Vec3* wt) {
    Float cos_theta_i = dot(n, wi);
    Float sin2_theta_i = fmax_(0.0f, 1.0f - cos_theta_i * cos_theta_i);
    Float sin2_theta_t = eta * eta * sin2_theta_i;
    if (sin2_theta_t >= 1.0f) return false;
    Float cos_theta_t = sqrtf(1.0f - sin2_theta_t);
    *wt = eta * -wi + (eta * cos_theta_i - cos_theta_t) * n;
    return true;
}
inline Vec3 reflect(Vec3 wo, Vec3 n) { return -wo + 2.0f * dot(wo, n) * n; }   // :85-87
inline bool same_hemisphere(Vec3 a, Vec3 b) { return is_sign_positive(a.z) == is_sign_positive(b.z); }  // :89-91

// ---- Fresnel: src/fresnel.rs
inline Float fresnel_dielectric(Float cos_theta_i, Float eta_i, Float eta_t) {     // :4-22
    cos_theta_i = clampf(cos_theta_i, -1.0f, 1.0f);
    bool entering = cos_theta_i > 0.0f;
    if (!entering) { std::swap(eta_i, eta_t); cos_theta_i = fabsf(cos_theta_i); }
    Float sin_theta_i = sqrtf(fmax_(1.0f - cos_theta_i * cos_theta_i, 0.0f));
    Float sin_theta_t = eta_i / eta_t * sin_theta_i;
    if (sin_theta_t >= 1.0f) return 1.0f;
    Float cos_theta_t = sqrtf(fmax_(1.0f - sin_theta_t * sin_theta_t, 0.0f));
    Float r_parallel = ((eta_t * cos_theta_i) - (eta_i * cos_theta_t)) / ((eta_t * cos_theta_i) + (eta_i * cos_theta_t));
    Float r_perp = ((eta_i * cos_theta_i) - (eta_t * cos_theta_t)) / ((eta_i * cos_theta_i) + (eta_t * cos_theta_t));
    return (r_parallel * r_parallel + r_perp * r_perp) / 2.0f;
}
inline Spectrum fresnel_conductor(Float cos_theta_i, Spectrum eta_i, Spectrum eta_t, Spectrum k) {  // :25-48
    cos_theta_i = clampf(cos_theta_i, -1.0f, 1.0f);
    Spectrum eta = eta_t / eta_i;
    Spectrum eta_k = k / eta_i;
    Float cos_theta_i2 = cos_theta_i * cos_theta_i;
    Float sin_theta_i2 = 1.0f - cos_theta_i2;
    Spectrum eta2 = eta * eta;
    Spectrum eta_k2 = eta_k * eta_k;
    Spectrum t0 = eta2 - eta_k2 - sin_theta_i2;
    Spectrum a2plusb2 = (t0 * t0 + 4.0f * eta2 * eta_k2).sqrt();
    Spectrum t1 = a2plusb2 + cos_theta_i2;
    Spectrum a = (0.5f * (a2plusb2 + t0)).sqrt();
    Spectrum t2 = 2.0f * cos_theta_i * a;
    Spectrum Rs = (t1 - t2) / (t1 + t2);
    Spectrum t3 = cos_theta_i2 * a2plusb2 + sin_theta_i2 * sin_theta_i2;
    Spectrum t4 = t2 * sin_theta_i2;
    Spectrum Rp = Rs * (t3 - t4) / (t3 + t4);
    return 0.5f * (Rp + Rs);
}
struct Fresnel {
    enum Kind { NOOP, DIELECTRIC, CONDUCTOR } kind = NOOP;
    Float d_eta_i = 1, d_eta_t = 1;
    Spectrum c_eta_i, c_eta_t, c_k;
    Spectrum evaluate(Float cos_i) const {                                 // :67-103
        switch (kind) {
            case DIELECTRIC: return Spectrum(fresnel_dielectric(cos_i, d_eta_i, d_eta_t));
            case CONDUCTOR: return fresnel_conductor(fabsf(cos_i), c_eta_i, c_eta_t, c_k);
            default: return Spectrum(1.0f);
        }
    }
};

// ---- TrowbridgeReitzDistribution: src/reflection/microfacet.rs:119-187 (Beckmann is never instantiated)
inline Float roughness_to_alpha(Float roughness) {                         // :40-45
    Float rough = fmax_(roughness, 1.0e-3f);
    Float x = m_ln(rough);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}
struct TrowbridgeReitz {
    Float alpha_x, alpha_y;
    Float d(Vec3 wh) const {                                               // :131-142
        Float t2 = tan2_theta(wh);
        if (std::isinf(t2)) return 0.0f;
        Float cos4_theta = cos2_theta(wh) * cos2_theta(wh);
        Float e = (cos2_phi(wh) / (alpha_x * alpha_x) + sin2_phi(wh) / (alpha_y * alpha_y)) * t2;
        return 1.0f / (PI * alpha_x * alpha_y * cos4_theta * (1.0f + e) * (1.0f + e));
    }
    Float lambda(Vec3 w) const {                                           // :144-156
        Float abs_tan_theta = fabsf(tan_theta(w));
        if (std::isinf(abs_tan_theta)) return 0.0f;
        Float alpha = sqrtf(cos2_phi(w) * alpha_x * alpha_x + sin2_phi(w) * alpha_y * alpha_y);
        Float a2t2 = (alpha * abs_tan_theta) * (alpha * abs_tan_theta);
        return (-1.0f + sqrtf(1.0f + a2t2)) / 2.0f;
    }
    Float g(Vec3 wo, Vec3 wi) const { return 1.0f / (1.0f + lambda(wo) + lambda(wi)); }   // :22-24
    Float pdf(Vec3 /*wo*/, Vec3 wh) const { return d(wh) * abs_cos_theta(wh); }            // :29-32
    Vec3 sample_wh(Vec3 wo, Vec2 u) const {                                // :158-186
        Float cos_t, phi;
        if (alpha_x == alpha_y) {
            Float tan_theta2 = (alpha_x * alpha_x) * u.x / (1.0f - u.x);
            cos_t = 1.0f / sqrtf(1.0f + tan_theta2);
            phi = 2.0f * PI * u.y;
        } else {
            phi = m_atan(alpha_y / alpha_x * m_tan(2.0f * PI * u.y + 0.5f * PI));
            if (u.y > 0.5f) phi += PI;
            Float sp = m_sin(phi), cp = m_cos(phi);
            Float alpha2 = 1.0f / ((cp * cp) / (alpha_x * alpha_x) + (sp * sp) / (alpha_y * alpha_y));
            Float tan_theta2 = alpha2 * u.x / (1.0f - u.x);
            cos_t = 1.0f / sqrtf(1.0f + tan_theta2);
        }
        Float sin_t = sqrtf(fmax_(0.0f, 1.0f - (cos_t * cos_t)));
        Vec3 wh = spherical_direction(sin_t, cos_t, phi);
        return same_hemisphere(wo, wh) ? wh : -wh;
    }
};

struct ScatterSample { Spectrum f; Vec3 wi; Float pdf; uint8_t sampled_type; };

// ---- BxDFs: src/reflection/mod.rs:101-439
struct BxDF {
    enum Kind { LAMBERTIAN, OREN_NAYAR, SPECULAR_REFLECTION, SPECULAR_TRANSMISSION, MICROFACET_REFLECTION, MICROFACET_TRANSMISSION } kind;
    Spectrum r;            // r or t
    Float a = 0, b = 0;    // OrenNayar
    Fresnel fresnel;
    TrowbridgeReitz distribution{0, 0};
    Float eta_a = 1, eta_b = 1;   // transmission

    uint8_t get_type() const {
        switch (kind) {
            case LAMBERTIAN: case OREN_NAYAR: return BSDF_REFLECTION | BSDF_DIFFUSE;
            case SPECULAR_REFLECTION: return BSDF_REFLECTION | BSDF_SPECULAR;
            case SPECULAR_TRANSMISSION: return BSDF_TRANSMISSION | BSDF_SPECULAR;
            case MICROFACET_REFLECTION: return BSDF_REFLECTION | BSDF_GLOSSY;
            default: return BSDF_TRANSMISSION | BSDF_GLOSSY;
        }
    }
    bool matches_flags(uint8_t t) const { return (t & get_type()) == get_type(); }   // t.contains(self.get_type())
    Float get_eta(Vec3 wo) const { return cos_theta(wo) > 0.0f ? eta_b / eta_a : eta_a / eta_b; }   // :378-380

    Spectrum f(Vec3 wo, Vec3 wi) const {
        switch (kind) {
            case LAMBERTIAN: return r * FRAC_1_PI;                          // :159-161
            case OREN_NAYAR: {                                              // :273-297
                Float sin_theta_i = sin_theta(wi), sin_theta_o = sin_theta(wo);
                Float max_cos = 0.0f;
                if (sin_theta_i > 1.0e-4f && sin_theta_o > 1.0e-4f) {
                    Float d_cos = cos_phi(wi) * cos_phi(wo) + sin_phi(wi) * sin_phi(wo);
                    max_cos = fmax_(0.0f, d_cos);
                }
                Float sin_alpha, tan_beta;
                if (abs_cos_theta(wi) > abs_cos_theta(wo)) { sin_alpha = sin_theta_o; tan_beta = sin_theta_i / abs_cos_theta(wi); }
                else { sin_alpha = sin_theta_i; tan_beta = sin_theta_o / abs_cos_theta(wo); }
                return r * FRAC_1_PI * (a + (b * max_cos * sin_alpha * tan_beta));
            }
            case SPECULAR_REFLECTION: case SPECULAR_TRANSMISSION: return Spectrum(0.0f);
            case MICROFACET_REFLECTION: {                                   // :318-336
                Float cos_theta_o = abs_cos_theta(wo), cos_theta_i = abs_cos_theta(wi);
                Vec3 wh = wi + wo;
                if (cos_theta_i == 0.0f || cos_theta_o == 0.0f || (wh == Vec3(0, 0, 0))) return Spectrum(0.0f);
                wh = normalize(wh);
                Spectrum fr = fresnel.evaluate(dot(wi, faceforward(wh, Vec3(0, 0, 1))));
                return r * distribution.d(wh) * distribution.g(wo, wi) * fr / (4.0f * cos_theta_i * cos_theta_o);
            }
            default: {                                                      // MicrofacetTransmission :388-406 (mode = Radiance)
                if (same_hemisphere(wo, wi)) return Spectrum(0.0f);
                Float cos_theta_o = cos_theta(wo), cos_theta_i = cos_theta(wi);
                if (cos_theta_o == 0.0f || cos_theta_i == 0.0f) return Spectrum(0.0f);
                Float eta = get_eta(wo);
                Vec3 wh = normalize(wo + wi * eta);
                if (wh.z < 0.0f) wh = -wh;
                Spectrum fr = fresnel.evaluate(dot(wo, wh));
                Float sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
                Float factor = 1.0f / eta;
                return (Spectrum(1.0f) - fr) * r *
                       fabsf(distribution.d(wh) * distribution.g(wo, wi) * (eta * eta) * abs_dot(wi, wh) * abs_dot(wo, wh) * (factor * factor) /
                             (cos_theta_i * cos_theta_o * (sqrt_denom * sqrt_denom)));
            }
        }
    }
    Float pdf(Vec3 wo, Vec3 wi) const {
        switch (kind) {
            case LAMBERTIAN: case OREN_NAYAR:                               // :140-146
                return same_hemisphere(wo, wi) ? abs_cos_theta(wi) * FRAC_1_PI : 0.0f;
            case SPECULAR_REFLECTION: case SPECULAR_TRANSMISSION: return 0.0f;
            case MICROFACET_REFLECTION: {                                   // :354-360
                if (!same_hemisphere(wo, wi)) return 0.0f;
                Vec3 wh = normalize(wo + wi);
                return distribution.pdf(wo, wh) / (4.0f * dot(wo, wh));
            }
            default: {                                                      // :429-438
                if (same_hemisphere(wo, wi)) return 0.0f;
                Float eta = get_eta(wo);
                Vec3 wh = normalize(wo + wi * eta);
                Float sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
                Float dwh_dwi = fabsf(((eta * eta) * dot(wi, wh)) / (sqrt_denom * sqrt_denom));
                return distribution.pdf(wo, wh) * dwh_dwi;
            }
        }
    }
    bool sample_f(Vec3 wo, Vec2 sample, ScatterSample* out) const {
        switch (kind) {
            case LAMBERTIAN: case OREN_NAYAR: {                             // :130-138
                Vec3 wi = cosine_sample_hemisphere(sample);
                if (wo.z < 0.0f) wi.z *= -1.0f;
                out->pdf = pdf(wo, wi); out->f = f(wo, wi); out->wi = wi; out->sampled_type = get_type();
                return true;
            }
            case SPECULAR_REFLECTION: {                                     // :187-194
                Vec3 wi(-wo.x, -wo.y, wo.z);
                out->f = fresnel.evaluate(cos_theta(wi)) * r / abs_cos_theta(wi);
                out->wi = wi; out->pdf = 1.0f; out->sampled_type = get_type();
                return true;
            }
            case SPECULAR_TRANSMISSION: {                                   // :225-245
                bool entering = cos_theta(wo) > 0.0f;
                Float eta_i = entering ? eta_a : eta_b, eta_t = entering ? eta_b : eta_a;
                Vec3 n(0, 0, 1); if (dot(n, wo) < 0.0f) n = -n;             // Normal3::faceforward
                Vec3 wi;
                if (!refract(wo, n, eta_i / eta_t, &wi)) return false;
                Spectrum ft = r * (Spectrum(1.0f) - fresnel.evaluate(cos_theta(wi)));
                out->f = ft / abs_cos_theta(wi); out->wi = wi; out->pdf = 1.0f; out->sampled_type = get_type();
                return true;
            }
            case MICROFACET_REFLECTION: {                                   // :338-352
                Vec3 wh = distribution.sample_wh(wo, sample);
                Vec3 wi = reflect(wo, wh);
                if (!same_hemisphere(wo, wi)) return false;
                out->pdf = distribution.pdf(wo, wh) / (4.0f * dot(wo, wh));
                out->f = f(wo, wi); out->wi = wi; out->sampled_type = get_type();
                return true;
            }
            default: {                                                      // :408-427
                if (wo.z == 0.0f) return false;
                Vec3 wh = distribution.sample_wh(wo, sample);
                if (dot(wo, wh) < 0.0f) return false;
                Float eta = get_eta(-wo);
                Vec3 wi;
                if (!refract(wo, wh, eta, &wi)) return false;
                out->f = f(wo, wi); out->wi = wi; out->pdf = pdf(wo, wi); out->sampled_type = get_type();
                return true;
            }
        }
    }
};

// ---- Bsdf: src/reflection/bsdf.rs
struct Bsdf {
    Float eta; Vec3 ns, ng, ss, ts;
    BxDF bxdfs[8]; int n_bxdfs = 0;
    Bsdf() : eta(1.0f) {}
    Bsdf(const SurfaceInteraction& si, Float eta_) {                       // :32-48
        eta = eta_; ns = si.shading_n; ng = si.hit.n;
        ss = normalize(si.shading_geom.dpdu);
        ts = normalize(cross(ns, ss));
    }
    void add(const BxDF& b) { bxdfs[n_bxdfs++] = b; }
    int num_components(uint8_t flags) const { int n = 0; for (int i = 0; i < n_bxdfs; i++) n += bxdfs[i].matches_flags(flags); return n; }
    Vec3 world_to_local(Vec3 v) const { return Vec3(dot(v, ss), dot(v, ts), dot(v, ns)); }
    Vec3 local_to_world(Vec3 v) const {
        return Vec3(ss.x * v.x + ts.x * v.y + ns.x * v.z, ss.y * v.x + ts.y * v.y + ns.y * v.z, ss.z * v.x + ts.z * v.y + ns.z * v.z);
    }
    Spectrum f(Vec3 wo_world, Vec3 wi_world, uint8_t flags) const {        // :67-82
        Vec3 wi = world_to_local(wi_world), wo = world_to_local(wo_world);
        if (wo.z == 0.0f) return Spectrum(0.0f);
        bool refl = dot(wi_world, ng) * dot(wo_world, ng) > 0.0f;
        Spectrum sum(0.0f);
        for (int i = 0; i < n_bxdfs; i++) {
            const BxDF& b = bxdfs[i];
            if (!b.matches_flags(flags)) continue;
            if ((refl && (b.get_type() & BSDF_REFLECTION)) || (!refl && (b.get_type() & BSDF_TRANSMISSION))) sum = sum + b.f(wo, wi);
        }
        return sum;
    }
    bool sample_f(Vec3 wo_world, Vec2 u, uint8_t flags, ScatterSample* out) const {   // :85-129
        Float matching_comps = (Float)num_components(flags);
        if (matching_comps == 0.0f) return false;
        int comp = (int)f2usize(fmin_(floorf(u.x * matching_comps), matching_comps - 1.0f));
        const BxDF* bxdf = nullptr; int cnt = comp;
        for (int i = 0; i < n_bxdfs; i++) if (bxdfs[i].matches_flags(flags)) { if (cnt-- == 0) { bxdf = &bxdfs[i]; break; } }
        Vec2 u_remapped(u.x * matching_comps - (Float)comp, u.y);
        Vec3 wo = world_to_local(wo_world);
        ScatterSample s;
        if (!bxdf->sample_f(wo, u_remapped, &s)) return false;
        Float pdf = s.pdf; Vec3 wi = s.wi; Spectrum f = s.f;
        if (pdf == 0.0f) return false;
        Vec3 wi_world = local_to_world(wi);
        if (!(bxdf->get_type() & BSDF_SPECULAR) && matching_comps > 1.0f) {
            Float extra = 0.0f;   // .sum::<Float>() starts from 0.0
            for (int i = 0; i < n_bxdfs; i++) if (bxdfs[i].matches_flags(flags) && &bxdfs[i] != bxdf) extra = extra + bxdfs[i].pdf(wo, wi);
            pdf += extra;
        }
        if (matching_comps > 1.0f) pdf /= matching_comps;
        if (!(bxdf->get_type() & BSDF_SPECULAR)) {
            bool refl = dot(wi_world, ng) * dot(wo_world, ng) > 0.0f;
            Spectrum sum(0.0f);
            for (int i = 0; i < n_bxdfs; i++) {
                const BxDF& b = bxdfs[i];
                if (!b.matches_flags(flags)) continue;
                if ((refl && (b.get_type() & BSDF_REFLECTION)) || (!refl && (b.get_type() & BSDF_TRANSMISSION))) sum = sum + b.f(wo, wi);
            }
            f = sum;
        }
        out->f = f; out->wi = wi_world; out->pdf = pdf; out->sampled_type = s.sampled_type;
        return true;
    }
    Float pdf(Vec3 wo_world, Vec3 wi_world, uint8_t flags) const {          // :131-144
        Vec3 wo = world_to_local(wo_world), wi = world_to_local(wi_world);
        if (wo.z == 0.0f) return 0.0f;
        Float n_matching = (Float)num_components(flags);
        Float pdf = 0.0f;
        for (int i = 0; i < n_bxdfs; i++) if (bxdfs[i].matches_flags(flags)) pdf = pdf + bxdfs[i].pdf(wo, wi);
        return n_matching > 0.0f ? pdf / n_matching : 0.0f;
    }
};

// ---- Materials: src/material/*.rs (all textures constant)
enum MaterialError { MAT_OK = 0, MAT_UNSUPPORTED_SPECULAR_GLASS = 1 };
inline MaterialError compute_scattering_functions(const ftn_material& m, const SurfaceInteraction& si, bool allow_multiple_lobes, Bsdf* bsdf_out) {
    Spectrum a(m.a[0], m.a[1], m.a[2]), b(m.b[0], m.b[1], m.b[2]);
    switch (m.type) {
        case FTN_MAT_MATTE: {                                               // matte.rs:35-52
            Bsdf bsdf(si, 1.0f);
            Spectrum r = a.clamp_positive();
            Float sigma = clampf(m.s0, 0.0f, 90.0f);
            if (!r.is_black()) {
                BxDF x;
                if (sigma == 0.0f) { x.kind = BxDF::LAMBERTIAN; x.r = r; }
                else {                                                      // OrenNayar::new(r, Deg(sigma)) mod.rs:260-267
                    Float sg = deg_to_rad_cgmath(sigma);
                    Float sigma2 = sg * sg;
                    x.kind = BxDF::OREN_NAYAR; x.r = r;
                    x.a = 1.0f - (sigma2 / (2.0f * (sigma2 + 0.33f)));
                    x.b = 0.45f * sigma2 / (sigma2 + 0.09f);
                }
                bsdf.add(x);
            }
            *bsdf_out = bsdf; return MAT_OK;
        }
        case FTN_MAT_METAL: {                                               // metal.rs:37-65
            Float u_rough = m.s1, v_rough = m.s2;
            if (m.remap_roughness) { u_rough = roughness_to_alpha(u_rough); v_rough = roughness_to_alpha(v_rough); }
            Bsdf bsdf(si, 1.0f);
            BxDF x; x.kind = BxDF::MICROFACET_REFLECTION; x.r = Spectrum(1.0f);
            x.distribution = TrowbridgeReitz{u_rough, v_rough};
            x.fresnel.kind = Fresnel::CONDUCTOR; x.fresnel.c_eta_i = Spectrum(1.0f); x.fresnel.c_eta_t = a; x.fresnel.c_k = b;
            bsdf.add(x);
            *bsdf_out = bsdf; return MAT_OK;
        }
        case FTN_MAT_MIRROR: {                                              // mirror.rs:21-30
            Bsdf bsdf(si, 1.0f);
            Spectrum r = a.clamp_positive();
            if (!r.is_black()) { BxDF x; x.kind = BxDF::SPECULAR_REFLECTION; x.r = r; x.fresnel.kind = Fresnel::NOOP; bsdf.add(x); }
            *bsdf_out = bsdf; return MAT_OK;
        }
        case FTN_MAT_PLASTIC: {                                             // plastic.rs:24-48
            Bsdf bsdf(si, 1.0f);
            if (!a.is_black()) { BxDF x; x.kind = BxDF::LAMBERTIAN; x.r = a; bsdf.add(x); }
            if (!b.is_black()) {
                Float rough = m.s1;
                if (m.remap_roughness) rough = roughness_to_alpha(rough);
                BxDF x; x.kind = BxDF::MICROFACET_REFLECTION; x.r = b;
                x.distribution = TrowbridgeReitz{rough, rough};
                x.fresnel.kind = Fresnel::DIELECTRIC; x.fresnel.d_eta_i = 1.5f; x.fresnel.d_eta_t = 1.0f;
                bsdf.add(x);
            }
            *bsdf_out = bsdf; return MAT_OK;
        }
        case FTN_MAT_GLASS: {                                               // glass.rs:51-93
            Float eta = m.s0;
            Spectrum r = a.clamp_positive(), t = b.clamp_positive();
            Float u_rough = m.s1, v_rough = m.s2;
            if (m.remap_roughness) { u_rough = roughness_to_alpha(u_rough); v_rough = roughness_to_alpha(v_rough); }
            Bsdf bsdf(si, eta);
            bool is_specular = u_rough == 0.0f && v_rough == 0.0f;
            if (is_specular && allow_multiple_lobes) return MAT_UNSUPPORTED_SPECULAR_GLASS;   // todo!("FresnelSpecular")
            if (!r.is_black()) {
                BxDF x; x.r = r; x.fresnel.kind = Fresnel::DIELECTRIC; x.fresnel.d_eta_i = 1.0f; x.fresnel.d_eta_t = eta;
                if (is_specular) x.kind = BxDF::SPECULAR_REFLECTION;
                else { x.kind = BxDF::MICROFACET_REFLECTION; x.distribution = TrowbridgeReitz{u_rough, v_rough}; }
                bsdf.add(x);
            }
            if (!t.is_black()) {
                BxDF x; x.r = t; x.eta_a = 1.0f; x.eta_b = eta;
                x.fresnel.kind = Fresnel::DIELECTRIC; x.fresnel.d_eta_i = 1.0f; x.fresnel.d_eta_t = eta;
                if (is_specular) x.kind = BxDF::SPECULAR_TRANSMISSION;
                else { x.kind = BxDF::MICROFACET_TRANSMISSION; x.distribution = TrowbridgeReitz{u_rough, v_rough}; }
                bsdf.add(x);
            }
            *bsdf_out = bsdf; return MAT_OK;
        }
    }
    *bsdf_out = Bsdf(si, 1.0f);
    return MAT_OK;
}

}  // namespace orc
