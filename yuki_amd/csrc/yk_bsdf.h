// yk_bsdf.h — closest-hit shading math for the wavefront `shade` kernel.
//
// The reference builds a `Bsdf` (a Vec of dyn BxDF lobes in a bump arena) per hit
// (materials/{matte,glass,metal,glossy}.rs -> materials/bsdfs/mod.rs:87-223).
// Here the lobe list is implied by the material tag, so `Bsdf::f` / `Bsdf::sample_f`
// collapse to a switch; the arithmetic of every lobe is kept operation for
// operation (file:line cited at each function).
#pragma once
#include "yk_libm.h"
#include "yk_math.h"

namespace yk {

#define YK_PI 3.14159265358979323846f
#define YK_FRAC_1_PI 0.318309886183790671537767526745028724f
#define YK_FRAC_PI_2 1.57079632679489661923132169163975144f
#define YK_FRAC_PI_4 0.785398163397448309615660845819875721f

// BxdfType bits, bsdfs/mod.rs:24-34
enum : int { BX_NONE = 0, BX_REFLECTION = 1, BX_TRANSMISSION = 2, BX_DIFFUSE = 4, BX_GLOSSY = 8, BX_SPECULAR = 16 };

// Device material record (44 B): yk_material_desc with the per-hit constants the
// reference recomputes on every hit folded in where that is bit-identical.
struct Material {
    unsigned kind;   // 0 matte(lambert) 1 glass 2 metal 3 glossy 4 matte(oren-nayar) 5 black (no lobes)
    float a[3];      // Kd | R | eta | Rs
    float b[3];      //    | T | k   |
    float c;         // oren-nayar A | glass eta | alpha (after remap / square / clamp)
    float d;         // oren-nayar B
    unsigned tex;    // matte: 1 + index of the ImageTexture that supplies Kd, 0 = constant
};
enum : unsigned { MK_LAMBERT = 0, MK_GLASS = 1, MK_METAL = 2, MK_GLOSSY = 3, MK_OREN_NAYAR = 4, MK_BLACK = 5 };

// sampling/mod.rs:68-87
YK_HD void concentric_sample_disk(float ux, float uy, float& dx, float& dy) {
    float ox = ux * 2.0f - 1.0f;
    float oy = uy * 2.0f - 1.0f;
    if (ox == 0.0f && oy == 0.0f) {
        dx = 0.0f;
        dy = 0.0f;
        return;
    }
    float theta, r;
    if (fabsf(ox) > fabsf(oy)) {
        theta = YK_FRAC_PI_4 * (oy / ox);
        r = ox;
    } else {
        theta = YK_FRAC_PI_2 - YK_FRAC_PI_4 * (ox / oy);
        r = oy;
    }
    float sn, cs;
    det_sincosf(theta, sn, cs);  // sinf(theta), cosf(theta) with the argument reduced once: the same bits as two calls
    dx = cs * r;
    dy = sn * r;
}
// sampling/mod.rs:62-66
YK_HD V3 cosine_sample_hemisphere(float ux, float uy) {
    float dx, dy;
    concentric_sample_disk(ux, uy, dx, dy);
    float z = sqrtf(rmax(1.0f - dx * dx - dy * dy, 0.0f));
    return V3{dx, dy, z};
}

// bsdfs/mod.rs:225-282 — local-frame trigonometry
YK_HD float cos_theta(V3 w) { return w.z; }
YK_HD float cos_2_theta(V3 w) { return w.z * w.z; }
YK_HD float sin_2_theta(V3 w) { return rmax(1.0f - cos_2_theta(w), 0.0f); }
YK_HD float sin_theta(V3 w) { return sqrtf(sin_2_theta(w)); }
YK_HD float tan_theta(V3 w) { return sin_theta(w) / cos_theta(w); }
YK_HD float tan_2_theta(V3 w) { return sin_2_theta(w) / cos_2_theta(w); }
YK_HD float sin_phi(V3 w) {
    float st = sin_theta(w);
    return st == 0.0f ? 1.0f : rclamp(w.y / st, -1.0f, 1.0f);
}
YK_HD float cos_phi(V3 w) {
    float st = sin_theta(w);
    return st == 0.0f ? 1.0f : rclamp(w.x / st, -1.0f, 1.0f);
}
YK_HD float sin_2_phi(V3 w) { return sin_phi(w) * sin_phi(w); }
YK_HD float cos_2_phi(V3 w) { return cos_phi(w) * cos_phi(w); }
YK_HD bool same_hemisphere(V3 w, V3 wp) { return w.z * wp.z > 0.0f; }

// fresnel.rs:22-51
YK_HD RGB fresnel_dielectric(float eta_i_in, float eta_t_in, float cos_theta_i) {
    cos_theta_i = rclamp(cos_theta_i, -1.0f, 1.0f);
    float eta_i = eta_i_in, eta_t = eta_t_in;
    if (!(cos_theta_i > 0.0f)) {
        eta_i = eta_t_in;
        eta_t = eta_i_in;
        cos_theta_i = fabsf(cos_theta_i);
    }
    float sin_theta_i = sqrtf(rmax(1.0f - cos_theta_i * cos_theta_i, 0.0f));
    float sin_theta_t = eta_i / eta_t * sin_theta_i;
    if (sin_theta_t >= 1.0f) return RGB{1.0f, 1.0f, 1.0f};
    float cos_theta_t = sqrtf(rmax(1.0f - sin_theta_t * sin_theta_t, 0.0f));
    float r_parallel = ((eta_t * cos_theta_i) - (eta_i * cos_theta_t)) / ((eta_t * cos_theta_i) + (eta_i * cos_theta_t));
    float r_perpendicular = ((eta_i * cos_theta_i) - (eta_t * cos_theta_t)) / ((eta_i * cos_theta_i) + (eta_t * cos_theta_t));
    return RGB{1.0f, 1.0f, 1.0f} * (r_parallel * r_parallel + r_perpendicular * r_perpendicular) / 2.0f;
}
// fresnel.rs:65-95 with eta_i = (1,1,1) as metal.rs:46-50 passes it
YK_HD RGB fresnel_conductor(RGB eta_t, RGB k, float cos_theta_i) {
    const RGB eta_i = RGB{1.0f, 1.0f, 1.0f};
    cos_theta_i = rmin(fabsf(cos_theta_i), 1.0f);
    RGB eta = eta_t / eta_i;
    RGB eta_k = k / eta_i;
    float cos_theta_i_2 = cos_theta_i * cos_theta_i;
    float sin_theta_i_2 = 1.0f - cos_theta_i_2;
    RGB eta_2 = eta * eta;
    RGB eta_k_2 = eta_k * eta_k;
    RGB t0 = eta_2 - eta_k_2 - sin_theta_i_2;
    RGB a_2_plus_b_2 = rgb_sqrt(t0 * t0 + eta_2 * eta_k_2 * 4.0f);
    RGB t1 = a_2_plus_b_2 + cos_theta_i_2;
    RGB a = rgb_sqrt((a_2_plus_b_2 + t0) * 0.5f);
    RGB t2 = a * cos_theta_i * 2.0f;
    RGB rs = (t1 - t2) / (t1 + t2);
    RGB t3 = a_2_plus_b_2 * cos_theta_i_2 + sin_theta_i_2 * sin_theta_i_2;
    RGB t4 = t2 * sin_theta_i_2;
    RGB rp = rs * (t3 - t4) / (t3 + t4);
    return (rp + rs) * 0.5f;
}
// fresnel.rs:107-117
YK_HD RGB fresnel_schlick(RGB rs, float cos_theta_i) {
    cos_theta_i = rclamp(cos_theta_i, -1.0f, 1.0f);
    float v = 1.0f - cos_theta_i;
    float p5 = (v * v) * (v * v) * v;
    return rs + (RGB{1.0f, 1.0f, 1.0f} - rs) * p5;
}

// trowbridge_reitz.rs:23-30 — evaluated once per material on the host
YK_HD float roughness_to_alpha(float roughness) {
    float x = det_logf(rmax(roughness, 0.001f));
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}
// trowbridge_reitz.rs:34-44
YK_HD float tr_d(float alpha, V3 wh) {
    float t2 = tan_2_theta(wh);
    if (isinf(t2)) return 0.0f;
    float alpha_2 = alpha * alpha;
    float cos_4_theta = cos_2_theta(wh) * cos_2_theta(wh);
    float e = (cos_2_phi(wh) / alpha_2 + sin_2_phi(wh) / alpha_2) * t2;
    return 1.0f / (YK_PI * alpha_2 * cos_4_theta * (1.0f + e) * (1.0f + e));
}
// trowbridge_reitz.rs:46-58
YK_HD float tr_lambda(float alpha, V3 w) {
    float abs_tan_theta = fabsf(tan_theta(w));
    if (isinf(abs_tan_theta)) return 0.0f;
    float a = sqrtf(cos_2_phi(w) * alpha * alpha + sin_2_phi(w) * alpha * alpha);
    float a2t2 = (a * abs_tan_theta) * (a * abs_tan_theta);
    return (-1.0f + sqrtf(1.0f + a2t2)) / 2.0f;
}
// microfacet.rs:25-27
YK_HD float tr_g(float alpha, V3 wo, V3 wi) { return 1.0f / (1.0f + tr_lambda(alpha, wo) + tr_lambda(alpha, wi)); }
// trowbridge_reitz.rs:60-74 and bsdfs/mod.rs:279-281
YK_HD V3 tr_sample_wh(float alpha, V3 wo, float u0, float u1) {
    float tan_theta_2 = alpha * alpha * u0 / (1.0f - u0);
    float cos_t = 1.0f / sqrtf(1.0f + tan_theta_2);
    float phi = 2.0f * YK_PI * u1;
    float sin_t = sqrtf(rmax(1.0f - cos_t * cos_t, 0.0f));
    float sn, cs;
    det_sincosf(phi, sn, cs);
    V3 wh = V3{sin_t * cs, sin_t * sn, cos_t};
    return same_hemisphere(wo, wh) ? wh : -wh;
}

YK_HD RGB mat_fresnel(const Material& m, float c) {
    if (m.kind == MK_METAL) return fresnel_conductor(RGB{m.a[0], m.a[1], m.a[2]}, RGB{m.b[0], m.b[1], m.b[2]}, c);
    return fresnel_schlick(RGB{m.a[0], m.a[1], m.a[2]}, c);
}

// MicrofacetReflection::f, microfacet.rs:51-72 (r = ones)
YK_HD RGB microfacet_f(const Material& m, V3 wo, V3 wi) {
    float cos_theta_o = fabsf(cos_theta(wo));
    float cos_theta_i = fabsf(cos_theta(wi));
    if (cos_theta_i == 0.0f || cos_theta_o == 0.0f) return RGB{0.0f, 0.0f, 0.0f};
    V3 wh = wi + wo;
    if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return RGB{0.0f, 0.0f, 0.0f};
    wh = normalize(wh);
    RGB fr = mat_fresnel(m, dot(wi, faceforward_v(wh, V3{0.0f, 0.0f, 1.0f})));
    const RGB r = RGB{1.0f, 1.0f, 1.0f};
    return r * tr_d(m.c, wh) * tr_g(m.c, wo, wi) * fr / (4.0f * cos_theta_i * cos_theta_o);
}

// OrenNayar::f, oren_nayar.rs:30-54; first/second follow the call position
YK_HD RGB oren_nayar_f(const Material& m, V3 first, V3 second) {
    float sin_theta_1 = sin_theta(first);
    float sin_theta_2 = sin_theta(second);
    float max_cos = 0.0f;
    if (sin_theta_1 > 1e-4f && sin_theta_2 > 1e-4f) {
        float sin_phi_1 = sin_phi(first), cos_phi_1 = cos_phi(first);
        float sin_phi_2 = sin_phi(second), cos_phi_2 = cos_phi(second);
        float d_cos = cos_phi_1 * cos_phi_2 + sin_phi_1 * sin_phi_2;
        max_cos = rmax(d_cos, 0.0f);
    }
    float sin_alpha, tan_beta;
    if (fabsf(cos_theta(first)) > fabsf(cos_theta(second))) {
        sin_alpha = sin_theta_2;
        tan_beta = sin_theta_1 / fabsf(cos_theta(first));
    } else {
        sin_alpha = sin_theta_1;
        tan_beta = sin_theta_2 / fabsf(cos_theta(second));
    }
    return RGB{m.a[0], m.a[1], m.a[2]} * YK_FRAC_1_PI * (m.c + m.d * max_cos * sin_alpha * tan_beta);
}

// lobe f() in the local frame for the single non-specular lobe of a material
YK_HD RGB lobe_f(const Material& m, V3 wo, V3 wi) {
    switch (m.kind) {
        case MK_LAMBERT: return RGB{m.a[0], m.a[1], m.a[2]} * YK_FRAC_1_PI;  // lambertian.rs:21-23
        case MK_OREN_NAYAR: return oren_nayar_f(m, wo, wi);
        case MK_METAL:
        case MK_GLOSSY: return microfacet_f(m, wo, wi);
        default: return RGB{0.0f, 0.0f, 0.0f};
    }
}

// Shading frame of Bsdf::new, bsdfs/mod.rs:87-99
struct Frame {
    V3 s, t, n;  // s_shading, t_shading, n_shading
    V3 ng;       // n_geom
};
YK_HD Frame make_frame(V3 n_geom, V3 n_shading, V3 shading_dpdu) {
    Frame f;
    f.n = n_shading;
    f.s = normalize(shading_dpdu);
    f.t = cross(n_shading, f.s);
    f.ng = n_geom;
    return f;
}
// bsdfs/mod.rs:107-122
YK_HD V3 world_to_local(const Frame& f, V3 v) { return V3{dot(v, f.s), dot(v, f.t), dot_nv(v, f.n)}; }
YK_HD V3 local_to_world(const Frame& f, V3 v) {
    return V3{f.s.x * v.x + f.t.x * v.y + f.n.x * v.z, f.s.y * v.x + f.t.y * v.y + f.n.y * v.z,
              f.s.z * v.x + f.t.z * v.y + f.n.z * v.z};
}

// Bsdf::f(wo, wi, BxdfType::all()), bsdfs/mod.rs:125-147
YK_HD RGB bsdf_f(const Material& m, const Frame& fr, V3 wo_world, V3 wi_world) {
    RGB f = RGB{0.0f, 0.0f, 0.0f};
    if (m.kind == MK_BLACK) return f;
    if (m.kind == MK_GLASS) {
        // two specular lobes; exactly one passes the reflect/transmit filter and
        // contributes Spectrum::zeros() (specular.rs:20-22,64-66)
        return f + RGB{0.0f, 0.0f, 0.0f};
    }
    V3 wo = world_to_local(fr, wo_world);
    V3 wi = world_to_local(fr, wi_world);
    bool reflect = dot_nv(wi_world, fr.ng) * dot_nv(wo_world, fr.ng) > 0.0f;
    if (reflect) f = f + lobe_f(m, wo, wi);  // all remaining lobes are REFLECTION lobes
    return f;
}

struct BsdfSample {
    V3 wi;
    RGB f;
    float pdf;
    int type;
};
YK_HD BsdfSample bsdf_sample_none() { return BsdfSample{V3{0.0f, 0.0f, 0.0f}, RGB{0.0f, 0.0f, 0.0f}, 0.0f, BX_NONE}; }

// bsdfs/mod.rs:284-296
YK_HD bool refract(V3 wi, V3 n, float eta, V3& wt) {
    float cos_theta_i = dot_nv(n, wi);
    float sin_2_theta_i = rmax(1.0f - cos_theta_i * cos_theta_i, 0.0f);
    float sin_2_theta_t = eta * eta * sin_2_theta_i;
    if (sin_2_theta_t >= 1.0f) return false;
    float cos_theta_t = sqrtf(1.0f - sin_2_theta_t);
    wt = (-wi) * eta + n * (eta * cos_theta_i - cos_theta_t);
    return true;
}

// Bsdf::sample_f(wo, u, BxdfType::all()), bsdfs/mod.rs:150-223
YK_HD BsdfSample bsdf_sample_f(const Material& m, const Frame& fr, V3 wo_world, float u0, float u1) {
    if (m.kind == MK_BLACK) return bsdf_sample_none();  // matching_comps == 0
    V3 wo = world_to_local(fr, wo_world);
    BsdfSample s;
    if (m.kind == MK_GLASS) {
        // matching_comps = 2 ; comp = min(floor(u0*2), 1)
        float fl = floorf(u0 * 2.0f);
        int comp = (fl != fl || fl <= 0.0f) ? 0 : 1;
        V3 wi;
        RGB f;
        if (comp == 0) {  // specular::Reflection::sample_f, specular.rs:24-34
            wi = V3{-wo.x, -wo.y, wo.z};
            f = RGB{m.a[0], m.a[1], m.a[2]} * fresnel_dielectric(1.0f, m.c, cos_theta(wi)) / fabsf(cos_theta(wi));
            s.type = BX_SPECULAR | BX_REFLECTION;
        } else {  // specular::Transmission::sample_f, specular.rs:68-92
            bool entering = cos_theta(wo) > 0.0f;
            float eta_i = entering ? 1.0f : m.c;
            float eta_t = entering ? m.c : 1.0f;
            if (!refract(wo, faceforward_v(V3{0.0f, 0.0f, 1.0f}, wo), eta_i / eta_t, wi)) return bsdf_sample_none();
            f = RGB{m.b[0], m.b[1], m.b[2]} * (RGB{1.0f, 1.0f, 1.0f} - fresnel_dielectric(1.0f, m.c, cos_theta(wi))) /
                fabsf(cos_theta(wi));
            s.type = BX_SPECULAR | BX_TRANSMISSION;
        }
        float pdf = 1.0f;
        s.wi = local_to_world(fr, wi);
        pdf /= 2.0f;  // matching_comps > 1
        s.pdf = pdf;
        s.f = f;
        return s;
    }
    // single lobe: comp = 0, u_remapped = (u0 * 1, u1)
    float ur0 = u0 * 1.0f;
    V3 wi;
    float pdf;
    RGB f;
    if (m.kind == MK_LAMBERT || m.kind == MK_OREN_NAYAR) {  // lambertian.rs:25-47, oren_nayar.rs:56-78
        wi = cosine_sample_hemisphere(ur0, u1);
        if (wo.z < 0.0f) wi.z *= -1.0f;
        pdf = same_hemisphere(wo, wi) ? fabsf(cos_theta(wi)) * YK_FRAC_1_PI : 0.0f;
        f = lobe_f(m, wo, wi);
        s.type = BX_DIFFUSE | BX_REFLECTION;
    } else {  // MicrofacetReflection::sample_f, microfacet.rs:74-99
        if (wo.z == 0.0f) return bsdf_sample_none();
        V3 wh = tr_sample_wh(m.c, wo, ur0, u1);
        if (dot(wo, wh) < 0.0f) return bsdf_sample_none();
        wi = (-wo) + wh * 2.0f * dot(wo, wh);  // reflect(), bsdfs/mod.rs:298-300
        if (!same_hemisphere(wo, wi)) return bsdf_sample_none();
        pdf = (tr_d(m.c, wh) * cos_theta(wh)) / (4.0f * dot(wo, wh));
        f = microfacet_f(m, wo, wi);
        s.type = BX_REFLECTION | BX_GLOSSY;
    }
    if (pdf == 0.0f) return bsdf_sample_none();
    s.wi = local_to_world(fr, wi);
    s.pdf = pdf;
    s.f = f;
    return s;
}

// Bsdf::sample_f(wo, Point2(0, 0), SPECULAR | REFLECTION or SPECULAR | TRANSMISSION) — what
// Whitted::specular_contribution asks for (whitted.rs:49-51).  Only lobes whose type is
// inside the mask match (bsdfs/mod.rs:150-160): glass is the only material with specular
// lobes and exactly one of its two matches, so matching_comps = 1, the lobe's pdf of 1 is not
// divided, and u is irrelevant.  BX_NONE = no matching lobe, or total internal reflection.
YK_HD BsdfSample bsdf_sample_specular(const Material& m, const Frame& fr, V3 wo_world, int direction) {
    if (m.kind != MK_GLASS) return bsdf_sample_none();
    V3 wo = world_to_local(fr, wo_world);
    BsdfSample s;
    V3 wi;
    if (direction == BX_REFLECTION) {  // specular.rs:24-34
        wi = V3{-wo.x, -wo.y, wo.z};
        s.f = RGB{m.a[0], m.a[1], m.a[2]} * fresnel_dielectric(1.0f, m.c, cos_theta(wi)) / fabsf(cos_theta(wi));
        s.type = BX_SPECULAR | BX_REFLECTION;
    } else {  // specular.rs:68-92
        bool entering = cos_theta(wo) > 0.0f;
        float eta_i = entering ? 1.0f : m.c;
        float eta_t = entering ? m.c : 1.0f;
        if (!refract(wo, faceforward_v(V3{0.0f, 0.0f, 1.0f}, wo), eta_i / eta_t, wi)) return bsdf_sample_none();
        s.f = RGB{m.b[0], m.b[1], m.b[2]} * (RGB{1.0f, 1.0f, 1.0f} - fresnel_dielectric(1.0f, m.c, cos_theta(wi))) / fabsf(cos_theta(wi));
        s.type = BX_SPECULAR | BX_TRANSMISSION;
    }
    s.wi = local_to_world(fr, wi);
    s.pdf = 1.0f;
    return s;
}

}  // namespace yk
