// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of yuki/src/materials/{matte,glass,metal,glossy}.rs and
// yuki/src/materials/bsdfs/*.rs, plus the sampling helpers of
// yuki/src/sampling/mod.rs:62-87.  Parity unpinned by the reference.
#pragma once
#include <vector>
#include "olibm.h"
#include "omath.h"
#include "oshapes.h"

namespace orc {

const float O_PI = 3.14159265358979323846f;
const float O_FRAC_1_PI = 0.318309886183790671537767526745028724f;
const float O_FRAC_PI_2 = 1.57079632679489661923132169163975144f;
const float O_FRAC_PI_4 = 0.785398163397448309615660845819875721f;

// sampling/mod.rs:68-87
inline Point2f concentric_sample_disk(Point2f u) {
    Point2f offset = u * 2.0f - Vec2<float>(1.0f, 1.0f);
    if (offset == Point2f(0.0f, 0.0f)) return Point2f(0.0f, 0.0f);
    float theta, r;
    if (std::fabs(offset.x) > std::fabs(offset.y)) {
        theta = O_FRAC_PI_4 * (offset.y / offset.x);
        r = offset.x;
    } else {
        theta = O_FRAC_PI_2 - O_FRAC_PI_4 * (offset.x / offset.y);
        r = offset.y;
    }
    return Point2f(lm::cosf_(theta), lm::sinf_(theta)) * r;
}
// sampling/mod.rs:62-66
inline Vec3f cosine_sample_hemisphere(Point2f u) {
    Point2f d = concentric_sample_disk(u);
    float z = std::sqrt(rmax(1.0f - d.x * d.x - d.y * d.y, 0.0f));
    return Vec3f(d.x, d.y, z);
}

// bsdfs/mod.rs:24-34
enum BxdfType {
    BX_NONE = 0,
    BX_REFLECTION = 1,
    BX_TRANSMISSION = 2,
    BX_DIFFUSE = 4,
    BX_GLOSSY = 8,
    BX_SPECULAR = 16,
    BX_ALL = 31
};

struct BxdfSample {
    Vec3f wi;
    Spectrumf f;
    float pdf;
    int sample_type;
    BxdfSample() : wi(0, 0, 0), f(0, 0, 0), pdf(0.0f), sample_type(BX_NONE) {}
};

// bsdfs/mod.rs:225-300 trig helpers
inline float cos_theta(Vec3f w) { return w.z; }
inline float cos_2_theta(Vec3f w) { return w.z * w.z; }
inline float sin_2_theta(Vec3f w) { return rmax(1.0f - cos_2_theta(w), 0.0f); }
inline float sin_theta(Vec3f w) { return std::sqrt(sin_2_theta(w)); }
inline float tan_theta(Vec3f w) { return sin_theta(w) / cos_theta(w); }
inline float tan_2_theta(Vec3f w) { return sin_2_theta(w) / cos_2_theta(w); }
inline float sin_phi(Vec3f w) {
    float st = sin_theta(w);
    return st == 0.0f ? 1.0f : rclamp(w.y / st, -1.0f, 1.0f);
}
inline float sin_2_phi(Vec3f w) { return sin_phi(w) * sin_phi(w); }
inline float cos_phi(Vec3f w) {
    float st = sin_theta(w);
    return st == 0.0f ? 1.0f : rclamp(w.x / st, -1.0f, 1.0f);
}
inline float cos_2_phi(Vec3f w) { return cos_phi(w) * cos_phi(w); }
inline bool same_hemisphere(Vec3f w, Vec3f wp) { return w.z * wp.z > 0.0f; }
inline Vec3f spherical_direction(float sin_t, float cos_t, float phi) {
    return Vec3f(sin_t * lm::cosf_(phi), sin_t * lm::sinf_(phi), cos_t);
}
// bsdfs/mod.rs:284-296
inline bool refract(Vec3f wi, Normalf n, float eta, Vec3f& wt) {
    float cos_theta_i = n.dot_v(wi);
    float sin_2_theta_i = rmax(1.0f - cos_theta_i * cos_theta_i, 0.0f);
    float sin_2_theta_t = eta * eta * sin_2_theta_i;
    if (sin_2_theta_t >= 1.0f) return false;
    float cos_theta_t = std::sqrt(1.0f - sin_2_theta_t);
    wt = (-wi) * eta + Vec3f(n) * (eta * cos_theta_i - cos_theta_t);
    return true;
}
// bsdfs/mod.rs:298-300
inline Vec3f reflect(Vec3f wo, Vec3f n) { return -wo + n * 2.0f * wo.dot(n); }

// ---- Fresnel (bsdfs/fresnel.rs) ------------------------------------------
enum FresnelKind { FR_DIELECTRIC = 0, FR_CONDUCTOR = 1, FR_SCHLICK = 2 };
struct Fresnel {
    int kind;
    float eta_i_s, eta_t_s;        // dielectric
    Spectrumf eta_i, eta_t, k;     // conductor
    Spectrumf rs;                  // schlick
};
inline Spectrumf ssqrt(Spectrumf v) { return Spectrumf(std::sqrt(v.r), std::sqrt(v.g), std::sqrt(v.b)); }
// fresnel.rs:22-51
inline Spectrumf fresnel_dielectric(float eta_i_in, float eta_t_in, float cos_theta_i) {
    cos_theta_i = rclamp(cos_theta_i, -1.0f, 1.0f);
    bool entering = cos_theta_i > 0.0f;
    float eta_i, eta_t;
    if (entering) {
        eta_i = eta_i_in;
        eta_t = eta_t_in;
    } else {
        eta_i = eta_t_in;
        eta_t = eta_i_in;
        cos_theta_i = std::fabs(cos_theta_i);
    }
    float sin_theta_i = std::sqrt(rmax(1.0f - cos_theta_i * cos_theta_i, 0.0f));
    float sin_theta_t = eta_i / eta_t * sin_theta_i;
    if (sin_theta_t >= 1.0f) return Spectrumf::ones();
    float cos_theta_t = std::sqrt(rmax(1.0f - sin_theta_t * sin_theta_t, 0.0f));
    float r_parallel = ((eta_t * cos_theta_i) - (eta_i * cos_theta_t)) / ((eta_t * cos_theta_i) + (eta_i * cos_theta_t));
    float r_perpendicular =
        ((eta_i * cos_theta_i) - (eta_t * cos_theta_t)) / ((eta_i * cos_theta_i) + (eta_t * cos_theta_t));
    return Spectrumf::ones() * (r_parallel * r_parallel + r_perpendicular * r_perpendicular) / 2.0f;
}
// fresnel.rs:65-95
inline Spectrumf fresnel_conductor(Spectrumf eta_i, Spectrumf eta_t, Spectrumf k, float cos_theta_i) {
    cos_theta_i = rmin(std::fabs(cos_theta_i), 1.0f);
    Spectrumf eta = eta_t / eta_i;
    Spectrumf eta_k = k / eta_i;
    float cos_theta_i_2 = cos_theta_i * cos_theta_i;
    float sin_theta_i_2 = 1.0f - cos_theta_i_2;
    Spectrumf eta_2 = eta * eta;
    Spectrumf eta_k_2 = eta_k * eta_k;
    Spectrumf t0 = eta_2 - eta_k_2 - sin_theta_i_2;
    Spectrumf a_2_plus_b_2 = ssqrt(t0 * t0 + eta_2 * eta_k_2 * 4.0f);
    Spectrumf t1 = a_2_plus_b_2 + cos_theta_i_2;
    Spectrumf a = ssqrt((a_2_plus_b_2 + t0) * 0.5f);
    Spectrumf t2 = a * cos_theta_i * 2.0f;
    Spectrumf rs = (t1 - t2) / (t1 + t2);
    Spectrumf t3 = a_2_plus_b_2 * cos_theta_i_2 + sin_theta_i_2 * sin_theta_i_2;
    Spectrumf t4 = t2 * sin_theta_i_2;
    Spectrumf rp = rs * (t3 - t4) / (t3 + t4);
    return (rp + rs) * 0.5f;
}
// fresnel.rs:107-117
inline Spectrumf fresnel_schlick(Spectrumf rs, float cos_theta_i) {
    cos_theta_i = rclamp(cos_theta_i, -1.0f, 1.0f);
    float v = 1.0f - cos_theta_i;
    float p5 = (v * v) * (v * v) * v;
    return rs + (Spectrumf::ones() - rs) * p5;
}
inline Spectrumf fresnel_eval(const Fresnel& f, float c) {
    switch (f.kind) {
        case FR_DIELECTRIC: return fresnel_dielectric(f.eta_i_s, f.eta_t_s, c);
        case FR_CONDUCTOR: return fresnel_conductor(f.eta_i, f.eta_t, f.k, c);
        default: return fresnel_schlick(f.rs, c);
    }
}

// ---- Trowbridge-Reitz (bsdfs/trowbridge_reitz.rs) -------------------------
struct TrowbridgeReitz {
    float alpha;
    // :15-20
    static TrowbridgeReitz make(float a) {
        TrowbridgeReitz d;
        d.alpha = rmax(a, 0.001f);
        return d;
    }
    // :23-30
    static float roughness_to_alpha(float roughness) {
        float x = lm::logf_(rmax(roughness, 0.001f));
        return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
    }
    // :34-44
    float d(Vec3f wh) const {
        float t2 = tan_2_theta(wh);
        if (std::isinf(t2)) return 0.0f;
        float alpha_2 = alpha * alpha;
        float cos_4_theta = cos_2_theta(wh) * cos_2_theta(wh);
        float e = (cos_2_phi(wh) / alpha_2 + sin_2_phi(wh) / alpha_2) * t2;
        return 1.0f / (O_PI * alpha_2 * cos_4_theta * (1.0f + e) * (1.0f + e));
    }
    // :46-58
    float lambda(Vec3f w) const {
        float abs_tan_theta = std::fabs(tan_theta(w));
        if (std::isinf(abs_tan_theta)) return 0.0f;
        float a = std::sqrt(cos_2_phi(w) * alpha * alpha + sin_2_phi(w) * alpha * alpha);
        float a2t2 = (a * abs_tan_theta) * (a * abs_tan_theta);
        return (-1.0f + std::sqrt(1.0f + a2t2)) / 2.0f;
    }
    // microfacet.rs:25-27
    float g(Vec3f wo, Vec3f wi) const { return 1.0f / (1.0f + lambda(wo) + lambda(wi)); }
    // :60-74
    Vec3f sample_wh(Vec3f wo, Point2f u) const {
        float tan_theta_2 = alpha * alpha * u[0] / (1.0f - u[0]);
        float cos_t = 1.0f / std::sqrt(1.0f + tan_theta_2);
        float phi = 2.0f * O_PI * u[1];
        float sin_t = std::sqrt(rmax(1.0f - cos_t * cos_t, 0.0f));
        Vec3f wh = spherical_direction(sin_t, cos_t, phi);
        return same_hemisphere(wo, wh) ? wh : -wh;
    }
    // :76-78
    float pdf(Vec3f, Vec3f wh) const { return d(wh) * cos_theta(wh); }
};

// ---- BxDFs ------------------------------------------------------------------
enum BxdfKind { BXDF_LAMBERT = 0, BXDF_OREN_NAYAR = 1, BXDF_SPEC_REFL = 2, BXDF_SPEC_TRANS = 3, BXDF_MICROFACET = 4 };

struct Bxdf {
    int kind;
    Spectrumf r;        // reflectance / R / T
    float a, b;         // Oren-Nayar
    float eta_i, eta_t; // transmission
    Fresnel fresnel;
    TrowbridgeReitz dist;

    int flags() const {
        switch (kind) {
            case BXDF_LAMBERT:
            case BXDF_OREN_NAYAR: return BX_DIFFUSE | BX_REFLECTION;
            case BXDF_SPEC_REFL: return BX_SPECULAR | BX_REFLECTION;
            case BXDF_SPEC_TRANS: return BX_SPECULAR | BX_TRANSMISSION;
            default: return BX_REFLECTION | BX_GLOSSY;
        }
    }
    bool matches(int t) const { return (t & flags()) == flags(); }

    Spectrumf f(Vec3f wo, Vec3f wi) const {
        switch (kind) {
            case BXDF_LAMBERT: return r * O_FRAC_1_PI;  // lambertian.rs:21-23
            case BXDF_OREN_NAYAR: {                      // oren_nayar.rs:30-54 (params named wi, wo there)
                Vec3f p = wo, q = wi;
                float sin_theta_p = sin_theta(p);
                float sin_theta_q = sin_theta(q);
                float max_cos = 0.0f;
                if (sin_theta_p > 1e-4f && sin_theta_q > 1e-4f) {
                    float sin_phi_p = sin_phi(p), cos_phi_p = cos_phi(p);
                    float sin_phi_q = sin_phi(q), cos_phi_q = cos_phi(q);
                    float d_cos = cos_phi_p * cos_phi_q + sin_phi_p * sin_phi_q;
                    max_cos = rmax(d_cos, 0.0f);
                }
                float sin_alpha, tan_beta;
                if (std::fabs(cos_theta(p)) > std::fabs(cos_theta(q))) {
                    sin_alpha = sin_theta_q;
                    tan_beta = sin_theta_p / std::fabs(cos_theta(p));
                } else {
                    sin_alpha = sin_theta_p;
                    tan_beta = sin_theta_q / std::fabs(cos_theta(q));
                }
                return r * O_FRAC_1_PI * (a + b * max_cos * sin_alpha * tan_beta);
            }
            case BXDF_SPEC_REFL:
            case BXDF_SPEC_TRANS: return Spectrumf::zeros();
            default: {  // microfacet.rs:51-72
                float cos_theta_o = std::fabs(cos_theta(wo));
                float cos_theta_i = std::fabs(cos_theta(wi));
                if (cos_theta_i == 0.0f || cos_theta_o == 0.0f) return Spectrumf::zeros();
                Vec3f wh = wi + wo;
                if (wh == Vec3f(0, 0, 0)) return Spectrumf::zeros();
                wh = wh.normalized();
                Spectrumf fr = fresnel_eval(fresnel, wi.dot(Vec3f(Normalf(wh).faceforward_v(Vec3f(0.0f, 0.0f, 1.0f)))));
                return r * dist.d(wh) * dist.g(wo, wi) * fr / (4.0f * cos_theta_i * cos_theta_o);
            }
        }
    }

    float pdf(Vec3f wo, Vec3f wi) const {
        switch (kind) {
            case BXDF_LAMBERT:
            case BXDF_OREN_NAYAR: return same_hemisphere(wo, wi) ? std::fabs(cos_theta(wi)) * O_FRAC_1_PI : 0.0f;
            case BXDF_SPEC_REFL:
            case BXDF_SPEC_TRANS: return 1.0f;
            default: {  // microfacet.rs:101-108
                if (!same_hemisphere(wo, wi)) return 0.0f;
                Vec3f wh = (wo + wi).normalized();
                return dist.pdf(wo, wh) / (4.0f * wo.dot(wh));
            }
        }
    }

    BxdfSample sample_f(Vec3f wo, Point2f u) const {
        BxdfSample s;
        switch (kind) {
            case BXDF_LAMBERT:
            case BXDF_OREN_NAYAR: {  // lambertian.rs:25-39, oren_nayar.rs:56-70
                Vec3f wi = cosine_sample_hemisphere(u);
                if (wo.z < 0.0f) wi.z *= -1.0f;
                s.wi = wi;
                s.pdf = pdf(wo, wi);
                s.f = f(wo, wi);
                s.sample_type = flags();
                return s;
            }
            case BXDF_SPEC_REFL: {  // specular.rs:24-34
                Vec3f wi(-wo.x, -wo.y, wo.z);
                s.wi = wi;
                s.f = r * fresnel_eval(fresnel, cos_theta(wi)) / std::fabs(cos_theta(wi));
                s.pdf = 1.0f;
                s.sample_type = flags();
                return s;
            }
            case BXDF_SPEC_TRANS: {  // specular.rs:68-92
                bool entering = cos_theta(wo) > 0.0f;
                float ei = entering ? eta_i : eta_t;
                float et = entering ? eta_t : eta_i;
                Vec3f wi;
                if (!refract(wo, Normalf(0.0f, 0.0f, 1.0f).faceforward_v(wo), ei / et, wi)) return BxdfSample();
                s.wi = wi;
                s.f = r * (Spectrumf::ones() - fresnel_dielectric(eta_i, eta_t, cos_theta(wi))) / std::fabs(cos_theta(wi));
                s.pdf = 1.0f;
                s.sample_type = flags();
                return s;
            }
            default: {  // microfacet.rs:74-99
                if (wo.z == 0.0f) return BxdfSample();
                Vec3f wh = dist.sample_wh(wo, u);
                if (wo.dot(wh) < 0.0f) return BxdfSample();
                Vec3f wi = reflect(wo, wh);
                if (!same_hemisphere(wo, wi)) return BxdfSample();
                s.pdf = dist.pdf(wo, wh) / (4.0f * wo.dot(wh));
                s.f = f(wo, wi);
                s.wi = wi;
                s.sample_type = flags();
                return s;
            }
        }
    }
};

// ---- materials ---------------------------------------------------------------
enum MaterialKind { MAT_MATTE = 0, MAT_GLASS = 1, MAT_METAL = 2, MAT_GLOSSY = 3 };
struct Material {
    int kind;
    Spectrumf a, b;  // matte: a=Kd ; glass: a=R,b=T ; metal: a=eta,b=k ; glossy: a=Rs
    float c;         // matte: sigma (radians) ; glass: eta ; metal/glossy: roughness
    bool remap_roughness;
    int a_texture = -1;  // matte: Kd is an ImageTexture (index into the scene's textures) when >= 0
};

// textures/image_texture.rs:49-111
struct ImageTexture {
    std::vector<Spectrumf> data;
    size_t width = 0, height = 0;
    // `f as usize`: saturating, NaN -> 0
    static size_t as_usize(float f) {
        if (!(f > 0.0f)) return 0;
        if (f >= 18446744073709551616.0f) return ~(size_t)0;
        return (size_t)f;
    }
    Spectrumf evaluate(const SurfaceInteraction& si) const {
        Point2f st = si.uv;
        // Repeat
        st.x = st.x - std::trunc(st.x);  // f32::fract
        if (st.x < 0.0f) st.x = 1.0f + st.x;
        st.y = st.y - std::trunc(st.y);
        if (st.y < 0.0f) st.y = 1.0f + st.y;
        // Flip y
        st.y = 1.0f - st.y;
        st.x = st.x * (float)width - 0.5f;
        st.y = st.y * (float)height - 0.5f;
        return data.at(as_usize(st.y) * width + as_usize(st.x));  // the reference panics when out of range
    }
};

// bsdfs/mod.rs:74-223
struct Bsdf {
    Bxdf bxdfs[2];
    int n;
    Normalf n_geom, n_shading;
    Vec3f s_shading, t_shading;

    // bsdfs/mod.rs:87-99
    explicit Bsdf(const SurfaceInteraction& si) : n(0) {
        n_shading = si.shading.n;
        s_shading = si.shading.dpdu.normalized();
        t_shading = Vec3f(n_shading).cross(s_shading);
        n_geom = si.n;
    }
    void add(const Bxdf& b) { bxdfs[n++] = b; }
    Vec3f world_to_local(Vec3f v) const { return Vec3f(v.dot(s_shading), v.dot(t_shading), v.dot_n(n_shading)); }
    Vec3f local_to_world(Vec3f v) const {
        return Vec3f(s_shading.x * v.x + t_shading.x * v.y + n_shading.x * v.z,
                     s_shading.y * v.x + t_shading.y * v.y + n_shading.y * v.z,
                     s_shading.z * v.x + t_shading.z * v.y + n_shading.z * v.z);
    }
    // bsdfs/mod.rs:125-147
    Spectrumf f(Vec3f wo_world, Vec3f wi_world, int bxdf_type) const {
        Vec3f wo = world_to_local(wo_world);
        Vec3f wi = world_to_local(wi_world);
        bool refl = wi_world.dot_n(n_geom) * wo_world.dot_n(n_geom) > 0.0f;
        Spectrumf f = Spectrumf::zeros();
        for (int i = 0; i < n; ++i) {
            const Bxdf& bx = bxdfs[i];
            if (bx.matches(bxdf_type) &&
                ((refl && (bx.flags() & BX_REFLECTION)) || (!refl && (bx.flags() & BX_TRANSMISSION))))
                f += bx.f(wo, wi);
        }
        return f;
    }
    // bsdfs/mod.rs:150-223
    BxdfSample sample_f(Vec3f wo_world, Point2f u, int sample_type) const {
        int matching = 0;
        for (int i = 0; i < n; ++i)
            if (bxdfs[i].matches(sample_type)) ++matching;
        if (matching == 0) return BxdfSample();
        // Rust: (u0 * k).floor() as usize — saturating, NaN -> 0
        float fl = std::floor(u[0] * (float)matching);
        size_t ci = (fl != fl || fl <= 0.0f) ? 0 : (fl >= 1.8446744e19f ? (size_t)-1 : (size_t)fl);
        int comp = (int)(ci < (size_t)(matching - 1) ? ci : (size_t)(matching - 1));
        const Bxdf* bxdf = nullptr;
        int cnt = comp;
        for (int i = 0; i < n; ++i)
            if (bxdfs[i].matches(sample_type)) {
                if (cnt-- == 0) {
                    bxdf = &bxdfs[i];
                    break;
                }
            }
        Vec3f wo = world_to_local(wo_world);
        Point2f u_remapped(u[0] * (float)(matching - comp), u[1]);  // sic (quirk 9)
        BxdfSample bs = bxdf->sample_f(wo, u_remapped);
        Vec3f wi_local = bs.wi;
        Spectrumf f = bs.f;
        float pdf = bs.pdf;
        if (pdf == 0.0f) return BxdfSample();
        Vec3f wi_world = local_to_world(wi_local);
        if (!(bxdf->flags() & BX_SPECULAR) && matching > 1) {
            for (int i = 0; i < n; ++i)
                if (&bxdfs[i] != bxdf && bxdfs[i].matches(sample_type)) pdf += bxdfs[i].pdf(wo, wi_local);
        }
        if (matching > 1) pdf /= (float)matching;
        if (!(bxdf->flags() & BX_SPECULAR) && matching > 1) {
            bool refl = wi_world.dot_n(n_geom) * wo_world.dot_n(n_geom) > 0.0f;
            f = Spectrumf::zeros();
            for (int i = 0; i < n; ++i) {
                const Bxdf& b = bxdfs[i];
                if (b.matches(sample_type) &&
                    ((refl && (b.flags() & BX_REFLECTION)) || (!refl && (b.flags() & BX_TRANSMISSION))))
                    f += b.f(wo, wi_local);
            }
        }
        BxdfSample out;
        out.wi = wi_world;
        out.f = f;
        out.pdf = pdf;
        out.sample_type = bs.sample_type;
        return out;
    }
};

// Material::compute_scattering_functions — matte.rs:22-40, glass.rs:27-45,
// metal.rs:34-61, glossy.rs:32-58
inline Bsdf compute_scattering_functions(const Material& m, const SurfaceInteraction& si, const std::vector<ImageTexture>* textures = nullptr) {
    Bsdf bsdf(si);
    switch (m.kind) {
        case MAT_MATTE: {
            Spectrumf reflectance = (m.a_texture >= 0 && textures) ? textures->at((size_t)m.a_texture).evaluate(si) : m.a;
            float sigma = m.c;
            if (!reflectance.is_black()) {
                Bxdf b;
                b.r = reflectance;
                if (sigma == 0.0f) {
                    b.kind = BXDF_LAMBERT;
                } else {  // oren_nayar.rs:20-27
                    b.kind = BXDF_OREN_NAYAR;
                    float sigma2 = sigma * sigma;
                    b.a = 1.0f - (sigma2 / (2.0f * (sigma2 + 0.33f)));
                    b.b = 0.45f * sigma2 / (sigma2 + 0.09f);
                }
                bsdf.add(b);
            }
            break;
        }
        case MAT_GLASS: {
            Bxdf r;
            r.kind = BXDF_SPEC_REFL;
            r.r = m.a;
            r.fresnel.kind = FR_DIELECTRIC;
            r.fresnel.eta_i_s = 1.0f;
            r.fresnel.eta_t_s = m.c;
            bsdf.add(r);
            Bxdf t;
            t.kind = BXDF_SPEC_TRANS;
            t.r = m.b;
            t.eta_i = 1.0f;
            t.eta_t = m.c;
            bsdf.add(t);
            break;
        }
        case MAT_METAL: {
            float roughness = m.remap_roughness ? TrowbridgeReitz::roughness_to_alpha(m.c) : m.c;
            Bxdf b;
            b.kind = BXDF_MICROFACET;
            b.r = Spectrumf(1.0f, 1.0f, 1.0f);
            b.fresnel.kind = FR_CONDUCTOR;
            b.fresnel.eta_i = Spectrumf(1.0f, 1.0f, 1.0f);
            b.fresnel.eta_t = m.a;
            b.fresnel.k = m.b;
            b.dist = TrowbridgeReitz::make(roughness);
            bsdf.add(b);
            break;
        }
        default: {  // glossy
            float roughness = m.remap_roughness ? TrowbridgeReitz::roughness_to_alpha(m.c) : m.c;
            Bxdf b;
            b.kind = BXDF_MICROFACET;
            b.r = Spectrumf(1.0f, 1.0f, 1.0f);
            b.fresnel.kind = FR_SCHLICK;
            b.fresnel.rs = m.a;
            b.dist = TrowbridgeReitz::make(roughness * roughness);
            bsdf.add(b);
            break;
        }
    }
    return bsdf;
}

}  // namespace orc
