// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of yuki/src/lights/*.rs and yuki/src/visibility.rs.
// Parity unpinned by the reference (no tests for lights).
#pragma once
#include "obvh.h"
#include "omath.h"
#include "oshapes.h"

namespace orc {

enum LightKind { LIGHT_POINT = 0, LIGHT_SPOT = 1, LIGHT_DISTANT = 2, LIGHT_RECT = 3 };

struct Light {
    int kind;
    Point3f p;        // point / spot position
    Vec3f w;          // distant direction
    Spectrumf i;      // intensity (point/spot) or radiance (distant/rect)
    float cos_total_width, cos_falloff_start;  // spot
    Transformf world_to_light;                 // spot
    Transformf sample_to_world;                // rect
    float area;                                // rect
};

// visibility.rs:6-24
struct VisibilityTester {
    Interaction p0, p1;
    int area_light;  // -1 = None
    Rayf ray() const { return p0.spawn_ray_to(p1); }
    bool unoccluded(const BVH& bvh, size_t* node_tests = nullptr, size_t* shape_tests = nullptr) const {
        return !bvh.any_intersect(ray(), area_light, node_tests, shape_tests);
    }
};

struct LightSample {
    Vec3f l;
    Spectrumf li;
    bool has_vis;
    VisibilityTester vis;
    float pdf;
};

// lights/mod.rs:29-32 `Light::sample_li`
inline LightSample sample_li(const Light& L, int light_index, const SurfaceInteraction& si, Point2f u) {
    LightSample s;
    s.has_vis = true;
    s.vis.p0 = Interaction(si.p, si.n);
    s.vis.area_light = -1;
    switch (L.kind) {
        case LIGHT_POINT: {  // point_light.rs:27-50
            Vec3f to_light = L.p - si.p;
            float dist_sqr = to_light.len_sqr();
            s.li = L.i / dist_sqr;
            float dist = std::sqrt(dist_sqr);
            s.l = to_light / dist;
            s.vis.p1 = Interaction(L.p, Normalf(0.0f, 0.0f, 1.0f));
            s.pdf = 1.0f;
            break;
        }
        case LIGHT_SPOT: {  // spot_light.rs:32-80
            Vec3f to_light = L.p - si.p;
            float dist_sqr = to_light.len_sqr();
            float dist = std::sqrt(dist_sqr);
            s.l = to_light / dist;
            float falloff;
            {
                Vec3f dir_local = L.world_to_light.apply(-s.l).normalized();
                float ct = dir_local.z;
                if (ct < L.cos_total_width)
                    falloff = 0.0f;
                else if (ct > L.cos_falloff_start)
                    falloff = 1.0f;
                else {
                    float delta = (ct - L.cos_total_width) / (L.cos_falloff_start - L.cos_total_width);
                    falloff = (delta * delta) * (delta * delta);
                }
            }
            s.li = L.i * falloff / dist_sqr;
            if (s.li.is_black()) s.has_vis = false;
            s.vis.p1 = Interaction(L.p, Normalf(0.0f, 0.0f, 1.0f));
            s.pdf = 1.0f;
            break;
        }
        case LIGHT_DISTANT: {  // distant_light.rs:24-43
            s.li = L.i;
            s.l = L.w;
            s.vis.p1 = Interaction(si.p + L.w * 10000.0f, Normalf(0.0f, 0.0f, 1.0f));
            s.pdf = 1.0f;
            break;
        }
        default: {  // rectangular_light.rs:46-71
            Point3f p = L.sample_to_world.apply(Point3f(u.x, 0.0f, u.y));
            Normalf n = L.sample_to_world.apply(Normalf(0.0f, -1.0f, 0.0f));
            Vec3f wi = (p - si.p).normalized();
            s.li = n.dot_v(-wi) > 0.0f ? L.i : Spectrumf::zeros();
            s.vis.p1 = Interaction(p, n);
            s.vis.area_light = light_index;
            s.pdf = si.p.dist_sqr(p) / (std::fabs(n.dot_v(-wi)) * L.area);
            s.l = wi;
            break;
        }
    }
    return s;
}

// rectangular_light.rs:75-81 via interaction.rs:134-138
inline Spectrumf emitted_radiance(const std::vector<Light>& lights, const SurfaceInteraction& si, Vec3f w) {
    if (si.area_light < 0) return Spectrumf::zeros();
    const Light& L = lights[si.area_light];
    return si.n.dot_v(w) > 0.0f ? L.i : Spectrumf::zeros();
}

}  // namespace orc
