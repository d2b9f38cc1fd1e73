// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of yuki/src/camera.rs, yuki/src/math/transforms.rs,
// yuki/src/integrators/{mod,path,whitted,geometry_normals,bvh_heatmap,
// shading_normals}.rs and the tile helpers of yuki/src/film.rs:299-376.
// Parity unpinned by the reference (no tests above the math layer).
#pragma once
#include <cstdint>
#include <vector>

#include "obsdf.h"
#include "obvh.h"
#include "olights.h"
#include "osampler.h"

namespace orc {

// ---- math/transforms.rs ------------------------------------------------------
template <class T> inline Transform<T> translation(Vec3<T> d) {
    T a[4][4] = {{1, 0, 0, d.x}, {0, 1, 0, d.y}, {0, 0, 1, d.z}, {0, 0, 0, 1}};
    T b[4][4] = {{1, 0, 0, -d.x}, {0, 1, 0, -d.y}, {0, 0, 1, -d.z}, {0, 0, 0, 1}};
    return Transform<T>(Matrix4x4<T>::from_rows(a), Matrix4x4<T>::from_rows(b));
}
template <class T> inline Transform<T> scale(T x, T y, T z) {
    T a[4][4] = {{x, 0, 0, 0}, {0, y, 0, 0}, {0, 0, z, 0}, {0, 0, 0, 1}};
    T b[4][4] = {{T(1) / x, 0, 0, 0}, {0, T(1) / y, 0, 0}, {0, 0, T(1) / z, 0}, {0, 0, 0, 1}};
    return Transform<T>(Matrix4x4<T>::from_rows(a), Matrix4x4<T>::from_rows(b));
}
// sin/cos of the generic T: f32 goes through olibm, f64 through the host libm
inline float t_sin(float x) { return lm::sinf_(x); }
inline float t_cos(float x) { return lm::cosf_(x); }
inline double t_sin(double x) { return std::sin(x); }
inline double t_cos(double x) { return std::cos(x); }
template <class T> inline Transform<T> rotation_x(T theta) {
    T c = t_cos(theta), s = t_sin(theta);
    T a[4][4] = {{1, 0, 0, 0}, {0, c, -s, 0}, {0, s, c, 0}, {0, 0, 0, 1}};
    Matrix4x4<T> m = Matrix4x4<T>::from_rows(a);
    return Transform<T>(m, m.transposed());
}
template <class T> inline Transform<T> rotation_y(T theta) {
    T c = t_cos(theta), s = t_sin(theta);
    T a[4][4] = {{c, 0, s, 0}, {0, 1, 0, 0}, {-s, 0, c, 0}, {0, 0, 0, 1}};
    Matrix4x4<T> m = Matrix4x4<T>::from_rows(a);
    return Transform<T>(m, m.transposed());
}
template <class T> inline Transform<T> rotation_z(T theta) {
    T c = t_cos(theta), s = t_sin(theta);
    T a[4][4] = {{c, -s, 0, 0}, {s, c, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    Matrix4x4<T> m = Matrix4x4<T>::from_rows(a);
    return Transform<T>(m, m.transposed());
}
// transforms.rs:98-127
template <class T> inline Transform<T> rotation(T theta, Vec3<T> axis) {
    Vec3<T> a = axis.normalized();
    T c = t_cos(theta), s = t_sin(theta);
    T r[4][4] = {{a.x * a.x + (T(1) - a.x * a.x) * c, a.x * a.y * (T(1) - c) - a.z * s, a.x * a.z * (T(1) - c) + a.y * s, 0},
                 {a.x * a.y * (T(1) - c) + a.z * s, a.y * a.y + (T(1) - a.y * a.y) * c, a.y * a.z * (T(1) - c) - a.x * s, 0},
                 {a.x * a.z * (T(1) - c) - a.y * s, a.y * a.z * (T(1) - c) + a.x * s, a.z * a.z + (T(1) - a.z * a.z) * c, 0},
                 {0, 0, 0, 1}};
    Matrix4x4<T> m = Matrix4x4<T>::from_rows(r);
    return Transform<T>(m, m.transposed());
}
// transforms.rs:130-136
template <class T> inline Transform<T> rotation_euler(Vec3<T> theta) {
    return rotation_x(theta.x) * (rotation_y(theta.y) * rotation_z(theta.z));
}
// transforms.rs:138-153 — returns world_to_camera (m = inverse, m_inv = camera_to_world)
template <class T> inline Transform<T> look_at(Point3<T> pos, Point3<T> target, Vec3<T> up) {
    Vec3<T> dir = (target - pos).normalized();
    Vec3<T> right = up.normalized().cross(dir).normalized();
    Vec3<T> new_up = dir.cross(right);
    T a[4][4] = {{right.x, new_up.x, dir.x, pos.x}, {right.y, new_up.y, dir.y, pos.y}, {right.z, new_up.z, dir.z, pos.z}, {0, 0, 0, 1}};
    Matrix4x4<T> c2w = Matrix4x4<T>::from_rows(a);
    return Transform<T>(c2w.inverted(), c2w);
}

// ---- camera.rs -----------------------------------------------------------------
struct Camera {
    Transformf camera_to_world, raster_to_camera;
    // camera.rs:52-102.  fov_axis: 0 = FoV::X, 1 = FoV::Y ; degrees.
    static Camera make(Point3f position, Point3f target, Vec3f up, int fov_axis, float fov_angle, uint16_t res_x,
                       uint16_t res_y) {
        Camera cam;
        cam.camera_to_world = look_at(position, target, up).inverted();
        float near = 1e-2f, far = 1000.0f;
        // f32::to_radians: self * (PI / 180)
        float rad = fov_angle * (O_PI / 180.0f);
        float inv_tan = 1.0f / lm::tanf_(rad / 2.0f);
        float pm[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, far / (far - near), -(far * near) / (far - near)}, {0, 0, 1, 0}};
        Transformf camera_to_screen = scale(inv_tan, inv_tan, 1.0f) * Transformf::from_rows(pm);
        float film_x = (float)res_x, film_y = (float)res_y;
        float smin_x, smin_y, smax_x, smax_y;
        if (fov_axis == 0) {
            float ar = film_x / film_y;
            smin_x = -1.0f;
            smin_y = -1.0f / ar;
            smax_x = 1.0f;
            smax_y = 1.0f / ar;
        } else {
            float ar = film_y / film_x;
            smin_x = -1.0f / ar;
            smin_y = -1.0f;
            smax_x = 1.0f / ar;
            smax_y = 1.0f;
        }
        Transformf screen_to_raster =
            scale(film_x, film_y, 1.0f) * (scale(1.0f / (smax_x - smin_x), 1.0f / (smin_y - smax_y), 1.0f) *
                                           translation(Vec3f(-smin_x, -smax_y, 0.0f)));
        Transformf raster_to_screen = screen_to_raster.inverted();
        cam.raster_to_camera = camera_to_screen.inverted() * raster_to_screen;
        return cam;
    }
    // camera.rs:105-114
    Rayf ray(Point2f p_film) const {
        Point3f pf(p_film.x, p_film.y, 0.0f);
        Point3f p_camera = raster_to_camera.apply(pf);
        Rayf r(Point3f(), Vec3f(p_camera).normalized(), std::numeric_limits<float>::infinity());
        return camera_to_world.apply(r);
    }
};

// ---- scene -----------------------------------------------------------------------
struct Scene {
    Geometry geom;
    BVH bvh;
    std::vector<Material> materials;
    std::vector<Light> lights;
    std::vector<ImageTexture> textures;
    Spectrumf background;
    // owned copies of the caller's arrays
    std::vector<float> points, normals, uvs;
};

enum IntegratorKind {
    INTEGRATOR_WHITTED = 0,
    INTEGRATOR_PATH = 1,
    INTEGRATOR_BVH_INTERSECTIONS = 2,
    INTEGRATOR_GEOMETRY_NORMALS = 3,
    INTEGRATOR_SHADING_NORMALS = 4
};

struct IntegratorParams {
    int kind;
    uint32_t max_depth;
    bool has_clamp;
    float indirect_clamp;
};

struct RadianceResult {
    Spectrumf li;
    size_t ray_scene_intersections;
};

// traversal statistics for the roofline; not in the reference
struct TraceStats {
    size_t closest_rays, closest_node_tests, closest_shape_tests;
    size_t shadow_rays, shadow_node_tests, shadow_shape_tests;
    TraceStats() : closest_rays(0), closest_node_tests(0), closest_shape_tests(0), shadow_rays(0), shadow_node_tests(0), shadow_shape_tests(0) {}
    void add(const TraceStats& o) {
        closest_rays += o.closest_rays;
        closest_node_tests += o.closest_node_tests;
        closest_shape_tests += o.closest_shape_tests;
        shadow_rays += o.shadow_rays;
        shadow_node_tests += o.shadow_node_tests;
        shadow_shape_tests += o.shadow_shape_tests;
    }
};

// NEE over all lights — path.rs:102-119 and whitted.rs:113-133 (identical folds)
inline Spectrumf direct_lighting(const Scene& scene, const SurfaceInteraction& si, const Bsdf& bsdf, Sampler& sampler,
                                 TraceStats* st) {
    Spectrumf c = Spectrumf::zeros();
    for (size_t li_idx = 0; li_idx < scene.lights.size(); ++li_idx) {
        LightSample ls = sample_li(scene.lights[li_idx], (int)li_idx, si, sampler.get_2d());
        if (!ls.li.is_black()) {
            Spectrumf f = bsdf.f(si.wo, ls.l, BX_ALL);
            if (ls.has_vis) {
                if (!f.is_black()) {
                    bool vis;
                    if (st) {
                        st->shadow_rays += 1;
                        vis = ls.vis.unoccluded(scene.bvh, &st->shadow_node_tests, &st->shadow_shape_tests);
                    } else {
                        vis = ls.vis.unoccluded(scene.bvh);
                    }
                    if (vis) c = c + f * ls.li * rclamp(si.shading.n.dot_v(ls.l), 0.0f, 1.0f) / ls.pdf;
                }
            }
        }
    }
    return c;
}

// integrators/path.rs:49-178
inline RadianceResult path_li(const IntegratorParams& prm, Rayf ray, const Scene& scene, Sampler& sampler, TraceStats* st) {
    Spectrumf incoming_radiance = Spectrumf::zeros();
    Spectrumf beta = Spectrumf::ones();
    uint32_t bounces = 0;
    bool specular_bounce = false;
    size_t ray_count = 0;
    while (bounces < prm.max_depth) {
        ray_count += 1;
        IntersectionResult ir = scene.bvh.intersect(ray);
        if (st) {
            st->closest_rays += 1;
            st->closest_node_tests += ir.intersection_test_count;
            st->closest_shape_tests += ir.shape_test_count;
        }
        if (ir.has_hit) {
            const SurfaceInteraction& si = ir.hit.si;
            Bsdf bsdf = compute_scattering_functions(scene.materials[ir.hit.shape->material], si, &scene.textures);
            Spectrumf radiance = direct_lighting(scene, si, bsdf, sampler, st);
            if (bounces == 0 || specular_bounce) radiance += beta * emitted_radiance(scene.lights, si, -ray.d);
            if (bounces > 0 && prm.has_clamp) radiance = radiance.smin(Spectrumf::ones() * prm.indirect_clamp);
            incoming_radiance += beta * radiance;

            Vec3f wo = -ray.d;
            BxdfSample bs = bsdf.sample_f(wo, sampler.get_2d(), BX_ALL);
            if (bs.f.is_black() || bs.pdf == 0.0f) break;
            specular_bounce = (bs.sample_type & BX_SPECULAR) != 0;
            beta *= bs.f * std::fabs(bs.wi.dot_n(si.shading.n)) / bs.pdf;
            ray = Interaction(si.p, si.n).spawn_ray(bs.wi);
        } else {
            incoming_radiance += beta * scene.background;
            break;
        }
        if (bounces > 3) {
            float q = rmax(1.0f - beta.g, 0.05f);
            if (sampler.get_1d() < q) break;
            beta *= Spectrumf::ones() / (1.0f - q);
        }
        bounces += 1;
    }
    RadianceResult r;
    r.li = incoming_radiance;
    r.ray_scene_intersections = ray_count;
    return r;
}

// integrators/whitted.rs:39-181
inline RadianceResult whitted_li(const IntegratorParams& prm, Rayf ray, const Scene& scene, uint32_t depth, Sampler& sampler,
                                 bool is_specular, TraceStats* st) {
    IntersectionResult ir = scene.bvh.intersect(ray);
    if (st) {
        st->closest_rays += 1;
        st->closest_node_tests += ir.intersection_test_count;
        st->closest_shape_tests += ir.shape_test_count;
    }
    RadianceResult out;
    if (!ir.has_hit) {
        out.li = scene.background;
        out.ray_scene_intersections = 1;
        return out;
    }
    const SurfaceInteraction& si = ir.hit.si;
    Bsdf bsdf = compute_scattering_functions(scene.materials[ir.hit.shape->material], si, &scene.textures);
    size_t ray_count = 1;
    Spectrumf sum_li = direct_lighting(scene, si, bsdf, sampler, st);
    if (depth == 0 || is_specular) sum_li += emitted_radiance(scene.lights, si, -ray.d);
    if (depth + 1 < prm.max_depth) {
        const int types[2] = {BX_REFLECTION, BX_TRANSMISSION};
        for (int k = 0; k < 2; ++k) {
            BxdfSample bs = bsdf.sample_f(si.wo, Point2f(0.0f, 0.0f), BX_SPECULAR | types[k]);
            if (bs.sample_type != BX_NONE) {
                Rayf refl = Interaction(si.p, si.n).spawn_ray(bs.wi);
                RadianceResult sub =
                    whitted_li(prm, refl, scene, depth + 1, sampler, (bs.sample_type & BX_SPECULAR) != 0, st);
                sub.li = bs.f * sub.li * std::fabs(bs.wi.dot_n(si.shading.n));
                sum_li += sub.li;
                ray_count += sub.ray_scene_intersections;
            }
        }
    }
    out.li = sum_li;
    out.ray_scene_intersections = ray_count;
    return out;
}

// trait Integrator::li — integrators/mod.rs:94-101 + the debug integrators
inline RadianceResult integrator_li(const IntegratorParams& prm, Rayf ray, const Scene& scene, Sampler& sampler,
                                    TraceStats* st) {
    switch (prm.kind) {
        case INTEGRATOR_PATH: return path_li(prm, ray, scene, sampler, st);
        case INTEGRATOR_WHITTED: return whitted_li(prm, ray, scene, 0, sampler, false, st);
        default: {
            IntersectionResult ir = scene.bvh.intersect(ray);
            if (st) {
                st->closest_rays += 1;
                st->closest_node_tests += ir.intersection_test_count;
                st->closest_shape_tests += ir.shape_test_count;
            }
            RadianceResult r;
            r.ray_scene_intersections = 1;
            if (prm.kind == INTEGRATOR_BVH_INTERSECTIONS) {  // bvh_heatmap.rs:25-40
                r.li = Spectrumf((float)ir.intersection_test_count, (float)ir.intersection_count,
                                 ir.has_hit ? (float)ir.intersection_count : 0.0f);
            } else if (!ir.has_hit) {
                r.li = Spectrumf::zeros();
            } else if (prm.kind == INTEGRATOR_GEOMETRY_NORMALS) {  // geometry_normals.rs:24-33
                Normalf n = ir.hit.si.n;
                r.li = Spectrumf(n.x, n.y, n.z) / 2.0f + 0.5f;
            } else {  // shading_normals.rs
                Normalf n = ir.hit.si.shading.n;
                r.li = Spectrumf(n.x, n.y, n.z) / 2.0f + 0.5f;
            }
            return r;
        }
    }
}

struct Tile {
    uint16_t x0, y0, x1, y1;  // bb.p_min, bb.p_max (exclusive)
};

// trait Integrator::render — integrators/mod.rs:120-185 (non-accumulating film).
// `accumulating_sample` >= 0 reproduces the accumulate path (one sample with
// global index tile.sample, raw value stored).
inline size_t render_tile(const IntegratorParams& prm, const Scene& scene, const Camera& camera, const Sampler& sampler_proto,
                          const Tile& tile, float* tile_pixels, int accumulating_sample, TraceStats* st,
                          float* per_sample /* optional: 3*spp floats per pixel */ = nullptr) {
    uint32_t tile_width = (uint32_t)tile.x1 - tile.x0;
    Sampler sampler = sampler_proto;
    size_t ray_count = 0;
    for (uint32_t py = tile.y0; py < tile.y1; ++py) {
        for (uint32_t px = tile.x0; px < tile.x1; ++px) {
            Spectrumf color = Spectrumf::zeros();
            uint32_t sample_count = accumulating_sample >= 0 ? 1u : sampler.samples_per_pixel();
            uint32_t pixel_offset = (py - tile.y0) * tile_width + (px - tile.x0);
            for (uint32_t sample_index = 0; sample_index < sample_count; ++sample_index) {
                uint32_t global_index = accumulating_sample >= 0 ? (uint32_t)accumulating_sample : sample_index;
                sampler.start_pixel_sample((uint16_t)px, (uint16_t)py, global_index, 0);
                Point2f p_film = Point2f((float)px, (float)py) + sampler.get_2d();
                Rayf ray = camera.ray(p_film);
                RadianceResult res = integrator_li(prm, ray, scene, sampler, st);
                color += res.li;
                ray_count += res.ray_scene_intersections;
                if (per_sample) {
                    float* o = per_sample + ((size_t)pixel_offset * sample_count + sample_index) * 3;
                    o[0] = res.li.r;
                    o[1] = res.li.g;
                    o[2] = res.li.b;
                }
            }
            color /= (float)sample_count;
            tile_pixels[3 * pixel_offset + 0] = color.r;
            tile_pixels[3 * pixel_offset + 1] = color.g;
            tile_pixels[3 * pixel_offset + 2] = color.b;
        }
    }
    return ray_count;
}

// film.rs:299-376 — tiles clipped to the film, ordered as an outward spiral
inline std::vector<Tile> film_tiles(uint16_t res_x, uint16_t res_y, uint16_t tile_dim) {
    int h_tiles = (int)std::ceil((float)res_x / (float)tile_dim);
    int v_tiles = (int)std::ceil((float)res_y / (float)tile_dim);
    std::vector<Tile> grid((size_t)h_tiles * v_tiles);
    std::vector<char> present((size_t)h_tiles * v_tiles, 0);
    for (uint32_t j = 0; j < res_y; j += tile_dim)
        for (uint32_t i = 0; i < res_x; i += tile_dim) {
            Tile t;
            t.x0 = (uint16_t)i;
            t.y0 = (uint16_t)j;
            t.x1 = (uint16_t)((i + tile_dim) < res_x ? (i + tile_dim) : res_x);
            t.y1 = (uint16_t)((j + tile_dim) < res_y ? (j + tile_dim) : res_y);
            size_t k = (size_t)(j / tile_dim) * h_tiles + (i / tile_dim);
            grid[k] = t;
            present[k] = 1;
        }
    int center_x = (h_tiles / 2) - (1 - h_tiles % 2);
    int center_y = (v_tiles / 2) - (1 - v_tiles % 2);
    int max_dim = h_tiles > v_tiles ? h_tiles : v_tiles;
    int x = 0, y = 0, dx = 0, dy = -1;
    std::vector<Tile> queue;
    queue.reserve(grid.size());
    for (int it = 0; it < max_dim * max_dim; ++it) {
        int tile_x = center_x + x, tile_y = center_y + y;
        if (tile_x >= 0 && tile_x < h_tiles && tile_y >= 0 && tile_y < v_tiles) {
            size_t k = (size_t)tile_y * h_tiles + tile_x;
            if (present[k]) {
                queue.push_back(grid[k]);
                present[k] = 0;
            }
        }
        if (x == y || (x < 0 && x == -y) || (x > 0 && x == 1 - y)) {
            int tmp = dx;
            dx = dy;
            dy = tmp;
            dx *= -1;
        }
        x += dx;
        y += dy;
    }
    return queue;
}

}  // namespace orc
