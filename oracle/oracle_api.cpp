// ORACLE — TEST INFRASTRUCTURE ONLY.  C entry points over the header-only
// restatement (see oracle_api.h).  Build: make -C oracle
#include "oracle_api.h"

#include <atomic>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "orender.h"

using namespace orc;

struct orc_scene {
    Scene scene;
    std::vector<Shape> source_shapes;
};

static Transformf xf_from(const float* m, const float* mi) {
    Matrix4x4f a, b;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            a.m[i][j] = m[4 * i + j];
            b.m[i][j] = mi[4 * i + j];
        }
    return Transformf(a, b);
}
static void xf_to(const Transformf& t, float* m, float* mi) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            if (m) m[4 * i + j] = t.m.m[i][j];
            if (mi) mi[4 * i + j] = t.m_inv.m[i][j];
        }
}
static Spectrumf sp(const float* v) { return Spectrumf(v[0], v[1], v[2]); }

template <class T> static Matrix4x4<T> m_from(const T* m) {
    Matrix4x4<T> a;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) a.m[i][j] = m[4 * i + j];
    return a;
}
template <class T> static void m_to(const Matrix4x4<T>& a, T* m) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) m[4 * i + j] = a.m[i][j];
}
template <class T> static Transform<T> rot_any(int axis, T theta, const T* av) {
    switch (axis) {
        case 0: return rotation_x(theta);
        case 1: return rotation_y(theta);
        case 2: return rotation_z(theta);
        default: return rotation(theta, Vec3<T>(av[0], av[1], av[2]));
    }
}
extern "C" {

static Light light_from(const orc_light_desc& l) {
    Light L;
    L.kind = (int)l.kind;
    L.p = Point3f(l.p[0], l.p[1], l.p[2]);
    L.w = Vec3f(l.p[0], l.p[1], l.p[2]);
    L.i = sp(l.i);
    L.cos_total_width = l.cos_total_width;
    L.cos_falloff_start = l.cos_falloff_start;
    L.world_to_light = xf_from(l.world_to_light, l.world_to_light);
    L.sample_to_world = xf_from(l.sample_to_world, l.sample_to_world_inv);
    L.area = l.area;
    return L;
}

int orc_scene_create(const orc_scene_desc* d, orc_scene** out) {
    if (!d || !out) return 1;
    orc_scene* s = new orc_scene();
    Scene& sc = s->scene;
    sc.points.assign(d->points, d->points + 3 * (size_t)d->n_vertices);
    if (d->normals) sc.normals.assign(d->normals, d->normals + 3 * (size_t)d->n_vertices);
    if (d->uvs) sc.uvs.assign(d->uvs, d->uvs + 2 * (size_t)d->n_vertices);
    sc.geom.points = sc.points.data();
    sc.geom.normals = d->normals ? sc.normals.data() : nullptr;
    sc.geom.uvs = d->uvs ? sc.uvs.data() : nullptr;
    for (uint32_t i = 0; i < d->n_meshes; ++i) {
        Mesh m;
        m.has_normals = d->meshes[i].has_normals != 0;
        m.has_uvs = d->meshes[i].has_uvs != 0;
        m.swaps_handedness = d->meshes[i].swaps_handedness != 0;
        sc.geom.meshes.push_back(m);
    }
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        Material m;
        m.kind = (int)d->materials[i].kind;
        m.a = sp(d->materials[i].a);
        m.b = sp(d->materials[i].b);
        m.c = d->materials[i].c;
        m.remap_roughness = (d->materials[i].flags & 1u) != 0;
        m.a_texture = (m.kind == MAT_MATTE && (d->materials[i].flags & 2u)) ? (int)d->materials[i].a_texture : -1;
        if (m.a_texture >= (int)d->n_textures) {
            delete s;
            return 5;
        }
        sc.materials.push_back(m);
    }
    for (uint32_t i = 0; i < d->n_textures; ++i) {
        ImageTexture t;
        t.width = d->textures[i].width;
        t.height = d->textures[i].height;
        const float* q = d->textures[i].rgb;
        for (size_t k = 0; k < t.width * t.height; ++k) t.data.push_back(Spectrumf(q[3 * k], q[3 * k + 1], q[3 * k + 2]));
        sc.textures.push_back(t);
    }
    for (uint32_t i = 0; i < d->n_lights; ++i) sc.lights.push_back(light_from(d->lights[i]));
    sc.background = sp(d->background);
    std::vector<Shape>& shapes = s->source_shapes;
    shapes.reserve((size_t)d->n_triangles + d->n_spheres);
    for (uint32_t i = 0; i < d->n_triangles; ++i) {
        Shape sh;
        sh.kind = SHAPE_TRIANGLE;
        sh.v[0] = d->indices[3 * i];
        sh.v[1] = d->indices[3 * i + 1];
        sh.v[2] = d->indices[3 * i + 2];
        sh.mesh = d->tri_mesh ? d->tri_mesh[i] : 0;
        sh.material = d->tri_material ? d->tri_material[i] : 0;
        sh.area_light = d->tri_area_light ? d->tri_area_light[i] : -1;
        sh.radius = 0.0f;
        sh.sphere_swaps = false;
        sh.source_index = i;
        shapes.push_back(sh);
    }
    for (uint32_t i = 0; i < d->n_spheres; ++i) {
        Shape sh;
        sh.kind = SHAPE_SPHERE;
        sh.v[0] = sh.v[1] = sh.v[2] = 0;
        sh.mesh = 0;
        sh.object_to_world = xf_from(d->spheres[i].object_to_world, d->spheres[i].world_to_object);
        sh.world_to_object = sh.object_to_world.inverted();
        sh.radius = d->spheres[i].radius;
        sh.sphere_swaps = sh.object_to_world.swaps_handedness();
        sh.material = d->spheres[i].material;
        sh.area_light = -1;
        sh.source_index = d->n_triangles + i;
        shapes.push_back(sh);
    }
    if (shapes.empty()) {
        delete s;
        return 2;
    }
    if (d->shape_order) {
        std::vector<Shape> ordered;
        ordered.reserve(shapes.size());
        for (size_t i = 0; i < shapes.size(); ++i) {
            if (d->shape_order[i] >= shapes.size()) {
                delete s;
                return 4;
            }
            ordered.push_back(shapes[d->shape_order[i]]);
        }
        shapes.swap(ordered);
    }
    bool ok = sc.bvh.build(&sc.geom, shapes, d->max_shapes_in_node, (int)d->split_method);
    if (!ok) {
        delete s;
        return 3;
    }
    *out = s;
    return 0;
}

void orc_scene_destroy(orc_scene* s) { delete s; }
size_t orc_scene_node_count(const orc_scene* s) { return s->scene.bvh.nodes.size(); }
size_t orc_scene_shape_count(const orc_scene* s) { return s->scene.bvh.shapes.size(); }

int orc_scene_export_bvh(const orc_scene* s, orc_bvh_node* nodes, uint32_t* shape_order) {
    const BVH& b = s->scene.bvh;
    if (nodes)
        for (size_t i = 0; i < b.nodes.size(); ++i) {
            const BVHNode& n = b.nodes[i];
            orc_bvh_node& o = nodes[i];
            o.bmin[0] = n.bounds.p_min.x; o.bmin[1] = n.bounds.p_min.y; o.bmin[2] = n.bounds.p_min.z;
            o.bmax[0] = n.bounds.p_max.x; o.bmax[1] = n.bounds.p_max.y; o.bmax[2] = n.bounds.p_max.z;
            o.a = n.a;
            o.count = n.count;
            o.axis = n.axis;
            o.is_leaf = n.is_leaf;
        }
    if (shape_order)
        for (size_t i = 0; i < b.shapes.size(); ++i) shape_order[i] = b.shapes[i].source_index;
    return 0;
}

int orc_camera_make(const orc_camera_params* p, orc_camera* out) {
    Camera c = Camera::make(Point3f(p->position[0], p->position[1], p->position[2]),
                            Point3f(p->target[0], p->target[1], p->target[2]), Vec3f(p->up[0], p->up[1], p->up[2]),
                            (int)p->fov_axis, p->fov_degrees, p->res_x, p->res_y);
    xf_to(c.camera_to_world, out->camera_to_world, out->camera_to_world_inv);
    xf_to(c.raster_to_camera, out->raster_to_camera, out->raster_to_camera_inv);
    return 0;
}

size_t orc_film_tiles(uint16_t res_x, uint16_t res_y, uint16_t tile_dim, orc_tile* out, size_t cap) {
    std::vector<Tile> t = film_tiles(res_x, res_y, tile_dim);
    if (out)
        for (size_t i = 0; i < t.size() && i < cap; ++i) {
            out[i].x0 = t[i].x0;
            out[i].y0 = t[i].y0;
            out[i].x1 = t[i].x1;
            out[i].y1 = t[i].y1;
        }
    return t.size();
}

// lights/rectangular_light.rs:31-42
void orc_make_rect_light(const float l2w[16], const float l2w_inv[16], const float radiance[3], const float size[2],
                         orc_light_desc* out) {
    std::memset(out, 0, sizeof(*out));
    Transformf light_to_world = xf_from(l2w, l2w_inv);
    Transformf sample_to_light = scale(size[0], 1.0f, size[1]) * translation(Vec3f(-0.5f, 0.0f, -0.5f));
    Transformf sample_to_world = light_to_world * sample_to_light;
    out->kind = LIGHT_RECT;
    for (int k = 0; k < 3; ++k) out->i[k] = radiance[k];
    xf_to(sample_to_world, out->sample_to_world, out->sample_to_world_inv);
    out->area = size[0] * size[1];
}
// lights/spot_light.rs:20-36
void orc_make_spot_light(const float l2w[16], const float l2w_inv[16], const float intensity[3], float total_width_deg,
                         float falloff_start_deg, orc_light_desc* out) {
    std::memset(out, 0, sizeof(*out));
    Transformf light_to_world = xf_from(l2w, l2w_inv);
    Transformf world_to_light = light_to_world.inverted();
    Point3f p = light_to_world.apply(Point3f(0.0f, 0.0f, 0.0f));
    out->kind = LIGHT_SPOT;
    out->p[0] = p.x; out->p[1] = p.y; out->p[2] = p.z;
    for (int k = 0; k < 3; ++k) out->i[k] = intensity[k];
    out->cos_total_width = lm::cosf_(total_width_deg * (O_PI / 180.0f));
    out->cos_falloff_start = lm::cosf_(falloff_start_deg * (O_PI / 180.0f));
    xf_to(world_to_light, out->world_to_light, nullptr);
}
// lights/point_light.rs:18-24
void orc_make_point_light(const float l2w[16], const float intensity[3], orc_light_desc* out) {
    std::memset(out, 0, sizeof(*out));
    Transformf light_to_world = xf_from(l2w, l2w);
    Point3f p = light_to_world.apply(Point3f(0.0f, 0.0f, 0.0f));
    out->kind = LIGHT_POINT;
    out->p[0] = p.x; out->p[1] = p.y; out->p[2] = p.z;
    for (int k = 0; k < 3; ++k) out->i[k] = intensity[k];
}

static Camera cam_from(const orc_camera* c) {
    Camera cam;
    cam.camera_to_world = xf_from(c->camera_to_world, c->camera_to_world_inv);
    cam.raster_to_camera = xf_from(c->raster_to_camera, c->raster_to_camera_inv);
    return cam;
}
static Sampler sampler_from(const orc_sampler_desc* d) {
    Sampler s;
    s.kind = (int)d->kind;
    s.nx = d->nx;
    s.ny = d->kind == SAMPLER_UNIFORM ? 1 : d->ny;
    s.jitter = d->jitter != 0;
    s.rng_seed = d->seed;
    s.px = s.py = 0;
    s.sample_index = s.dimension = 0;
    s.rng = Pcg32(d->seed, 0);
    return s;
}
static IntegratorParams integ_from(const orc_integrator_desc* d) {
    IntegratorParams p;
    p.kind = (int)d->kind;
    p.max_depth = d->max_depth;
    p.has_clamp = d->has_clamp != 0;
    p.indirect_clamp = d->indirect_clamp;
    return p;
}

// Integrator::li (integrators/mod.rs:94-101) for caller-supplied rays: the sampler is started at
// (pixel, sample_index) and `dimension` draws are consumed, as a render leaves it after the camera
// sample (integrators/mod.rs:152-166), then li runs.  out_li: 3 floats per ray; out_rays: li's ray counts.
int orc_li(const orc_scene* s, const orc_sampler_desc* smp, const orc_integrator_desc* integ, size_t n, const float* ray_o, const float* ray_d,
           const uint16_t* pixel_xy, const uint32_t* sample_index, uint32_t dimension, float* out_li, uint64_t* out_rays) {
    Sampler sampler = sampler_from(smp);
    IntegratorParams prm = integ_from(integ);
    for (size_t i = 0; i < n; ++i) {
        // the state li finds in a render: started at dimension 0, then `dimension` one-dimensional draws consumed
        // by the caller (two for the camera sample, integrators/mod.rs:152-166)
        sampler.start_pixel_sample(pixel_xy[2 * i], pixel_xy[2 * i + 1], sample_index[i], 0);
        for (uint32_t k = 0; k < dimension; ++k) (void)sampler.get_1d();
        Rayf ray(Point3f(ray_o[3 * i], ray_o[3 * i + 1], ray_o[3 * i + 2]), Vec3f(ray_d[3 * i], ray_d[3 * i + 1], ray_d[3 * i + 2]),
                 std::numeric_limits<float>::infinity());
        RadianceResult r = integrator_li(prm, ray, s->scene, sampler, nullptr);
        out_li[3 * i] = r.li.r;
        out_li[3 * i + 1] = r.li.g;
        out_li[3 * i + 2] = r.li.b;
        if (out_rays) out_rays[i] = r.ray_scene_intersections;
    }
    return 0;
}

}  // extern "C"

static int render_tiles_common(const orc_scene* s, const orc_camera* cam, const orc_sampler_desc* smp, const orc_integrator_desc* integ,
                               const orc_tile* tiles, const uint16_t* tile_samples, size_t n_tiles, float* out_rgb, uint64_t* out_ray_count,
                               orc_trace_stats* stats, int n_threads, float* per_sample);

extern "C" {

int orc_render_tiles(const orc_scene* s, const orc_camera* cam, const orc_sampler_desc* smp, const orc_integrator_desc* integ,
                     const orc_tile* tiles, size_t n_tiles, float* out_rgb, uint64_t* out_ray_count, orc_trace_stats* stats,
                     int n_threads, float* per_sample) {
    return render_tiles_common(s, cam, smp, integ, tiles, nullptr, n_tiles, out_rgb, out_ray_count, stats, n_threads, per_sample);
}

int orc_render_tiles_accumulating(const orc_scene* s, const orc_camera* cam, const orc_sampler_desc* smp, const orc_integrator_desc* integ,
                                  const orc_tile* tiles, const uint16_t* tile_samples, size_t n_tiles, float* out_rgb, uint64_t* out_ray_count,
                                  int n_threads) {
    return render_tiles_common(s, cam, smp, integ, tiles, tile_samples, n_tiles, out_rgb, out_ray_count, nullptr, n_threads, nullptr);
}

}  // extern "C"

static int render_tiles_common(const orc_scene* s, const orc_camera* cam, const orc_sampler_desc* smp, const orc_integrator_desc* integ,
                               const orc_tile* tiles, const uint16_t* tile_samples, size_t n_tiles, float* out_rgb, uint64_t* out_ray_count,
                               orc_trace_stats* stats, int n_threads, float* per_sample) {
    Camera camera = cam_from(cam);
    Sampler sampler = sampler_from(smp);
    IntegratorParams prm = integ_from(integ);
    uint32_t spp = sampler.samples_per_pixel();
    std::vector<size_t> offsets(n_tiles + 1, 0);
    for (size_t i = 0; i < n_tiles; ++i)
        offsets[i + 1] = offsets[i] + (size_t)(tiles[i].x1 - tiles[i].x0) * (size_t)(tiles[i].y1 - tiles[i].y0);
    if (n_threads <= 0) {
        unsigned hc = std::thread::hardware_concurrency();
        n_threads = hc > 1 ? (int)hc - 1 : 1;  // render_manager.rs:78
    }
    std::mutex queue_mutex;  // render_worker.rs:172-180: pop under a mutex
    size_t next_tile = 0;
    std::atomic<uint64_t> total_rays(0);
    std::mutex stats_mutex;
    TraceStats total_stats;
    auto worker = [&]() {
        TraceStats local;
        uint64_t rays = 0;
        for (;;) {
            size_t ti;
            {
                std::lock_guard<std::mutex> lk(queue_mutex);
                if (next_tile >= n_tiles) break;
                ti = next_tile++;
            }
            Tile t;
            t.x0 = tiles[ti].x0; t.y0 = tiles[ti].y0; t.x1 = tiles[ti].x1; t.y1 = tiles[ti].y1;
            rays += render_tile(prm, s->scene, camera, sampler, t, out_rgb + 3 * offsets[ti], tile_samples ? (int)tile_samples[ti] : -1, stats ? &local : nullptr,
                                per_sample ? per_sample + 3 * offsets[ti] * spp : nullptr);
        }
        total_rays += rays;
        if (stats) {
            std::lock_guard<std::mutex> lk(stats_mutex);
            total_stats.add(local);
        }
    };
    if (n_threads == 1) {
        worker();
    } else {
        std::vector<std::thread> th;
        for (int i = 0; i < n_threads; ++i) th.emplace_back(worker);
        for (auto& t : th) t.join();
    }
    if (out_ray_count) *out_ray_count = total_rays.load();
    if (stats) {
        stats->closest_rays = total_stats.closest_rays;
        stats->closest_node_tests = total_stats.closest_node_tests;
        stats->closest_shape_tests = total_stats.closest_shape_tests;
        stats->shadow_rays = total_stats.shadow_rays;
        stats->shadow_node_tests = total_stats.shadow_node_tests;
        stats->shadow_shape_tests = total_stats.shadow_shape_tests;
    }
    return 0;
}

extern "C" {

void orc_camera_rays(const orc_camera* cam, const orc_sampler_desc* smp, const orc_tile* tile, uint32_t sample_index,
                     float* out_o, float* out_d) {
    Camera camera = cam_from(cam);
    Sampler sampler = sampler_from(smp);
    size_t k = 0;
    for (uint32_t py = tile->y0; py < tile->y1; ++py)
        for (uint32_t px = tile->x0; px < tile->x1; ++px, ++k) {
            sampler.start_pixel_sample((uint16_t)px, (uint16_t)py, sample_index, 0);
            Point2f p_film = Point2f((float)px, (float)py) + sampler.get_2d();
            Rayf r = camera.ray(p_film);
            out_o[3 * k] = r.o.x; out_o[3 * k + 1] = r.o.y; out_o[3 * k + 2] = r.o.z;
            out_d[3 * k] = r.d.x; out_d[3 * k + 1] = r.d.y; out_d[3 * k + 2] = r.d.z;
        }
}

void orc_intersect(const orc_scene* s, size_t n, const float* o, const float* d, const float* t_max, int32_t* out_shape,
                   float* out_t, float* out_n, float* out_ns, float* out_p, uint32_t* out_node_tests,
                   uint32_t* out_node_hits, uint32_t* out_shape_tests) {
    for (size_t i = 0; i < n; ++i) {
        Rayf r(Point3f(o[3 * i], o[3 * i + 1], o[3 * i + 2]), Vec3f(d[3 * i], d[3 * i + 1], d[3 * i + 2]),
               t_max ? t_max[i] : std::numeric_limits<float>::infinity());
        IntersectionResult ir = s->scene.bvh.intersect(r);
        if (out_node_tests) out_node_tests[i] = (uint32_t)ir.intersection_test_count;
        if (out_node_hits) out_node_hits[i] = (uint32_t)ir.intersection_count;
        if (out_shape_tests) out_shape_tests[i] = (uint32_t)ir.shape_test_count;
        if (ir.has_hit) {
            out_shape[i] = (int32_t)ir.hit.shape->source_index;
            if (out_t) out_t[i] = ir.hit.t;
            const SurfaceInteraction& si = ir.hit.si;
            if (out_n) { out_n[3 * i] = si.n.x; out_n[3 * i + 1] = si.n.y; out_n[3 * i + 2] = si.n.z; }
            if (out_ns) { out_ns[3 * i] = si.shading.n.x; out_ns[3 * i + 1] = si.shading.n.y; out_ns[3 * i + 2] = si.shading.n.z; }
            if (out_p) { out_p[3 * i] = si.p.x; out_p[3 * i + 1] = si.p.y; out_p[3 * i + 2] = si.p.z; }
        } else {
            out_shape[i] = -1;
            if (out_t) out_t[i] = std::numeric_limits<float>::infinity();
            for (int k = 0; k < 3; ++k) {
                if (out_n) out_n[3 * i + k] = 0.0f;
                if (out_ns) out_ns[3 * i + k] = 0.0f;
                if (out_p) out_p[3 * i + k] = 0.0f;
            }
        }
    }
}

void orc_any_intersect(const orc_scene* s, size_t n, const float* o, const float* d, const float* t_max,
                       const int32_t* area_light, uint8_t* out_hit) {
    for (size_t i = 0; i < n; ++i) {
        Rayf r(Point3f(o[3 * i], o[3 * i + 1], o[3 * i + 2]), Vec3f(d[3 * i], d[3 * i + 1], d[3 * i + 2]), t_max[i]);
        out_hit[i] = s->scene.bvh.any_intersect(r, area_light ? area_light[i] : -1) ? 1 : 0;
    }
}

uint64_t orc_siphash13(const uint8_t* msg, size_t len) {
    SipHasher13 h;
    h.write(msg, (unsigned)len);
    return h.finish();
}
void orc_pcg32_sequence(uint64_t state, uint64_t stream, uint64_t advance, uint32_t* out, size_t n) {
    Pcg32 r(state, stream);
    r.advance(advance);
    for (size_t i = 0; i < n; ++i) out[i] = r.next_u32();
}
uint32_t orc_permutation_element(uint32_t i, uint32_t l, uint32_t p) { return permutation_element(i, l, p); }
void orc_sampler_sequence(const orc_sampler_desc* smp, uint16_t px, uint16_t py, uint32_t sample_index, const uint8_t* dims,
                          size_t n_draws, float* out) {
    Sampler s = sampler_from(smp);
    s.start_pixel_sample(px, py, sample_index, 0);
    for (size_t i = 0; i < n_draws; ++i) {
        if (dims[i] == 1) {
            out[2 * i] = s.get_1d();
            out[2 * i + 1] = 0.0f;
        } else {
            Point2f p = s.get_2d();
            out[2 * i] = p.x;
            out[2 * i + 1] = p.y;
        }
    }
}

float orc_sinf(float x) { return lm::sinf_(x); }
float orc_cosf(float x) { return lm::cosf_(x); }
float orc_tanf(float x) { return lm::tanf_(x); }
float orc_logf(float x) { return lm::logf_(x); }
float orc_atan2f(float y, float x) { return lm::atan2f_(y, x); }
float orc_acosf(float x) { return lm::acosf_(x); }
float orc_expf(float x) { return lm::expf_(x); }
// fn 0 sin, 1 cos, 2 tan, 3 log, 4 acos, 5 atan2(x[i], y[i]), 6 exp over arrays (the exhaustive comparisons of tests/ and tools/)
void orc_libm_array(int fn, size_t n, const float* x, const float* y, float* out) {
    for (size_t i = 0; i < n; ++i) {
        switch (fn) {
            case 0: out[i] = lm::sinf_(x[i]); break;
            case 1: out[i] = lm::cosf_(x[i]); break;
            case 2: out[i] = lm::tanf_(x[i]); break;
            case 3: out[i] = lm::logf_(x[i]); break;
            case 4: out[i] = lm::acosf_(x[i]); break;
            case 6: out[i] = lm::expf_(x[i]); break;
            default: out[i] = lm::atan2f_(x[i], y ? y[i] : 0.0f); break;
        }
    }
}

void orc_mat4_inverse_f32(const float* m, float* out) { m_to(m_from(m).inverted(), out); }
void orc_mat4_inverse_f64(const double* m, double* out) { m_to(m_from(m).inverted(), out); }
void orc_mat4_mul_f32(const float* a, const float* b, float* out) { m_to(m_from(a) * m_from(b), out); }
void orc_transform_apply_f32(const float* m, const float* m_inv, int what, const float* v, float* out) {
    Transformf t(m_from(m), m_from(m_inv));
    if (what == 0) {
        Vec3f r = t.apply(Vec3f(v[0], v[1], v[2]));
        out[0] = r.x; out[1] = r.y; out[2] = r.z;
    } else if (what == 1) {
        Point3f r = t.apply(Point3f(v[0], v[1], v[2]));
        out[0] = r.x; out[1] = r.y; out[2] = r.z;
    } else {
        Normalf r = t.apply(Normalf(v[0], v[1], v[2]));
        out[0] = r.x; out[1] = r.y; out[2] = r.z;
    }
}
void orc_transform_bounds_f32(const float* m, const float* m_inv, const float* bmin, const float* bmax, float* out6) {
    Transformf t(m_from(m), m_from(m_inv));
    Bounds3f b(Point3f(bmin[0], bmin[1], bmin[2]), Point3f(bmax[0], bmax[1], bmax[2]));
    Bounds3f r = t.apply(b);
    out6[0] = r.p_min.x; out6[1] = r.p_min.y; out6[2] = r.p_min.z;
    out6[3] = r.p_max.x; out6[4] = r.p_max.y; out6[5] = r.p_max.z;
}
void orc_look_at_f64(const double* pos, const double* target, const double* up, double* m, double* m_inv) {
    Transform<double> t = look_at(Point3<double>(pos[0], pos[1], pos[2]), Point3<double>(target[0], target[1], target[2]),
                                  Vec3<double>(up[0], up[1], up[2]));
    m_to(t.m, m);
    m_to(t.m_inv, m_inv);
}
void orc_look_at_f32(const float* pos, const float* target, const float* up, float* m, float* m_inv) {
    Transformf t = look_at(Point3f(pos[0], pos[1], pos[2]), Point3f(target[0], target[1], target[2]), Vec3f(up[0], up[1], up[2]));
    m_to(t.m, m);
    m_to(t.m_inv, m_inv);
}
void orc_rotation_f64(int axis, double theta, const double* axis_v, double* m, double* m_inv) {
    Transform<double> t = rot_any<double>(axis, theta, axis_v);
    m_to(t.m, m);
    m_to(t.m_inv, m_inv);
}
void orc_rotation_f32(int axis, float theta, const float* axis_v, float* m, float* m_inv) {
    Transformf t = rot_any<float>(axis, theta, axis_v);
    m_to(t.m, m);
    m_to(t.m_inv, m_inv);
}
void orc_vec3_ops_f32(const float* a, const float* b, float* out) {
    Vec3f va(a[0], a[1], a[2]), vb(b[0], b[1], b[2]);
    Vec3f c = va.cross(vb);
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
    out[3] = va.dot(vb);
    out[4] = va.len();
    Vec3f nn = va.normalized();
    out[5] = nn.x; out[6] = nn.y; out[7] = nn.z;
    out[8] = (float)va.max_dimension();
}
void orc_bounds_ops_f32(const float* bmin, const float* bmax, const float* p, float* out) {
    Bounds3f b;
    b.p_min = Point3f(bmin[0], bmin[1], bmin[2]);
    b.p_max = Point3f(bmax[0], bmax[1], bmax[2]);
    Vec3f o = b.offset(Point3f(p[0], p[1], p[2]));
    out[0] = o.x; out[1] = o.y; out[2] = o.z;
    out[3] = b.surface_area();
    out[4] = b.volume();
    out[5] = (float)b.maximum_extent();
}
void orc_math_kat_f32(int op, const float* a, const float* b, float* out) {
    Vec3f va(a[0], a[1], a[2]);
    Vec3f vb = b ? Vec3f(b[0], b[1], b[2]) : Vec3f();
    auto put3 = [&](float x, float y, float z) { out[0] = x; out[1] = y; out[2] = z; };
    auto box = [](const float* q) {
        Bounds3f r;
        r.p_min = Point3f(q[0], q[1], q[2]);
        r.p_max = Point3f(q[3], q[4], q[5]);
        return r;
    };
    auto put_box = [&](const Bounds3f& r) {
        out[0] = r.p_min.x; out[1] = r.p_min.y; out[2] = r.p_min.z;
        out[3] = r.p_max.x; out[4] = r.p_max.y; out[5] = r.p_max.z;
    };
    switch (op) {
        case 0: { Vec3f r = va.vmin(vb); put3(r.x, r.y, r.z); break; }
        case 1: { Vec3f r = va.vmax(vb); put3(r.x, r.y, r.z); break; }
        case 2: out[0] = va.min_comp(); break;
        case 3: out[0] = va.max_comp(); break;
        case 4: out[0] = (float)va.max_dimension(); break;
        case 5: { Vec3f r = va.permuted((int)b[0], (int)b[1], (int)b[2]); put3(r.x, r.y, r.z); break; }
        case 6: { Vec3f r = va.abs(); put3(r.x, r.y, r.z); break; }
        case 7: { Vec3f r = -va; put3(r.x, r.y, r.z); break; }
        case 8: {  // impl_point.rs:39-49: (1 - t) * self + t * other, per component
            const float t = b[3];
            put3((1.0f - t) * a[0] + t * b[0], (1.0f - t) * a[1] + t * b[1], (1.0f - t) * a[2] + t * b[2]);
            break;
        }
        case 20:  // Bounds3::lerp, impl_bounds.rs:101-111: (1 - t.c) * p_min.c + t.c * p_max.c
            put3((1.0f - b[0]) * a[0] + b[0] * a[3], (1.0f - b[1]) * a[1] + b[1] * a[4], (1.0f - b[2]) * a[2] + b[2] * a[5]);
            break;
        case 9: put_box(box(a).union_b(box(b))); break;
        case 10: put_box(box(a).union_p(Point3f(b[0], b[1], b[2]))); break;
        case 11: { Vec3f r = box(a).diagonal(); put3(r.x, r.y, r.z); break; }
        case 12: out[0] = box(a).inside(Point3f(b[0], b[1], b[2])) ? 1.0f : 0.0f; break;
        case 13: { Rayf r(Point3f(a[0], a[1], a[2]), Vec3f(a[3], a[4], a[5]), 1.0f); Point3f q = r.point(b[0]); put3(q.x, q.y, q.z); break; }
        case 14: out[0] = Normalf(a[0], a[1], a[2]).dot(Normalf(b[0], b[1], b[2])); break;
        case 15: out[0] = Normalf(a[0], a[1], a[2]).len_sqr(); break;
        case 16: out[0] = Point3f(a[0], a[1], a[2]).dist(Point3f(b[0], b[1], b[2])); break;
        case 17: out[0] = Point3f(a[0], a[1], a[2]).dist_sqr(Point3f(b[0], b[1], b[2])); break;
        case 18: out[0] = va.len_sqr(); break;
        case 19: put_box(Bounds3f(Point3f(a[0], a[1], a[2]), Point3f(b[0], b[1], b[2]))); break;
        // operator traits of Vec3 / Point3 / Normal (impl_vec.rs macros: one IEEE operation per component)
        case 21: { Vec3f r = va + vb; put3(r.x, r.y, r.z); break; }
        case 22: { Vec3f r = va - vb; put3(r.x, r.y, r.z); break; }
        case 23: { Vec3f r = va * b[0]; put3(r.x, r.y, r.z); break; }
        case 24: { Vec3f r = va / b[0]; put3(r.x, r.y, r.z); break; }
        case 25: out[0] = Normalf(a[0], a[1], a[2]).len(); break;
        case 26: { Normalf r = Normalf(a[0], a[1], a[2]).normalized(); put3(r.x, r.y, r.z); break; }
        case 27: {  // Transform::swaps_handedness, transform.rs:85-91 (a: 16 floats, row major)
            float rows[4][4];
            for (int i = 0; i < 16; ++i) rows[i / 4][i % 4] = a[i];
            out[0] = Transformf(Matrix4x4<float>::from_rows(rows), Matrix4x4<float>::from_rows(rows)).swaps_handedness() ? 1.0f : 0.0f;
            break;
        }
        case 28: {  // Matrix4x4::transposed (out: 16 floats)
            float rows[4][4];
            for (int i = 0; i < 16; ++i) rows[i / 4][i % 4] = a[i];
            Matrix4x4<float> t = Matrix4x4<float>::from_rows(rows).transposed();
            for (int i = 0; i < 16; ++i) out[i] = t.m[i / 4][i % 4];
            break;
        }
        default: out[0] = 0.0f;
    }
}
void orc_coordinate_system_f32(const float* v, float* v1, float* v2) {
    Vec3f a, b;
    coordinate_system(Vec3f(v[0], v[1], v[2]), a, b);
    v1[0] = a.x; v1[1] = a.y; v1[2] = a.z;
    v2[0] = b.x; v2[1] = b.y; v2[2] = b.z;
}
int orc_slab_test_f32(const float* bmin, const float* bmax, const float* o, const float* d, float t_max, float* tmin, float* tmax) {
    Bounds3f b;
    b.p_min = Point3f(bmin[0], bmin[1], bmin[2]);
    b.p_max = Point3f(bmax[0], bmax[1], bmax[2]);
    Rayf r(Point3f(o[0], o[1], o[2]), Vec3f(d[0], d[1], d[2]), t_max);
    Vec3f inv_dir(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
    b.slab_test(r, inv_dir, *tmin, *tmax);
    return *tmin <= *tmax ? 1 : 0;
}

static SurfaceInteraction si_for_bsdf(const float* n_geom, const float* n_shading, const float* dpdu) {
    SurfaceInteraction si;
    si.n = Normalf(n_geom[0], n_geom[1], n_geom[2]);
    si.shading.n = Normalf(n_shading[0], n_shading[1], n_shading[2]);
    si.shading.dpdu = Vec3f(dpdu[0], dpdu[1], dpdu[2]);
    si.dpdu = si.shading.dpdu;
    return si;
}
static Material mat_from(const orc_material_desc* m) {
    Material r;
    r.kind = (int)m->kind;
    r.a = sp(m->a);
    r.b = sp(m->b);
    r.c = m->c;
    r.remap_roughness = (m->flags & 1u) != 0;
    return r;
}
void orc_bsdf_eval(const orc_material_desc* m, const float* n_geom, const float* n_shading, const float* dpdu,
                   const float* wo, const float* wi, float* out_f) {
    SurfaceInteraction si = si_for_bsdf(n_geom, n_shading, dpdu);
    Bsdf b = compute_scattering_functions(mat_from(m), si);
    Spectrumf f = b.f(Vec3f(wo[0], wo[1], wo[2]), Vec3f(wi[0], wi[1], wi[2]), BX_ALL);
    out_f[0] = f.r; out_f[1] = f.g; out_f[2] = f.b;
}
void orc_bsdf_sample(const orc_material_desc* m, const float* n_geom, const float* n_shading, const float* dpdu,
                     const float* wo, const float* u, float* out) {
    SurfaceInteraction si = si_for_bsdf(n_geom, n_shading, dpdu);
    Bsdf b = compute_scattering_functions(mat_from(m), si);
    BxdfSample s = b.sample_f(Vec3f(wo[0], wo[1], wo[2]), Point2f(u[0], u[1]), BX_ALL);
    out[0] = s.wi.x; out[1] = s.wi.y; out[2] = s.wi.z;
    out[3] = s.f.r; out[4] = s.f.g; out[5] = s.f.b;
    out[6] = s.pdf;
    out[7] = (float)s.sample_type;
}

// Light::sample_li (lights/mod.rs:29-32) and the ray of its VisibilityTester (visibility.rs:21-23) for n surface points
void orc_light_sample(const orc_light_desc* light, int32_t light_index, size_t n, const float* p, const float* n_geom, const float* u,
                      float* out18) {
    const Light L = light_from(*light);
    for (size_t i = 0; i < n; ++i) {
        SurfaceInteraction si;
        si.p = Point3f(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
        si.n = Normalf(n_geom[3 * i], n_geom[3 * i + 1], n_geom[3 * i + 2]);
        LightSample s = sample_li(L, light_index, si, Point2f(u[2 * i], u[2 * i + 1]));
        Rayf r = s.vis.ray();
        float* o = out18 + 18 * i;
        o[0] = s.l.x; o[1] = s.l.y; o[2] = s.l.z;
        o[3] = s.li.r; o[4] = s.li.g; o[5] = s.li.b;
        o[6] = s.pdf;
        o[7] = s.has_vis ? 1.0f : 0.0f;
        o[8] = (float)s.vis.area_light;
        o[9] = s.vis.p1.p.x; o[10] = s.vis.p1.p.y; o[11] = s.vis.p1.p.z;
        o[12] = r.o.x; o[13] = r.o.y; o[14] = r.o.z;
        o[15] = r.d.x; o[16] = r.d.y; o[17] = r.d.z;
    }
}

void orc_texture_eval(const orc_texture_desc* tex, size_t n, const float* uv, float* out_rgb) {
    ImageTexture t;
    t.width = tex->width;
    t.height = tex->height;
    for (size_t k = 0; k < t.width * t.height; ++k) t.data.push_back(Spectrumf(tex->rgb[3 * k], tex->rgb[3 * k + 1], tex->rgb[3 * k + 2]));
    for (size_t i = 0; i < n; ++i) {
        SurfaceInteraction si;
        si.uv = Point2f(uv[2 * i], uv[2 * i + 1]);
        Spectrumf c = t.evaluate(si);
        out_rgb[3 * i] = c.r;
        out_rgb[3 * i + 1] = c.g;
        out_rgb[3 * i + 2] = c.b;
    }
}

size_t orc_sizeof(int what) {
    switch (what) {
        case 0: return sizeof(orc_scene_desc);
        case 1: return sizeof(orc_material_desc);
        case 2: return sizeof(orc_light_desc);
        case 3: return sizeof(orc_sphere_desc);
        case 4: return sizeof(orc_camera);
        case 5: return sizeof(orc_camera_params);
        case 6: return sizeof(orc_sampler_desc);
        case 7: return sizeof(orc_integrator_desc);
        case 8: return sizeof(orc_tile);
        case 9: return sizeof(orc_bvh_node);
        case 10: return sizeof(orc_mesh_desc);
        case 11: return sizeof(orc_trace_stats);
        default: return 0;
    }
}

}  // extern "C"
