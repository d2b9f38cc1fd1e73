/* ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * C entry points of the CPU restatement (liboracle.so), loaded through ctypes by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
 * (libyuki_hip.so) never includes, links or calls anything declared here.
 *
 * The POD layouts deliberately equal those of include/yuki_hip.h so one ctypes
 * definition feeds both sides; tests/test_abi.py checks the sizes agree.
 */
#ifndef ORACLE_API_H
#define ORACLE_API_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_mesh_desc {
    uint8_t has_normals, has_uvs, swaps_handedness, pad;
} orc_mesh_desc;

typedef struct orc_sphere_desc {
    float object_to_world[16];
    float world_to_object[16];
    float radius;
    int32_t material;
} orc_sphere_desc;

typedef struct orc_material_desc {
    uint32_t kind;   /* 0 matte, 1 glass, 2 metal, 3 glossy */
    float a[3];      /* matte Kd | glass R | metal eta | glossy Rs */
    float b[3];      /*          | glass T | metal k   |           */
    float c;         /* matte sigma(rad) | glass eta | metal/glossy roughness */
    uint32_t flags;  /* bit0: remap_roughness ; bit1: matte Kd = textures[a_texture] */
    uint32_t a_texture;
} orc_material_desc;

/* ImageTexture<Spectrum<f32>> (textures/image_texture.rs:49-56): row-major RGB, row 0 = top */
typedef struct orc_texture_desc {
    uint32_t width, height;
    const float* rgb;
} orc_texture_desc;

typedef struct orc_light_desc {
    uint32_t kind;  /* 0 point, 1 spot, 2 distant, 3 rectangular */
    float p[3];     /* point/spot position ; distant: direction w */
    float i[3];     /* intensity (point/spot) | radiance (distant/rect) */
    float cos_total_width, cos_falloff_start;
    float world_to_light[16];      /* spot */
    float sample_to_world[16];     /* rect */
    float sample_to_world_inv[16]; /* rect */
    float area;                    /* rect */
} orc_light_desc;

typedef struct orc_scene_desc {
    uint32_t n_vertices;
    const float* points;  /* 3*n_vertices, world space */
    const float* normals; /* 3*n_vertices or NULL */
    const float* uvs;     /* 2*n_vertices or NULL */
    uint32_t n_triangles;
    const uint32_t* indices;       /* 3*n_triangles */
    const uint32_t* tri_mesh;      /* n_triangles */
    const int32_t* tri_material;   /* n_triangles */
    const int32_t* tri_area_light; /* n_triangles, -1 = none */
    uint32_t n_meshes;
    const orc_mesh_desc* meshes;
    uint32_t n_spheres;
    const orc_sphere_desc* spheres;
    uint32_t n_materials;
    const orc_material_desc* materials;
    uint32_t n_lights;
    const orc_light_desc* lights;
    float background[3];
    uint32_t split_method; /* 0 SAH, 1 Middle, 2 EqualCounts */
    uint32_t max_shapes_in_node;
    /* order of Scene.shapes as handed to BoundingVolumeHierarchy::new (the pbrt loader keeps
     * file order, pbrt/mod.rs:807-822); ids: triangles first, then spheres.  NULL = natural */
    const uint32_t* shape_order;
    uint32_t n_textures;
    const orc_texture_desc* textures;
} orc_scene_desc;

typedef struct orc_camera {
    float camera_to_world[16], camera_to_world_inv[16];
    float raster_to_camera[16], raster_to_camera_inv[16];
} orc_camera;

typedef struct orc_camera_params {
    float position[3], target[3], up[3];
    uint32_t fov_axis; /* 0 X, 1 Y */
    float fov_degrees;
    uint16_t res_x, res_y;
} orc_camera_params;

typedef struct orc_sampler_desc {
    uint32_t kind; /* 0 uniform, 1 stratified */
    uint32_t nx, ny;
    uint32_t jitter;
    uint64_t seed;
} orc_sampler_desc;

typedef struct orc_integrator_desc {
    uint32_t kind; /* 0 whitted, 1 path, 2 bvh intersections, 3 geometry normals, 4 shading normals */
    uint32_t max_depth;
    uint32_t has_clamp;
    float indirect_clamp;
} orc_integrator_desc;

typedef struct orc_tile {
    uint16_t x0, y0, x1, y1;
} orc_tile;

/* 32-byte BVH node in the reference's layout (bvh.rs:536-556) */
typedef struct orc_bvh_node {
    float bmin[3], bmax[3];
    uint32_t a;
    uint16_t count;
    uint8_t axis, is_leaf;
} orc_bvh_node;

typedef struct orc_trace_stats {
    uint64_t closest_rays, closest_node_tests, closest_shape_tests;
    uint64_t shadow_rays, shadow_node_tests, shadow_shape_tests;
} orc_trace_stats;

typedef struct orc_scene orc_scene;

/* scene */
int orc_scene_create(const orc_scene_desc* desc, orc_scene** out);
void orc_scene_destroy(orc_scene* s);
size_t orc_scene_node_count(const orc_scene* s);
size_t orc_scene_shape_count(const orc_scene* s);
int orc_scene_export_bvh(const orc_scene* s, orc_bvh_node* nodes, uint32_t* shape_order);

/* host-side helpers restated from camera.rs / transforms.rs / lights */
int orc_camera_make(const orc_camera_params* p, orc_camera* out);
size_t orc_film_tiles(uint16_t res_x, uint16_t res_y, uint16_t tile_dim, orc_tile* out, size_t cap);
void orc_make_rect_light(const float l2w[16], const float l2w_inv[16], const float radiance[3], const float size[2],
                         orc_light_desc* out);
void orc_make_spot_light(const float l2w[16], const float l2w_inv[16], const float intensity[3], float total_width_deg,
                         float falloff_start_deg, orc_light_desc* out);
void orc_make_point_light(const float l2w[16], const float intensity[3], orc_light_desc* out);

/* Integrator::render over a list of tiles; out_rgb is tile-major, each tile
 * row-major, 3 floats per pixel.  n_threads <= 0: hardware_concurrency-1
 * workers popping tiles from a mutex-guarded queue like render_worker.rs. */
int orc_render_tiles(const orc_scene* s, const orc_camera* cam, const orc_sampler_desc* smp, const orc_integrator_desc* integ,
                     const orc_tile* tiles, size_t n_tiles, float* out_rgb, uint64_t* out_ray_count, orc_trace_stats* stats,
                     int n_threads, float* per_sample /* nullable */);

/* Integrator::render(accumulating = true): one sample per pixel with global index
 * tile_samples[t] (FilmTile.sample), raw value stored (integrators/mod.rs:146-182) */
int orc_render_tiles_accumulating(const orc_scene* s, const orc_camera* cam, const orc_sampler_desc* smp, const orc_integrator_desc* integ,
                                  const orc_tile* tiles, const uint16_t* tile_samples, size_t n_tiles, float* out_rgb, uint64_t* out_ray_count,
                                  int n_threads);
/* Integrator::li (integrators/mod.rs:94-101) for caller-supplied rays; the sampler is started at (pixel, sample_index) and `dimension` draws are consumed first */
int orc_li(const orc_scene* s, const orc_sampler_desc* smp, const orc_integrator_desc* integ, size_t n, const float* ray_o, const float* ray_d,
           const uint16_t* pixel_xy, const uint32_t* sample_index, uint32_t dimension, float* out_li, uint64_t* out_rays);

/* per-stage entry points */
void orc_camera_rays(const orc_camera* cam, const orc_sampler_desc* smp, const orc_tile* tile, uint32_t sample_index,
                     float* out_o, float* out_d);
void orc_intersect(const orc_scene* s, size_t n, const float* o, const float* d, const float* t_max, int32_t* out_shape,
                   float* out_t, float* out_n, float* out_ns, float* out_p, uint32_t* out_node_tests,
                   uint32_t* out_node_hits, uint32_t* out_shape_tests);
void orc_any_intersect(const orc_scene* s, size_t n, const float* o, const float* d, const float* t_max,
                       const int32_t* area_light, uint8_t* out_hit);

/* sampler / rng / hash KAT hooks */
uint64_t orc_siphash13(const uint8_t* msg, size_t len);
void orc_pcg32_sequence(uint64_t state, uint64_t stream, uint64_t advance, uint32_t* out, size_t n);
uint32_t orc_permutation_element(uint32_t i, uint32_t l, uint32_t p);
void orc_sampler_sequence(const orc_sampler_desc* smp, uint16_t px, uint16_t py, uint32_t sample_index,
                          const uint8_t* dims /* 1 or 2 per draw */, size_t n_draws, float* out /* 2 per draw */);

/* libm KAT hooks */
float orc_sinf(float x);
float orc_cosf(float x);
float orc_tanf(float x);
float orc_logf(float x);
float orc_atan2f(float y, float x);
float orc_acosf(float x);
float orc_expf(float x);
void orc_libm_array(int fn, size_t n, const float* x, const float* y, float* out);

/* math KAT hooks (replay of the reference's tests/src/ *.rs) */
void orc_mat4_inverse_f32(const float* m, float* out);
void orc_mat4_inverse_f64(const double* m, double* out);
void orc_mat4_mul_f32(const float* a, const float* b, float* out);
void orc_transform_apply_f32(const float* m, const float* m_inv, int what /*0 vec,1 point,2 normal*/, const float* v,
                             float* out);
void orc_transform_bounds_f32(const float* m, const float* m_inv, const float* bmin, const float* bmax, float* out6);
void orc_look_at_f64(const double* pos, const double* target, const double* up, double* m, double* m_inv);
void orc_look_at_f32(const float* pos, const float* target, const float* up, float* m, float* m_inv);
void orc_rotation_f64(int axis /*0 x,1 y,2 z,3 arbitrary*/, double theta, const double* axis_v, double* m, double* m_inv);
void orc_rotation_f32(int axis, float theta, const float* axis_v, float* m, float* m_inv);
void orc_vec3_ops_f32(const float* a, const float* b, float* out /* cross[3], dot, len(a), normalized(a)[3], max_dim(a) */);
void orc_bounds_ops_f32(const float* bmin, const float* bmax, const float* p, float* out /* offset[3], area, volume, max_extent */);
void orc_coordinate_system_f32(const float* v, float* v1, float* v2);
/* The rest of the reference's math unit tests (tests/src/{bounds,point,normal,vector,ray}.rs) replayed one operation at a
 * time.  a, b: up to 6 floats each (a box is min xyz, max xyz); out: up to 6 floats.  op:
 *   0 Vec3::min  1 Vec3::max  2 min_comp  3 max_comp  4 max_dimension  5 permuted(a; b = indices)  6 abs  7 neg
 *   8 Point3::lerp(a, b, t = b[3])  9 Bounds3::union_b  10 Bounds3::union_p(a, p = b)  11 Bounds3::diagonal
 *   12 Bounds3::inside(a, p = b) -> 0/1  13 Ray::point(o = a[0..3], d = a[3..6], t = b[0])  14 Normal::dot  15 Normal::len_sqr
 *   16 Point3::dist  17 Point3::dist_sqr  18 Vec3::len_sqr  19 Bounds3::new(p0 = a, p1 = b) (sorted corners)  20 Bounds3::lerp(a, t = b) */
void orc_math_kat_f32(int op, const float* a, const float* b, float* out);
int orc_slab_test_f32(const float* bmin, const float* bmax, const float* o, const float* d, float t_max, float* tmin,
                      float* tmax);

/* BSDF KAT hooks */
void orc_bsdf_eval(const orc_material_desc* m, const float* n_geom, const float* n_shading, const float* dpdu,
                   const float* wo, const float* wi, float* out_f);
void orc_bsdf_sample(const orc_material_desc* m, const float* n_geom, const float* n_shading, const float* dpdu,
                     const float* wo, const float* u, float* out /* wi[3], f[3], pdf, type */);

/* Light::sample_li + VisibilityTester::ray for n surface points against one light; 18 floats per point:
 * l[3], li[3], pdf, has_vis, area_light, p1[3], shadow-ray origin[3], direction[3] */
void orc_light_sample(const orc_light_desc* light, int32_t light_index, size_t n, const float* p, const float* n_geom, const float* u,
                      float* out18);

/* ImageTexture::evaluate at n uv pairs (KAT hook) */
void orc_texture_eval(const orc_texture_desc* tex, size_t n, const float* uv, float* out_rgb);

size_t orc_sizeof(int what);

#ifdef __cplusplus
}
#endif
#endif
