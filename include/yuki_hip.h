/* yuki_hip.h — C ABI of the MI355X-native wavefront Path integrator.
 *
 * Drop-in boundary for the hot path of sndels/yuki: the private Rust trait
 *
 *     trait Integrator { fn li(..); fn render(&self, scratch, scene, camera,
 *         sampler, accumulating, tile, tile_pixels, early_termination_predicate)
 *         -> usize }                    (yuki/src/integrators/mod.rs:92-186)
 *
 * and the data its sibling traits describe (Sampler sampling/mod.rs:46-57,
 * Material materials/mod.rs:20-27, Shape shapes/mod.rs:26-39, Light
 * lights/mod.rs:29-37).  A Rust shim `impl Integrator for HipPath` binds exactly
 * these entry points; our own host code (C++ wrapper
 * yuki_hip.hpp, Python mirror yuki_amd/) and the parity tests call the same ones.
 *
 * Conventions
 *   - plain C, caller-owned buffers, no hidden allocation handed back;
 *   - every call returns a yk_status (the reference panics/asserts instead:
 *     integrators/mod.rs:131,141 -> YK_ERR_INVALID_ARGUMENT);
 *   - matrices are row-major float[16] (math/matrix.rs:16);
 *   - radiance buffers are tightly packed RGB float triples (Spectrum<f32>,
 *     math/spectrum.rs:45-55), tile-major, each tile row-major — the layout of
 *     `tile_pixels[ty*tile_width+tx]` (integrators/mod.rs:177-182);
 *   - the sampler seed is explicit (the reference draws it from thread_rng(),
 *     sampling/uniform.rs:37 — quirk 20 of SURVEY.md).
 */
#ifndef YUKI_HIP_H
#define YUKI_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YK_ABI_VERSION 1

typedef enum yk_status {
    YK_OK = 0,
    YK_ERR_INVALID_ARGUMENT = 1, /* contract violation (reference: assert!/panic) */
    YK_ERR_NO_DEVICE = 2,        /* HIP runtime / device unavailable */
    YK_ERR_DEVICE = 3,           /* a HIP call failed; see yk_last_error */
    YK_ERR_OUT_OF_MEMORY = 4,
    YK_ERR_UNSUPPORTED = 5,      /* e.g. Whitted deeper than 16, yk_li with a debug integrator, unsupported file content */
    YK_ERR_BVH_BUILD = 6,        /* reference: assert_ne!(mid,start) bvh.rs:368 */
    YK_ERR_CANCELLED = 7,        /* early_termination_predicate returned true */
    YK_ERR_STACK_OVERFLOW = 8    /* traversal stack > 64, reference: assert bvh.rs:174 */
} yk_status;

/* ---- scene description: flattened, world space, caller owned, read only ---- */

/* shapes/mesh.rs:7-43 — per `Mesh` flags (normals / uvs presence is per mesh) */
typedef struct yk_mesh_desc {
    uint8_t has_normals, has_uvs, swaps_handedness, pad;
} yk_mesh_desc;

/* shapes/sphere.rs:14-35 */
typedef struct yk_sphere_desc {
    float object_to_world[16];
    float world_to_object[16];
    float radius;
    int32_t material;
} yk_sphere_desc;

/* materials/{matte,glass,metal,glossy}.rs with ConstantTexture inputs
 * (textures/constant.rs) folded in */
typedef enum yk_material_kind { YK_MAT_MATTE = 0, YK_MAT_GLASS = 1, YK_MAT_METAL = 2, YK_MAT_GLOSSY = 3 } yk_material_kind;
typedef struct yk_material_desc {
    uint32_t kind;
    float a[3];     /* matte Kd | glass R | metal eta | glossy Rs */
    float b[3];     /*          | glass T | metal k   |           */
    float c;        /* matte sigma (radians) | glass eta | metal/glossy roughness */
    uint32_t flags; /* bit0: remap_roughness ; bit1: matte Kd comes from textures[a_texture] */
    uint32_t a_texture; /* ImageTexture index when bit1 is set (scene/pbrt/mod.rs:887-902) */
} yk_material_desc;
#define YK_MAT_FLAG_REMAP 1u
#define YK_MAT_FLAG_TEXTURED_A 2u

/* textures/image_texture.rs:49-56 `ImageTexture<Spectrum<f32>>`: row-major RGB, row 0 = top
 * row of the image file; evaluated point-sampled, repeat, v flipped (:81-111) */
typedef struct yk_texture_desc {
    uint32_t width, height;
    const float* rgb; /* 3 * width * height */
} yk_texture_desc;

/* lights/{point,spot,distant,rectangular}_light.rs — build with yk_make_*_light */
typedef enum yk_light_kind { YK_LIGHT_POINT = 0, YK_LIGHT_SPOT = 1, YK_LIGHT_DISTANT = 2, YK_LIGHT_RECT = 3 } yk_light_kind;
typedef struct yk_light_desc {
    uint32_t kind;
    float p[3]; /* point/spot position ; distant: direction w */
    float i[3]; /* intensity (point/spot) | radiance (distant/rect) */
    float cos_total_width, cos_falloff_start;
    float world_to_light[16];      /* spot */
    float sample_to_world[16];     /* rect */
    float sample_to_world_inv[16]; /* rect */
    float area;                    /* rect */
} yk_light_desc;

typedef enum yk_split_method { YK_SPLIT_SAH = 0, YK_SPLIT_MIDDLE = 1, YK_SPLIT_EQUAL_COUNTS = 2 } yk_split_method;

/* scene/mod.rs:41-49 `Scene` + SceneLoadSettings (:25-39) */
typedef struct yk_scene_desc {
    uint32_t n_vertices;
    const float* points;  /* 3*n_vertices, world space (Mesh::new pre-transforms, mesh.rs:27-33) */
    const float* normals; /* 3*n_vertices or NULL */
    const float* uvs;     /* 2*n_vertices or NULL */
    uint32_t n_triangles;
    const uint32_t* indices;       /* 3*n_triangles */
    const uint32_t* tri_mesh;      /* n_triangles -> meshes[] */
    const int32_t* tri_material;   /* n_triangles -> materials[] */
    const int32_t* tri_area_light; /* n_triangles -> lights[] or -1 (Triangle.area_light, triangle.rs:22) */
    uint32_t n_meshes;
    const yk_mesh_desc* meshes;
    uint32_t n_spheres; /* shape ids: triangles first, then spheres (see shape_order) */
    const yk_sphere_desc* spheres;
    uint32_t n_materials;
    const yk_material_desc* materials;
    uint32_t n_lights;
    const yk_light_desc* lights;
    float background[3];
    uint32_t split_method;       /* yk_split_method */
    uint32_t max_shapes_in_node; /* scene/mod.rs:36 default 1 */
    /* Order in which the shapes enter BoundingVolumeHierarchy::new (Scene.shapes, which the
     * pbrt loader fills in file order): n_triangles + n_spheres entries, entry < n_triangles
     * = that triangle, otherwise sphere (entry - n_triangles).  NULL = triangles, then spheres. */
    const uint32_t* shape_order;
    uint32_t n_textures;
    const yk_texture_desc* textures;
} yk_scene_desc;

/* camera.rs:19-22 `Camera` = two Transforms */
typedef struct yk_camera {
    float camera_to_world[16], camera_to_world_inv[16];
    float raster_to_camera[16], raster_to_camera_inv[16];
} yk_camera;

/* camera.rs:24-30 `CameraParameters` + the film resolution Camera::new reads */
typedef struct yk_camera_params {
    float position[3], target[3], up[3];
    uint32_t fov_axis; /* 0 = FoV::X, 1 = FoV::Y */
    float fov_degrees;
    uint16_t res_x, res_y;
} yk_camera_params;

/* sampling/mod.rs:16-19 `SamplerType` */
typedef enum yk_sampler_kind { YK_SAMPLER_UNIFORM = 0, YK_SAMPLER_STRATIFIED = 1 } yk_sampler_kind;
typedef struct yk_sampler_desc {
    uint32_t kind;
    uint32_t nx, ny; /* uniform: nx = pixel_samples ; stratified: pixel_samples.{x,y} */
    uint32_t jitter; /* stratified jitter_samples */
    uint64_t seed;   /* rng_seed */
} yk_sampler_desc;

/* integrators/mod.rs:33-40 `IntegratorType` */
typedef enum yk_integrator_kind {
    YK_INTEGRATOR_WHITTED = 0, /* whitted.rs:39-181; max_depth <= 16 on the device */
    YK_INTEGRATOR_PATH = 1,
    YK_INTEGRATOR_BVH_INTERSECTIONS = 2,
    YK_INTEGRATOR_GEOMETRY_NORMALS = 3,
    YK_INTEGRATOR_SHADING_NORMALS = 4
} yk_integrator_kind;
typedef struct yk_integrator_desc {
    uint32_t kind;
    uint32_t max_depth;   /* path.rs:20-23 Params */
    uint32_t has_clamp;   /* indirect_clamp.is_some() */
    float indirect_clamp;
} yk_integrator_desc;

/* film.rs:43-65 `FilmTile.bb` (Bounds2<u16>, max exclusive) */
typedef struct yk_tile {
    uint16_t x0, y0, x1, y1;
} yk_tile;

/* bvh.rs:536-556 — the reference's 32-byte node, exported for inspection/tests */
typedef struct yk_bvh_node {
    float bmin[3], bmax[3];
    uint32_t a;     /* interior: second_child_index ; leaf: first_shape_index */
    uint16_t count; /* leaf: shape_count */
    uint8_t axis, is_leaf;
} yk_bvh_node;

typedef struct yk_scene_info {
    uint64_t n_nodes, n_interior, n_shapes;
    float bounds_min[3], bounds_max[3];
    double build_seconds, upload_seconds;
    uint64_t device_bytes;
    uint32_t max_leaf_shapes, tree_depth;
} yk_scene_info;

typedef struct yk_render_stats {
    uint64_t rays;          /* closest-hit rays == the reference's ray_count (path.rs:87) */
    uint64_t shadow_rays;   /* any-hit rays, not part of the metric */
    uint64_t samples;       /* camera samples rendered */
    double seconds_total;   /* first launch -> film resolved (device time, HIP events) */
    double seconds_trace;   /* summed duration of the closest-hit traversal launches */
    double seconds_shadow;  /* summed duration of the any-hit traversal launches */
    double seconds_shade;   /* summed duration of the shade (BSDF+NEE) launches */
    uint32_t trace_launches, batches;
    uint32_t shadow_launches, reserved; /* any-hit traversal launches (two on bounces whose shadow rays are split) */
} yk_render_stats;

typedef struct yk_context yk_context;
typedef struct yk_scene yk_scene;

/* early_termination_predicate (integrators/mod.rs:129,153: the reference polls it once per pixel
 * sample; render_worker.rs:240-255 relies on that for "low latency kills").  Returning non-zero
 * aborts the render with YK_ERR_CANCELLED (tile contents undefined, as in render_worker.rs:252-255).
 * When is it polled:
 *   - before every batch is enqueued (all calls);
 *   - in a SYNCHRONOUS call — one that returns pixels to the host or is given `stats` — about every
 *     100 us while the GPU works.  On a non-zero answer a word in pinned host memory is set; one
 *     wave of every running traversal launch reads it whenever it claims work, raises a word in
 *     device memory and poisons the launch's queue head, so no wave claims again; every kernel (every
 *     block of the grid-stride ones) reads the device word when it starts and finds its queue
 *     empty; a job of many batches is enqueued two batches at a time.  The call returns after the
 *     drain (3-5 ms into a 1.5-s job; ~10 ms for a context's first interruption).  The next render on the context is unaffected.
 *   - an ASYNCHRONOUS submission (device output, stats == NULL) has returned before the GPU
 *     started: the caller interrupts it with yk_context_interrupt from any thread.
 * The predicate is called from the thread that made the call, never concurrently. */
typedef int (*yk_cancel_fn)(void* user);

/* ---- library / context ---------------------------------------------------- */
uint32_t yk_abi_version(void);
const char* yk_status_string(yk_status s);
yk_status yk_context_create(int device, yk_context** out);
void yk_context_destroy(yk_context* ctx);
yk_status yk_last_error(const yk_context* ctx, char* buf, size_t cap);
/* The hipStream_t every entry point of this context runs on when it is given no stream of the
 * caller's (a render also uses a side stream that joins it again before the call's last launch).
 * Lets a caller order its own device work — a collective, a film read-back — after a render
 * without a second stream: wrap the handle (torch.cuda.ExternalStream, hipStreamWaitEvent ...).
 * Owned by the context; NULL for a NULL context. */
void* yk_context_stream(const yk_context* ctx);
/* Stop what the context has enqueued (see yk_cancel_fn): callable from any thread while another one is
 * inside a render call on the same context.  Kernels stop at their next look at the word; the context's next
 * submission first waits for the interrupted one to drain. */
yk_status yk_context_interrupt(yk_context* ctx);
/* tuning knobs (none changes any result): "batch_paths" (camera samples per batch, 64 .. 2^29),
 * "streams" (1|2 work sets), "sample_buf_cap" (bytes), "time_kernels" (0 | 1: per-kernel
 * seconds in yk_render_stats for jobs of at least 2^20 samples | 2: always),
 * "packet_bounces" / "packet_shadow_bounces" (leading bounces traced by the wave-packet
 * kernels), "overlap_shadow" (0|1), "shade_reorder" (0|1: paths of a shade block dealt to
 * lanes by material kind), "top_nodes" (tree-top nodes the traversal kernels keep
 * in LDS, 0..1023; the kernels hold at most what they were built for) and "wide_bvh" (0: binary nodes only | 1: traverse the 4-wide collapse of the
 * BVH | 2, default: keep both, jobs of up to 6 M paths use the 4-wide one) — the last two apply
 * to scenes created afterwards. */
yk_status yk_context_set_option(yk_context* ctx, const char* key, int64_t value);

/* ---- host-side restatements (no GPU needed) -------------------------------- */
/* Camera::new, camera.rs:52-102 */
yk_status yk_camera_init(const yk_camera_params* params, yk_camera* out);
/* film_tiles / generate_tiles / outward_spiral, film.rs:299-376,409-475.
 * Returns the number of tiles; writes min(cap, n) of them in spiral order. */
size_t yk_film_tiles(uint16_t res_x, uint16_t res_y, uint16_t tile_dim, yk_tile* out, size_t cap);
/* RectangularLight::new rectangular_light.rs:31-42, SpotLight::new spot_light.rs:20-36,
 * PointLight::new point_light.rs:18-24 */
yk_status yk_make_rect_light(const float light_to_world[16], const float light_to_world_inv[16], const float radiance[3],
                             const float size[2], yk_light_desc* out);
yk_status yk_make_spot_light(const float light_to_world[16], const float light_to_world_inv[16], const float intensity[3],
                             float total_width_degrees, float falloff_start_degrees, yk_light_desc* out);
yk_status yk_make_point_light(const float light_to_world[16], const float intensity[3], yk_light_desc* out);
/* Film::update_tile (film.rs:210-282), host buffers: tile-major -> row-major film */
yk_status yk_film_update_tiles(const yk_tile* tiles, size_t n_tiles, const float* tile_rgb, uint16_t res_x, uint16_t res_y,
                               float* film_rgb);

/* ---- scene ------------------------------------------------------------------ */
/* BoundingVolumeHierarchy::new (bvh.rs:39-115) on the host, then upload.
 * ctx may be NULL: host-only scene (BVH build/export without a GPU).
 * A scene (and a yk_tile_list) is read-only after creation and belongs to the DEVICE of `ctx`:
 * any context on that device may render it, also concurrently from several threads — two
 * contexts with their own streams keep two renders in flight, so the latency tail of one
 * overlaps the bulk of the next.  Destroy it only after every render that uses
 * it has completed (the library does not reference-count scenes). */
yk_status yk_scene_create(yk_context* ctx, const yk_scene_desc* desc, yk_scene** out);
void yk_scene_destroy(yk_scene* scene);
yk_status yk_scene_get_info(const yk_scene* scene, yk_scene_info* out);
/* nodes: n_nodes entries in the reference's depth-first layout; shape_order:
 * n_shapes source indices in leaf order (bvh.rs:96).  Either may be NULL. */
yk_status yk_scene_export_bvh(const yk_scene* scene, yk_bvh_node* nodes, uint32_t* shape_order);

/* ---- the hot path -------------------------------------------------------------- */
/* Integrator::render for a batch of tiles (integrators/mod.rs:120-185, non-accumulating
 * film).  out_rgb: host buffer, tile-major, 3 floats per pixel. */
yk_status yk_render_tiles(yk_context* ctx, const yk_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                          const yk_integrator_desc* integrator, const yk_tile* tiles, size_t n_tiles, float* out_rgb,
                          yk_render_stats* stats, yk_cancel_fn cancel, void* user);
/* Same, radiance left in device memory (d_out_rgb: device pointer, same layout)
 * on `stream` (a hipStream_t, NULL = the context's stream).  Asynchronous unless
 * stats != NULL. */
yk_status yk_render_tiles_device(yk_context* ctx, const yk_scene* scene, const yk_camera* camera,
                                 const yk_sampler_desc* sampler, const yk_integrator_desc* integrator, const yk_tile* tiles,
                                 size_t n_tiles, void* d_out_rgb, void* stream, yk_render_stats* stats, yk_cancel_fn cancel,
                                 void* user);
/* Integrator::render(accumulating = true) (integrators/mod.rs:146-161; the tile queue of
 * render_manager.rs:135-143): ONE sample per pixel whose global sample index is the tile's
 * FilmTile.sample (tile_samples[t], u16 like film.rs:52, which must be below the sampler's
 * samples per pixel as in render_manager.rs:135-143 — YK_ERR_INVALID_ARGUMENT otherwise); the
 * raw radiance is stored (divided by 1).  Fold the result into the film with yk_film_accumulate_tiles[_device]; the displayed
 * image is film / samples (tonemap.rs:240-241). */
yk_status yk_render_tiles_accumulating(yk_context* ctx, const yk_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                                       const yk_integrator_desc* integrator, const yk_tile* tiles, const uint16_t* tile_samples, size_t n_tiles,
                                       float* out_rgb, yk_render_stats* stats, yk_cancel_fn cancel, void* user);
yk_status yk_render_tiles_accumulating_device(yk_context* ctx, const yk_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                                              const yk_integrator_desc* integrator, const yk_tile* tiles, const uint16_t* tile_samples,
                                              size_t n_tiles, void* d_out_rgb, void* stream, yk_render_stats* stats, yk_cancel_fn cancel,
                                              void* user);
/* Film::update_tile with `samples` present (film.rs:260-272): film += tile pixels and
 * tile_sample_counts[t] += 1 (position t in `tiles` plays FilmTile.index; may be NULL). */
yk_status yk_film_accumulate_tiles(const yk_tile* tiles, size_t n_tiles, const float* tile_rgb, uint16_t res_x, uint16_t res_y, float* film_rgb,
                                   uint32_t* tile_sample_counts);
yk_status yk_film_accumulate_tiles_device(yk_context* ctx, const yk_tile* tiles, size_t n_tiles, const void* d_tile_rgb, uint16_t res_x,
                                          uint16_t res_y, void* d_film_rgb, void* stream);
/* A tile list prepared once and reused every frame — what a GPU worker does with the film's
 * tile queue (render_manager.rs:125-143).  The list keeps a device-resident pixel table, so
 * yk_render_tile_list_device (with stats == NULL) and yk_film_update_tile_list_device enqueue
 * their work on `stream` and return without any host synchronisation; results are identical
 * to the yk_tile-array entry points.  tile_samples != NULL makes it an accumulating-film list
 * (FilmTile.sample per tile). */
typedef struct yk_tile_list yk_tile_list;
yk_status yk_tile_list_create(yk_context* ctx, const yk_tile* tiles, const uint16_t* tile_samples, size_t n_tiles, yk_tile_list** out);
void yk_tile_list_destroy(yk_tile_list* list);
yk_status yk_render_tile_list_device(yk_context* ctx, const yk_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                                     const yk_integrator_desc* integrator, const yk_tile_list* list, void* d_out_rgb, void* stream,
                                     yk_render_stats* stats, yk_cancel_fn cancel, void* user);
yk_status yk_film_update_tile_list_device(yk_context* ctx, const yk_tile_list* list, const void* d_tile_rgb, uint16_t res_x, uint16_t res_y,
                                          void* d_film_rgb, void* stream, int accumulate);
/* Several passes of the accumulating film in ONE submission: passes FilmTile.sample,
 * FilmTile.sample + 1, ... + n_passes - 1 of every tile (what render_manager.rs:125-143 does by
 * re-queueing the tiles n_passes times).  A single 1080p pass is 2 M camera rays — every launch
 * of it lasts as long as its longest ray — so an interactive GPU worker renders a handful of
 * passes per submission (8 passes: 2.6x the rays per second of pass-by-pass).
 * out: n_passes x (pixels of the list) x RGB, pass-major; each pass is bit for bit what the
 * one-pass call with that sample index returns.  yk_film_accumulate_tile_list_passes_device
 * adds them to the film pass after pass (film.rs:260-272), so the film equals the one n_passes
 * single submissions would have produced. */
yk_status yk_render_tiles_accumulating_passes(yk_context* ctx, const yk_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                                              const yk_integrator_desc* integrator, const yk_tile* tiles, const uint16_t* tile_samples,
                                              size_t n_tiles, uint32_t n_passes, float* out_rgb, yk_render_stats* stats, yk_cancel_fn cancel,
                                              void* user);
yk_status yk_render_tile_list_passes_device(yk_context* ctx, const yk_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                                            const yk_integrator_desc* integrator, const yk_tile_list* list, uint32_t n_passes, void* d_out_rgb,
                                            void* stream, yk_render_stats* stats, yk_cancel_fn cancel, void* user);
/* The same for a PLAIN tile list (tile_samples == NULL) whose tiles all stand at the same sample: passes first_sample ..
 * first_sample + n_passes - 1 of every tile — the worker's accumulate loop (render_manager.rs:125-143 re-queues all
 * tiles with the next sample index) without a new list per pass. */
yk_status yk_render_tile_list_samples_device(yk_context* ctx, const yk_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                                             const yk_integrator_desc* integrator, const yk_tile_list* list, uint32_t first_sample,
                                             uint32_t n_passes, void* d_out_rgb, void* stream, yk_render_stats* stats, yk_cancel_fn cancel,
                                             void* user);
yk_status yk_film_accumulate_tile_list_passes_device(yk_context* ctx, const yk_tile_list* list, const void* d_passes_rgb, uint32_t n_passes,
                                                     uint16_t res_x, uint16_t res_y, void* d_film_rgb, void* stream);

/* Film output (app/util.rs:90-111 write_exr -> exr::prelude::write_rgb_file): an OpenEXR 2
 * scan-line file with three FLOAT channels B, G, R, uncompressed, increasing Y — readable by
 * the tools the reference targets (readme.md:46-47); and a little-endian PFM ("PF") writer.
 * pixels: row-major RGB, row 0 = top. */
yk_status yk_write_exr(const char* path, uint32_t width, uint32_t height, const float* rgb);
yk_status yk_write_pfm(const char* path, uint32_t width, uint32_t height, const float* rgb);

/* Exactly the trait method: one tile, returns the ray count through *out_rays. */
yk_status yk_render_tile(yk_context* ctx, const yk_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                         const yk_integrator_desc* integrator, const yk_tile* tile, float* tile_pixels, uint64_t* out_rays);
/* Film::update_tile on the device: scatter tile-major radiance into a row-major
 * film (both device pointers).  Used on rank 0 after the RCCL gather. */
yk_status yk_film_update_tiles_device(yk_context* ctx, const yk_tile* tiles, size_t n_tiles, const void* d_tile_rgb,
                                      uint16_t res_x, uint16_t res_y, void* d_film_rgb, void* stream);
/* Integrator::li (integrators/mod.rs:94-101) for n caller-supplied rays; the
 * sampler is started at (pixel, sample_index) and advanced by `dimension`
 * draws already consumed by the caller (2 for a camera ray). */
yk_status yk_li(yk_context* ctx, const yk_scene* scene, const yk_sampler_desc* sampler, const yk_integrator_desc* integrator,
                size_t n, const float* ray_o, const float* ray_d, const uint16_t* pixel_xy, const uint32_t* sample_index,
                uint32_t dimension, float* out_li, uint32_t* out_ray_counts);

/* ---- per-stage entry points (parity tests, profiling) ------------------------- */
/* BoundingVolumeHierarchy::intersect (bvh.rs:160-232) for n host rays.
 * out_shape: source shape index or -1; counters as IntersectionResult. */
yk_status yk_trace_closest(yk_context* ctx, const yk_scene* scene, size_t n, const float* ray_o, const float* ray_d,
                           const float* t_max /* NULL = inf */, int32_t* out_shape, float* out_t, float* out_bary /* 3n */,
                           uint32_t* out_node_tests, uint32_t* out_node_hits, uint32_t* out_shape_tests);
/* BoundingVolumeHierarchy::any_intersect (bvh.rs:235-302) */
yk_status yk_trace_any(yk_context* ctx, const yk_scene* scene, size_t n, const float* ray_o, const float* ray_d,
                       const float* t_max, const int32_t* area_light /* NULL = none */, uint8_t* out_hit);
/* Sampler start_pixel_sample + draws, evaluated on the device */
yk_status yk_sampler_sequence(yk_context* ctx, const yk_sampler_desc* sampler, uint16_t px, uint16_t py, uint32_t sample_index,
                              const uint8_t* dims, size_t n_draws, float* out /* 2 per draw */);
/* Camera::ray for every pixel of a tile at one sample index (camera.rs:105-114) */
yk_status yk_camera_rays(yk_context* ctx, const yk_camera* camera, const yk_sampler_desc* sampler, const yk_tile* tile,
                         uint32_t sample_index, float* out_o, float* out_d);
/* device libm used by the kernels: fn 0 sin, 1 cos, 2 tan, 3 log, 4 acos, 5 atan2(x=y_in,y=x_in),
 * 6 sqrt, 7 a/b, 8 f64-sqrt helper, 9 / 10 f32::min / max; 11..27 work on packed triples (n = 3 x count):
 * 11 Vec3::dot, 12 cross, 13 len, 14 normalized, 15 max_dimension, 16 abs, 17 Normal::dot_v, 18 the kx/ky/kz
 * permutation of Triangle::intersect, 19 / 20 Vec3::min / max, 21 Normal::faceforward_v, 22 a + b, 23 a - b,
 * 24 a * b.x, 25 a / b.x, 26 -a, 27 len_sqr (result in out[3k..3k+2]); scalar again: 28 / 29 the sine / cosine of the
 * shared-reduction pair the shading code calls (the same bits as fn 0 / 1) */
yk_status yk_device_math(yk_context* ctx, int fn, size_t n, const float* a, const float* b, float* out);
/* The HOST instance of the same scalar functions (fn 0..5, 28, 29 as above; 30 expf): what the loaders, the camera, the spot
 * light and the roughness remap evaluate on the CPU.  Needs no device. */
yk_status yk_host_math(int fn, size_t n, const float* a, const float* b, float* out);
/* Bsdf::f and Bsdf::sample_f on the device for n (wo, wi|u) pairs against one material */
yk_status yk_bsdf_eval(yk_context* ctx, const yk_material_desc* material, size_t n, const float* n_geom,
                       const float* n_shading, const float* dpdu, const float* wo, const float* wi, float* out_f);
yk_status yk_bsdf_sample(yk_context* ctx, const yk_material_desc* material, size_t n, const float* n_geom,
                         const float* n_shading, const float* dpdu, const float* wo, const float* u, float* out8);

/* Light::sample_li (lights/mod.rs:29-32; point_light.rs:27-50, spot_light.rs:32-80, distant_light.rs:24-43,
 * rectangular_light.rs:46-71) on the device for n surface points against one light, and the ray of the
 * VisibilityTester it returns (visibility.rs:21-23, interaction.rs:44-59).  out: 18 floats per point —
 * l[3], li[3], pdf, has_vis (0/1), area_light (light_index or -1), p1[3], shadow-ray origin[3], direction[3]. */
yk_status yk_light_sample(yk_context* ctx, const yk_light_desc* light, int32_t light_index, size_t n, const float* p,
                          const float* n_geom, const float* u, float* out18);

size_t yk_sizeof(int what);

/* ---- several GPUs ------------------------------------------------------
 * The reference renders from ONE process: RenderManager spawns its workers
 * (renderer/render_manager.rs:78-97), hands out the film's tiles — "interleave tiles" is its own
 * TODO (render_manager.rs:206-210) — and every finished tile is written back by
 * Film::update_tile (film.rs:210-282).  yk_multi is that for the GPUs of a node:
 *   - one context and one host thread per device (the thread enqueues that device's render);
 *   - the scene's BVH is built once on the host and copied to every device;
 *   - tile i of the film's outward spiral (film.rs:333-376) belongs to device i mod G; every
 *     device renders its tiles into a dense tile-major slab in its own HBM;
 *   - ONE exchange: the slabs move into device 0's memory with RCCL point-to-point calls
 *     (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd — a gather: tiles are disjoint, so
 *     nothing is reduced) issued on the contexts' own streams, i.e. ordered after each render
 *     without host synchronisation; xGMI links are point to point, every slab takes its own;
 *   - device 0 scatters the slabs into the row-major film (Film::update_tile).
 * Results are bit for bit those of a single-device render of the same film.
 * RCCL is loaded on first use (dlopen: a process that already has it — PyTorch — shares it);
 * without it yk_multi_create on more than one device returns YK_ERR_UNSUPPORTED. */
typedef struct yk_multi yk_multi;
typedef struct yk_multi_scene yk_multi_scene;
typedef struct yk_multi_film yk_multi_film;
/* devices: HIP ordinals, devices[0] assembles the film.  n_devices == 1 is a plain one-GPU
 * render through the same code (no communicator unless "rccl_loopback" is set). */
yk_status yk_multi_create(const int* devices, uint32_t n_devices, yk_multi** out);
/* The same with flags:
 *   YK_MULTI_PEER_COPY       the slabs travel by hipMemcpyPeerAsync on device 0's stream (behind an event of the
 *                            sender's stream) instead of RCCL send / recv — no RCCL needed; also option "peer_copy";
 *   YK_MULTI_SHARED_DEVICES  ranks may name the same device (RCCL refuses two ranks on one device, so this implies
 *                            the peer copy): G ranks on ONE GPU run the deal, the per-rank tile lists, the slab
 *                            layout, the exchange ordering and device 0's scatter exactly as G GPUs would — what a
 *                            one-GPU box can test of the G > 1 path (tests/test_multi_gpu.py). */
#define YK_MULTI_SHARED_DEVICES 1u
#define YK_MULTI_PEER_COPY 2u
yk_status yk_multi_create_ex(const int* devices, uint32_t n_devices, uint32_t flags, yk_multi** out);
/* The deal, without a device: the tiles of `rank` among `n_ranks` — spiral tile i of film_tiles(res, tile_dim)
 * (film.rs:333-376, 409-475) goes to rank i mod n_ranks (render_manager.rs:206-210).  Returns the number of tiles
 * of the rank (0 on a bad argument), writes up to `cap` of them to `out` (may be NULL) and the rank's pixel count
 * to *out_pixels (may be NULL): its slab holds 3 x that many floats, tile after tile, rows top to bottom. */
size_t yk_multi_deal(uint16_t res_x, uint16_t res_y, uint16_t tile_dim, uint32_t n_ranks, uint32_t rank, yk_tile* out, size_t cap,
                     uint64_t* out_pixels);
/* Destroy the films and scenes made from it first or afterwards, in any order; a render must not be in flight. */
void yk_multi_destroy(yk_multi* m);
uint32_t yk_multi_device_count(const yk_multi* m);
/* The context of rank r (options, yk_last_error); owned by the yk_multi. */
yk_context* yk_multi_context(yk_multi* m, uint32_t rank);
/* yk_context_set_option on every context; plus "rccl_loopback" (0 | 1): rank 0's own slab also
 * takes the exchange path (to itself) — exercises the collective on a single GPU; "peer_copy" (0 | 1). */
yk_status yk_multi_set_option(yk_multi* m, const char* key, int64_t value);
yk_status yk_multi_last_error(const yk_multi* m, char* buf, size_t cap);
/* BoundingVolumeHierarchy::new once, one copy per device. */
yk_status yk_multi_scene_create(yk_multi* m, const yk_scene_desc* desc, yk_multi_scene** out);
void yk_multi_scene_destroy(yk_multi_scene* scene);
yk_status yk_multi_scene_get_info(const yk_multi_scene* scene, yk_scene_info* out);
/* The film (film.rs:67-113) and its tile queue: film_tiles(res, tile_dim) dealt round-robin,
 * prepared per device (yk_tile_list), slabs, and the row-major RGB film in device 0's memory. */
yk_status yk_multi_film_create(yk_multi* m, uint16_t res_x, uint16_t res_y, uint16_t tile_dim, yk_multi_film** out);
void yk_multi_film_destroy(yk_multi_film* film);
/* device-0 pointer to res_x * res_y RGB float triples, row-major */
void* yk_multi_film_device_ptr(const yk_multi_film* film);
/* Render the whole film.  film_rgb: host buffer (res_x * res_y * 3 floats) or NULL; stats: sums
 * over the devices (seconds: the slowest device) or NULL.  With both NULL the call only
 * enqueues work (renders, exchange, scatter) and returns: yk_multi_sync waits for it, after
 * which yk_multi_film_device_ptr holds the frame.
 * cancel: the user's predicate is never called concurrently and never again after it returned non-zero — the
 * reference's is a consuming FnMut polled by one thread (render_worker.rs:240-249) — although every device's host
 * thread polls: they share a latch, and the first non-zero answer stops ALL devices (YK_ERR_CANCELLED).  With a
 * predicate the per-device renders are synchronous (it is polled about every 100 us while the GPUs work, see
 * yk_cancel_fn); the film then holds nothing defined. */
yk_status yk_multi_render_film(yk_multi* m, const yk_multi_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                               const yk_integrator_desc* integrator, yk_multi_film* film, float* film_rgb, yk_render_stats* stats,
                               yk_cancel_fn cancel, void* user);
/* The accumulating film over all devices (integrators/mod.rs:146-161 accumulating = true + film.rs:260-272): passes
 * first_sample .. first_sample + n_passes - 1 of EVERY tile in one submission, each added to the film on device 0 —
 * bit for bit the film n_passes single-device submissions produce.  yk_multi_film_clear zeroes the film (a new
 * accumulation: FilmSettings.clear), stream-ordered on device 0. */
yk_status yk_multi_accumulate_film(yk_multi* m, const yk_multi_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                                   const yk_integrator_desc* integrator, yk_multi_film* film, uint32_t first_sample, uint32_t n_passes,
                                   float* film_rgb, yk_render_stats* stats, yk_cancel_fn cancel, void* user);
yk_status yk_multi_film_clear(yk_multi* m, yk_multi_film* film);
/* yk_context_interrupt on every rank's context: stops a frame in flight from any thread (the frame's call returns
 * YK_ERR_CANCELLED if it was synchronous); the next frame is unaffected. */
yk_status yk_multi_interrupt(yk_multi* m);
yk_status yk_multi_sync(yk_multi* m);

/* One process per GPU (MPI-style hosts, torch.distributed launchers): the same exchange between
 * processes.  Rank 0 obtains an id (ncclGetUniqueId) and hands it to the other ranks by its own
 * means; every rank then joins with its context (ncclCommInitRank).  yk_dist_gather moves
 * `count` floats from every rank's d_send into rank 0's d_recv[rank * count ...] on `stream`
 * (NULL: the context's stream), without host synchronisation; d_recv is ignored elsewhere.
 * `count` MUST be the same on every rank (the padded maximum of the slab sizes: a rank's send and rank 0's
 * receive are matched by count, unequal values hang the exchange).  The RCCL found at run time must report a 2.x
 * version >= 2.7 (ncclGetVersion), anything else is YK_ERR_UNSUPPORTED. */
#define YK_DIST_ID_BYTES 128
typedef struct yk_dist yk_dist;
yk_status yk_dist_unique_id(uint8_t id[YK_DIST_ID_BYTES]);
yk_status yk_dist_create(yk_context* ctx, const uint8_t id[YK_DIST_ID_BYTES], uint32_t rank, uint32_t world, yk_dist** out);
void yk_dist_destroy(yk_dist* dist);
yk_status yk_dist_gather(yk_dist* dist, const void* d_send, void* d_recv, size_t count, void* stream);

/* ---- many render workers, one device ---------------------------------------------------------------------------
 * The reference renders with num_cpus - 1 worker threads, each calling Integrator::render for ONE tile at a time
 * (render_manager.rs:78-97, render_worker.rs:205-256).  yk_combiner_render_tile is that call for such a thread: it blocks
 * until its tile is rendered, and the calls that are waiting at the same time are merged into one yk_render_tiles
 * submission (the first waiter leads it, the others follow) on one of the combiner's contexts ("lanes": up to
 * n_contexts submissions in flight; all on one device, each context once).  Every caller receives exactly the
 * pixels a single-tile call returns.  max_tiles: most tiles per submission (0 = 64); linger_us: how long a caller
 * that finds itself alone waits for company before it submits.
 *   accumulating_sample < 0: Integrator::render(accumulating = false) — all samples of the pixel, the mean stored;
 *   otherwise accumulating = true with FilmTile.sample = accumulating_sample (one sample, raw value).
 *   Only calls for the same scene, camera, sampler, integrator and mode share a submission.
 *   stats (may be NULL): times are the submission's; rays / shadow_rays / samples are the submission's counts shared
 *   out by tile area with the remainder to the leading call — exact in sum over the callers, which is how the
 *   reference uses them (render_manager.rs:277-281).
 *   cancel: polled by the calling thread itself about every 100 us while it waits (the reference's predicate consumes
 *   a channel message, render_worker.rs:240-249, so it must fire in its own worker).  Fired while the tile is still
 *   queued: the call returns YK_ERR_CANCELLED at once.  Fired while the tile is part of a running submission: that
 *   submission is interrupted; callers whose predicate has fired return YK_ERR_CANCELLED, the other callers' tiles
 *   are queued again — nobody is handed pixels of an interrupted job. */
typedef struct yk_combiner yk_combiner;
typedef struct yk_combiner_info {
    uint64_t submissions, tiles, requeued; /* submissions made, tiles rendered through them, tiles queued again after an interruption */
    uint32_t largest_submission, lanes;
} yk_combiner_info;
yk_status yk_combiner_create(yk_context* const* contexts, uint32_t n_contexts, uint32_t max_tiles, uint32_t linger_us, yk_combiner** out);
void yk_combiner_destroy(yk_combiner* combiner); /* no call may be waiting in it */
yk_status yk_combiner_render_tile(yk_combiner* combiner, const yk_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                                  const yk_integrator_desc* integrator, const yk_tile* tile, int32_t accumulating_sample, float* tile_pixels,
                                  yk_render_stats* stats, yk_cancel_fn cancel, void* user);
yk_status yk_combiner_get_info(const yk_combiner* combiner, yk_combiner_info* out);
yk_status yk_combiner_last_error(const yk_combiner* combiner, char* buf, size_t cap);

/* ---- scene input -------------------------------------------
 * The reference's loaders, host-only (no device needed): they produce the flattened
 * scene description yk_scene_create consumes plus the camera and film settings the
 * reference's `load` functions return.
 *   yk_load_ply   Scene::ply          scene/mod.rs:99-152 + ply::load scene/ply.rs:19-130
 *                 (white matte, fit-to-unit-cube transform, point light, 640x480 camera)
 *   yk_load_pbrt  scene::pbrt::load   scene/pbrt/mod.rs:94-857 — the subset the reference
 *                 implements: perspective Camera, Film resolution, LookAt, Translate/Scale/
 *                 Rotate, Attribute/Transform blocks, Include, (Make)NamedMaterial/Material
 *                 {matte,glass,glossy,metal}, LightSource {infinite,distant,point}, Shape
 *                 {sphere,trianglemesh,plymesh}, Texture "spectrum" "imagemap" for matte Kd.
 * split_method / max_shapes_in_node are SceneLoadSettings (scene/mod.rs:25-39) and are
 * copied into the description.  Errors: where the reference returns LoadError or panics
 * the call returns non-zero and yk_loader_last_error() (thread-local) holds the reason. */
/* ImageTexture::new(path) (textures/image_texture.rs:66-70,114-141): decode an image file
 * into RGB f32 (u8 / 255, u16 / 65535, float as is, no gamma, alpha dropped).  The decoder
 * is chosen from the file extension like image::io::Reader::open: .png, .bmp, .tga,
 * .ppm/.pnm, .qoi, .ff (farbfeld), .exr (scan-line, none/ZIPS/ZIP); the `image` crate's
 * remaining formats (JPEG, GIF, TIFF, WebP, HDR ...) return YK_ERR_UNSUPPORTED; grey files
 * are the reference's "Unsupported image format".  out->rgb is owned by the library:
 * yk_image_texture_free. */
yk_status yk_image_texture_load(const char* path, yk_texture_desc* out);
void yk_image_texture_free(yk_texture_desc* tex);

typedef struct yk_loaded_scene yk_loaded_scene;
yk_status yk_load_ply(const char* path, uint32_t split_method, uint32_t max_shapes_in_node, yk_loaded_scene** out);
yk_status yk_load_pbrt(const char* path, uint32_t split_method, uint32_t max_shapes_in_node, yk_loaded_scene** out);
/* Pointers written into *desc stay valid until yk_loaded_scene_destroy.  camera->res_x/res_y
 * carry FilmSettings.res; *tile_dim its tile_dim (16).  camera / tile_dim may be NULL. */
yk_status yk_loaded_scene_get(const yk_loaded_scene* loaded, yk_scene_desc* desc, yk_camera_params* camera, uint16_t* tile_dim);
void yk_loaded_scene_destroy(yk_loaded_scene* loaded);
const char* yk_loader_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* YUKI_HIP_H */
