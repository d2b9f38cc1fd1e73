// yk_api.cpp — the C ABI (include/yuki_hip.h): context, scene upload, and the
// batch scheduler that drives the wavefront kernels.
//
// The scheduler plays the role of the reference's RenderManager/RenderWorker
// pair (renderer/render_manager.rs:69-193, render_worker.rs:62-137): instead of
// num_cpus-1 threads popping 16x16 tiles, ALL pixels x samples of the submitted
// tiles become one work range that is cut into batches of `batch_paths` camera
// samples; each batch runs raygen + max_depth x (trace, shade, shadow,
// accumulate) without any host synchronisation — queue lengths live in device
// memory and the persistent kernels read them there.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "yk_internal.h"

static double now_seconds() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <class T> static yk_status upload(yk_context* ctx, DevBuf& buf, const T* src, size_t count) {
    size_t bytes = std::max<size_t>(count * sizeof(T), 16);
    HIP_TRY(ctx, buf.ensure(bytes));
    if (count) HIP_TRY(ctx, hipMemcpy(buf.p, src, count * sizeof(T), hipMemcpyHostToDevice));
    return YK_OK;
}

extern "C" {

uint32_t yk_abi_version(void) { return YK_ABI_VERSION; }

const char* yk_status_string(yk_status s) {
    switch (s) {
        case YK_OK: return "ok";
        case YK_ERR_INVALID_ARGUMENT: return "invalid argument";
        case YK_ERR_NO_DEVICE: return "no HIP device";
        case YK_ERR_DEVICE: return "HIP error";
        case YK_ERR_OUT_OF_MEMORY: return "out of device memory";
        case YK_ERR_UNSUPPORTED: return "unsupported on the device path";
        case YK_ERR_BVH_BUILD: return "BVH build failed";
        case YK_ERR_CANCELLED: return "cancelled";
        case YK_ERR_STACK_OVERFLOW: return "traversal stack overflow";
    }
    return "unknown";
}

yk_status yk_context_create(int device, yk_context** out) {
    if (!out) return YK_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return YK_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return YK_ERR_INVALID_ARGUMENT;
    if (hipSetDevice(device) != hipSuccess) return YK_ERR_NO_DEVICE;
    yk_context* ctx = new yk_context();
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
    // Two streams per context (main + side); the second work set's pair is created on first use.
    // HIP multiplexes streams onto few hardware queues (GPU_MAX_HW_QUEUES, default 4) and streams
    // that share a queue serialise, so a context never holds streams it does not run work on.
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ws[0].done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ws[1].done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_out, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->ws[0].side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ws[0].ev_shade, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ws[1].ev_shade, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ws[0].ev_acc, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ws[1].ev_acc, hipEventDisableTiming) != hipSuccess) {
        delete ctx;
        return YK_ERR_DEVICE;
    }
    ctx->ws[0].stream = ctx->stream;
    if (const char* w = std::getenv("YK_WIDE_BVH")) ctx->wide_bvh = std::min(std::max(std::atoi(w), 0), 2);  // experiments; same as set_option("wide_bvh")
    if (const char* w = std::getenv("YK_PACKET_BOUNCES")) ctx->packet_bounces = std::max(std::atoi(w), 0);
    if (const char* w = std::getenv("YK_PACKET_SHADOW_BOUNCES")) ctx->packet_shadow_bounces = std::max(std::atoi(w), 0);
    if (const char* w = std::getenv("YK_TOP_NODES")) ctx->top_nodes = std::min(std::max(std::atoi(w), 0), YK_TOP_MAX);
    *out = ctx;
    return YK_OK;
}

void yk_context_destroy(yk_context* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->ws[1].stream) (void)hipStreamSynchronize(ctx->ws[1].stream);
    for (WorkSet& w : ctx->ws) {
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 4; ++b) w.path[a][b].release();
        DevBuf* wb[] = {&w.hit, &w.pend, &w.shO, &w.shD, &w.shC, &w.vis, &w.shq, &w.shO2, &w.shD2, &w.shq2, &w.ctrl, &w.spill, &w.spill_side};
        for (DevBuf* b : wb) b->release();
        if (w.done) (void)hipEventDestroy(w.done);
        if (w.ev_shade) (void)hipEventDestroy(w.ev_shade);
        if (w.ev_acc) (void)hipEventDestroy(w.ev_acc);
        if (w.side) {
            (void)hipStreamSynchronize(w.side);
            (void)hipStreamDestroy(w.side);
        }
    }
    if (ctx->ws[1].stream) (void)hipStreamDestroy(ctx->ws[1].stream);
    if (ctx->ev_in) (void)hipEventDestroy(ctx->ev_in);
    if (ctx->ev_out) (void)hipEventDestroy(ctx->ev_out);
    DevBuf* all[] = {&ctx->sample_buf, &ctx->pixel_xy, &ctx->pixel_aux, &ctx->tiles, &ctx->tile_off, &ctx->counters, &ctx->stats4, &ctx->hit4};
    for (DevBuf* b : all) b->release();
    for (DevBuf& b : ctx->scratch) b.release();
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

yk_status yk_last_error(const yk_context* ctx, char* buf, size_t cap) {
    if (!ctx || !buf || cap == 0) return YK_ERR_INVALID_ARGUMENT;
    std::snprintf(buf, cap, "%s", ctx->last_error.c_str());
    return YK_OK;
}

void* yk_context_stream(const yk_context* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

yk_status yk_context_set_option(yk_context* ctx, const char* key, int64_t value) {
    if (!ctx || !key) return YK_ERR_INVALID_ARGUMENT;
    YK_LOCK(ctx);
    std::string k(key);
    if (k == "batch_paths") {
        if (value < 64 || value > ((int64_t)1 << 29)) return YK_ERR_INVALID_ARGUMENT;  // 2^29: a path's slot in its batch shares a word with three flag bits
        ctx->batch_paths = value;
    } else if (k == "sample_buf_cap") {
        if (value < (1 << 20)) return YK_ERR_INVALID_ARGUMENT;
        ctx->sample_buf_cap = value;
    } else if (k == "streams") {
        if (value < 1 || value > 2) return YK_ERR_INVALID_ARGUMENT;
        ctx->streams = value;
    } else if (k == "packet_bounces") {
        if (value < 0) return YK_ERR_INVALID_ARGUMENT;
        ctx->packet_bounces = value;
    } else if (k == "packet_shadow_bounces") {
        if (value < 0) return YK_ERR_INVALID_ARGUMENT;
        ctx->packet_shadow_bounces = value;
    } else if (k == "shade_reorder") {
        ctx->shade_reorder = value != 0;
    } else if (k == "overlap_shadow") {
        ctx->overlap_shadow = value != 0;
    } else if (k == "top_nodes") {
        if (value < 0 || value > YK_TOP_MAX) return YK_ERR_INVALID_ARGUMENT;
        ctx->top_nodes = value;
    } else if (k == "wide_bvh") {
        if (value < 0 || value > 2) return YK_ERR_INVALID_ARGUMENT;
        ctx->wide_bvh = value;
    } else if (k == "time_kernels") {
        ctx->time_kernels = value;
    } else {
        return YK_ERR_INVALID_ARGUMENT;
    }
    return YK_OK;
}

// ------------------------------------------------------------------ host helpers
yk_status yk_camera_init(const yk_camera_params* params, yk_camera* out) { return camera_init(params, out); }

size_t yk_film_tiles(uint16_t res_x, uint16_t res_y, uint16_t tile_dim, yk_tile* out, size_t cap) try {
    std::vector<yk_tile> t = film_tiles(res_x, res_y, tile_dim);
    if (out)
        for (size_t i = 0; i < t.size() && i < cap; ++i) out[i] = t[i];
    return t.size();
} catch (const std::exception&) {
    return 0;
}

yk_status yk_make_rect_light(const float l2w[16], const float l2w_inv[16], const float radiance[3], const float size[2], yk_light_desc* out) {
    if (!l2w || !l2w_inv || !radiance || !size || !out) return YK_ERR_INVALID_ARGUMENT;
    std::memset(out, 0, sizeof(*out));
    Xf light_to_world = xf_from(l2w, l2w_inv);
    Xf sample_to_light = xf_mul(xf_scale(size[0], 1.0f, size[1]), xf_translation(-0.5f, 0.0f, -0.5f));
    Xf sample_to_world = xf_mul(light_to_world, sample_to_light);
    out->kind = YK_LIGHT_RECT;
    for (int k = 0; k < 3; ++k) out->i[k] = radiance[k];
    std::memcpy(out->sample_to_world, sample_to_world.m, 64);
    std::memcpy(out->sample_to_world_inv, sample_to_world.mi, 64);
    out->area = size[0] * size[1];
    return YK_OK;
}

yk_status yk_make_spot_light(const float l2w[16], const float l2w_inv[16], const float intensity[3], float total_width_degrees,
                             float falloff_start_degrees, yk_light_desc* out) {
    if (!l2w || !l2w_inv || !intensity || !out) return YK_ERR_INVALID_ARGUMENT;
    std::memset(out, 0, sizeof(*out));
    V3 p = xf_point(l2w, V3{0.0f, 0.0f, 0.0f});
    out->kind = YK_LIGHT_SPOT;
    out->p[0] = p.x;
    out->p[1] = p.y;
    out->p[2] = p.z;
    for (int k = 0; k < 3; ++k) out->i[k] = intensity[k];
    out->cos_total_width = det_cosf(total_width_degrees * (YK_PI / 180.0f));
    out->cos_falloff_start = det_cosf(falloff_start_degrees * (YK_PI / 180.0f));
    std::memcpy(out->world_to_light, l2w_inv, 64);
    return YK_OK;
}

yk_status yk_make_point_light(const float l2w[16], const float intensity[3], yk_light_desc* out) {
    if (!l2w || !intensity || !out) return YK_ERR_INVALID_ARGUMENT;
    std::memset(out, 0, sizeof(*out));
    V3 p = xf_point(l2w, V3{0.0f, 0.0f, 0.0f});
    out->kind = YK_LIGHT_POINT;
    out->p[0] = p.x;
    out->p[1] = p.y;
    out->p[2] = p.z;
    for (int k = 0; k < 3; ++k) out->i[k] = intensity[k];
    return YK_OK;
}

yk_status yk_film_update_tiles(const yk_tile* tiles, size_t n_tiles, const float* tile_rgb, uint16_t res_x, uint16_t res_y, float* film_rgb) {
    if (!tiles || !tile_rgb || !film_rgb) return YK_ERR_INVALID_ARGUMENT;
    size_t off = 0;
    for (size_t t = 0; t < n_tiles; ++t) {
        const yk_tile& tl = tiles[t];
        if (tl.x1 > res_x || tl.y1 > res_y || tl.x0 >= tl.x1 || tl.y0 >= tl.y1) return YK_ERR_INVALID_ARGUMENT;  // film.rs:227-234
        size_t w = (size_t)tl.x1 - tl.x0;
        for (size_t y = tl.y0; y < tl.y1; ++y) {
            std::memcpy(film_rgb + 3 * (y * res_x + tl.x0), tile_rgb + 3 * off, 3 * w * sizeof(float));
            off += w;
        }
    }
    return YK_OK;
}

// Film::update_tile with accumulation on (film.rs:260-272): film += tile ; samples[tile] += 1
yk_status yk_film_accumulate_tiles(const yk_tile* tiles, size_t n_tiles, const float* tile_rgb, uint16_t res_x, uint16_t res_y, float* film_rgb,
                                   uint32_t* tile_sample_counts) {
    if (!tiles || !tile_rgb || !film_rgb) return YK_ERR_INVALID_ARGUMENT;
    size_t off = 0;
    for (size_t t = 0; t < n_tiles; ++t) {
        const yk_tile& tl = tiles[t];
        if (tl.x1 > res_x || tl.y1 > res_y || tl.x0 >= tl.x1 || tl.y0 >= tl.y1) return YK_ERR_INVALID_ARGUMENT;
        size_t w = (size_t)tl.x1 - tl.x0;
        for (size_t y = tl.y0; y < tl.y1; ++y) {
            float* dst = film_rgb + 3 * (y * res_x + tl.x0);
            const float* src = tile_rgb + 3 * off;
            for (size_t k = 0; k < 3 * w; ++k) dst[k] += src[k];
            off += w;
        }
        if (tile_sample_counts) tile_sample_counts[t] += 1;
    }
    return YK_OK;
}

// ------------------------------------------------------------------ scene
static Material make_material(const yk_material_desc& m) {
    Material r;
    std::memset(&r, 0, sizeof(r));
    for (int k = 0; k < 3; ++k) {
        r.a[k] = m.a[k];
        r.b[k] = m.b[k];
    }
    const bool remap = (m.flags & YK_MAT_FLAG_REMAP) != 0;
    const bool textured = m.kind == YK_MAT_MATTE && (m.flags & YK_MAT_FLAG_TEXTURED_A) != 0;
    r.tex = textured ? m.a_texture + 1u : 0u;
    switch (m.kind) {
        case YK_MAT_MATTE: {  // matte.rs:27-39 (a textured Kd is tested for black per hit)
            if (!textured && m.a[0] == 0.0f && m.a[1] == 0.0f && m.a[2] == 0.0f) {
                r.kind = MK_BLACK;
            } else if (m.c == 0.0f) {
                r.kind = MK_LAMBERT;
            } else {  // oren_nayar.rs:20-27
                r.kind = MK_OREN_NAYAR;
                float sigma2 = m.c * m.c;
                r.c = 1.0f - (sigma2 / (2.0f * (sigma2 + 0.33f)));
                r.d = 0.45f * sigma2 / (sigma2 + 0.09f);
            }
            break;
        }
        case YK_MAT_GLASS:
            r.kind = MK_GLASS;
            r.c = m.c;
            break;
        case YK_MAT_METAL: {  // metal.rs:39-50, trowbridge_reitz.rs:15-20
            r.kind = MK_METAL;
            float roughness = remap ? roughness_to_alpha(m.c) : m.c;
            r.c = rmax(roughness, 0.001f);
            break;
        }
        default: {  // glossy.rs:37-49
            r.kind = MK_GLOSSY;
            float roughness = remap ? roughness_to_alpha(m.c) : m.c;
            r.c = rmax(roughness * roughness, 0.001f);
            break;
        }
    }
    return r;
}

static DevLight make_light(const yk_light_desc& l) {
    DevLight d;
    std::memset(&d, 0, sizeof(d));
    d.kind = l.kind;
    for (int k = 0; k < 3; ++k) {
        d.p[k] = l.p[k];
        d.i[k] = l.i[k];
    }
    d.cos_total_width = l.cos_total_width;
    d.cos_falloff_start = l.cos_falloff_start;
    std::memcpy(d.w2l, l.world_to_light, 64);
    std::memcpy(d.s2w, l.sample_to_world, 64);
    V3 n = xf_normal(l.sample_to_world_inv, V3{0.0f, -1.0f, 0.0f});  // rectangular_light.rs:48
    d.n[0] = n.x;
    d.n[1] = n.y;
    d.n[2] = n.z;
    d.area = l.area;
    return d;
}

}  // extern "C"

// Everything yk_scene_create derives from a scene description on the host (yk_internal.h).
struct SceneImage {
    std::shared_ptr<const HostBvh> bvh;
    const yk_scene_desc* d = nullptr;  // BORROWED: the caller's arrays (indices, points, normals, uvs, tri_material) are uploaded straight
                                       // from the description, so an image is only valid inside the call that built it
    uint32_t n_triangles = 0, n_spheres = 0, n_lights = 0, n_delta_lights = 0;
    yk_scene_info info;  // host part: node counts, bounds, build time
    bool has_device_records = false, wide = false, wide_auto = false;
    uint32_t root_ref = 0;
    std::vector<DevNode> dn, top, top_any;
    std::vector<DevNode4> dn4;
    std::vector<float4> tris, texels, prim_attr;
    std::vector<uint4> prim_shade, tex_info;
    std::vector<uint32_t> mesh_flags, tri_mesh;
    std::vector<int32_t> tri_al;
    std::vector<Material> mats;
    std::vector<DevSphere> spheres;
    std::vector<DevLight> lights;
};

// Host half of yk_scene_create: validation, BoundingVolumeHierarchy::new (bvh.rs:39-115) and — when `ctx` is given (its
// "top_nodes" / "wide_bvh" options apply) — the device records laid out from the tree.
yk_status yk_build_scene_image(yk_context* ctx, const yk_scene_desc* d, std::shared_ptr<SceneImage>& out) try {
    if (!d) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null scene description");
    out.reset();
    if ((uint64_t)d->n_triangles + d->n_spheres == 0) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "empty scene");
    if (d->n_triangles && (!d->points || !d->indices)) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "missing geometry arrays");
    if (d->max_shapes_in_node == 0 || d->max_shapes_in_node > 65535u || d->split_method > 2) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "bad BVH settings");
    for (uint32_t i = 0; i < d->n_triangles; ++i) {
        for (int k = 0; k < 3; ++k)
            if (d->indices[3 * i + k] >= d->n_vertices) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "vertex index out of range");
        if (d->tri_mesh && d->tri_mesh[i] >= d->n_meshes) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "mesh index out of range");
        if (d->tri_material && (d->tri_material[i] < 0 || (uint32_t)d->tri_material[i] >= d->n_materials))
            return fail(ctx, YK_ERR_INVALID_ARGUMENT, "material index out of range");
        if (d->tri_area_light && d->tri_area_light[i] >= (int32_t)d->n_lights) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "light index out of range");
    }
    if ((d->n_spheres && !d->spheres) || (d->n_materials && !d->materials) || (d->n_lights && !d->lights) || (d->n_meshes && !d->meshes))
        return fail(ctx, YK_ERR_INVALID_ARGUMENT, "a count is non-zero but its array is NULL");
    if (d->tri_area_light)  // Triangle.area_light is Option<Arc<RectangularLight>> (triangle.rs:22): -1 or a rectangular light
        for (uint32_t i = 0; i < d->n_triangles; ++i) {
            const int32_t al = d->tri_area_light[i];
            if (al < -1 || (al >= 0 && d->lights[al].kind != YK_LIGHT_RECT)) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "tri_area_light must be -1 or index a rectangular light");
        }
    for (uint32_t k = 0; k < d->n_spheres; ++k)
        if (d->spheres[k].material < 0 || (uint32_t)d->spheres[k].material >= d->n_materials) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "sphere material out of range");
    for (uint32_t m = 0; m < d->n_materials; ++m)
        if ((d->materials[m].flags & YK_MAT_FLAG_TEXTURED_A) && d->materials[m].kind == YK_MAT_MATTE && d->materials[m].a_texture >= d->n_textures)
            return fail(ctx, YK_ERR_INVALID_ARGUMENT, "material texture index out of range");
    if (d->n_materials >= (1u << 26)) return fail(ctx, YK_ERR_UNSUPPORTED, "more than 2^26 materials");
    for (uint32_t t = 0; t < d->n_textures; ++t)
        if (!d->textures || !d->textures[t].rgb || d->textures[t].width == 0 || d->textures[t].height == 0 || d->textures[t].width >= (1u << 24) ||
            d->textures[t].height >= (1u << 24))
            return fail(ctx, YK_ERR_INVALID_ARGUMENT, "bad texture");
    if (d->n_triangles && (!d->tri_material || d->n_materials == 0 || d->n_meshes == 0))
        return fail(ctx, YK_ERR_INVALID_ARGUMENT, "triangles need materials and meshes");
    for (uint32_t m = 0; m < d->n_meshes; ++m) {
        if (d->meshes[m].has_normals && !d->normals) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "mesh has_normals without a normals array");
        if (d->meshes[m].has_uvs && !d->uvs) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "mesh has_uvs without a uvs array");
    }

    std::shared_ptr<SceneImage> img = std::make_shared<SceneImage>();
    SceneImage* s = img.get();
    std::shared_ptr<HostBvh> bvh = std::make_shared<HostBvh>();
    s->bvh = bvh;
    s->d = d;
    s->n_triangles = d->n_triangles;
    s->n_spheres = d->n_spheres;
    s->n_lights = d->n_lights;
    for (uint32_t l = 0; l < d->n_lights; ++l) s->n_delta_lights += d->lights[l].kind != YK_LIGHT_RECT ? 1u : 0u;
    std::memset(&s->info, 0, sizeof(s->info));

    // world bounds of every shape: Triangle::world_bound (triangle.rs:229-235),
    // Sphere::world_bound (sphere.rs:121-123)
    std::vector<ShapeBounds> sb((size_t)d->n_triangles + d->n_spheres);
    for (uint32_t i = 0; i < d->n_triangles; ++i) {
        const float* p0 = d->points + 3 * (size_t)d->indices[3 * i];
        const float* p1 = d->points + 3 * (size_t)d->indices[3 * i + 1];
        const float* p2 = d->points + 3 * (size_t)d->indices[3 * i + 2];
        for (int k = 0; k < 3; ++k) {
            sb[i].bmin[k] = rmin(rmin(p0[k], p1[k]), p2[k]);
            sb[i].bmax[k] = rmax(rmax(p0[k], p1[k]), p2[k]);
        }
    }
    for (uint32_t i = 0; i < d->n_spheres; ++i) {
        const yk_sphere_desc& sp = d->spheres[i];
        const float r = sp.radius;
        const float lo[3] = {-r, -r, -r}, hi[3] = {r, r, r};
        const float big = 3.40282347e+38f;
        ShapeBounds b = {{big, big, big}, {-big, -big, -big}};
        const int corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}};  // transform.rs:194-206
        for (int c = 0; c < 8; ++c) {
            V3 q = xf_point(sp.object_to_world, V3{corner[c][0] ? hi[0] : lo[0], corner[c][1] ? hi[1] : lo[1], corner[c][2] ? hi[2] : lo[2]});
            const float qq[3] = {q.x, q.y, q.z};
            for (int k = 0; k < 3; ++k) {
                b.bmin[k] = rmin(b.bmin[k], qq[k]);
                b.bmax[k] = rmax(b.bmax[k], qq[k]);
            }
        }
        sb[(size_t)d->n_triangles + i] = b;
    }
    if (d->shape_order) {  // the caller's Scene.shapes order (a permutation of all shapes)
        std::vector<uint8_t> seen(sb.size(), 0);
        std::vector<ShapeBounds> ordered(sb.size());
        for (size_t i = 0; i < sb.size(); ++i) {
            const uint32_t src = d->shape_order[i];
            if (src >= sb.size() || seen[src]) {
                return fail(ctx, YK_ERR_INVALID_ARGUMENT, "shape_order is not a permutation of the shapes");
            }
            seen[src] = 1;
            ordered[i] = sb[src];
        }
        sb.swap(ordered);
    }
    double t0 = now_seconds();
    build_bvh(sb, d->max_shapes_in_node, d->split_method, *bvh);
    s->info.build_seconds = now_seconds() - t0;
    if (d->shape_order)  // leaf order -> position in Scene.shapes -> source shape
        for (uint32_t& o : bvh->shape_order) o = d->shape_order[o];
    if (bvh->split_failed || bvh->nodes.empty()) {
        return fail(ctx, YK_ERR_BVH_BUILD, "BVH split failed (reference: assert_ne!(mid, start))");
    }
    s->info.n_nodes = bvh->nodes.size();
    s->info.n_shapes = bvh->shape_order.size();
    s->info.max_leaf_shapes = bvh->max_leaf_shapes;
    s->info.tree_depth = bvh->depth;
    for (int k = 0; k < 3; ++k) {
        s->info.bounds_min[k] = bvh->nodes[0].bmin[k];
        s->info.bounds_max[k] = bvh->nodes[0].bmax[k];
    }
    uint64_t n_interior = 0;
    for (const yk_bvh_node& n : bvh->nodes) n_interior += n.is_leaf ? 0 : 1;
    s->info.n_interior = n_interior;

    if (ctx) {  // device records (a host-only scene — ctx == NULL — stops at the tree)
        const std::vector<yk_bvh_node>& nodes = bvh->nodes;
        // interior index of each reference node = number of interior nodes before it
        std::vector<uint32_t> interior_index(nodes.size());
        uint32_t cnt = 0;
        for (size_t i = 0; i < nodes.size(); ++i) {
            interior_index[i] = cnt;
            if (!nodes[i].is_leaf) ++cnt;
        }
        if (nodes.size() > YK_REF_INDEX_MAX || bvh->shape_order.size() > YK_REF_INDEX_MAX) {
            return fail(ctx, YK_ERR_UNSUPPORTED, "more than 2^28 BVH nodes or shapes");
        }
        auto ref_of = [&](uint32_t idx) -> uint32_t { return nodes[idx].is_leaf ? (YK_LEAF_BIT | nodes[idx].a) : interior_index[idx]; };
        std::vector<DevNode>& dn = s->dn;
        dn.assign(std::max<size_t>(n_interior, 1), DevNode());
        for (size_t i = 0; i < nodes.size(); ++i) {
            if (nodes[i].is_leaf) continue;
            const yk_bvh_node& c0 = nodes[i + 1];
            const yk_bvh_node& c1 = nodes[nodes[i].a];
            DevNode& o = dn[interior_index[i]];
            o.q0 = make_float4(c0.bmin[0], c0.bmin[1], c0.bmin[2], c0.bmax[0]);
            o.q1 = make_float4(c0.bmax[1], c0.bmax[2], c1.bmin[0], c1.bmin[1]);
            o.q2 = make_float4(c1.bmin[2], c1.bmax[0], c1.bmax[1], c1.bmax[2]);
            o.q3 = make_uint4(ref_of((uint32_t)i + 1), ref_of(nodes[i].a) | ((uint32_t)nodes[i].axis << YK_AXIS_SHIFT), 0u, 0u);
        }
        // top of the tree, breadth first, for the LDS-resident copies (YK_TOP_BIT refs).  Two sets: the closest-hit kernels
        // keep 8-byte stack entries (ref, entry distance) in LDS and have room for trace_top_nodes() nodes beside them; the
        // any-hit kernel's entries are a bare ref (4 bytes), which leaves room for trace_top_nodes_any() — more than twice as many.
        auto build_top = [&](size_t cap, std::vector<DevNode>& top) {
            top.clear();
            if (nodes[0].is_leaf || cap == 0) return;
            std::vector<uint32_t> order;  // reference node indices, breadth first
            std::vector<uint32_t> top_id(nodes.size(), 0xffffffffu);
            order.push_back(0);
            top_id[0] = 0;
            for (size_t q = 0; q < order.size() && order.size() < cap; ++q) {
                const uint32_t P = order[q];
                for (uint32_t c : {P + 1, nodes[P].a}) {
                    if (!nodes[c].is_leaf && order.size() < cap) {
                        top_id[c] = (uint32_t)order.size();
                        order.push_back(c);
                    }
                }
            }
            for (uint32_t P : order) {
                DevNode t = dn[interior_index[P]];
                const uint32_t c0 = P + 1, c1 = nodes[P].a;
                if (top_id[c0] != 0xffffffffu) t.q3.x = YK_TOP_BIT | top_id[c0];
                if (top_id[c1] != 0xffffffffu) t.q3.y = YK_TOP_BIT | top_id[c1] | ((uint32_t)nodes[P].axis << YK_AXIS_SHIFT);
                top.push_back(t);
            }
        };
        build_top((size_t)std::min<int64_t>(ctx->top_nodes, trace_top_nodes()), s->top);
        build_top((size_t)std::min<int64_t>(ctx->top_nodes, trace_top_nodes_any()), s->top_any);
        // 4-wide collapse (DevNode4): one node per reference interior node reached at even depth
        // below the root.  Built only while the traversal stack of the collapsed tree is
        // guaranteed to fit (the reference asserts on its own stack depth, bvh.rs:172-174).
        std::vector<DevNode4>& dn4 = s->dn4;
        const bool wide = s->wide = ctx->wide_bvh != 0 && !nodes[0].is_leaf && bvh->depth <= 64;
        if (wide) {
            dn4.reserve(n_interior / 2 + 1);
            struct Todo {
                uint32_t binary;  // reference node index of P
                uint32_t slot;    // DevNode4 index to fill
            };
            std::vector<Todo> stack;
            dn4.emplace_back();
            stack.push_back(Todo{0u, 0u});
            while (!stack.empty()) {
                const Todo td = stack.back();
                stack.pop_back();
                const uint32_t P = td.binary, A = P + 1, B = nodes[P].a;
                uint32_t child[4] = {YK_REF_NONE, YK_REF_NONE, YK_REF_NONE, YK_REF_NONE};  // reference node index per slot
                if (nodes[A].is_leaf) {
                    child[0] = A;
                } else {
                    child[0] = A + 1;
                    child[1] = nodes[A].a;
                }
                if (nodes[B].is_leaf) {
                    child[2] = B;
                } else {
                    child[2] = B + 1;
                    child[3] = nodes[B].a;
                }
                float box[4][6] = {};
                uint32_t ref[4];
                for (int k = 0; k < 4; ++k) {
                    ref[k] = YK_REF_NONE;
                    if (child[k] == YK_REF_NONE) continue;
                    const yk_bvh_node& c = nodes[child[k]];
                    for (int a = 0; a < 3; ++a) {
                        box[k][a] = c.bmin[a];
                        box[k][3 + a] = c.bmax[a];
                    }
                    if (c.is_leaf) {
                        ref[k] = YK_LEAF_BIT | c.a;
                    } else {
                        ref[k] = (uint32_t)dn4.size();
                        dn4.emplace_back();
                    }
                }
                // children are expanded so that the first visited subtree (for a positive ray) follows in memory
                for (int k = 3; k >= 0; --k)
                    if (ref[k] != YK_REF_NONE && !(ref[k] & YK_LEAF_BIT)) stack.push_back(Todo{child[k], ref[k]});
                DevNode4& o = dn4[td.slot];
                o.q0 = make_float4(box[0][0], box[0][1], box[0][2], box[0][3]);
                o.q1 = make_float4(box[0][4], box[0][5], box[1][0], box[1][1]);
                o.q2 = make_float4(box[1][2], box[1][3], box[1][4], box[1][5]);
                o.q3 = make_float4(box[2][0], box[2][1], box[2][2], box[2][3]);
                o.q4 = make_float4(box[2][4], box[2][5], box[3][0], box[3][1]);
                o.q5 = make_float4(box[3][2], box[3][3], box[3][4], box[3][5]);
                o.q6 = make_uint4(ref[0], ref[1], ref[2], ref[3]);
                const uint32_t axA = nodes[A].is_leaf ? 0u : nodes[A].axis, axB = nodes[B].is_leaf ? 0u : nodes[B].axis;
                o.q7 = make_uint4((uint32_t)nodes[P].axis | (axA << 2) | (axB << 4), 0u, 0u, 0u);
            }
        }
        const size_t np = bvh->shape_order.size();
        std::vector<float4>& tris = s->tris;
        tris.assign(3 * np, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        std::vector<uint4>& prim_shade = s->prim_shade;
        prim_shade.assign(np, make_uint4(0u, 0u, 0u, 0u));
        std::vector<uint32_t> mat_kind(std::max<uint32_t>(d->n_materials, 1), 0u);  // device BSDF kind (MK_*) per material
        for (uint32_t m = 0; m < d->n_materials; ++m) mat_kind[m] = make_material(d->materials[m]).kind & 7u;
        std::vector<uint8_t> last(np, 0);
        for (const yk_bvh_node& n : nodes)
            if (n.is_leaf) last[(size_t)n.a + n.count - 1] = 1;
        for (size_t p = 0; p < np; ++p) {
            uint32_t src = bvh->shape_order[p];
            if (src >= d->n_triangles) {  // sphere: only the source index and the flags are read
                uint32_t none = 0xffffffffu, fl = (last[p] ? YK_PRIM_LAST : 0u) | YK_PRIM_SPHERE | (mat_kind[d->spheres[src - d->n_triangles].material] << YK_PRIM_KIND_SHIFT);
                float w0, w1, w2;
                std::memcpy(&w0, &none, 4);
                std::memcpy(&w1, &src, 4);
                std::memcpy(&w2, &fl, 4);
                tris[3 * p + 0] = make_float4(0.0f, 0.0f, 0.0f, w0);
                tris[3 * p + 1] = make_float4(0.0f, 0.0f, 0.0f, w1);
                tris[3 * p + 2] = make_float4(0.0f, 0.0f, 0.0f, w2);
                prim_shade[p] = make_uint4(0u, 0u, 0u, ((uint32_t)d->spheres[src - d->n_triangles].material << 6) | (mat_kind[d->spheres[src - d->n_triangles].material] << 3));
                continue;
            }
            const float* p0 = d->points + 3 * (size_t)d->indices[3 * src];
            const float* p1 = d->points + 3 * (size_t)d->indices[3 * src + 1];
            const float* p2 = d->points + 3 * (size_t)d->indices[3 * src + 2];
            int al = d->tri_area_light ? d->tri_area_light[src] : -1;
            uint32_t alb = (uint32_t)al, lastb = (last[p] ? YK_PRIM_LAST : 0u) | (mat_kind[d->tri_material[src]] << YK_PRIM_KIND_SHIFT);
            float w0, w1, w2;
            std::memcpy(&w0, &alb, 4);
            std::memcpy(&w1, &src, 4);
            std::memcpy(&w2, &lastb, 4);
            tris[3 * p + 0] = make_float4(p0[0], p0[1], p0[2], w0);
            tris[3 * p + 1] = make_float4(p1[0], p1[1], p1[2], w1);
            tris[3 * p + 2] = make_float4(p2[0], p2[1], p2[2], w2);
            const yk_mesh_desc& md = d->meshes[d->tri_mesh ? d->tri_mesh[src] : 0];
            const uint32_t mfl = (md.has_normals ? YK_MESH_NORMALS : 0u) | (md.has_uvs ? YK_MESH_UVS : 0u) | (md.swaps_handedness ? YK_MESH_SWAPS : 0u);
            prim_shade[p] = make_uint4(d->indices[3 * src], d->indices[3 * src + 1], d->indices[3 * src + 2],
                                       ((uint32_t)d->tri_material[src] << 6) | (mat_kind[d->tri_material[src]] << 3) | mfl);
        }
        if (d->normals || d->uvs) {  // leaf-order copy of the per-vertex normals / uvs (yk_device.h: DevScene::prim_attr)
            s->prim_attr.assign(4 * np, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
            for (size_t p = 0; p < np; ++p) {
                const uint32_t src = bvh->shape_order[p];
                if (src >= d->n_triangles) continue;
                const yk_mesh_desc& md = d->meshes[d->tri_mesh ? d->tri_mesh[src] : 0];
                float nrm[3][3] = {}, uv[3][2] = {};
                for (int k = 0; k < 3; ++k) {
                    const size_t vi = d->indices[3 * (size_t)src + k];
                    if (md.has_normals)
                        for (int c = 0; c < 3; ++c) nrm[k][c] = d->normals[3 * vi + c];
                    if (md.has_uvs)
                        for (int c = 0; c < 2; ++c) uv[k][c] = d->uvs[2 * vi + c];
                }
                s->prim_attr[4 * p + 0] = make_float4(nrm[0][0], nrm[0][1], nrm[0][2], uv[0][0]);
                s->prim_attr[4 * p + 1] = make_float4(nrm[1][0], nrm[1][1], nrm[1][2], uv[0][1]);
                s->prim_attr[4 * p + 2] = make_float4(nrm[2][0], nrm[2][1], nrm[2][2], uv[1][0]);
                s->prim_attr[4 * p + 3] = make_float4(uv[1][1], uv[2][0], uv[2][1], 0.0f);
            }
        }
        std::vector<uint32_t>& mesh_flags = s->mesh_flags;
        mesh_flags.assign(std::max<uint32_t>(d->n_meshes, 1), 0);
        for (uint32_t m = 0; m < d->n_meshes; ++m)
            mesh_flags[m] = (d->meshes[m].has_normals ? YK_MESH_NORMALS : 0u) | (d->meshes[m].has_uvs ? YK_MESH_UVS : 0u) |
                            (d->meshes[m].swaps_handedness ? YK_MESH_SWAPS : 0u);
        std::vector<Material>& mats = s->mats;
        mats.resize(std::max<uint32_t>(d->n_materials, 1));
        for (uint32_t m = 0; m < d->n_materials; ++m) mats[m] = make_material(d->materials[m]);
        std::vector<DevSphere>& spheres = s->spheres;
        spheres.resize(std::max<uint32_t>(d->n_spheres, 1));
        for (uint32_t k = 0; k < d->n_spheres; ++k) {
            DevSphere& o = spheres[k];
            std::memcpy(o.o2w, d->spheres[k].object_to_world, 64);
            std::memcpy(o.w2o, d->spheres[k].world_to_object, 64);
            o.radius = d->spheres[k].radius;
            o.material = d->spheres[k].material;
            const float* m = o.o2w;  // Transform::swaps_handedness, transform.rs:85-91
            float det = m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) + m[2] * (m[4] * m[9] - m[5] * m[8]);
            o.swaps_handedness = det < 0.0f ? 1u : 0u;
            o.pad = 0;
        }
        std::vector<DevLight>& lights = s->lights;
        lights.resize(std::max<uint32_t>(d->n_lights, 1));
        for (uint32_t l = 0; l < d->n_lights; ++l) lights[l] = make_light(d->lights[l]);
        std::vector<uint32_t>& tri_mesh = s->tri_mesh;
        tri_mesh.assign(d->n_triangles, 0);
        if (d->tri_mesh) std::memcpy(tri_mesh.data(), d->tri_mesh, sizeof(uint32_t) * d->n_triangles);
        std::vector<int32_t>& tri_al = s->tri_al;
        tri_al.assign(d->n_triangles, -1);
        if (d->tri_area_light) std::memcpy(tri_al.data(), d->tri_area_light, sizeof(int32_t) * d->n_triangles);

        for (uint32_t t = 0; t < d->n_textures; ++t) {
            const yk_texture_desc& td = d->textures[t];
            s->tex_info.push_back(make_uint4((unsigned)s->texels.size(), td.width, td.height, 0u));
            const size_t n = (size_t)td.width * td.height;
            if (s->texels.size() + n > 0xffffffffull) return fail(ctx, YK_ERR_UNSUPPORTED, "more than 2^32 texels");
            for (size_t k = 0; k < n; ++k) s->texels.push_back(make_float4(td.rgb[3 * k], td.rgb[3 * k + 1], td.rgb[3 * k + 2], 0.0f));
        }
        s->root_ref = ref_of(0);
        s->wide_auto = wide && ctx->wide_bvh == 2;
        s->has_device_records = true;
    }
    out = img;
    return YK_OK;
} YK_CATCH(ctx)

// Device half: one copy of the image in the HBM of ctx's device.
yk_status yk_upload_scene_image(yk_context* ctx, const std::shared_ptr<SceneImage>& img, yk_scene** out) try {
    if (!out) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null out");
    *out = nullptr;
    if (!img || !img->bvh) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null scene image");
    yk_scene* s = new yk_scene();
    struct SceneGuard {  // frees the half-built scene on every early return and on an exception
        yk_scene* s;
        ~SceneGuard() {
            if (s) yk_scene_destroy(s);
        }
    } guard{s};
    s->device = ctx ? ctx->device : -1;
    s->bvh = img->bvh;
    s->n_triangles = img->n_triangles;
    s->n_spheres = img->n_spheres;
    s->n_lights = img->n_lights;
    s->n_delta_lights = img->n_delta_lights;
    s->info = img->info;
    if (ctx) {
        if (!img->has_device_records) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "scene image was built without device records");
        const yk_scene_desc* d = img->d;
        (void)hipSetDevice(ctx->device);
        double u0 = now_seconds();
        yk_status st;
#define UP(buf, ptr, n) \
    if ((st = upload(ctx, s->buf, ptr, n)) != YK_OK) return st;
        UP(nodes, img->dn.data(), img->dn.size());
        UP(nodes4, img->dn4.data(), img->dn4.size());
        UP(top_nodes, img->top.data(), img->top.size());
        UP(top_nodes_any, img->top_any.data(), img->top_any.size());
        UP(tris, img->tris.data(), img->tris.size());
        UP(prim_shade, img->prim_shade.data(), img->prim_shade.size());
        UP(prim_attr, img->prim_attr.data(), img->prim_attr.size());
        UP(indices, d->indices, 3 * (size_t)d->n_triangles);
        UP(points, d->points, 3 * (size_t)d->n_vertices);
        UP(normals, d->normals, d->normals ? 3 * (size_t)d->n_vertices : 0);
        UP(uvs, d->uvs, d->uvs ? 2 * (size_t)d->n_vertices : 0);
        UP(tri_mesh, img->tri_mesh.data(), img->tri_mesh.size());
        UP(tri_material, d->tri_material, (size_t)d->n_triangles);
        UP(tri_area_light, img->tri_al.data(), img->tri_al.size());
        UP(mesh_flags, img->mesh_flags.data(), img->mesh_flags.size());
        UP(materials, img->mats.data(), img->mats.size());
        UP(lights, img->lights.data(), img->lights.size());
        UP(spheres, img->spheres.data(), img->spheres.size());
        UP(texels, img->texels.data(), img->texels.size());
        UP(tex_info, img->tex_info.data(), img->tex_info.size());
#undef UP
        const std::vector<yk_bvh_node>& nodes = img->bvh->nodes;
        DevScene& ds = s->dev;
        ds.nodes = s->nodes.as<DevNode>();
        ds.nodes4 = img->wide ? s->nodes4.as<DevNode4>() : nullptr;
        s->wide_auto = img->wide_auto;
        ds.top_nodes = s->top_nodes.as<DevNode>();
        ds.n_top = (uint32_t)img->top.size();
        ds.top_nodes_any = s->top_nodes_any.as<DevNode>();
        ds.n_top_any = (uint32_t)img->top_any.size();
        ds.tris = s->tris.as<float4>();
        ds.prim_shade = s->prim_shade.as<uint4>();
        ds.prim_attr = img->prim_attr.empty() ? nullptr : s->prim_attr.as<float4>();
        ds.spheres = d->n_spheres ? s->spheres.as<DevSphere>() : nullptr;
        ds.n_triangles = d->n_triangles;
        ds.root_ref = img->root_ref;
        for (int k = 0; k < 3; ++k) {
            ds.root_bmin[k] = nodes[0].bmin[k];
            ds.root_bmax[k] = nodes[0].bmax[k];
            ds.background[k] = d->background[k];
        }
        ds.indices = s->indices.as<uint32_t>();
        ds.points = s->points.as<float>();
        ds.normals = s->normals.as<float>();
        ds.uvs = s->uvs.as<float>();
        ds.tri_mesh = s->tri_mesh.as<uint32_t>();
        ds.tri_material = s->tri_material.as<int32_t>();
        ds.tri_area_light = s->tri_area_light.as<int32_t>();
        ds.mesh_flags = s->mesh_flags.as<uint32_t>();
        ds.materials = s->materials.as<Material>();
        ds.lights = s->lights.as<DevLight>();
        ds.n_lights = d->n_lights;
        ds.texels = d->n_textures ? s->texels.as<float4>() : nullptr;
        ds.tex_info = d->n_textures ? s->tex_info.as<uint4>() : nullptr;
        s->on_device = true;
        s->info.upload_seconds = now_seconds() - u0;
        DevBuf* all[] = {&s->nodes, &s->nodes4, &s->top_nodes, &s->top_nodes_any, &s->tris, &s->prim_shade, &s->prim_attr, &s->indices, &s->points, &s->normals, &s->uvs, &s->tri_mesh, &s->tri_material, &s->tri_area_light,
                         &s->mesh_flags, &s->materials, &s->lights, &s->spheres, &s->texels, &s->tex_info};
        for (DevBuf* b : all) s->info.device_bytes += b->bytes;
    }
    guard.s = nullptr;
    *out = s;
    return YK_OK;
} YK_CATCH(ctx)

extern "C" {

yk_status yk_scene_create(yk_context* ctx, const yk_scene_desc* d, yk_scene** out) {
    std::unique_lock<std::recursive_mutex> yk_lock_;
    if (ctx) yk_lock_ = std::unique_lock<std::recursive_mutex>(ctx->mu);
    if (!d || !out) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null scene description");
    *out = nullptr;
    std::shared_ptr<SceneImage> img;
    yk_status st = yk_build_scene_image(ctx, d, img);
    if (st != YK_OK) return st;
    return yk_upload_scene_image(ctx, img, out);
}


void yk_scene_destroy(yk_scene* s) {
    if (!s) return;
    if (s->device >= 0) (void)hipSetDevice(s->device);
    DevBuf* all[] = {&s->nodes, &s->nodes4, &s->top_nodes, &s->top_nodes_any, &s->tris, &s->prim_shade, &s->prim_attr, &s->indices, &s->points, &s->normals, &s->uvs, &s->tri_mesh, &s->tri_material, &s->tri_area_light,
                     &s->mesh_flags, &s->materials, &s->lights, &s->spheres, &s->texels, &s->tex_info};
    for (DevBuf* b : all) b->release();
    delete s;
}

yk_status yk_scene_get_info(const yk_scene* s, yk_scene_info* out) {
    if (!s || !out) return YK_ERR_INVALID_ARGUMENT;
    *out = s->info;
    return YK_OK;
}

yk_status yk_scene_export_bvh(const yk_scene* s, yk_bvh_node* nodes, uint32_t* shape_order) {
    if (!s) return YK_ERR_INVALID_ARGUMENT;
    if (nodes) std::memcpy(nodes, s->bvh->nodes.data(), s->bvh->nodes.size() * sizeof(yk_bvh_node));
    if (shape_order) std::memcpy(shape_order, s->bvh->shape_order.data(), s->bvh->shape_order.size() * sizeof(uint32_t));
    return YK_OK;
}

// ------------------------------------------------------------------ render
// ctx->counters: 8 x u64 (closest-hit rays, shadow rays, ...) followed by a 4-word error block whose word
// YK_CTRL_ERR the traversal kernels set on a stack overflow.  Both are zeroed ONCE per call (begin_call) — the
// per-batch control blocks of the work sets are zeroed with every batch and must not hold the flag.
#define YK_COUNTER_BYTES 96
static unsigned* error_block(yk_context* ctx) { return reinterpret_cast<unsigned*>(ctx->counters.as<unsigned long long>() + 8); }

static yk_status ensure_work_buffers(yk_context* ctx, WorkSet& ws, size_t paths, unsigned n_lights, unsigned n_delta_lights) {
    HIP_TRY(ctx, ctx->counters.ensure(YK_COUNTER_BYTES));
    unsigned nl = std::max(1u, n_lights);
    unsigned na = nl, nd = std::max(1u, n_delta_lights);  // queue 1 holds every light's rays on the bounces that are not split
    if (paths <= ws.cap_paths && nl <= ws.cap_lights && na <= ws.cap_area && nd <= ws.cap_delta) return YK_OK;
    paths = std::max(paths, ws.cap_paths);
    nl = std::max(nl, ws.cap_lights);
    na = std::max(na, ws.cap_area);
    nd = std::max(nd, ws.cap_delta);
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 4; ++b) HIP_TRY(ctx, ws.path[a][b].ensure(paths * 16));
    HIP_TRY(ctx, ws.hit.ensure(paths * 4));
    HIP_TRY(ctx, ws.pend.ensure(paths * 16));
    HIP_TRY(ctx, ws.shC.ensure(paths * nl * 16));
    HIP_TRY(ctx, ws.vis.ensure(paths * YK_VIS_STRIDE(nl)));
    // two shadow queues: rays towards area lights / towards point, spot and distant lights
    HIP_TRY(ctx, ws.shO.ensure(paths * na * 16));
    HIP_TRY(ctx, ws.shD.ensure(paths * na * 16));
    HIP_TRY(ctx, ws.shq.ensure(paths * na * 4));
    HIP_TRY(ctx, ws.shO2.ensure(paths * nd * 16));
    HIP_TRY(ctx, ws.shD2.ensure(paths * nd * 16));
    HIP_TRY(ctx, ws.shq2.ensure(paths * nd * 4));
    HIP_TRY(ctx, ws.ctrl.ensure(YK_CTRL_ALLOC_WORDS * 4));
    ws.cap_paths = paths;
    ws.cap_lights = nl;
    ws.cap_area = na;
    ws.cap_delta = nd;
    return YK_OK;
}

static unsigned trace_grid(const yk_context* ctx) { return (unsigned)ctx->n_cu * trace_blocks_per_cu(); }

static yk_status ensure_spill(yk_context* ctx, WorkSet& ws) {
    size_t threads = (size_t)trace_grid(ctx) * trace_block_size();
    HIP_TRY(ctx, ws.spill.ensure(threads * trace_spill_depth() * 8));
    HIP_TRY(ctx, ws.spill_side.ensure(threads * trace_spill_depth() * 8));
    return YK_OK;
}

static PathBuffers path_buffers(WorkSet& ws, int which) {
    PathBuffers p;
    p.rayO = ws.path[which][0].as<float4>();
    p.rayD = ws.path[which][1].as<float4>();
    p.thru = ws.path[which][2].as<float4>();
    p.rngs = ws.path[which][3].as<uint4>();
    return p;
}

static yk_status make_params(yk_context* ctx, const yk_sampler_desc* smp, const yk_integrator_desc* integ, RenderParams& prm) {
    if (!smp || !integ) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null sampler/integrator");
    std::memset(&prm, 0, sizeof(prm));
    prm.sampler.kind = smp->kind;
    prm.sampler.nx = smp->nx;
    prm.sampler.ny = smp->kind == YK_SAMPLER_UNIFORM ? 1 : smp->ny;
    prm.sampler.jitter = smp->jitter;
    prm.sampler.seed = smp->seed;
    if (smp->kind > 1 || prm.sampler.nx == 0 || prm.sampler.ny == 0) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "bad sampler");
    uint64_t spp = (uint64_t)prm.sampler.nx * prm.sampler.ny;
    if (spp > 0xFFFFu) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "samples per pixel exceed u16 (integrators/mod.rs:139)");
    prm.sampler.spp = (unsigned)spp;
    prm.spe = (unsigned)spp;  // plain film: every sample of the pixel (callers with a sample-index table overwrite it)
    prm.max_depth = integ->max_depth;
    prm.has_clamp = integ->has_clamp;
    prm.clamp = integ->indirect_clamp;
    prm.integrator = integ->kind;
    if (integ->kind == YK_INTEGRATOR_WHITTED && integ->max_depth > whitted_max_depth())
        return fail(ctx, YK_ERR_UNSUPPORTED, "Whitted: max_depth above 16 is not supported on the device");
    if (integ->kind > YK_INTEGRATOR_SHADING_NORMALS) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "bad integrator kind");
    return YK_OK;
}

struct KernelTimer {
    yk_context* ctx;
    bool on;
    std::vector<std::pair<int, int>> spans[3];  // 0 trace, 1 shadow, 2 shade
    size_t used = 0;
    int begin(hipStream_t s) {
        if (!on) return -1;
        if (used + 2 > ctx->ev_pool.size()) {
            size_t old = ctx->ev_pool.size();
            ctx->ev_pool.resize(old + 256);
            for (size_t i = old; i < ctx->ev_pool.size(); ++i) (void)hipEventCreate(&ctx->ev_pool[i]);
        }
        int a = (int)used;
        used += 2;
        (void)hipEventRecord(ctx->ev_pool[a], s);
        return a;
    }
    void end(int a, int cls, hipStream_t s) {
        if (a < 0) return;
        (void)hipEventRecord(ctx->ev_pool[a + 1], s);
        spans[cls].push_back(std::make_pair(a, a + 1));
    }
    double total(int cls) {
        double ms = 0.0;
        for (auto& sp : spans[cls]) {
            float t = 0.0f;
            if (hipEventElapsedTime(&t, ctx->ev_pool[sp.first], ctx->ev_pool[sp.second]) == hipSuccess) ms += t;
        }
        return ms * 1e-3;
    }
};

// camera rays (and the leading bounces the "packet_bounces" option names) go to the wave-packet kernel
static bool packet_kernel_traces_bounce(const yk_context* ctx, const yk_scene* scene, unsigned b) {
    return b < (unsigned)ctx->packet_bounces && scene->bvh->depth <= 64;
}

// one batch of `n` paths already generated into buffer 0; runs the bounce loop
static void run_bounces(yk_context* ctx, WorkSet& ws, hipStream_t st, const yk_scene* scene, const RenderParams& prm, const uint32_t* pixel_xy,
                        const uint32_t* sample_index_tab, float4* sample_buf, KernelTimer& kt, unsigned long long* counters, bool coherent,
                        uint32_t n_paths, uint32_t sid_base, bool lean_camera_bounce, uint32_t* n_shadow_launches = nullptr) {
    unsigned* ctrl = ws.ctrl.as<unsigned>();
    unsigned* errblk = error_block(ctx);  // outlives the batch (ctrl is zeroed per batch)
    // Two node layouts: the binary 64-byte nodes win when the machine is full (one 4-wide node
    // costs the loads of two binary ones, and throughput is bound by per-lane loads);
    // the 4-wide collapse halves the dependent steps of a ray, which is what a job too small to
    // fill the machine waits for (a 16x16 tile: 1.83 -> 1.38 ms, a 1080p pass: 7.7 -> 7.0 ms,
    // equal at 8 M paths, 9 % slower for the 132 M-path frame).
    const DevScene ds = dev_scene_for(scene, n_paths);
    // Queue lengths are only known on the device, but none exceeds the batch's path count
    // (x lights for shadow rays).  A small job — one 16x16 tile of the reference's per-tile
    // calls is 16 K paths — gets grids of that size instead of machine-filling ones: every wave
    // of a persistent kernel pays one atomic on the queue head before it can find out that
    // there is nothing for it (7168 waves x 8 bounces x 3 kernels per tile otherwise).
    auto fit = [](unsigned full, uint64_t items) { return (unsigned)std::min<uint64_t>(full, std::max<uint64_t>(1, (items + 255) / 256)); };
    const uint64_t n_shadow_max = (uint64_t)n_paths * std::max(1u, scene->n_lights);
    const unsigned tg = fit(trace_grid(ctx), n_paths), tg_any = fit(trace_grid(ctx), n_shadow_max);
    const unsigned pg_full = (unsigned)ctx->n_cu * packet_blocks_per_cu();
    const unsigned pg = fit(pg_full, n_paths), pg_any = fit(pg_full, n_shadow_max);
    // k_shade / k_accumulate are grid-stride kernels: 256 blocks per CU (three are resident) let the block
    // scheduler even out the iterations' very different costs; 8 persistent-style blocks per CU were 2.9 % slower
    // on the frame (sweep 3..1024: 147.3, 146.9, 148.0 (8), 146.6, 145.4 (24), 145.1 (96), 143.7 (256), 144.2, 144.3 ms)
    static const unsigned shade_bpc = std::getenv("YK_SHADE_BPC") ? (unsigned)std::atoi(std::getenv("YK_SHADE_BPC")) : 256u;
    const unsigned sg = fit((unsigned)ctx->n_cu * shade_bpc, n_paths);  // k_accumulate: 256 paths per block and step
    const unsigned spill_stride = trace_grid(ctx) * trace_block_size();
    unsigned cur = 0;
    // Bounce b: trace_closest -> shade on `st`; then {trace_any, accumulate}(b) go to the side
    // stream while `st` already traces bounce b+1 — two persistent kernels whose drained
    // CUs are picked up by the other one (the tail of a small launch is one long ray).
    // shade(b+1) overwrites what accumulate(b) reads (the other path buffer, pend, shC, vis,
    // the shadow queue and its counter), so it waits for ev_acc.
    const bool overlap = ctx->overlap_shadow != 0 && ws.side != nullptr;
    hipStream_t sb = overlap ? ws.side : st;
    for (unsigned b = 0; b < prm.max_depth; ++b) {
        PathBuffers pc = path_buffers(ws, (int)cur), pn = path_buffers(ws, (int)(cur ^ 1u));
        // camera rays (consecutive samples of a pixel) and the shadow rays they spawn are coherent:
        // the wave walks the tree once for all 64 of them (yk_packet.hip)
        const bool packet = coherent && packet_kernel_traces_bounce(ctx, scene, b);
        // lean camera bounce (yk_device.h, YK_CTRL_CAM_O): raygen stored neither origins nor throughputs
        const float4* lean_origin = (b == 0 && lean_camera_bounce) ? reinterpret_cast<const float4*>(ctrl + YK_CTRL_CAM_O) : nullptr;
        const bool packet_shadow = coherent && b < (unsigned)ctx->packet_shadow_bounces && scene->bvh->depth <= 64 && scene->n_delta_lights > 0;
        // shadow rays are split into two queues only when the second one gets the packet kernel;
        // otherwise everything goes to the first queue and one launch traces it
        const bool split = packet_shadow && scene->n_lights > scene->n_delta_lights;
        const bool all_delta = packet_shadow && !split;  // no area lights: the single queue is all coherent
        unsigned* bc = ctrl + YK_CTRL_BOUNCE(b);  // this bounce's counters and queue heads, zeroed with the batch
        int e = kt.begin(st);
        if (packet)
            launch_trace_closest_packet(st, pg, ds, lean_origin ? nullptr : pc.rayO, pc.rayD, bc, bc + YK_CTRL_HEAD, ws.hit.as<int>(), counters, lean_origin);
        else
            launch_trace_closest(st, tg, ds, pc.rayO, pc.rayD, nullptr, bc, bc + YK_CTRL_HEAD, ws.hit.as<int>(), nullptr, nullptr,
                                 ws.spill.as<uint2>(), spill_stride, errblk, counters);
        kt.end(e, 0, st);
        if (overlap && b > 0) (void)hipStreamWaitEvent(st, ws.ev_acc, 0);
        e = kt.begin(st);
        launch_shade(st, sg, ds, prm, pixel_xy, sample_index_tab, pc, pn, ws.hit.as<int>(), ws.pend.as<float4>(), ws.shO.as<float4>(),
                     ws.shD.as<float4>(), ws.shC.as<float4>(), ws.vis.as<unsigned char>(), ws.shq.as<unsigned>(), ws.shO2.as<float4>(),
                     ws.shD2.as<float4>(), ws.shq2.as<unsigned>(), bc, split ? 1u : 0u, (b > 0 && ctx->shade_reorder) ? 1u : 0u, 3u * (unsigned)ctx->n_cu, sid_base, lean_origin);
        kt.end(e, 2, st);
        if (overlap) {
            (void)hipEventRecord(ws.ev_shade, st);
            (void)hipStreamWaitEvent(sb, ws.ev_shade, 0);
        }
        e = kt.begin(sb);
        uint2* any_spill = (overlap ? ws.spill_side : ws.spill).as<uint2>();
        if (all_delta) {
            launch_trace_any_packet(sb, pg_any, ds, ws.shO.as<float4>(), ws.shD.as<float4>(), ws.shq.as<unsigned>(), bc + YK_CTRL_SHQ,
                                    bc + YK_CTRL_HEAD + 1, ws.vis.as<unsigned char>(), counters + 1);
        } else {
            launch_trace_any(sb, tg_any, ds, ws.shO.as<float4>(), ws.shD.as<float4>(), ws.shq.as<unsigned>(), bc + YK_CTRL_SHQ,
                             bc + YK_CTRL_HEAD + 1, ws.vis.as<unsigned char>(), any_spill, spill_stride, errblk, counters + 1);
            if (split && n_shadow_launches) ++*n_shadow_launches;
            if (split)  // rays converging on a point / spot / distant light: wave packets
                launch_trace_any_packet(sb, pg_any, ds, ws.shO2.as<float4>(), ws.shD2.as<float4>(), ws.shq2.as<unsigned>(), bc + YK_CTRL_SHQ2,
                                        bc + YK_CTRL_HEAD + 2, ws.vis.as<unsigned char>(), counters + 1);
        }
        kt.end(e, 1, sb);
        if (n_shadow_launches) ++*n_shadow_launches;
        launch_accumulate(sb, sg, prm, pc, ws.pend.as<float4>(), ws.shC.as<float4>(), ws.vis.as<unsigned char>(), ds.n_lights, sample_buf, bc, b == 0 ? 1u : 0u, sid_base);
        if (overlap) (void)hipEventRecord(ws.ev_acc, sb);
        if (kt.on && std::getenv("YK_DEBUG_BOUNCES")) {  // per-bounce breakdown (synchronises; diagnostics only)
            unsigned h[YK_CTRL_STRIDE + 1];
            (void)hipStreamSynchronize(st);
            (void)hipStreamSynchronize(sb);
            (void)hipMemcpy(h, bc, sizeof(h), hipMemcpyDeviceToHost);
            float tt = 0, ts = 0, th = 0;
            (void)hipEventElapsedTime(&tt, ctx->ev_pool[kt.spans[0].back().first], ctx->ev_pool[kt.spans[0].back().second]);
            (void)hipEventElapsedTime(&ts, ctx->ev_pool[kt.spans[1].back().first], ctx->ev_pool[kt.spans[1].back().second]);
            (void)hipEventElapsedTime(&th, ctx->ev_pool[kt.spans[2].back().first], ctx->ev_pool[kt.spans[2].back().second]);
            std::fprintf(stderr, "bounce %u: rays %u trace %.3f ms (%.0f Mray/s) | shadow rays %u %.3f ms (%.0f Mray/s) | shade %.3f ms | survivors %u\n", b, h[0],
                         tt, h[0] / (tt * 1e3), h[YK_CTRL_SHQ] + h[YK_CTRL_SHQ2], ts, (h[YK_CTRL_SHQ] + h[YK_CTRL_SHQ2]) / (ts * 1e3), th, h[YK_CTRL_STRIDE]);
        }
        cur ^= 1u;
    }
    if (overlap) (void)hipStreamWaitEvent(st, ws.ev_acc, 0);  // the batch is complete on `st` once its last accumulate is
}

}  // extern "C"

// Integrator::render for a list of tiles.  tile_samples == nullptr: the plain film (all
// samples of a pixel, mean stored).  Otherwise the accumulating film (integrators/mod.rs:
// 146-161): one sample per pixel with global index tile_samples[t], raw value stored.
static yk_status render_tiles_impl(yk_context* ctx, const yk_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                                   const yk_integrator_desc* integrator, const yk_tile* tiles, const uint16_t* tile_samples, size_t n_tiles,
                                   void* d_out_rgb, void* stream, yk_render_stats* stats, yk_cancel_fn cancel, void* user,
                                   const yk_tile_list* prepared = nullptr, uint32_t n_passes = 1) try {
    if (!ctx) return YK_ERR_INVALID_ARGUMENT;
    YK_LOCK(ctx);
    if (prepared) {
        if (prepared->device != ctx->device) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "tile list was not created on this context's device");
        tiles = prepared->tiles.data();
        tile_samples = prepared->samples.empty() ? nullptr : prepared->samples.data();
        n_tiles = prepared->tiles.size();
    }
    if (!scene || !camera || !tiles || n_tiles == 0 || !d_out_rgb) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null argument");
    if (!scene->on_device || scene->device != ctx->device) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "scene was not created on this context's device");
    RenderParams prm;
    yk_status ps = make_params(ctx, sampler, integrator, prm);
    if (ps != YK_OK) return ps;
    if (prm.integrator == YK_INTEGRATOR_PATH && prm.max_depth > YK_CTRL_MAX_DEPTH)
        return fail(ctx, YK_ERR_INVALID_ARGUMENT, "max_depth too large");
    (void)hipSetDevice(ctx->device);
    // The render always runs on the context's own streams; a caller's stream hands over to them
    // and takes over again at the end (two event waits), so the work is ordered on it as if it
    // had been launched there — and the main / side stream pair keeps its own hardware queues.
    hipStream_t caller = (hipStream_t)stream;
    hipStream_t st = ctx->stream;
    if (caller) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_in, caller));
        HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_in, 0));
    }
    struct HandBack {  // on every exit path: whatever was enqueued is ordered before the caller's next work
        yk_context* c;
        hipStream_t caller, st;
        ~HandBack() {
            if (!caller) return;
            if (hipEventRecord(c->ev_out, st) == hipSuccess) (void)hipStreamWaitEvent(caller, c->ev_out, 0);
        }
    } hand_back{ctx, caller, st};

    // tiles -> pixel ranges (assert!(tile_pixels.len() >= tile.bb.area()), integrators/mod.rs:131)
    std::vector<uint32_t> off(n_tiles + 1, 0);
    uint64_t total_px = 0;
    for (size_t t = 0; t < n_tiles; ++t) {
        if (tiles[t].x0 >= tiles[t].x1 || tiles[t].y0 >= tiles[t].y1) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "Bounds2 with a dimension <= 0");
        total_px += (uint64_t)(tiles[t].x1 - tiles[t].x0) * (uint64_t)(tiles[t].y1 - tiles[t].y0);
        if (total_px > 0xFFFFFFFFull) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "too many pixels in one call");
        off[t + 1] = (uint32_t)total_px;
    }
    const bool accumulating = tile_samples != nullptr;
    if (n_passes == 0 || n_passes > 0xFFFFu || (!accumulating && n_passes != 1)) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "bad number of passes");
    // samples rendered per pixel by this call: the accumulating film renders passes FilmTile.sample .. + n_passes - 1
    const uint32_t spp = accumulating ? n_passes : prm.sampler.spp;
    prm.spe = spp;
    if (accumulating) {
        // render_manager.rs:135-143 queues samples 0 .. spp-1 of a tile and nothing else; an index beyond that is
        // outside the samplers' domain (the stratified permutation walks cycles of [0, spp) and need not terminate)
        for (size_t t = 0; t < n_tiles; ++t)
            if ((uint64_t)tile_samples[t] + n_passes > prm.sampler.spp)
                return fail(ctx, YK_ERR_INVALID_ARGUMENT, "FilmTile.sample (+ passes) beyond the sampler's samples per pixel");
    }
    // chunk so that sample ids fit u32 and the sample buffer stays under the cap
    uint64_t max_px_chunk = std::min<uint64_t>(0xFFFFFFF0ull / spp, (uint64_t)ctx->sample_buf_cap / (16ull * spp));
    if (max_px_chunk == 0) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "sample_buf_cap too small for one pixel");

    DevCamera cam;
    std::memcpy(cam.c2w, camera->camera_to_world, 64);
    std::memcpy(cam.r2c, camera->raster_to_camera, 64);

    const bool is_path = prm.integrator == YK_INTEGRATOR_PATH;
    // Work is cut into batches of <= batch_paths camera samples.  With streams == 2
    // batches alternate between two work sets / HIP streams, so the latency-bound
    // tail launches of one batch (late bounces, few rays) run beside the bulk
    // launches of the other.  A job that fits one batch stays on one work set: splitting it
    // gains nothing once the side stream overlaps shadow rays with the next bounce (measured).
    const uint64_t total_work = total_px * spp;
    size_t batch = (size_t)std::min<uint64_t>((uint64_t)ctx->batch_paths, total_work);
    const int n_ws = (ctx->streams >= 2 && is_path && !stream && total_work > batch) ? 2 : 1;
    {
        // keep the per-batch work buffers (148 + 53*n_lights bytes per path) within half of the free HBM
        size_t free_b = 0, total_b = 0;
        if (batch > ctx->ws[0].cap_paths && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t per_path = (148 + 53 * (size_t)std::max(1u, scene->n_lights) + 36) * (size_t)n_ws;
            const size_t fit = (free_b / 2) / per_path;
            if (fit >= 65536 && batch > fit) batch = fit;
        }
    }
    if (n_ws == 2 && !ctx->ws[1].stream) {  // the second work set's stream pair, on first use
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->ws[1].stream, hipStreamNonBlocking));
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->ws[1].side, hipStreamNonBlocking));
    }
    for (int w = 0; w < n_ws; ++w) {
        yk_status wb = ensure_work_buffers(ctx, ctx->ws[w], batch, scene->n_lights, scene->n_delta_lights);
        if (wb != YK_OK) return wb;
        if ((wb = ensure_spill(ctx, ctx->ws[w])) != YK_OK) return wb;
    }
    HIP_TRY(ctx, ctx->tiles.ensure(n_tiles * sizeof(yk_tile)));
    HIP_TRY(ctx, ctx->tile_off.ensure((n_tiles + 1) * 4));
    if (!is_path && prm.integrator == YK_INTEGRATOR_BVH_INTERSECTIONS) HIP_TRY(ctx, ctx->stats4.ensure(batch * 16));

    unsigned long long* counters = ctx->counters.as<unsigned long long>();
    HIP_TRY(ctx, hipMemsetAsync(counters, 0, YK_COUNTER_BYTES, st));
    unsigned* errblk = error_block(ctx);
    KernelTimer kt;
    kt.ctx = ctx;
    // per-kernel HIP-event timings: two event records per launch and one elapsed-time query per
    // kernel — not for jobs so small (a tile, a few tiles) that this bookkeeping is the cost
    kt.on = stats != nullptr && ctx->time_kernels != 0 && (total_work >= (1u << 20) || ctx->time_kernels > 1);
    struct EventPair {  // destroyed on every exit path
        hipEvent_t a = nullptr, b = nullptr;
        ~EventPair() {
            if (a) (void)hipEventDestroy(a);
            if (b) (void)hipEventDestroy(b);
        }
    } frame_ev;
    hipEvent_t& ev0 = frame_ev.a;
    hipEvent_t& ev1 = frame_ev.b;
    if (stats) {
        HIP_TRY(ctx, hipEventCreate(&ev0));
        HIP_TRY(ctx, hipEventCreate(&ev1));
        HIP_TRY(ctx, hipEventRecord(ev0, st));
    }
    uint32_t n_batches = 0, n_trace = 0, n_shadow = 0;
    float* out = reinterpret_cast<float*>(d_out_rgb);

    size_t t_begin = 0;
    while (t_begin < n_tiles) {
        size_t t_end = t_begin;
        while (t_end < n_tiles && (uint64_t)(off[t_end + 1] - off[t_begin]) <= max_px_chunk) ++t_end;
        if (t_end == t_begin) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "a single tile exceeds sample_buf_cap");
        const uint32_t px0 = off[t_begin], npx = off[t_end] - off[t_begin];
        HIP_TRY(ctx, ctx->sample_buf.ensure((size_t)npx * spp * 16));
        float4* sample_buf = ctx->sample_buf.as<float4>();
        uint32_t* pixel_xy = nullptr;
        uint32_t* pixel_sample = nullptr;
        const uint16_t* d_tile_sample = nullptr;
        if (prepared) {  // the pixel table of the whole list is already on the device
            pixel_xy = prepared->pixel_xy.as<uint32_t>() + px0;
            if (accumulating) pixel_sample = prepared->pixel_sample.as<uint32_t>() + px0;
        } else if (t_end - t_begin == 1) {  // one tile (the reference's per-tile call): it travels as a kernel argument
            HIP_TRY(ctx, ctx->pixel_xy.ensure((size_t)npx * 4));
            pixel_xy = ctx->pixel_xy.as<uint32_t>();
            if (accumulating) {
                HIP_TRY(ctx, ctx->scratch[5].ensure((size_t)npx * 4));
                pixel_sample = ctx->scratch[5].as<uint32_t>();
            }
            launch_pixel_table_one(st, tiles[t_begin], npx, pixel_xy, accumulating ? tile_samples[t_begin] : 0u, pixel_sample);
        } else {
        std::vector<uint32_t> loc(t_end - t_begin + 1);
        for (size_t t = t_begin; t <= t_end; ++t) loc[t - t_begin] = off[t] - px0;
        HIP_TRY(ctx, hipMemcpyAsync(ctx->tiles.p, tiles + t_begin, (t_end - t_begin) * sizeof(yk_tile), hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->tile_off.p, loc.data(), loc.size() * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));  // `loc` is a stack-lifetime staging buffer
        HIP_TRY(ctx, ctx->pixel_xy.ensure((size_t)npx * 4));
        pixel_xy = ctx->pixel_xy.as<uint32_t>();
        if (accumulating) {
            HIP_TRY(ctx, ctx->scratch[4].ensure((t_end - t_begin) * 2));
            HIP_TRY(ctx, ctx->scratch[5].ensure((size_t)npx * 4));
            HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[4].p, tile_samples + t_begin, (t_end - t_begin) * 2, hipMemcpyHostToDevice, st));
            d_tile_sample = ctx->scratch[4].as<uint16_t>();
            pixel_sample = ctx->scratch[5].as<uint32_t>();
        }
        launch_pixel_table(st, ctx->tiles.as<yk_tile>(), ctx->tile_off.as<uint32_t>(), (uint32_t)(t_end - t_begin), npx, pixel_xy, d_tile_sample, pixel_sample);
        }
        // the pixel's share of every camera sample's sampler start, once per pixel (yk_rng.h, PixelSampler)
        HIP_TRY(ctx, ctx->pixel_aux.ensure((size_t)npx * 16));
        launch_pixel_sampler(st, prm.sampler, pixel_xy, npx, ctx->pixel_aux.as<uint4>());
        // the second stream starts after the pixel table exists
        if (n_ws == 2) {
            HIP_TRY(ctx, hipEventRecord(ctx->ws[0].done, st));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->ws[1].stream, ctx->ws[0].done, 0));
        }

        const uint64_t work = (uint64_t)npx * spp;
        int which = 0;
        for (uint64_t w0 = 0; w0 < work; w0 += batch) {
            if (cancel && cancel(user)) {
                (void)hipStreamSynchronize(st);
                if (n_ws == 2) (void)hipStreamSynchronize(ctx->ws[1].stream);
                return fail(ctx, YK_ERR_CANCELLED, "cancelled by early_termination_predicate");
            }
            WorkSet& ws = ctx->ws[which];
            hipStream_t bs = n_ws == 2 ? ws.stream : st;
            unsigned* ctrl = ws.ctrl.as<unsigned>();
            const uint32_t n = (uint32_t)std::min<uint64_t>(batch, work - w0);
            HIP_TRY(ctx, hipMemsetAsync(ctrl, 0, YK_CTRL_WORDS * 4, bs));
            // Path, camera rays traced by the packet kernel: the lean camera bounce (yk_device.h, YK_CTRL_CAM_O)
            const bool lean = is_path && prm.max_depth > 0 && packet_kernel_traces_bounce(ctx, scene, 0);
            launch_raygen(bs, cam, prm, pixel_xy, pixel_sample, w0, n, path_buffers(ws, 0), sample_buf, ctrl + YK_CTRL_BOUNCE(0),
                          lean ? reinterpret_cast<float4*>(ctrl + YK_CTRL_CAM_O) : nullptr, ctx->pixel_aux.as<uint4>());
            ++n_batches;
            if (is_path) {
                run_bounces(ctx, ws, bs, scene, prm, pixel_xy, pixel_sample, sample_buf, kt, counters, true, n, (uint32_t)w0, lean, &n_shadow);
                n_trace += prm.max_depth;
            } else if (prm.integrator == YK_INTEGRATOR_WHITTED) {
                // one lane per camera sample runs the whole recursion (whitted.rs:74-181)
                int e = kt.begin(bs);
                launch_whitted(bs, trace_grid(ctx), scene->dev, prm, pixel_xy, pixel_sample, path_buffers(ws, 0), n, sample_buf, ws.spill.as<uint2>(),
                               trace_grid(ctx) * trace_block_size(), errblk, counters);
                kt.end(e, 0, bs);
                ++n_trace;
            } else {
                PathBuffers pc = path_buffers(ws, 0);
                const bool want_stats = prm.integrator == YK_INTEGRATOR_BVH_INTERSECTIONS;
                int e = kt.begin(bs);
                launch_trace_closest(bs, trace_grid(ctx), dev_scene_for(scene, n), pc.rayO, pc.rayD, nullptr, ctrl + YK_CTRL_BOUNCE(0), ctrl + YK_CTRL_BOUNCE(0) + YK_CTRL_HEAD, ws.hit.as<int>(), nullptr,
                                     want_stats ? ctx->stats4.as<uint4>() : nullptr, ws.spill.as<uint2>(), trace_grid(ctx) * trace_block_size(), errblk,
                                     counters);
                kt.end(e, 0, bs);
                launch_debug_shade(bs, scene->dev, prm.integrator, pc, ws.hit.as<int>(), ctx->stats4.as<uint4>(), n, sample_buf);
                ++n_trace;
            }
            if (n_ws == 2) which ^= 1;
        }
        if (n_ws == 2) {  // resolve (on the caller-visible stream) waits for the second stream
            HIP_TRY(ctx, hipEventRecord(ctx->ws[1].done, ctx->ws[1].stream));
            HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ws[1].done, 0));
        }
        if (accumulating)  // raw values, pass-major over the whole tile list
            launch_resolve_passes(st, sample_buf, npx, spp, out + 3 * (size_t)px0, 3 * (size_t)total_px);
        else
            launch_resolve(st, sample_buf, npx, spp, out + 3 * (size_t)px0);
        t_begin = t_end;
    }
    HIP_TRY(ctx, hipGetLastError());
    if (stats) {
        HIP_TRY(ctx, hipEventRecord(ev1, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        std::memset(stats, 0, sizeof(*stats));
        unsigned long long host_counters[YK_COUNTER_BYTES / 8];
        HIP_TRY(ctx, hipMemcpy(host_counters, counters, YK_COUNTER_BYTES, hipMemcpyDeviceToHost));
        unsigned host_err[4];
        std::memcpy(host_err, host_counters + 8, sizeof(host_err));
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, ev0, ev1);
        stats->rays = host_counters[0];
        stats->shadow_rays = host_counters[1];
        stats->samples = total_px * spp;
        stats->seconds_total = ms * 1e-3;
        stats->seconds_trace = kt.total(0);
        stats->seconds_shadow = kt.total(1);
        stats->seconds_shade = kt.total(2);
        stats->trace_launches = n_trace;
        stats->shadow_launches = n_shadow;
        stats->batches = n_batches;
        if (host_err[YK_CTRL_ERR] & 1u) return fail(ctx, YK_ERR_STACK_OVERFLOW, "BVH traversal stack exceeded 64 entries (bvh.rs:174)");
    } else if (scene->bvh->depth > 64) {
        // A tree deeper than the reference's 64-entry stack (bvh.rs:172-174) can overflow it: such a render is
        // not left asynchronous — the flag is read before the call returns, whoever the caller is.
        unsigned host_err[4];
        HIP_TRY(ctx, hipMemcpyAsync(host_err, errblk, sizeof(host_err), hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        if (host_err[YK_CTRL_ERR] & 1u) return fail(ctx, YK_ERR_STACK_OVERFLOW, "BVH traversal stack exceeded 64 entries (bvh.rs:174)");
    }
    return YK_OK;
} YK_CATCH(ctx)

static yk_status render_tiles_host(yk_context* ctx, const yk_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                                   const yk_integrator_desc* integrator, const yk_tile* tiles, const uint16_t* tile_samples, size_t n_tiles,
                                   float* out_rgb, yk_render_stats* stats, yk_cancel_fn cancel, void* user, uint32_t n_passes = 1);

extern "C" {

yk_status yk_render_tiles_device(yk_context* ctx, const yk_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                                 const yk_integrator_desc* integrator, const yk_tile* tiles, size_t n_tiles, void* d_out_rgb, void* stream,
                                 yk_render_stats* stats, yk_cancel_fn cancel, void* user) {
    return render_tiles_impl(ctx, scene, camera, sampler, integrator, tiles, nullptr, n_tiles, d_out_rgb, stream, stats, cancel, user);
}

yk_status yk_render_tiles_accumulating_device(yk_context* ctx, const yk_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                                              const yk_integrator_desc* integrator, const yk_tile* tiles, const uint16_t* tile_samples, size_t n_tiles,
                                              void* d_out_rgb, void* stream, yk_render_stats* stats, yk_cancel_fn cancel, void* user) {
    if (!ctx) return YK_ERR_INVALID_ARGUMENT;
    YK_LOCK(ctx);
    if (!tile_samples) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null tile_samples");
    return render_tiles_impl(ctx, scene, camera, sampler, integrator, tiles, tile_samples, n_tiles, d_out_rgb, stream, stats, cancel, user);
}

yk_status yk_render_tiles_accumulating(yk_context* ctx, const yk_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                                       const yk_integrator_desc* integrator, const yk_tile* tiles, const uint16_t* tile_samples, size_t n_tiles,
                                       float* out_rgb, yk_render_stats* stats, yk_cancel_fn cancel, void* user) {
    if (!ctx) return YK_ERR_INVALID_ARGUMENT;
    YK_LOCK(ctx);
    if (!tile_samples) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null tile_samples");
    return render_tiles_host(ctx, scene, camera, sampler, integrator, tiles, tile_samples, n_tiles, out_rgb, stats, cancel, user);
}

yk_status yk_render_tiles_accumulating_passes(yk_context* ctx, const yk_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                                              const yk_integrator_desc* integrator, const yk_tile* tiles, const uint16_t* tile_samples, size_t n_tiles,
                                              uint32_t n_passes, float* out_rgb, yk_render_stats* stats, yk_cancel_fn cancel, void* user) {
    if (!ctx) return YK_ERR_INVALID_ARGUMENT;
    YK_LOCK(ctx);
    if (!tile_samples) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null tile_samples");
    return render_tiles_host(ctx, scene, camera, sampler, integrator, tiles, tile_samples, n_tiles, out_rgb, stats, cancel, user, n_passes);
}

yk_status yk_tile_list_create(yk_context* ctx, const yk_tile* tiles, const uint16_t* tile_samples, size_t n_tiles, yk_tile_list** out) try {
    if (!ctx) return YK_ERR_INVALID_ARGUMENT;
    YK_LOCK(ctx);
    if (!tiles || !out || n_tiles == 0) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    yk_tile_list* l = new yk_tile_list();
    l->device = ctx->device;
    l->tiles.assign(tiles, tiles + n_tiles);
    if (tile_samples) l->samples.assign(tile_samples, tile_samples + n_tiles);
    l->off.assign(n_tiles + 1, 0);
    uint64_t total = 0;
    for (size_t t = 0; t < n_tiles; ++t) {
        if (tiles[t].x0 >= tiles[t].x1 || tiles[t].y0 >= tiles[t].y1) {
            delete l;
            return fail(ctx, YK_ERR_INVALID_ARGUMENT, "Bounds2 with a dimension <= 0");
        }
        total += (uint64_t)(tiles[t].x1 - tiles[t].x0) * (uint64_t)(tiles[t].y1 - tiles[t].y0);
        if (total > 0xFFFFFFFFull) {
            delete l;
            return fail(ctx, YK_ERR_INVALID_ARGUMENT, "too many pixels in one list");
        }
        l->off[t + 1] = (uint32_t)total;
    }
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    yk_status rc = YK_OK;
    auto tryhip = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == YK_OK) rc = fail(ctx, e == hipErrorOutOfMemory ? YK_ERR_OUT_OF_MEMORY : YK_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
    };
    tryhip(ctx->tiles.ensure(n_tiles * sizeof(yk_tile)), "tiles");
    tryhip(ctx->tile_off.ensure((n_tiles + 1) * 4), "tile offsets");
    tryhip(l->pixel_xy.ensure((size_t)total * 4), "pixel table");
    if (tile_samples) {
        tryhip(ctx->scratch[4].ensure(n_tiles * 2), "tile samples");
        tryhip(l->pixel_sample.ensure((size_t)total * 4), "pixel samples");
    }
    if (rc == YK_OK) {
        tryhip(hipMemcpyAsync(ctx->tiles.p, tiles, n_tiles * sizeof(yk_tile), hipMemcpyHostToDevice, st), "upload tiles");
        tryhip(hipMemcpyAsync(ctx->tile_off.p, l->off.data(), l->off.size() * 4, hipMemcpyHostToDevice, st), "upload offsets");
        if (tile_samples) tryhip(hipMemcpyAsync(ctx->scratch[4].p, tile_samples, n_tiles * 2, hipMemcpyHostToDevice, st), "upload samples");
    }
    if (rc == YK_OK) {
        launch_pixel_table(st, ctx->tiles.as<yk_tile>(), ctx->tile_off.as<uint32_t>(), (uint32_t)n_tiles, (uint32_t)total, l->pixel_xy.as<uint32_t>(),
                           tile_samples ? ctx->scratch[4].as<uint16_t>() : nullptr, tile_samples ? l->pixel_sample.as<uint32_t>() : nullptr);
        tryhip(hipGetLastError(), "pixel table kernel");
        tryhip(hipStreamSynchronize(st), "sync");
    }
    if (rc != YK_OK) {
        l->pixel_xy.release();
        l->pixel_sample.release();
        delete l;
        return rc;
    }
    *out = l;
    return YK_OK;
} YK_CATCH(ctx)

void yk_tile_list_destroy(yk_tile_list* l) {
    if (!l) return;
    if (l->device >= 0) (void)hipSetDevice(l->device);
    l->pixel_xy.release();
    l->pixel_sample.release();
    delete l;
}

yk_status yk_render_tile_list_device(yk_context* ctx, const yk_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                                     const yk_integrator_desc* integrator, const yk_tile_list* list, void* d_out_rgb, void* stream,
                                     yk_render_stats* stats, yk_cancel_fn cancel, void* user) {
    if (!ctx) return YK_ERR_INVALID_ARGUMENT;
    if (!list) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null tile list");
    return render_tiles_impl(ctx, scene, camera, sampler, integrator, list->tiles.data(), nullptr, list->tiles.size(), d_out_rgb, stream, stats, cancel,
                             user, list);
}

yk_status yk_render_tile_list_passes_device(yk_context* ctx, const yk_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                                            const yk_integrator_desc* integrator, const yk_tile_list* list, uint32_t n_passes, void* d_out_rgb, void* stream,
                                            yk_render_stats* stats, yk_cancel_fn cancel, void* user) {
    if (!ctx) return YK_ERR_INVALID_ARGUMENT;
    if (!list) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null tile list");
    if (list->samples.empty()) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "passes need an accumulating tile list (tile_samples)");
    return render_tiles_impl(ctx, scene, camera, sampler, integrator, list->tiles.data(), nullptr, list->tiles.size(), d_out_rgb, stream, stats, cancel,
                             user, list, n_passes);
}

static yk_status film_update_list(yk_context* ctx, const yk_tile_list* list, const void* d_tile_rgb, uint16_t res_x, uint16_t res_y, void* d_film_rgb,
                                  void* stream, int accumulate, uint32_t n_passes);

yk_status yk_film_accumulate_tile_list_passes_device(yk_context* ctx, const yk_tile_list* list, const void* d_passes_rgb, uint32_t n_passes, uint16_t res_x,
                                                     uint16_t res_y, void* d_film_rgb, void* stream) {
    return film_update_list(ctx, list, d_passes_rgb, res_x, res_y, d_film_rgb, stream, 1, n_passes);
}

yk_status yk_film_update_tile_list_device(yk_context* ctx, const yk_tile_list* list, const void* d_tile_rgb, uint16_t res_x, uint16_t res_y,
                                          void* d_film_rgb, void* stream, int accumulate) {
    return film_update_list(ctx, list, d_tile_rgb, res_x, res_y, d_film_rgb, stream, accumulate, 1);
}

static yk_status film_update_list(yk_context* ctx, const yk_tile_list* list, const void* d_tile_rgb, uint16_t res_x, uint16_t res_y, void* d_film_rgb,
                                  void* stream, int accumulate, uint32_t n_passes) {
    if (!ctx) return YK_ERR_INVALID_ARGUMENT;
    YK_LOCK(ctx);
    if (!list || !d_tile_rgb || !d_film_rgb) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null argument");
    if (list->device != ctx->device) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "tile list was not created on this context's device");
    for (const yk_tile& t : list->tiles)
        if (t.x1 > res_x || t.y1 > res_y) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "update_tile: Tile doesn't fit film");
    (void)hipSetDevice(ctx->device);
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    if (n_passes == 0 || n_passes > 0xFFFFu) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "bad number of passes");
    launch_film_scatter(st, list->pixel_xy.as<uint32_t>(), list->off.back(), reinterpret_cast<const float*>(d_tile_rgb), res_x,
                        reinterpret_cast<float*>(d_film_rgb), accumulate ? 1 : 0, n_passes, 3 * (size_t)list->off.back());
    HIP_TRY(ctx, hipGetLastError());
    return YK_OK;  // asynchronous: ordered on `stream`
}

yk_status yk_render_tiles(yk_context* ctx, const yk_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                          const yk_integrator_desc* integrator, const yk_tile* tiles, size_t n_tiles, float* out_rgb, yk_render_stats* stats,
                          yk_cancel_fn cancel, void* user) {
    return render_tiles_host(ctx, scene, camera, sampler, integrator, tiles, nullptr, n_tiles, out_rgb, stats, cancel, user);
}

}  // extern "C"

static yk_status render_tiles_host(yk_context* ctx, const yk_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                                   const yk_integrator_desc* integrator, const yk_tile* tiles, const uint16_t* tile_samples, size_t n_tiles,
                                   float* out_rgb, yk_render_stats* stats, yk_cancel_fn cancel, void* user, uint32_t n_passes) {
    if (!ctx) return YK_ERR_INVALID_ARGUMENT;
    YK_LOCK(ctx);
    if (!tiles || n_tiles == 0 || !out_rgb) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null argument");
    uint64_t total_px = 0;
    for (size_t t = 0; t < n_tiles; ++t) {
        if (tiles[t].x0 >= tiles[t].x1 || tiles[t].y0 >= tiles[t].y1) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "Bounds2 with a dimension <= 0");
        total_px += (uint64_t)(tiles[t].x1 - tiles[t].x0) * (uint64_t)(tiles[t].y1 - tiles[t].y0);
    }
    (void)hipSetDevice(ctx->device);
    if (n_passes == 0 || n_passes > 0xFFFFu) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "bad number of passes");
    HIP_TRY(ctx, ctx->scratch[0].ensure(total_px * 12 * n_passes));
    yk_render_stats local;
    yk_status st = render_tiles_impl(ctx, scene, camera, sampler, integrator, tiles, tile_samples, n_tiles, ctx->scratch[0].p, nullptr,
                                     stats ? stats : &local, cancel, user, nullptr, n_passes);
    if (st != YK_OK) return st;
    HIP_TRY(ctx, hipMemcpy(out_rgb, ctx->scratch[0].p, total_px * 12 * n_passes, hipMemcpyDeviceToHost));
    return YK_OK;
}

extern "C" {

yk_status yk_render_tile(yk_context* ctx, const yk_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                         const yk_integrator_desc* integrator, const yk_tile* tile, float* tile_pixels, uint64_t* out_rays) {
    yk_render_stats stats;
    yk_status st = yk_render_tiles(ctx, scene, camera, sampler, integrator, tile, 1, tile_pixels, &stats, nullptr, nullptr);
    if (st == YK_OK && out_rays) *out_rays = stats.rays;
    return st;
}

}  // extern "C"

static yk_status film_tiles_device(yk_context* ctx, const yk_tile* tiles, size_t n_tiles, const void* d_tile_rgb, uint16_t res_x, uint16_t res_y,
                                   void* d_film_rgb, void* stream, int accumulate);

extern "C" {

yk_status yk_film_update_tiles_device(yk_context* ctx, const yk_tile* tiles, size_t n_tiles, const void* d_tile_rgb, uint16_t res_x, uint16_t res_y,
                                      void* d_film_rgb, void* stream) {
    return film_tiles_device(ctx, tiles, n_tiles, d_tile_rgb, res_x, res_y, d_film_rgb, stream, 0);
}

yk_status yk_film_accumulate_tiles_device(yk_context* ctx, const yk_tile* tiles, size_t n_tiles, const void* d_tile_rgb, uint16_t res_x, uint16_t res_y,
                                          void* d_film_rgb, void* stream) {
    return film_tiles_device(ctx, tiles, n_tiles, d_tile_rgb, res_x, res_y, d_film_rgb, stream, 1);
}

}  // extern "C"

static yk_status film_tiles_device(yk_context* ctx, const yk_tile* tiles, size_t n_tiles, const void* d_tile_rgb, uint16_t res_x, uint16_t res_y,
                                   void* d_film_rgb, void* stream, int accumulate) try {
    if (!ctx) return YK_ERR_INVALID_ARGUMENT;
    YK_LOCK(ctx);
    if (!tiles || !d_tile_rgb || !d_film_rgb || n_tiles == 0) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null argument");
    (void)hipSetDevice(ctx->device);
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    std::vector<uint32_t> off(n_tiles + 1, 0);
    uint64_t total = 0;
    for (size_t t = 0; t < n_tiles; ++t) {
        if (tiles[t].x1 > res_x || tiles[t].y1 > res_y || tiles[t].x0 >= tiles[t].x1 || tiles[t].y0 >= tiles[t].y1)
            return fail(ctx, YK_ERR_INVALID_ARGUMENT, "update_tile: Tile doesn't fit film");
        total += (uint64_t)(tiles[t].x1 - tiles[t].x0) * (uint64_t)(tiles[t].y1 - tiles[t].y0);
        if (total > 0xFFFFFFFFull) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "too many pixels");
        off[t + 1] = (uint32_t)total;
    }
    HIP_TRY(ctx, ctx->scratch[1].ensure(n_tiles * sizeof(yk_tile)));
    HIP_TRY(ctx, ctx->scratch[2].ensure((n_tiles + 1) * 4));
    HIP_TRY(ctx, ctx->scratch[3].ensure(total * 4));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[1].p, tiles, n_tiles * sizeof(yk_tile), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[2].p, off.data(), off.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    launch_pixel_table(st, ctx->scratch[1].as<yk_tile>(), ctx->scratch[2].as<uint32_t>(), (uint32_t)n_tiles, (uint32_t)total, ctx->scratch[3].as<uint32_t>());
    launch_film_scatter(st, ctx->scratch[3].as<uint32_t>(), (uint32_t)total, reinterpret_cast<const float*>(d_tile_rgb), res_x,
                        reinterpret_cast<float*>(d_film_rgb), accumulate);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return YK_OK;
} YK_CATCH(ctx)

extern "C" {

yk_status yk_li(yk_context* ctx, const yk_scene* scene, const yk_sampler_desc* sampler, const yk_integrator_desc* integrator, size_t n,
                const float* ray_o, const float* ray_d, const uint16_t* pixel_xy, const uint32_t* sample_index, uint32_t dimension, float* out_li,
                uint32_t* out_ray_counts) try {
    if (!ctx) return YK_ERR_INVALID_ARGUMENT;
    YK_LOCK(ctx);
    if (!scene || !ray_o || !ray_d || !pixel_xy || !sample_index || !out_li || n == 0) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null argument");
    if (!scene->on_device || scene->device != ctx->device) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "scene was not created on this context's device");
    if (n > ((size_t)1 << 28)) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "too many rays");
    RenderParams prm;
    yk_status ps = make_params(ctx, sampler, integrator, prm);
    if (ps != YK_OK) return ps;
    if (prm.integrator != YK_INTEGRATOR_PATH && prm.integrator != YK_INTEGRATOR_WHITTED)
        return fail(ctx, YK_ERR_UNSUPPORTED, "yk_li implements the Path and Whitted integrators");
    prm.spe = 1;  // one table entry (pixel, sample index) per ray
    for (size_t i = 0; i < n; ++i)
        if (sample_index[i] >= prm.sampler.spp) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "sample_index >= samples per pixel");
    if (prm.max_depth > YK_CTRL_MAX_DEPTH) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "max_depth too large");
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    yk_status wb = ensure_work_buffers(ctx, ctx->ws[0], n, scene->n_lights, scene->n_delta_lights);
    if (wb != YK_OK) return wb;
    if ((wb = ensure_spill(ctx, ctx->ws[0])) != YK_OK) return wb;
    HIP_TRY(ctx, ctx->scratch[4].ensure(n * 12));
    HIP_TRY(ctx, ctx->scratch[5].ensure(n * 12));
    HIP_TRY(ctx, ctx->scratch[6].ensure(n * 4));
    HIP_TRY(ctx, ctx->scratch[7].ensure(n * 4));
    HIP_TRY(ctx, ctx->pixel_xy.ensure(n * 4));
    HIP_TRY(ctx, ctx->sample_buf.ensure(n * 16));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[4].p, ray_o, n * 12, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[5].p, ray_d, n * 12, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[6].p, pixel_xy, n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[7].p, sample_index, n * 4, hipMemcpyHostToDevice, st));
    unsigned* ctrl = ctx->ws[0].ctrl.as<unsigned>();
    unsigned long long* counters = ctx->counters.as<unsigned long long>();
    HIP_TRY(ctx, hipMemsetAsync(counters, 0, YK_COUNTER_BYTES, st));
    HIP_TRY(ctx, hipMemsetAsync(ctrl, 0, YK_CTRL_WORDS * 4, st));
    launch_raygen_user(st, prm, ctx->scratch[4].as<float>(), ctx->scratch[5].as<float>(), ctx->scratch[6].as<uint16_t>(), ctx->scratch[7].as<uint32_t>(),
                       dimension, (uint32_t)n, path_buffers(ctx->ws[0], 0), ctx->sample_buf.as<float4>(), ctx->pixel_xy.as<uint32_t>(), ctrl + YK_CTRL_BOUNCE(0));
    KernelTimer kt;
    kt.ctx = ctx;
    kt.on = false;
    if (prm.integrator == YK_INTEGRATOR_WHITTED)
        launch_whitted(st, trace_grid(ctx), scene->dev, prm, ctx->pixel_xy.as<uint32_t>(), ctx->scratch[7].as<uint32_t>(), path_buffers(ctx->ws[0], 0), (uint32_t)n,
                       ctx->sample_buf.as<float4>(), ctx->ws[0].spill.as<uint2>(), trace_grid(ctx) * trace_block_size(), error_block(ctx), counters);
    else
        run_bounces(ctx, ctx->ws[0], st, scene, prm, ctx->pixel_xy.as<uint32_t>(), ctx->scratch[7].as<uint32_t>(), ctx->sample_buf.as<float4>(), kt, counters, false, (uint32_t)n, 0u, false);
    HIP_TRY(ctx, hipGetLastError());
    std::vector<float> tmp(n * 4);
    unsigned host_err[4];
    HIP_TRY(ctx, hipMemcpyAsync(tmp.data(), ctx->sample_buf.p, n * 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(host_err, error_block(ctx), sizeof(host_err), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (host_err[YK_CTRL_ERR] & 1u) return fail(ctx, YK_ERR_STACK_OVERFLOW, "BVH traversal stack exceeded 64 entries (bvh.rs:174)");
    for (size_t i = 0; i < n; ++i) {
        out_li[3 * i] = tmp[4 * i];
        out_li[3 * i + 1] = tmp[4 * i + 1];
        out_li[3 * i + 2] = tmp[4 * i + 2];
    }
    if (out_ray_counts) std::memset(out_ray_counts, 0, n * 4);  // per-ray counts are not tracked by the wavefront
    return YK_OK;
} YK_CATCH(ctx)

// ------------------------------------------------------------------ per-stage entry points
yk_status yk_trace_closest(yk_context* ctx, const yk_scene* scene, size_t n, const float* ray_o, const float* ray_d, const float* t_max,
                           int32_t* out_shape, float* out_t, float* out_bary, uint32_t* out_node_tests, uint32_t* out_node_hits,
                           uint32_t* out_shape_tests) try {
    if (!ctx) return YK_ERR_INVALID_ARGUMENT;
    YK_LOCK(ctx);
    if (!scene || !ray_o || !ray_d || !out_shape || n == 0) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null argument");
    if (!scene->on_device || scene->device != ctx->device) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "scene was not created on this context's device");
    if (n > 0xFFFFFF00ull) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "too many rays");
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    yk_status wb = ensure_work_buffers(ctx, ctx->ws[0], n, scene->n_lights, scene->n_delta_lights);
    if (wb != YK_OK) return wb;
    if ((wb = ensure_spill(ctx, ctx->ws[0])) != YK_OK) return wb;
    const bool want_stats = out_node_tests || out_node_hits || out_shape_tests;
    HIP_TRY(ctx, ctx->scratch[4].ensure(n * 12));
    HIP_TRY(ctx, ctx->scratch[5].ensure(n * 12));
    HIP_TRY(ctx, ctx->scratch[6].ensure(n * 4));
    HIP_TRY(ctx, ctx->hit4.ensure(n * 16));
    if (want_stats) HIP_TRY(ctx, ctx->stats4.ensure(n * 16));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[4].p, ray_o, n * 12, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[5].p, ray_d, n * 12, hipMemcpyHostToDevice, st));
    if (t_max) HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[6].p, t_max, n * 4, hipMemcpyHostToDevice, st));
    PathBuffers pb = path_buffers(ctx->ws[0], 0);
    launch_pack_rays(st, n, ctx->scratch[4].as<float>(), ctx->scratch[5].as<float>(), pb.rayO, pb.rayD);
    unsigned* ctrl = ctx->ws[0].ctrl.as<unsigned>();
    HIP_TRY(ctx, hipMemsetAsync(ctrl, 0, YK_CTRL_WORDS * 4, st));
    unsigned nn = (unsigned)n;
    HIP_TRY(ctx, hipMemcpyAsync(ctrl, &nn, 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    launch_trace_closest(st, trace_grid(ctx), dev_scene_for(scene, n), pb.rayO, pb.rayD, t_max ? ctx->scratch[6].as<float>() : nullptr, ctrl, ctrl + YK_CTRL_HEADS,
                         ctx->ws[0].hit.as<int>(), ctx->hit4.as<float4>(), want_stats ? ctx->stats4.as<uint4>() : nullptr, ctx->ws[0].spill.as<uint2>(),
                         trace_grid(ctx) * trace_block_size(), ctrl, nullptr);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out_shape, ctx->ws[0].hit.p, n * 4, hipMemcpyDeviceToHost, st));
    std::vector<float> h4;
    if (out_t || out_bary) {
        h4.resize(n * 4);
        HIP_TRY(ctx, hipMemcpyAsync(h4.data(), ctx->hit4.p, n * 16, hipMemcpyDeviceToHost, st));
    }
    std::vector<uint32_t> s4;
    if (want_stats) {
        s4.resize(n * 4);
        HIP_TRY(ctx, hipMemcpyAsync(s4.data(), ctx->stats4.p, n * 16, hipMemcpyDeviceToHost, st));
    }
    unsigned host_ctrl[4];
    HIP_TRY(ctx, hipMemcpyAsync(host_ctrl, ctrl, 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    for (size_t i = 0; i < n; ++i) {
        if (out_t) out_t[i] = out_shape[i] >= 0 ? h4[4 * i] : __builtin_inff();
        if (out_bary) {
            out_bary[3 * i] = out_shape[i] >= 0 ? h4[4 * i + 1] : 0.0f;
            out_bary[3 * i + 1] = out_shape[i] >= 0 ? h4[4 * i + 2] : 0.0f;
            out_bary[3 * i + 2] = out_shape[i] >= 0 ? h4[4 * i + 3] : 0.0f;
        }
        if (out_node_tests) out_node_tests[i] = s4[4 * i];
        if (out_node_hits) out_node_hits[i] = s4[4 * i + 1];
        if (out_shape_tests) out_shape_tests[i] = s4[4 * i + 2];
    }
    if (host_ctrl[YK_CTRL_ERR] & 1u) return fail(ctx, YK_ERR_STACK_OVERFLOW, "BVH traversal stack exceeded 64 entries (bvh.rs:174)");
    return YK_OK;
} YK_CATCH(ctx)

yk_status yk_trace_any(yk_context* ctx, const yk_scene* scene, size_t n, const float* ray_o, const float* ray_d, const float* t_max,
                       const int32_t* area_light, uint8_t* out_hit) {
    if (!ctx) return YK_ERR_INVALID_ARGUMENT;
    YK_LOCK(ctx);
    if (!scene || !ray_o || !ray_d || !t_max || !out_hit || n == 0) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null argument");
    if (!scene->on_device || scene->device != ctx->device) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "scene was not created on this context's device");
    if (n > 0xFFFFFF00ull) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "too many rays");
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    yk_status wb = ensure_work_buffers(ctx, ctx->ws[0], n, scene->n_lights, scene->n_delta_lights);
    if (wb != YK_OK) return wb;
    if ((wb = ensure_spill(ctx, ctx->ws[0])) != YK_OK) return wb;
    HIP_TRY(ctx, ctx->scratch[4].ensure(n * 12));
    HIP_TRY(ctx, ctx->scratch[5].ensure(n * 12));
    HIP_TRY(ctx, ctx->scratch[6].ensure(n * 4));
    HIP_TRY(ctx, ctx->scratch[7].ensure(n * 4));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[4].p, ray_o, n * 12, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[5].p, ray_d, n * 12, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[6].p, t_max, n * 4, hipMemcpyHostToDevice, st));
    if (area_light) HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[7].p, area_light, n * 4, hipMemcpyHostToDevice, st));
    launch_pack_shadow_rays(st, n, ctx->scratch[4].as<float>(), ctx->scratch[5].as<float>(), ctx->scratch[6].as<float>(),
                            area_light ? ctx->scratch[7].as<int>() : nullptr, ctx->ws[0].shO.as<float4>(), ctx->ws[0].shD.as<float4>());
    unsigned* ctrl = ctx->ws[0].ctrl.as<unsigned>();
    HIP_TRY(ctx, hipMemsetAsync(ctrl, 0, YK_CTRL_WORDS * 4, st));
    unsigned nn = (unsigned)n;
    HIP_TRY(ctx, hipMemcpyAsync(ctrl, &nn, 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    launch_trace_any(st, trace_grid(ctx), dev_scene_for(scene, n), ctx->ws[0].shO.as<float4>(), ctx->ws[0].shD.as<float4>(), nullptr, ctrl, ctrl + YK_CTRL_HEADS,
                     ctx->ws[0].vis.as<unsigned char>(), ctx->ws[0].spill.as<uint2>(), trace_grid(ctx) * trace_block_size(), ctrl, nullptr);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out_hit, ctx->ws[0].vis.p, n, hipMemcpyDeviceToHost, st));
    unsigned host_ctrl[4];
    HIP_TRY(ctx, hipMemcpyAsync(host_ctrl, ctrl, 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (host_ctrl[YK_CTRL_ERR] & 1u) return fail(ctx, YK_ERR_STACK_OVERFLOW, "BVH traversal stack exceeded 64 entries (bvh.rs:174)");
    return YK_OK;
}

yk_status yk_sampler_sequence(yk_context* ctx, const yk_sampler_desc* sampler, uint16_t px, uint16_t py, uint32_t sample_index, const uint8_t* dims,
                              size_t n_draws, float* out) {
    if (!ctx) return YK_ERR_INVALID_ARGUMENT;
    YK_LOCK(ctx);
    if (!sampler || !dims || !out || n_draws == 0) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null argument");
    RenderParams prm;
    yk_integrator_desc dummy = {YK_INTEGRATOR_PATH, 1, 0, 0.0f};
    yk_status ps = make_params(ctx, sampler, &dummy, prm);
    if (ps != YK_OK) return ps;
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, ctx->scratch[4].ensure(n_draws));
    HIP_TRY(ctx, ctx->scratch[5].ensure(n_draws * 8));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[4].p, dims, n_draws, hipMemcpyHostToDevice, st));
    launch_sampler_sequence(st, prm.sampler, px, py, sample_index, ctx->scratch[4].as<uint8_t>(), n_draws, ctx->scratch[5].as<float>());
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->scratch[5].p, n_draws * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return YK_OK;
}

yk_status yk_camera_rays(yk_context* ctx, const yk_camera* camera, const yk_sampler_desc* sampler, const yk_tile* tile, uint32_t sample_index,
                         float* out_o, float* out_d) try {
    if (!ctx) return YK_ERR_INVALID_ARGUMENT;
    YK_LOCK(ctx);
    if (!camera || !sampler || !tile || !out_o || !out_d) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null argument");
    if (tile->x0 >= tile->x1 || tile->y0 >= tile->y1) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "Bounds2 with a dimension <= 0");
    RenderParams prm;
    yk_integrator_desc dummy = {YK_INTEGRATOR_PATH, 1, 0, 0.0f};
    yk_status ps = make_params(ctx, sampler, &dummy, prm);
    if (ps != YK_OK) return ps;
    if (sample_index >= prm.sampler.spp) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "sample_index >= samples per pixel");
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    const uint32_t npx = (uint32_t)(tile->x1 - tile->x0) * (uint32_t)(tile->y1 - tile->y0);
    const uint32_t spp = prm.sampler.spp;
    yk_status wb = ensure_work_buffers(ctx, ctx->ws[0], (size_t)npx * spp, 1, 0);
    if (wb != YK_OK) return wb;
    uint32_t off[2] = {0, npx};
    HIP_TRY(ctx, ctx->tiles.ensure(sizeof(yk_tile)));
    HIP_TRY(ctx, ctx->tile_off.ensure(8));
    HIP_TRY(ctx, ctx->pixel_xy.ensure((size_t)npx * 4));
    HIP_TRY(ctx, ctx->sample_buf.ensure((size_t)npx * spp * 16));
    HIP_TRY(ctx, ctx->scratch[4].ensure((size_t)npx * spp * 12));
    HIP_TRY(ctx, ctx->scratch[5].ensure((size_t)npx * spp * 12));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->tiles.p, tile, sizeof(yk_tile), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->tile_off.p, off, 8, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    launch_pixel_table(st, ctx->tiles.as<yk_tile>(), ctx->tile_off.as<uint32_t>(), 1, npx, ctx->pixel_xy.as<uint32_t>());
    DevCamera cam;
    std::memcpy(cam.c2w, camera->camera_to_world, 64);
    std::memcpy(cam.r2c, camera->raster_to_camera, 64);
    PathBuffers pb = path_buffers(ctx->ws[0], 0);
    launch_raygen(st, cam, prm, ctx->pixel_xy.as<uint32_t>(), nullptr, 0, npx * spp, pb, ctx->sample_buf.as<float4>(), ctx->ws[0].ctrl.as<unsigned>());
    launch_unpack_rays(st, (size_t)npx * spp, pb.rayO, pb.rayD, ctx->scratch[4].as<float>(), ctx->scratch[5].as<float>());
    HIP_TRY(ctx, hipGetLastError());
    std::vector<float> o((size_t)npx * spp * 3), d((size_t)npx * spp * 3);
    HIP_TRY(ctx, hipMemcpyAsync(o.data(), ctx->scratch[4].p, o.size() * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(d.data(), ctx->scratch[5].p, d.size() * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    for (uint32_t p = 0; p < npx; ++p)
        for (int k = 0; k < 3; ++k) {
            out_o[3 * p + k] = o[3 * ((size_t)p * spp + sample_index) + k];
            out_d[3 * p + k] = d[3 * ((size_t)p * spp + sample_index) + k];
        }
    return YK_OK;
} YK_CATCH(ctx)

yk_status yk_device_math(yk_context* ctx, int fn, size_t n, const float* a, const float* b, float* out) {
    if (!ctx) return YK_ERR_INVALID_ARGUMENT;
    YK_LOCK(ctx);
    if (!a || !out || n == 0) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null argument");
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, ctx->scratch[4].ensure(n * 4));
    HIP_TRY(ctx, ctx->scratch[5].ensure(n * 4));
    HIP_TRY(ctx, ctx->scratch[6].ensure(n * 4));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[4].p, a, n * 4, hipMemcpyHostToDevice, st));
    if (b) HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[5].p, b, n * 4, hipMemcpyHostToDevice, st));
    launch_device_math(st, fn, n, ctx->scratch[4].as<float>(), b ? ctx->scratch[5].as<float>() : nullptr, ctx->scratch[6].as<float>());
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->scratch[6].p, n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return YK_OK;
}

static yk_status bsdf_common(yk_context* ctx, const yk_material_desc* material, size_t n, const float* n_geom, const float* n_shading,
                             const float* dpdu, const float* wo, const float* x, int sample, float* out) {
    if (!ctx) return YK_ERR_INVALID_ARGUMENT;
    YK_LOCK(ctx);
    if (!material || !n_geom || !n_shading || !dpdu || !wo || !x || !out || n == 0) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null argument");
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    const float* src[5] = {n_geom, n_shading, dpdu, wo, x};
    size_t each[5] = {3, 3, 3, 3, (size_t)(sample ? 2 : 3)};
    for (int k = 0; k < 5; ++k) {
        HIP_TRY(ctx, ctx->scratch[k].ensure(n * each[k] * 4));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[k].p, src[k], n * each[k] * 4, hipMemcpyHostToDevice, st));
    }
    const size_t out_each = sample ? 8 : 3;
    HIP_TRY(ctx, ctx->scratch[5].ensure(n * out_each * 4));
    launch_bsdf_test(st, make_material(*material), n, ctx->scratch[0].as<float>(), ctx->scratch[1].as<float>(), ctx->scratch[2].as<float>(),
                     ctx->scratch[3].as<float>(), ctx->scratch[4].as<float>(), sample, ctx->scratch[5].as<float>());
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->scratch[5].p, n * out_each * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return YK_OK;
}

yk_status yk_bsdf_eval(yk_context* ctx, const yk_material_desc* material, size_t n, const float* n_geom, const float* n_shading, const float* dpdu,
                       const float* wo, const float* wi, float* out_f) {
    return bsdf_common(ctx, material, n, n_geom, n_shading, dpdu, wo, wi, 0, out_f);
}
yk_status yk_bsdf_sample(yk_context* ctx, const yk_material_desc* material, size_t n, const float* n_geom, const float* n_shading, const float* dpdu,
                         const float* wo, const float* u, float* out8) {
    return bsdf_common(ctx, material, n, n_geom, n_shading, dpdu, wo, u, 1, out8);
}

yk_status yk_light_sample(yk_context* ctx, const yk_light_desc* light, int32_t light_index, size_t n, const float* p, const float* n_geom,
                          const float* u, float* out18) {
    if (!ctx) return YK_ERR_INVALID_ARGUMENT;
    YK_LOCK(ctx);
    if (!light || !p || !n_geom || !u || !out18 || n == 0) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null argument");
    if (light->kind > YK_LIGHT_RECT) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "unknown light kind");
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    const float* src[3] = {p, n_geom, u};
    const size_t each[3] = {3, 3, 2};
    for (int k = 0; k < 3; ++k) {
        HIP_TRY(ctx, ctx->scratch[k].ensure(n * each[k] * 4));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[k].p, src[k], n * each[k] * 4, hipMemcpyHostToDevice, st));
    }
    HIP_TRY(ctx, ctx->scratch[5].ensure(n * 18 * 4));
    launch_light_test(st, make_light(*light), light_index, n, ctx->scratch[0].as<float>(), ctx->scratch[1].as<float>(), ctx->scratch[2].as<float>(),
                      ctx->scratch[5].as<float>());
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out18, ctx->scratch[5].p, n * 18 * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return YK_OK;
}

size_t yk_sizeof(int what) {
    switch (what) {
        case 0: return sizeof(yk_scene_desc);
        case 1: return sizeof(yk_material_desc);
        case 2: return sizeof(yk_light_desc);
        case 3: return sizeof(yk_sphere_desc);
        case 4: return sizeof(yk_camera);
        case 5: return sizeof(yk_camera_params);
        case 6: return sizeof(yk_sampler_desc);
        case 7: return sizeof(yk_integrator_desc);
        case 8: return sizeof(yk_tile);
        case 9: return sizeof(yk_bvh_node);
        case 10: return sizeof(yk_mesh_desc);
        case 11: return sizeof(yk_render_stats);
        case 12: return sizeof(yk_scene_info);
        default: return 0;
    }
}

}  // extern "C"
