// yk_context.cpp — the C ABI (include/yuki_hip.h): contexts and their options, and the host-only helpers
// (camera, film tiles, light constructors, the host-side Film::update_tile).
//
// Part of what used to be one file (yk_api.cpp); the others are yk_scene.cpp (scene description -> BVH ->
// device records), yk_render.cpp (the batch scheduler that drives the wavefront kernels) and yk_stages.cpp
// (per-stage entry points for the parity tests).
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
        hipEventCreateWithFlags(&ctx->ws[1].ev_acc, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ws[0].ev_batch, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ws[1].ev_batch, hipEventDisableTiming) != hipSuccess) {
        delete ctx;
        return YK_ERR_DEVICE;
    }
    ctx->ws[0].stream = ctx->stream;
    // the interruption word the kernels poll (yk_device.h, CancelRef): pinned, mapped, coherent host memory.  Without it
    // (allocation refused) renders still work and are interruptible between batches only.
    void* host_word = nullptr;
    if (hipHostMalloc(&host_word, 128, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
        void* dev_word = nullptr;
        if (hipHostGetDevicePointer(&dev_word, host_word, 0) == hipSuccess) {
            ctx->cancel_host = static_cast<unsigned*>(host_word);
            ctx->cancel_host_dev = static_cast<const unsigned*>(dev_word);
            std::memset(host_word, 0, 128);
            ctx->cancel_host[16] = 1u;  // (the stream that carries it is made when the first interruption needs it: a context holds no idle streams, see above)
        } else {
            (void)hipHostFree(host_word);
        }
    } else {
        (void)hipGetLastError();
    }
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
        if (w.ev_batch) (void)hipEventDestroy(w.ev_batch);
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
    if (ctx->cancel_stream) {
        (void)hipStreamSynchronize(ctx->cancel_stream);
        (void)hipStreamDestroy(ctx->cancel_stream);
    }
    if (ctx->cancel_host) (void)hipHostFree(ctx->cancel_host);
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
