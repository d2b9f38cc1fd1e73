// yk_stages.cpp — the per-stage entry points of the C ABI: the device functions the kernels call, one stage at a
// time, for the stage-level parity tests (tests/test_gpu_stages.py) and profiling.
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

yk_status yk_host_math(int fn, size_t n, const float* a, const float* b, float* out) {
    if (!a || !out) return YK_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < n; ++i) {
        const float x = a[i], y = b ? b[i] : 0.0f;
        float sn, cs;
        switch (fn) {
            case 0: out[i] = det_sinf(x); break;
            case 1: out[i] = det_cosf(x); break;
            case 2: out[i] = det_tanf(x); break;
            case 3: out[i] = det_logf(x); break;
            case 4: out[i] = det_acosf(x); break;
            case 5: out[i] = det_atan2f(x, y); break;
            case 28: det_sincosf(x, sn, cs); out[i] = sn; break;
            case 29: det_sincosf(x, sn, cs); out[i] = cs; break;
            case 30: out[i] = det_expf(x); break;
            default: return YK_ERR_INVALID_ARGUMENT;
        }
    }
    return YK_OK;
}

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

}  // extern "C"
