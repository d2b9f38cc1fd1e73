// yk_internal.h — library-private definitions shared by the host files of the library (yk_context.cpp,
// yk_scene.cpp, yk_render.cpp, yk_stages.cpp: the single-device entry points; yk_multi.cpp: several devices of
// one process, RCCL): the objects behind the opaque handles of include/yuki_hip.h.  Not installed.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "yk_device.h"
#include "yk_host.h"
#include "yk_kernels.h"

using namespace yk;

// ------------------------------------------------------------------ helpers
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t want) {
        if (want <= bytes) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) bytes = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct yk_context {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string last_error;
    int n_cu = 256;
    // options
    int64_t batch_paths = 128 << 20;
    int64_t packet_bounces = 1;         // leading bounces whose closest-hit rays use the wave-packet kernel (camera rays are coherent); 0 = never
    int64_t packet_shadow_bounces = 1;  // same for the shadow rays towards point / spot / distant lights (their own queue)
    int64_t shade_reorder = 1;   // deal the paths of a shade block to its lanes sorted by material kind (bounces > 0)
    int64_t overlap_shadow = 1;  // run {trace_any, accumulate}(b) on a side stream beside trace_closest(b+1)
    int64_t wide_bvh = 2;   // scenes created afterwards: 0 binary nodes only, 1 traverse the 4-wide collapse, 2 keep both and pick per job
    int64_t top_nodes = YK_TOP_MAX; // interior nodes (capped by what the kernels were built for) of the first tree levels the traversal kernels keep in LDS
    int64_t sample_buf_cap = (int64_t)64 << 30;
    int64_t time_kernels = 1;
    int64_t streams = 2;  // batches in flight (1 or 2): the second stream's launches fill the first one's tails
    // per-stream work buffers
    struct WorkSet {
        DevBuf path[2][4];
        DevBuf hit, pend, shO, shD, shC, vis, shq, shO2, shD2, shq2, ctrl, spill, spill_side;
        size_t cap_paths = 0;
        unsigned cap_lights = 0, cap_area = 0, cap_delta = 0;
        hipStream_t stream = nullptr;
        hipStream_t side = nullptr;  // shadow rays + accumulate of bounce b run here beside trace of bounce b+1
        hipEvent_t done = nullptr, ev_shade = nullptr, ev_acc = nullptr;
        hipEvent_t ev_batch = nullptr;  // end of the work set's latest batch (pacing of interruptible jobs, yk_render.cpp)
    } ws[2];
    DevBuf sample_buf, pixel_xy, pixel_aux, tiles, tile_off, counters, stats4, hit4, scratch[8];
    std::vector<hipEvent_t> ev_pool;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;  // hand-over between a caller's stream and the context's own
    // Interruption (yk_device.h, CancelRef): cancel_host[0] is the word the kernels poll across PCIe (pinned, mapped, coherent host
    // memory; cancel_host_dev is its device address); cancel_raised remembers that a submission left it set — the next one waits
    // for the context's streams before it clears the word, so that no kernel of the interrupted submission resumes.
    unsigned* cancel_host = nullptr;  // [0] the word, [16] a constant 1: the source of the host's copy into the device word
    const unsigned* cancel_host_dev = nullptr;
    hipStream_t cancel_stream = nullptr;  // carries that copy past the kernels in flight (made after the context's first interruption)
    bool want_cancel_stream = false;
    std::atomic<bool> cancel_raised{false};
    // every entry point that touches the context's buffers or streams holds this: calls on one
    // context from several host threads (the reference's tile workers) are serialised
    std::recursive_mutex mu;
};

typedef yk_context::WorkSet WorkSet;
static const uint32_t YK_WIDE_MAX_PATHS = 6u << 20;  // jobs up to this many paths traverse the 4-wide nodes (wide_bvh = 2)

struct yk_scene {
    int device = -1;  // a scene belongs to the device, not to the context that made it: any context there renders it, and it may outlive them
    std::shared_ptr<const HostBvh> bvh;  // one host tree may serve the copies of a scene on several devices (yk_multi_scene)
    uint32_t n_triangles = 0, n_spheres = 0, n_lights = 0, n_delta_lights = 0;
    bool wide_auto = false;  // both node layouts on the device: the 4-wide one is used for jobs below YK_WIDE_MAX_PATHS
    yk_scene_info info;
    // device
    DevBuf nodes, nodes4, top_nodes, top_nodes_any, tris, prim_shade, prim_attr, indices, points, normals, uvs, tri_mesh, tri_material, tri_area_light, mesh_flags, materials, lights, spheres, texels, tex_info;
    DevScene dev;
    bool on_device = false;
};

// The device scene a job of `n` rays traverses: with both node layouts present the 4-wide one
// serves small jobs only (see run_bounces).
static inline DevScene dev_scene_for(const yk_scene* scene, uint64_t n) {
    DevScene ds = scene->dev;
    if (scene->wide_auto && n > YK_WIDE_MAX_PATHS) ds.nodes4 = nullptr;
    return ds;
}

// A tile list prepared once and reused every frame (the GPU worker renders the same tiles
// over and over): host copy + the device pixel table, so that rendering and the film update
// need no upload and no host synchronisation.
struct yk_tile_list {
    int device = -1;
    std::vector<yk_tile> tiles;
    std::vector<uint16_t> samples;  // empty: plain film
    std::vector<uint32_t> off;      // n_tiles + 1 pixel offsets
    DevBuf pixel_xy, pixel_sample;
};

static inline yk_status fail(yk_context* ctx, yk_status st, const std::string& msg) {
    if (ctx) ctx->last_error = msg;
    return st;
}

#define YK_LOCK(ctx) std::lock_guard<std::recursive_mutex> yk_lock_((ctx)->mu)
#define HIP_TRY(ctx, expr)                                                                                           \
    do {                                                                                                             \
        hipError_t _e = (expr);                                                                                      \
        if (_e != hipSuccess) {                                                                                      \
            return fail(ctx, _e == hipErrorOutOfMemory ? YK_ERR_OUT_OF_MEMORY : YK_ERR_DEVICE,                       \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                                          \
        }                                                                                                            \
    } while (0)

// No exception crosses the C ABI (undefined behaviour for a Rust caller, an abort under ctypes): entry points that
// allocate host memory are function-try-blocks ending in this handler.
#define YK_CATCH(ctx)                                                                                         \
    catch (const std::bad_alloc&) { return fail(ctx, YK_ERR_OUT_OF_MEMORY, "host allocation failed"); }      \
    catch (const std::exception& e) { return fail(ctx, YK_ERR_INVALID_ARGUMENT, std::string("exception: ") + e.what()); }


// Everything yk_scene_create derives from a scene description on the host — the reference's BVH
// (BoundingVolumeHierarchy::new, bvh.rs:39-115) and the device records laid out from it.  Built once;
// uploaded to one device (yk_scene_create) or to every device of a yk_multi (yk_multi_scene_create).
struct SceneImage;
yk_status yk_build_scene_image(yk_context* opt_ctx, const yk_scene_desc* d, std::shared_ptr<SceneImage>& out);
yk_status yk_upload_scene_image(yk_context* ctx, const std::shared_ptr<SceneImage>& img, yk_scene** out);

static inline double now_seconds() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ------------------------------------------------------------------ yk_scene.cpp
Material make_material(const yk_material_desc& m);  // per-hit constants folded (GGX alpha, Oren-Nayar A / B)
DevLight make_light(const yk_light_desc& l);

// ------------------------------------------------------------------ yk_render.cpp (used by yk_stages.cpp too)
// ctx->counters: 8 x u64 (closest-hit rays, shadow rays, ...) followed by a 4-word error block whose word
// YK_CTRL_ERR the traversal kernels set on a stack overflow and whose word YK_CTRL_CANCELLED says that the render was interrupted.
// Both are zeroed ONCE per call — the per-batch control blocks of the work sets are zeroed with every batch and must not hold the flags.
// The error block has a 128-byte line of its own: its word YK_CTRL_CANCELLED is read by every kernel that starts, the counters
// before it take an atomic per wave.
#define YK_COUNTER_BYTES 256
unsigned* error_block(yk_context* ctx);
CancelRef cancel_ref(yk_context* ctx);  // the context's interruption words, as the kernels take them
yk_status ensure_work_buffers(yk_context* ctx, WorkSet& ws, size_t paths, unsigned n_lights, unsigned n_delta_lights);
yk_status ensure_spill(yk_context* ctx, WorkSet& ws);
unsigned trace_grid(const yk_context* ctx);
PathBuffers path_buffers(WorkSet& ws, int which);
yk_status make_params(yk_context* ctx, const yk_sampler_desc* smp, const yk_integrator_desc* integ, RenderParams& prm);

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

bool packet_kernel_traces_bounce(const yk_context* ctx, const yk_scene* scene, unsigned b);
// one batch of `n_paths` paths already generated into buffer 0; runs the bounce loop
void run_bounces(yk_context* ctx, WorkSet& ws, hipStream_t st, const yk_scene* scene, const RenderParams& prm, const uint32_t* pixel_xy,
                 const uint32_t* sample_index_tab, float4* sample_buf, KernelTimer& kt, unsigned long long* counters, bool coherent,
                 uint32_t n_paths, uint32_t sid_base, bool lean_camera_bounce, uint32_t* n_shadow_launches = nullptr);
