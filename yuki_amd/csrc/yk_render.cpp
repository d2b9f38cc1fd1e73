// yk_render.cpp — the batch scheduler that drives the wavefront kernels, behind the C ABI's render entry points.
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
#include <thread>
#include <vector>

#include "yk_internal.h"
#ifdef YK_EXPERIMENT_SORT  // timing builds only (tools/build_variant.sh sort -DYK_EXPERIMENT_SORT): DESIGN.md §9
#include "../../tools/micro/ray_sort_experiment.h"
#endif
#ifdef YK_EXPERIMENT_GRAPH
#include "../../tools/micro/graph_replay_experiment.h"
#endif

// ------------------------------------------------------------------ render
// ctx->counters: 8 x u64 (closest-hit rays, shadow rays, ...) followed by a 4-word error block whose word
// YK_CTRL_ERR the traversal kernels set on a stack overflow.  Both are zeroed ONCE per call (begin_call) — the
// per-batch control blocks of the work sets are zeroed with every batch and must not hold the flag.
unsigned* error_block(yk_context* ctx) { return reinterpret_cast<unsigned*>(ctx->counters.as<unsigned long long>() + 16); }
CancelRef cancel_ref(yk_context* ctx) { return CancelRef{ctx->cancel_host_dev, ctx->cancel_host_dev ? error_block(ctx) + YK_CTRL_CANCELLED : nullptr}; }

yk_status ensure_work_buffers(yk_context* ctx, WorkSet& ws, size_t paths, unsigned n_lights, unsigned n_delta_lights) {
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

unsigned trace_grid(const yk_context* ctx) { return (unsigned)ctx->n_cu * trace_blocks_per_cu(); }

yk_status ensure_spill(yk_context* ctx, WorkSet& ws) {
    size_t threads = (size_t)trace_grid(ctx) * trace_block_size();
    HIP_TRY(ctx, ws.spill.ensure(threads * trace_spill_depth() * 8));
    HIP_TRY(ctx, ws.spill_side.ensure(threads * trace_spill_depth() * 8));
    return YK_OK;
}

PathBuffers path_buffers(WorkSet& ws, int which) {
    PathBuffers p;
    p.rayO = ws.path[which][0].as<float4>();
    p.rayD = ws.path[which][1].as<float4>();
    p.thru = ws.path[which][2].as<float4>();
    p.rngs = ws.path[which][3].as<uint4>();
    return p;
}

yk_status make_params(yk_context* ctx, const yk_sampler_desc* smp, const yk_integrator_desc* integ, RenderParams& prm) {
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


// camera rays (and the leading bounces the "packet_bounces" option names) go to the wave-packet kernel
bool packet_kernel_traces_bounce(const yk_context* ctx, const yk_scene* scene, unsigned b) {
    return b < (unsigned)ctx->packet_bounces && scene->bvh->depth <= 64;
}

// one batch of `n` paths already generated into buffer 0; runs the bounce loop
void run_bounces(yk_context* ctx, WorkSet& ws, hipStream_t st, const yk_scene* scene, const RenderParams& prm, const uint32_t* pixel_xy,
                        const uint32_t* sample_index_tab, float4* sample_buf, KernelTimer& kt, unsigned long long* counters, bool coherent,
                        uint32_t n_paths, uint32_t sid_base, bool lean_camera_bounce, uint32_t* n_shadow_launches) {
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
            launch_trace_closest_packet(st, pg, ds, lean_origin ? nullptr : pc.rayO, pc.rayD, bc, bc + YK_CTRL_HEAD, ws.hit.as<int>(), counters, lean_origin, prm.cancel);
        else {
            unsigned* head = bc + YK_CTRL_HEAD;
#ifdef YK_EXPERIMENT_XCD
            head = yk_exp::sorter().heads(st, 0);
#endif
            launch_trace_closest(st, tg, ds, pc.rayO, pc.rayD, nullptr, bc, head, ws.hit.as<int>(), nullptr, nullptr,
                                 ws.spill.as<uint2>(), spill_stride, errblk, counters, prm.cancel.host);
        }
        kt.end(e, 0, st);
        if (overlap && b > 0) (void)hipStreamWaitEvent(st, ws.ev_acc, 0);
        e = kt.begin(st);
        launch_shade(st, sg, ds, prm, pixel_xy, sample_index_tab, pc, pn, ws.hit.as<int>(), ws.pend.as<float4>(), ws.shO.as<float4>(),
                     ws.shD.as<float4>(), ws.shC.as<float4>(), ws.vis.as<unsigned char>(), ws.shq.as<unsigned>(), ws.shO2.as<float4>(),
                     ws.shD2.as<float4>(), ws.shq2.as<unsigned>(), bc, split ? 1u : 0u, (b > 0 && ctx->shade_reorder) ? 1u : 0u, 3u * (unsigned)ctx->n_cu, sid_base, lean_origin);
        kt.end(e, 2, st);
#ifdef YK_EXPERIMENT_SORT  // order the queues k_shade just wrote (synchronises: timing experiment, tools/micro/ray_sort_experiment.h)
        YK_SORT_AFTER_SHADE
#endif
        if (overlap) {
            (void)hipEventRecord(ws.ev_shade, st);
            (void)hipStreamWaitEvent(sb, ws.ev_shade, 0);
        }
        e = kt.begin(sb);
        uint2* any_spill = (overlap ? ws.spill_side : ws.spill).as<uint2>();
        if (all_delta) {
            launch_trace_any_packet(sb, pg_any, ds, ws.shO.as<float4>(), ws.shD.as<float4>(), ws.shq.as<unsigned>(), bc + YK_CTRL_SHQ,
                                    bc + YK_CTRL_HEAD + 1, ws.vis.as<unsigned char>(), counters + 1, prm.cancel);
        } else {
            unsigned* any_head = bc + YK_CTRL_HEAD + 1;
#ifdef YK_EXPERIMENT_XCD
            any_head = yk_exp::sorter().heads(sb, 1);
#endif
            launch_trace_any(sb, tg_any, ds, ws.shO.as<float4>(), ws.shD.as<float4>(), ws.shq.as<unsigned>(), bc + YK_CTRL_SHQ,
                             any_head, ws.vis.as<unsigned char>(), any_spill, spill_stride, errblk, counters + 1, prm.cancel.host);
            if (split && n_shadow_launches) ++*n_shadow_launches;
            if (split)  // rays converging on a point / spot / distant light: wave packets
                launch_trace_any_packet(sb, pg_any, ds, ws.shO2.as<float4>(), ws.shD2.as<float4>(), ws.shq2.as<unsigned>(), bc + YK_CTRL_SHQ2,
                                        bc + YK_CTRL_HEAD + 2, ws.vis.as<unsigned char>(), counters + 1, prm.cancel);
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

// ------------------------------------------------------------------ interruption (yk_device.h, CancelRef)
// Raise the host's word: every traversal launch's relay wave sees it at its next claim (yk_device.h, CancelRef) and raises the
// device word that every later launch reads when it starts.
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static const bool g_debug_cancel = std::getenv("YK_DEBUG_CANCEL") != nullptr;

static void raise_cancel(yk_context* ctx, bool from_render_thread = true) {
    if (!ctx->cancel_host) return;
    const double t0 = g_debug_cancel ? now_ms() : 0.0;
    ctx->cancel_raised.store(true, std::memory_order_release);
    __atomic_store_n(ctx->cancel_host, 1u, __ATOMIC_RELEASE);
    // ... and the device word directly (a copy engine's job: it does not queue behind the kernels): k_shade looks at this one only.
    // The stream that carries the copy is NOT made here — the first interruption of a context relies on the relay wave alone (the
    // device word is up one traversal launch later at most) and leaves a note; the stream is created when the next submission
    // clears the word (clear_cancel), off the path whose latency the caller is waiting on.
    if (from_render_thread && !ctx->cancel_stream) ctx->want_cancel_stream = true;
    if (from_render_thread && ctx->cancel_stream && ctx->counters.p)
        (void)hipMemcpyAsync(error_block(ctx) + YK_CTRL_CANCELLED, ctx->cancel_host + 16, 4, hipMemcpyHostToDevice, ctx->cancel_stream);
    if (g_debug_cancel) std::fprintf(stderr, "[yk cancel] raise: %s, %.3f ms\n", ctx->cancel_stream ? "host word + device word" : "host word only (first interruption)", now_ms() - t0);
}

// Called by a submission before it enqueues anything: if an earlier one was interrupted, whatever it still has on the
// context's streams drains first — the word may only be cleared when no kernel that honoured it can run again.
static yk_status clear_cancel(yk_context* ctx) {
    if (!ctx->cancel_host || !ctx->cancel_raised.load(std::memory_order_acquire)) return YK_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->cancel_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->cancel_stream));  // a late copy of the 1 must not land in the next submission
    for (WorkSet& w : ctx->ws) {
        if (w.stream && w.stream != ctx->stream) HIP_TRY(ctx, hipStreamSynchronize(w.stream));
        if (w.side) HIP_TRY(ctx, hipStreamSynchronize(w.side));
    }
    __atomic_store_n(ctx->cancel_host, 0u, __ATOMIC_RELEASE);
    ctx->cancel_raised.store(false, std::memory_order_release);
    if (ctx->want_cancel_stream && !ctx->cancel_stream) {  // this context gets interrupted: later interruptions also write the device word directly
        if (hipStreamCreateWithFlags(&ctx->cancel_stream, hipStreamNonBlocking) != hipSuccess) ctx->cancel_stream = nullptr;
        ctx->want_cancel_stream = false;
    }
    return YK_OK;
}

// Wait for `done` (recorded on the stream that ends the submission) while polling the caller's predicate about every
// 100 us — the reference polls it once per pixel sample (integrators/mod.rs:153; render_worker.rs:240-255 relies on
// that for "low latency kills").  Returns true when the predicate fired: the word is raised and the streams have drained.
static bool poll_until(yk_context* ctx, hipEvent_t done, yk_cancel_fn cancel, void* user) {
    while (hipEventQuery(done) == hipErrorNotReady) {
        if (cancel(user)) {
            raise_cancel(ctx);
            return true;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    return false;
}
static bool wait_polling(yk_context* ctx, hipEvent_t done, hipStream_t st, yk_cancel_fn cancel, void* user) {
    const bool fired = cancel ? poll_until(ctx, done, cancel, user) : false;
    (void)hipStreamSynchronize(st);
    return fired;
}

// Integrator::render for a list of tiles.  tile_samples == nullptr: the plain film (all
// samples of a pixel, mean stored).  Otherwise the accumulating film (integrators/mod.rs:
// 146-161): one sample per pixel with global index tile_samples[t], raw value stored.
static yk_status render_tiles_impl(yk_context* ctx, const yk_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                                   const yk_integrator_desc* integrator, const yk_tile* tiles, const uint16_t* tile_samples, size_t n_tiles,
                                   void* d_out_rgb, void* stream, yk_render_stats* stats, yk_cancel_fn cancel, void* user,
                                   const yk_tile_list* prepared = nullptr, uint32_t n_passes = 1, int64_t uniform_first_sample = -1) try {
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
    {
        yk_status cs = clear_cancel(ctx);
        if (cs != YK_OK) return cs;
    }
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
    // accumulating film: FilmTile.sample per tile (tile_samples), or one sample index shared by all tiles
    const bool accumulating = tile_samples != nullptr || uniform_first_sample >= 0;
    if (tile_samples && uniform_first_sample >= 0) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "an accumulating tile list carries its own sample indices");
    if (n_passes == 0 || n_passes > 0xFFFFu || (!accumulating && n_passes != 1)) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "bad number of passes");
    // samples rendered per pixel by this call: the accumulating film renders passes FilmTile.sample .. + n_passes - 1
    const uint32_t spp = accumulating ? n_passes : prm.sampler.spp;
    prm.spe = spp;
    if (accumulating) {
        // render_manager.rs:135-143 queues samples 0 .. spp-1 of a tile and nothing else; an index beyond that is
        // outside the samplers' domain (the stratified permutation walks cycles of [0, spp) and need not terminate)
        for (size_t t = 0; tile_samples && t < n_tiles; ++t)
            if ((uint64_t)tile_samples[t] + n_passes > prm.sampler.spp)
                return fail(ctx, YK_ERR_INVALID_ARGUMENT, "FilmTile.sample (+ passes) beyond the sampler's samples per pixel");
        if (uniform_first_sample >= 0 && (uint64_t)uniform_first_sample + n_passes > prm.sampler.spp)
            return fail(ctx, YK_ERR_INVALID_ARGUMENT, "first_sample (+ passes) beyond the sampler's samples per pixel");
        if (uniform_first_sample >= 0) prm.sample_base = (uint32_t)uniform_first_sample;
    }
    const bool sample_table = tile_samples != nullptr;  // a per-pixel table of sample indices (absent: prm.sample_base for every pixel)
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
    HIP_TRY(ctx, hipMemsetAsync(counters, 0, YK_COUNTER_BYTES, st));  // the error block with it: stack-overflow flag, interruption word
    unsigned* errblk = error_block(ctx);
    prm.cancel = cancel_ref(ctx);
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
#ifdef YK_EXPERIMENT_GRAPH  // timing builds only (tools/micro/graph_replay_experiment.h)
    YK_GRAPH_BEGIN
#endif
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
            if (sample_table) pixel_sample = prepared->pixel_sample.as<uint32_t>() + px0;
        } else if (t_end - t_begin == 1) {  // one tile (the reference's per-tile call): it travels as a kernel argument
            HIP_TRY(ctx, ctx->pixel_xy.ensure((size_t)npx * 4));
            pixel_xy = ctx->pixel_xy.as<uint32_t>();
            if (sample_table) {
                HIP_TRY(ctx, ctx->scratch[5].ensure((size_t)npx * 4));
                pixel_sample = ctx->scratch[5].as<uint32_t>();
            }
            launch_pixel_table_one(st, tiles[t_begin], npx, pixel_xy, sample_table ? tile_samples[t_begin] : 0u, pixel_sample);
        } else {
        std::vector<uint32_t> loc(t_end - t_begin + 1);
        for (size_t t = t_begin; t <= t_end; ++t) loc[t - t_begin] = off[t] - px0;
        HIP_TRY(ctx, hipMemcpyAsync(ctx->tiles.p, tiles + t_begin, (t_end - t_begin) * sizeof(yk_tile), hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->tile_off.p, loc.data(), loc.size() * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));  // `loc` is a stack-lifetime staging buffer
        HIP_TRY(ctx, ctx->pixel_xy.ensure((size_t)npx * 4));
        pixel_xy = ctx->pixel_xy.as<uint32_t>();
        if (sample_table) {
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
        bool ws_busy[2] = {false, false};
        for (uint64_t w0 = 0; w0 < work; w0 += batch) {
            WorkSet& ws = ctx->ws[which];
            // An interruptible job of many batches is not enqueued all at once: a work set takes its next batch when its
            // previous one has finished (the other work set's batch keeps the GPU busy meanwhile), the predicate is polled
            // during the wait, and an interruption has at most two batches' launches left to drain — each of them an empty
            // launch of a few microseconds, but a 4K x 256 spp job holds 1300 of them.
            bool fired = cancel && ws_busy[which] && poll_until(ctx, ws.ev_batch, cancel, user);
            if (!fired && cancel && cancel(user)) {
                raise_cancel(ctx);
                fired = true;
            }
            if (fired) {  // what is already enqueued drains at once
                const double t0 = g_debug_cancel ? now_ms() : 0.0;
                (void)hipStreamSynchronize(st);
                const double t1 = g_debug_cancel ? now_ms() : 0.0;
                if (n_ws == 2) (void)hipStreamSynchronize(ctx->ws[1].stream);
                if (g_debug_cancel) {
                    unsigned words[4] = {0, 0, 0, 0};
                    (void)hipMemcpy(words, error_block(ctx), sizeof words, hipMemcpyDeviceToHost);
                    std::fprintf(stderr, "[yk cancel] drain: stream0 %.3f ms, stream1 %.3f ms, batch %llu of %llu, device word %u\n", t1 - t0, now_ms() - t1,
                                 (unsigned long long)(w0 / batch), (unsigned long long)((work + batch - 1) / batch), words[YK_CTRL_CANCELLED]);
                }
                return fail(ctx, YK_ERR_CANCELLED, "cancelled by early_termination_predicate");
            }
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
            if (cancel) {
                HIP_TRY(ctx, hipEventRecord(ws.ev_batch, bs));
                ws_busy[which] = true;
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
#ifdef YK_EXPERIMENT_GRAPH
    YK_GRAPH_END
#endif
    if (stats) {
        HIP_TRY(ctx, hipEventRecord(ev1, st));
        // a synchronous call: the predicate is polled while the GPU works
        if (wait_polling(ctx, ev1, st, cancel, user)) return fail(ctx, YK_ERR_CANCELLED, "cancelled by early_termination_predicate");
        if (ctx->cancel_raised.load(std::memory_order_acquire)) return fail(ctx, YK_ERR_CANCELLED, "interrupted (yk_context_interrupt)");
        HIP_TRY(ctx, hipGetLastError());
        std::memset(stats, 0, sizeof(*stats));
        unsigned long long host_counters[YK_COUNTER_BYTES / 8];
        HIP_TRY(ctx, hipMemcpy(host_counters, counters, YK_COUNTER_BYTES, hipMemcpyDeviceToHost));
        unsigned host_err[4];
        std::memcpy(host_err, host_counters + 16, sizeof(host_err));
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

yk_status yk_render_tile_list_samples_device(yk_context* ctx, const yk_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                                             const yk_integrator_desc* integrator, const yk_tile_list* list, uint32_t first_sample, uint32_t n_passes,
                                             void* d_out_rgb, void* stream, yk_render_stats* stats, yk_cancel_fn cancel, void* user) {
    if (!ctx) return YK_ERR_INVALID_ARGUMENT;
    if (!list) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null tile list");
    if (!list->samples.empty()) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "the list carries per-tile sample indices: use yk_render_tile_list_passes_device");
    return render_tiles_impl(ctx, scene, camera, sampler, integrator, list->tiles.data(), nullptr, list->tiles.size(), d_out_rgb, stream, stats, cancel,
                             user, list, n_passes, (int64_t)first_sample);
}

// An interruption from any thread (the call that renders holds the context's lock for its whole duration, this one
// takes none): what the context has enqueued stops at the kernels' next look at the word, the next submission waits
// for it to drain and clears the word.
yk_status yk_context_interrupt(yk_context* ctx) {
    if (!ctx) return YK_ERR_INVALID_ARGUMENT;
    raise_cancel(ctx, false);  // the host word only: no HIP call from a thread that may not have the device current
    return YK_OK;
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

}  // extern "C"
