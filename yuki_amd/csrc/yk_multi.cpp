// yk_multi.cpp — the film of one render spread over the GPUs of a node (include/yuki_hip.h, "several GPUs").
//
// The reference's RenderManager owns all workers of one process (renderer/render_manager.rs:78-97), deals the
// film's tiles to them (:125-143, "interleave tiles": :206-210) and Film::update_tile (film.rs:210-282) writes
// every finished tile back.  Here the workers are GPUs: one context + one host thread per device, the scene
// replicated, spiral tile i on device i mod G, and ONE exchange of the per-device slabs into device 0's memory —
// RCCL point-to-point calls on the contexts' own streams — followed by the scatter into the row-major film.
//
// xGMI is a set of point-to-point links (7 x ~153 GB/s per GPU), not a switch: a gather of disjoint slabs as
// G-1 independent send/recv pairs uses one link per peer and reduces nothing.  The payload is tiny (24.9 MB for a
// 1080p film, 3.1 MB per peer at G = 8); what matters is that nothing waits on the host between a device's last
// kernel and its send.
//
// RCCL is bound at run time (dlopen): the library has no link dependency on it, single-GPU users never load it
// (librccl is ~570 MB), and a host process that already carries a copy (PyTorch) shares that one.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <thread>

#include "yk_internal.h"

// ------------------------------------------------------------------ RCCL, bound lazily
namespace {

// the part of rccl.h this file needs (types only; no header dependency so the library builds without RCCL installed)
typedef struct ncclComm* ncclComm_t;
typedef struct {
    char internal[YK_DIST_ID_BYTES];
} ncclUniqueId;
static const int kNcclSuccess = 0;
static const int kNcclFloat = 7;  // ncclFloat32 (rccl.h: ncclDataType_t)

struct Rccl {
    void* handle = nullptr;
    std::string error;
    int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // A copy the process already holds first (RTLD_NOLOAD), then the usual names.  One process, one RCCL: the dynamic
        // linker matches libraries by soname (librccl.so.1), so whichever copy is loaded first — ours here, or the one a
        // host such as PyTorch ships and links — serves everybody afterwards.  A host that carries its own RCCL should
        // load it before the first yk_multi / yk_dist call (the Python tests import torch first for that reason).
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names)
            if ((r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL))) break;
        if (!r.handle)
            for (const char* n : names)
                if ((r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!r.handle) {
            const char* e = dlerror();
            r.error = std::string("RCCL not found (librccl.so.1): ") + (e ? e : "");
            return;
        }
        struct {
            const char* name;
            void** slot;
        } syms[] = {{"ncclCommInitAll", (void**)&r.CommInitAll}, {"ncclCommInitRank", (void**)&r.CommInitRank}, {"ncclGetUniqueId", (void**)&r.GetUniqueId},
                    {"ncclCommDestroy", (void**)&r.CommDestroy}, {"ncclGroupStart", (void**)&r.GroupStart},   {"ncclGroupEnd", (void**)&r.GroupEnd},
                    {"ncclSend", (void**)&r.Send},               {"ncclRecv", (void**)&r.Recv},               {"ncclGetErrorString", (void**)&r.GetErrorString}};
        for (auto& s : syms) {
            *s.slot = dlsym(r.handle, s.name);
            if (!*s.slot) {
                r.error = std::string("RCCL symbol missing: ") + s.name;
                r.handle = nullptr;
                return;
            }
        }
    });
    return r.handle ? &r : nullptr;
}

// One host thread per device: it sets its device once and runs the jobs posted to it, so that a frame's G
// render submissions (each ~1 ms of launches) proceed side by side instead of one after the other.
class Worker {
  public:
    explicit Worker(int device) : device_(device), thread_([this] { loop(); }) {}
    ~Worker() {
        {
            std::lock_guard<std::mutex> l(mu_);
            quit_ = true;
        }
        cv_.notify_all();
        thread_.join();
    }
    void post(std::function<yk_status()> job) {
        {
            std::lock_guard<std::mutex> l(mu_);
            job_ = std::move(job);
            busy_ = true;
        }
        cv_.notify_all();
    }
    yk_status wait() {
        std::unique_lock<std::mutex> l(mu_);
        cv_.wait(l, [this] { return !busy_; });
        return result_;
    }

  private:
    void loop() {
        (void)hipSetDevice(device_);
        for (;;) {
            std::function<yk_status()> job;
            {
                std::unique_lock<std::mutex> l(mu_);
                cv_.wait(l, [this] { return quit_ || (busy_ && job_); });
                if (quit_) return;
                job = std::move(job_);
                job_ = nullptr;
            }
            yk_status r;
            try {
                r = job();
            } catch (...) {
                r = YK_ERR_OUT_OF_MEMORY;
            }
            {
                std::lock_guard<std::mutex> l(mu_);
                result_ = r;
                busy_ = false;
            }
            cv_.notify_all();
        }
    }
    int device_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::function<yk_status()> job_;
    bool busy_ = false, quit_ = false;
    yk_status result_ = YK_OK;
    std::thread thread_;  // last: starts after the members above exist
};

}  // namespace

struct yk_multi {
    std::vector<int> devices;
    std::vector<yk_context*> ctx;
    std::vector<std::unique_ptr<Worker>> workers;
    std::vector<ncclComm_t> comms;  // empty until the first exchange needs them
    bool loopback = false;
    std::string last_error;
    std::mutex mu;
};

struct yk_multi_scene {
    yk_multi* owner = nullptr;
    std::vector<yk_scene*> per_device;
    yk_scene_info info;
};

struct yk_multi_film {
    yk_multi* owner = nullptr;
    std::vector<int> devices;            // copy of the owner's device list: the film frees its buffers without it
    uint16_t res_x = 0, res_y = 0, tile_dim = 0;
    std::vector<yk_tile_list*> lists;    // rank r's tiles, on device r (what it renders)
    std::vector<yk_tile_list*> lists0;   // the same tiles, on device 0 (what it scatters); lists0[0] == lists[0]
    std::vector<size_t> n_floats;        // 3 * pixels of rank r's slab
    std::vector<DevBuf> slab;            // on device r
    std::vector<DevBuf> gathered;        // on device 0 (rank 0's entry only used by the loopback option)
    DevBuf film;                         // on device 0: res_x * res_y * 3 floats
};

static yk_status mfail(yk_multi* m, yk_status st, const std::string& msg) {
    if (m) m->last_error = msg;
    return st;
}

static yk_status rccl_check(yk_multi* m, int rc, const char* what) {
    if (rc == kNcclSuccess) return YK_OK;
    Rccl* r = rccl();
    return mfail(m, YK_ERR_DEVICE, std::string(what) + ": " + (r ? r->GetErrorString(rc) : "RCCL error"));
}

// communicators of all devices, created by one call (single process: ncclCommInitAll)
static yk_status ensure_comms(yk_multi* m) {
    if (!m->comms.empty()) return YK_OK;
    Rccl* r = rccl();
    if (!r) return mfail(m, YK_ERR_UNSUPPORTED, "RCCL could not be loaded (librccl.so.1): multi-device exchange unavailable");
    std::vector<ncclComm_t> comms(m->devices.size(), nullptr);
    yk_status st = rccl_check(m, r->CommInitAll(comms.data(), (int)m->devices.size(), m->devices.data()), "ncclCommInitAll");
    if (st != YK_OK) return st;
    m->comms.swap(comms);
    return YK_OK;
}

extern "C" {

yk_status yk_multi_create(const int* devices, uint32_t n_devices, yk_multi** out) try {
    if (!devices || !out || n_devices == 0 || n_devices > 64) return YK_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    for (uint32_t a = 0; a < n_devices; ++a)
        for (uint32_t b = a + 1; b < n_devices; ++b)
            if (devices[a] == devices[b]) return YK_ERR_INVALID_ARGUMENT;  // one rank per GPU (RCCL refuses two ranks on one device)
    if (n_devices > 1 && !rccl()) return YK_ERR_UNSUPPORTED;
    std::unique_ptr<yk_multi> m(new yk_multi());
    m->devices.assign(devices, devices + n_devices);
    for (uint32_t r = 0; r < n_devices; ++r) {
        yk_context* c = nullptr;
        yk_status st = yk_context_create(devices[r], &c);
        if (st != YK_OK) {
            for (yk_context* x : m->ctx) yk_context_destroy(x);
            return st;
        }
        m->ctx.push_back(c);
    }
    for (uint32_t r = 0; r < n_devices; ++r) m->workers.emplace_back(new Worker(devices[r]));
    *out = m.release();
    return YK_OK;
} catch (const std::exception&) {
    return YK_ERR_OUT_OF_MEMORY;
}

void yk_multi_destroy(yk_multi* m) {
    if (!m) return;
    m->workers.clear();  // joins the threads
    if (!m->comms.empty()) {
        Rccl* r = rccl();
        for (size_t k = 0; k < m->comms.size(); ++k) {
            (void)hipSetDevice(m->devices[k]);
            (void)hipDeviceSynchronize();
            if (r && m->comms[k]) (void)r->CommDestroy(m->comms[k]);
        }
    }
    for (yk_context* c : m->ctx) yk_context_destroy(c);
    delete m;
}

uint32_t yk_multi_device_count(const yk_multi* m) { return m ? (uint32_t)m->devices.size() : 0u; }

yk_context* yk_multi_context(yk_multi* m, uint32_t rank) { return (m && rank < m->ctx.size()) ? m->ctx[rank] : nullptr; }

yk_status yk_multi_set_option(yk_multi* m, const char* key, int64_t value) {
    if (!m || !key) return YK_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> l(m->mu);
    if (std::strcmp(key, "rccl_loopback") == 0) {
        m->loopback = value != 0;
        return YK_OK;
    }
    for (yk_context* c : m->ctx) {
        yk_status st = yk_context_set_option(c, key, value);
        if (st != YK_OK) return mfail(m, st, std::string("bad option ") + key);
    }
    return YK_OK;
}

yk_status yk_multi_last_error(const yk_multi* m, char* buf, size_t cap) {
    if (!m || !buf || cap == 0) return YK_ERR_INVALID_ARGUMENT;
    std::snprintf(buf, cap, "%s", m->last_error.c_str());
    return YK_OK;
}

// ------------------------------------------------------------------ scene
yk_status yk_multi_scene_create(yk_multi* m, const yk_scene_desc* desc, yk_multi_scene** out) try {
    if (!m || !desc || !out) return YK_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> l(m->mu);
    *out = nullptr;
    // the host work — validation, BoundingVolumeHierarchy::new, device records — happens once
    std::shared_ptr<SceneImage> img;
    yk_status st = yk_build_scene_image(m->ctx[0], desc, img);
    if (st != YK_OK) return mfail(m, st, m->ctx[0]->last_error);
    std::unique_ptr<yk_multi_scene> s(new yk_multi_scene());
    s->owner = m;
    s->per_device.assign(m->ctx.size(), nullptr);
    // ... and every device's worker uploads its copy (PCIe links are per device)
    for (size_t r = 0; r < m->ctx.size(); ++r) {
        yk_context* c = m->ctx[r];
        yk_scene** slot = &s->per_device[r];
        m->workers[r]->post([c, img, slot] { return yk_upload_scene_image(c, img, slot); });
    }
    yk_status first = YK_OK;
    for (size_t r = 0; r < m->ctx.size(); ++r) {
        yk_status w = m->workers[r]->wait();
        if (w != YK_OK && first == YK_OK) first = mfail(m, w, "device " + std::to_string(m->devices[r]) + ": " + m->ctx[r]->last_error);
    }
    if (first != YK_OK) {
        for (yk_scene* x : s->per_device) yk_scene_destroy(x);
        return first;
    }
    (void)yk_scene_get_info(s->per_device[0], &s->info);
    for (size_t r = 1; r < s->per_device.size(); ++r) {
        yk_scene_info i;
        (void)yk_scene_get_info(s->per_device[r], &i);
        s->info.upload_seconds = std::max(s->info.upload_seconds, i.upload_seconds);
    }
    *out = s.release();
    return YK_OK;
} catch (const std::exception& e) {
    return mfail(m, YK_ERR_OUT_OF_MEMORY, e.what());
}

void yk_multi_scene_destroy(yk_multi_scene* s) {
    if (!s) return;
    for (yk_scene* x : s->per_device) yk_scene_destroy(x);
    delete s;
}

yk_status yk_multi_scene_get_info(const yk_multi_scene* s, yk_scene_info* out) {
    if (!s || !out) return YK_ERR_INVALID_ARGUMENT;
    *out = s->info;
    return YK_OK;
}

// ------------------------------------------------------------------ film
void yk_multi_film_destroy(yk_multi_film* f) {
    if (!f) return;
    for (size_t r = 0; r < f->lists.size(); ++r) {
        if (r < f->lists0.size() && f->lists0[r] && f->lists0[r] != f->lists[r]) yk_tile_list_destroy(f->lists0[r]);
        if (f->lists[r]) yk_tile_list_destroy(f->lists[r]);
    }
    for (size_t r = 0; r < f->slab.size() && r < f->devices.size(); ++r) {
        (void)hipSetDevice(f->devices[r]);
        f->slab[r].release();
    }
    if (!f->devices.empty()) (void)hipSetDevice(f->devices[0]);
    for (DevBuf& b : f->gathered) b.release();
    f->film.release();
    delete f;
}

yk_status yk_multi_film_create(yk_multi* m, uint16_t res_x, uint16_t res_y, uint16_t tile_dim, yk_multi_film** out) try {
    if (!m || !out || res_x == 0 || res_y == 0 || tile_dim == 0) return YK_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> l(m->mu);
    *out = nullptr;
    const size_t G = m->ctx.size();
    const std::vector<yk_tile> tiles = film_tiles(res_x, res_y, tile_dim);  // film.rs:409-475, outward spiral
    if (tiles.size() < G) return mfail(m, YK_ERR_INVALID_ARGUMENT, "fewer tiles than devices");
    std::unique_ptr<yk_multi_film, void (*)(yk_multi_film*)> f(new yk_multi_film(), yk_multi_film_destroy);
    f->owner = m;
    f->devices = m->devices;
    f->res_x = res_x;
    f->res_y = res_y;
    f->tile_dim = tile_dim;
    f->lists.assign(G, nullptr);
    f->lists0.assign(G, nullptr);
    f->n_floats.assign(G, 0);
    f->slab.resize(G);
    f->gathered.resize(G);
    for (size_t r = 0; r < G; ++r) {
        std::vector<yk_tile> mine;  // tile i -> device i mod G (render_manager.rs:206-210 "interleave tiles")
        for (size_t i = r; i < tiles.size(); i += G) mine.push_back(tiles[i]);
        size_t px = 0;
        for (const yk_tile& t : mine) px += (size_t)(t.x1 - t.x0) * (size_t)(t.y1 - t.y0);
        f->n_floats[r] = 3 * px;
        yk_status st = yk_tile_list_create(m->ctx[r], mine.data(), nullptr, mine.size(), &f->lists[r]);
        if (st != YK_OK) return mfail(m, st, m->ctx[r]->last_error);
        if (r == 0) {
            f->lists0[0] = f->lists[0];
        } else if ((st = yk_tile_list_create(m->ctx[0], mine.data(), nullptr, mine.size(), &f->lists0[r])) != YK_OK) {
            return mfail(m, st, m->ctx[0]->last_error);
        }
        (void)hipSetDevice(m->devices[r]);
        if (f->slab[r].ensure(f->n_floats[r] * sizeof(float)) != hipSuccess) return mfail(m, YK_ERR_OUT_OF_MEMORY, "slab");
        (void)hipSetDevice(m->devices[0]);
        if (f->gathered[r].ensure(f->n_floats[r] * sizeof(float)) != hipSuccess) return mfail(m, YK_ERR_OUT_OF_MEMORY, "gather buffer");
    }
    (void)hipSetDevice(m->devices[0]);
    if (f->film.ensure((size_t)res_x * res_y * 3 * sizeof(float)) != hipSuccess) return mfail(m, YK_ERR_OUT_OF_MEMORY, "film");
    if (hipMemset(f->film.p, 0, f->film.bytes) != hipSuccess) return mfail(m, YK_ERR_DEVICE, "film clear");
    *out = f.release();
    return YK_OK;
} catch (const std::exception& e) {
    return mfail(m, YK_ERR_OUT_OF_MEMORY, e.what());
}

void* yk_multi_film_device_ptr(const yk_multi_film* f) { return f ? f->film.p : nullptr; }

// ------------------------------------------------------------------ the frame
yk_status yk_multi_render_film(yk_multi* m, const yk_multi_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                               const yk_integrator_desc* integrator, yk_multi_film* film, float* film_rgb, yk_render_stats* stats, yk_cancel_fn cancel,
                               void* user) try {
    if (!m) return YK_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> l(m->mu);
    if (!scene || !camera || !sampler || !integrator || !film) return mfail(m, YK_ERR_INVALID_ARGUMENT, "null argument");
    if (scene->owner != m || film->owner != m) return mfail(m, YK_ERR_INVALID_ARGUMENT, "scene / film belong to another yk_multi");
    const size_t G = m->ctx.size();
    const bool exchange = G > 1 || m->loopback;
    if (exchange) {
        yk_status st = ensure_comms(m);
        if (st != YK_OK) return st;
    }
    // (1) every device renders its tiles; the host threads only enqueue (stats == NULL) or wait for their own device
    std::vector<yk_render_stats> per(G);
    for (size_t r = 0; r < G; ++r) {
        yk_context* c = m->ctx[r];
        const yk_scene* sc = scene->per_device[r];
        const yk_tile_list* tl = film->lists[r];
        void* dst = film->slab[r].p;
        yk_render_stats* ps = stats ? &per[r] : nullptr;
        m->workers[r]->post([=] { return yk_render_tile_list_device(c, sc, camera, sampler, integrator, tl, dst, nullptr, ps, cancel, user); });
    }
    yk_status first = YK_OK;
    for (size_t r = 0; r < G; ++r) {
        yk_status w = m->workers[r]->wait();
        if (w != YK_OK && first == YK_OK) first = mfail(m, w, "device " + std::to_string(m->devices[r]) + ": " + m->ctx[r]->last_error);
    }
    if (first != YK_OK) return first;
    // (2) slabs -> device 0: one group of point-to-point calls on the contexts' own streams (ordered after the renders)
    hipStream_t s0 = (hipStream_t)yk_context_stream(m->ctx[0]);
    if (exchange) {
        Rccl* r = rccl();
        yk_status st = rccl_check(m, r->GroupStart(), "ncclGroupStart");
        if (st != YK_OK) return st;
        for (size_t k = m->loopback ? 0 : 1; k < G && st == YK_OK; ++k) {
            (void)hipSetDevice(m->devices[k]);
            st = rccl_check(m, r->Send(film->slab[k].p, film->n_floats[k], kNcclFloat, 0, m->comms[k], (hipStream_t)yk_context_stream(m->ctx[k])), "ncclSend");
            if (st != YK_OK) break;
            (void)hipSetDevice(m->devices[0]);
            st = rccl_check(m, r->Recv(film->gathered[k].p, film->n_floats[k], kNcclFloat, (int)k, m->comms[0], s0), "ncclRecv");
        }
        yk_status ge = rccl_check(m, r->GroupEnd(), "ncclGroupEnd");
        if (st != YK_OK) return st;
        if (ge != YK_OK) return ge;
    }
    // (3) Film::update_tile for every tile, on device 0, behind the receives
    (void)hipSetDevice(m->devices[0]);
    for (size_t k = 0; k < G; ++k) {
        const void* src = (k == 0 && !m->loopback) ? film->slab[0].p : film->gathered[k].p;
        yk_status st = yk_film_update_tile_list_device(m->ctx[0], film->lists0[k], src, film->res_x, film->res_y, film->film.p, nullptr, 0);
        if (st != YK_OK) return mfail(m, st, m->ctx[0]->last_error);
    }
    if (film_rgb) {
        if (hipMemcpyAsync(film_rgb, film->film.p, (size_t)film->res_x * film->res_y * 3 * sizeof(float), hipMemcpyDeviceToHost, s0) != hipSuccess)
            return mfail(m, YK_ERR_DEVICE, "film read-back");
    }
    if (film_rgb || stats) {
        if (hipStreamSynchronize(s0) != hipSuccess) return mfail(m, YK_ERR_DEVICE, "synchronise device 0");
    }
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        for (const yk_render_stats& p : per) {
            stats->rays += p.rays;
            stats->shadow_rays += p.shadow_rays;
            stats->samples += p.samples;
            stats->seconds_total = std::max(stats->seconds_total, p.seconds_total);
            stats->seconds_trace = std::max(stats->seconds_trace, p.seconds_trace);
            stats->seconds_shadow = std::max(stats->seconds_shadow, p.seconds_shadow);
            stats->seconds_shade = std::max(stats->seconds_shade, p.seconds_shade);
            stats->trace_launches += p.trace_launches;
            stats->shadow_launches += p.shadow_launches;
            stats->batches += p.batches;
        }
    }
    return YK_OK;
} catch (const std::exception& e) {
    return mfail(m, YK_ERR_OUT_OF_MEMORY, e.what());
}

yk_status yk_multi_sync(yk_multi* m) {
    if (!m) return YK_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> l(m->mu);
    for (size_t r = m->ctx.size(); r-- > 0;) {  // device 0 last: its stream ends the frame
        (void)hipSetDevice(m->devices[r]);
        if (hipStreamSynchronize((hipStream_t)yk_context_stream(m->ctx[r])) != hipSuccess) return mfail(m, YK_ERR_DEVICE, "hipStreamSynchronize");
    }
    return YK_OK;
}

// ------------------------------------------------------------------ one process per GPU
}  // extern "C"

struct yk_dist {
    yk_context* ctx = nullptr;
    ncclComm_t comm = nullptr;
    uint32_t rank = 0, world = 1;
};

extern "C" {

yk_status yk_dist_unique_id(uint8_t id[YK_DIST_ID_BYTES]) {
    if (!id) return YK_ERR_INVALID_ARGUMENT;
    Rccl* r = rccl();
    if (!r) return YK_ERR_UNSUPPORTED;
    ncclUniqueId u;
    if (r->GetUniqueId(&u) != kNcclSuccess) return YK_ERR_DEVICE;
    std::memcpy(id, u.internal, YK_DIST_ID_BYTES);
    return YK_OK;
}

yk_status yk_dist_create(yk_context* ctx, const uint8_t id[YK_DIST_ID_BYTES], uint32_t rank, uint32_t world, yk_dist** out) {
    if (!ctx || !id || !out || world == 0 || rank >= world) return YK_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    Rccl* r = rccl();
    if (!r) return fail(ctx, YK_ERR_UNSUPPORTED, "RCCL could not be loaded (librccl.so.1)");
    (void)hipSetDevice(ctx->device);
    ncclUniqueId u;
    std::memcpy(u.internal, id, YK_DIST_ID_BYTES);
    ncclComm_t comm = nullptr;
    const int rc = r->CommInitRank(&comm, (int)world, u, (int)rank);
    if (rc != kNcclSuccess) return fail(ctx, YK_ERR_DEVICE, std::string("ncclCommInitRank: ") + r->GetErrorString(rc));
    yk_dist* d = new yk_dist();
    d->ctx = ctx;
    d->comm = comm;
    d->rank = rank;
    d->world = world;
    *out = d;
    return YK_OK;
}

void yk_dist_destroy(yk_dist* d) {
    if (!d) return;
    Rccl* r = rccl();
    (void)hipSetDevice(d->ctx->device);
    (void)hipDeviceSynchronize();
    if (r && d->comm) (void)r->CommDestroy(d->comm);
    delete d;
}

yk_status yk_dist_gather(yk_dist* d, const void* d_send, void* d_recv, size_t count, void* stream) {
    if (!d || !d_send || (d->rank == 0 && !d_recv)) return YK_ERR_INVALID_ARGUMENT;
    Rccl* r = rccl();
    yk_context* ctx = d->ctx;
    (void)hipSetDevice(ctx->device);
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    int rc = r->GroupStart();
    if (rc != kNcclSuccess) return fail(ctx, YK_ERR_DEVICE, std::string("ncclGroupStart: ") + r->GetErrorString(rc));
    rc = r->Send(d_send, count, kNcclFloat, 0, d->comm, st);
    if (rc == kNcclSuccess && d->rank == 0)
        for (uint32_t k = 0; k < d->world && rc == kNcclSuccess; ++k) rc = r->Recv(reinterpret_cast<float*>(d_recv) + (size_t)k * count, count, kNcclFloat, (int)k, d->comm, st);
    const int ge = r->GroupEnd();
    if (rc != kNcclSuccess) return fail(ctx, YK_ERR_DEVICE, std::string("ncclSend/ncclRecv: ") + r->GetErrorString(rc));
    if (ge != kNcclSuccess) return fail(ctx, YK_ERR_DEVICE, std::string("ncclGroupEnd: ") + r->GetErrorString(ge));
    return YK_OK;
}

}  // extern "C"
