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

#include <atomic>
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
    int (*GetVersion)(int*) = nullptr;
    int version = 0;
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
        // The prototypes above are declared by hand (ncclFloat32 = 7, a 128-byte ncclUniqueId by value, int results): that is
        // the ABI of NCCL / RCCL 2.x since point-to-point calls exist (2.7).  Anything else is refused instead of trusted.
        r.GetVersion = (int (*)(int*))dlsym(r.handle, "ncclGetVersion");
        if (!r.GetVersion || r.GetVersion(&r.version) != kNcclSuccess || r.version < 2700 || r.version >= 30000) {
            r.error = "RCCL version " + std::to_string(r.version) + " is outside the range this binding was written for (2.7 <= v < 3)";
            r.handle = nullptr;
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

// The caller's current device is part of ITS state (a host such as PyTorch allocates on it): every public entry
// point puts it back.
struct DeviceGuard {
    int saved = -1;
    DeviceGuard() {
        if (hipGetDevice(&saved) != hipSuccess) saved = -1;
    }
    ~DeviceGuard() {
        if (saved >= 0) (void)hipSetDevice(saved);
    }
};

// The reference's early-termination predicate is a consuming FnMut polled by ONE thread (render_worker.rs:240-249:
// `from_parent.try_recv()`); here G device threads poll.  The latch calls the user's predicate from one thread at a
// time, never again once it has fired, and every device sees the answer: the first poll that returns non-zero stops
// all of them.
struct CancelLatch {
    std::mutex mu;
    std::atomic<bool> hit{false};
    yk_cancel_fn fn = nullptr;
    void* user = nullptr;
    static int poll(void* p) {
        CancelLatch* l = static_cast<CancelLatch*>(p);
        if (l->hit.load(std::memory_order_acquire)) return 1;
        std::lock_guard<std::mutex> g(l->mu);
        if (l->hit.load(std::memory_order_relaxed)) return 1;
        if (l->fn && l->fn(l->user)) {
            l->hit.store(true, std::memory_order_release);
            return 1;
        }
        return 0;
    }
};

}  // namespace

struct yk_multi {
    std::vector<int> devices;
    std::vector<yk_context*> ctx;
    std::vector<std::unique_ptr<Worker>> workers;
    std::vector<ncclComm_t> comms;  // empty until the first exchange needs them
    std::vector<hipEvent_t> ev_rendered, ev_copied;  // peer-copy exchange: rank k's slab is complete / has left rank k
    bool loopback = false;
    bool peer_copy = false;  // slabs move by hipMemcpyPeerAsync on device 0's stream instead of RCCL send / recv
    bool shared = false;     // ranks may share a device (test rigs: G ranks on one GPU) — implies peer_copy
    mutable std::mutex mu;
    std::string last_error;
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
    uint32_t slab_passes = 1;            // passes a slab / gather buffer has room for (accumulating frames grow them)
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

// The tile deal without a device: spiral tile i of film_tiles(res, tile_dim) (film.rs:333-376, 409-475) belongs to
// rank i mod n_ranks (render_manager.rs:206-210, "interleave tiles").  yk_multi_film_create and bench.py's ranks
// use exactly this function.
size_t yk_multi_deal(uint16_t res_x, uint16_t res_y, uint16_t tile_dim, uint32_t n_ranks, uint32_t rank, yk_tile* out, size_t cap, uint64_t* out_pixels) try {
    if (out_pixels) *out_pixels = 0;
    if (res_x == 0 || res_y == 0 || tile_dim == 0 || n_ranks == 0 || rank >= n_ranks) return 0;
    const std::vector<yk_tile> tiles = film_tiles(res_x, res_y, tile_dim);
    size_t n = 0;
    uint64_t px = 0;
    for (size_t i = rank; i < tiles.size(); i += n_ranks) {
        if (out && n < cap) out[n] = tiles[i];
        px += (uint64_t)(tiles[i].x1 - tiles[i].x0) * (uint64_t)(tiles[i].y1 - tiles[i].y0);
        ++n;
    }
    if (out_pixels) *out_pixels = px;
    return n;
} catch (const std::exception&) {
    return 0;
}

yk_status yk_multi_create_ex(const int* devices, uint32_t n_devices, uint32_t flags, yk_multi** out) try {
    if (!devices || !out || n_devices == 0 || n_devices > 64 || (flags & ~(YK_MULTI_SHARED_DEVICES | YK_MULTI_PEER_COPY))) return YK_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    const bool shared = (flags & YK_MULTI_SHARED_DEVICES) != 0;
    const bool peer_copy = shared || (flags & YK_MULTI_PEER_COPY) != 0;
    if (!shared)
        for (uint32_t a = 0; a < n_devices; ++a)
            for (uint32_t b = a + 1; b < n_devices; ++b)
                if (devices[a] == devices[b]) return YK_ERR_INVALID_ARGUMENT;  // one rank per GPU (RCCL refuses two ranks on one device)
    if (n_devices > 1 && !peer_copy && !rccl()) return YK_ERR_UNSUPPORTED;
    DeviceGuard restore;
    // yk_multi_destroy copes with a half-built object: every failure path below frees the contexts made so far
    std::unique_ptr<yk_multi, void (*)(yk_multi*)> m(new yk_multi(), yk_multi_destroy);
    m->devices.assign(devices, devices + n_devices);
    m->shared = shared;
    m->peer_copy = peer_copy;
    for (uint32_t r = 0; r < n_devices; ++r) {
        yk_context* c = nullptr;
        yk_status st = yk_context_create(devices[r], &c);
        if (st != YK_OK) return st;
        m->ctx.push_back(c);
    }
    m->ev_rendered.assign(n_devices, nullptr);
    m->ev_copied.assign(n_devices, nullptr);
    for (uint32_t r = 0; r < n_devices; ++r) {
        if (hipSetDevice(devices[r]) != hipSuccess || hipEventCreateWithFlags(&m->ev_rendered[r], hipEventDisableTiming) != hipSuccess) return YK_ERR_DEVICE;
        if (hipSetDevice(devices[0]) != hipSuccess || hipEventCreateWithFlags(&m->ev_copied[r], hipEventDisableTiming) != hipSuccess) return YK_ERR_DEVICE;
    }
    for (uint32_t r = 0; r < n_devices; ++r) m->workers.emplace_back(new Worker(devices[r]));
    *out = m.release();
    return YK_OK;
} catch (const std::exception&) {
    return YK_ERR_OUT_OF_MEMORY;
}

yk_status yk_multi_create(const int* devices, uint32_t n_devices, yk_multi** out) { return yk_multi_create_ex(devices, n_devices, 0u, out); }

void yk_multi_destroy(yk_multi* m) {
    if (!m) return;
    DeviceGuard restore;
    m->workers.clear();  // joins the threads
    if (!m->comms.empty()) {
        Rccl* r = rccl();
        for (size_t k = 0; k < m->comms.size(); ++k) {
            (void)hipSetDevice(m->devices[k]);
            (void)hipDeviceSynchronize();
            if (r && m->comms[k]) (void)r->CommDestroy(m->comms[k]);
        }
    }
    for (size_t k = 0; k < m->ctx.size(); ++k) {  // the events may still be referenced by enqueued waits: idle streams first
        (void)hipSetDevice(m->devices[k]);
        (void)hipStreamSynchronize((hipStream_t)yk_context_stream(m->ctx[k]));
    }
    for (hipEvent_t e : m->ev_rendered)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : m->ev_copied)
        if (e) (void)hipEventDestroy(e);
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
    if (std::strcmp(key, "peer_copy") == 0) {
        if (m->shared && value == 0) return mfail(m, YK_ERR_INVALID_ARGUMENT, "ranks that share a device exchange by peer copy only");
        m->peer_copy = value != 0;
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
    std::lock_guard<std::mutex> l(m->mu);
    std::snprintf(buf, cap, "%s", m->last_error.c_str());
    return YK_OK;
}

// ------------------------------------------------------------------ scene
yk_status yk_multi_scene_create(yk_multi* m, const yk_scene_desc* desc, yk_multi_scene** out) try {
    if (!m || !desc || !out) return YK_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> l(m->mu);
    DeviceGuard restore;
    *out = nullptr;
    // the host work — validation, BoundingVolumeHierarchy::new, device records — happens once
    std::shared_ptr<SceneImage> img;
    yk_status st = yk_build_scene_image(m->ctx[0], desc, img);
    if (st != YK_OK) return mfail(m, st, m->ctx[0]->last_error);
    std::unique_ptr<yk_multi_scene> s(new yk_multi_scene());
    s->owner = m;
    s->per_device.assign(m->ctx.size(), nullptr);
    // ... and every device's worker uploads its copy (PCIe links are per device).  Whatever happens on this thread
    // in between, every posted worker is waited for before `s` (whose slots they write) can be freed.
    size_t posted = 0;
    struct WaitPosted {
        yk_multi* m;
        size_t* posted;
        ~WaitPosted() {
            for (size_t r = 0; r < *posted; ++r) (void)m->workers[r]->wait();
        }
    } wait_posted{m, &posted};
    for (size_t r = 0; r < m->ctx.size(); ++r) {
        yk_context* c = m->ctx[r];
        yk_scene** slot = &s->per_device[r];
        m->workers[r]->post([c, img, slot] { return yk_upload_scene_image(c, img, slot); });
        ++posted;
    }
    yk_status first = YK_OK;
    for (size_t r = 0; r < m->ctx.size(); ++r) {
        yk_status w = m->workers[r]->wait();
        if (w != YK_OK && first == YK_OK) first = mfail(m, w, "device " + std::to_string(m->devices[r]) + ": " + m->ctx[r]->last_error);
    }
    posted = 0;
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
    DeviceGuard restore;
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
    DeviceGuard restore;
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
    DeviceGuard restore;
    *out = nullptr;
    const size_t G = m->ctx.size();
    const size_t n_tiles = yk_film_tiles(res_x, res_y, tile_dim, nullptr, 0);  // film.rs:409-475, outward spiral
    if (n_tiles < G) return mfail(m, YK_ERR_INVALID_ARGUMENT, "fewer tiles than devices");
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
        // tile i -> rank i mod G (render_manager.rs:206-210 "interleave tiles"): yk_multi_deal, the same function a host calls
        uint64_t px = 0;
        std::vector<yk_tile> mine(yk_multi_deal(res_x, res_y, tile_dim, (uint32_t)G, (uint32_t)r, nullptr, 0, nullptr));
        (void)yk_multi_deal(res_x, res_y, tile_dim, (uint32_t)G, (uint32_t)r, mine.data(), mine.size(), &px);
        f->n_floats[r] = 3 * (size_t)px;
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
// One frame: renders -> exchange -> Film::update_tile on device 0.  first_sample < 0: the plain film (all samples of
// a pixel, mean stored, film overwritten).  Otherwise the accumulating film (integrators/mod.rs:146-161, film.rs:260-272):
// passes first_sample .. first_sample + n_passes - 1 of every tile, each ADDED to the film.
static yk_status render_frame(yk_multi* m, const yk_multi_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                              const yk_integrator_desc* integrator, yk_multi_film* film, int64_t first_sample, uint32_t n_passes, float* film_rgb,
                              yk_render_stats* stats, yk_cancel_fn cancel, void* user) try {
    if (!m) return YK_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> l(m->mu);
    DeviceGuard restore;
    if (!scene || !camera || !sampler || !integrator || !film) return mfail(m, YK_ERR_INVALID_ARGUMENT, "null argument");
    if (scene->owner != m || film->owner != m) return mfail(m, YK_ERR_INVALID_ARGUMENT, "scene / film belong to another yk_multi");
    const bool accumulating = first_sample >= 0;
    if (n_passes == 0 || n_passes > 0xFFFFu || (!accumulating && n_passes != 1)) return mfail(m, YK_ERR_INVALID_ARGUMENT, "bad number of passes");
    const size_t G = m->ctx.size();
    const bool peer = m->peer_copy;
    const bool exchange = G > 1 || m->loopback;
    if (exchange && !peer) {
        yk_status st = ensure_comms(m);
        if (st != YK_OK) return st;
    }
    if (accumulating && n_passes > film->slab_passes) {  // slabs hold n_passes x the rank's pixels (pass-major): grow them with nothing in flight
        for (size_t r = 0; r < G; ++r) {
            (void)hipSetDevice(m->devices[r]);
            (void)hipStreamSynchronize((hipStream_t)yk_context_stream(m->ctx[r]));
        }
        for (size_t r = 0; r < G; ++r) {
            (void)hipSetDevice(m->devices[r]);
            if (film->slab[r].ensure(film->n_floats[r] * sizeof(float) * n_passes) != hipSuccess) return mfail(m, YK_ERR_OUT_OF_MEMORY, "slab");
            (void)hipSetDevice(m->devices[0]);
            if (film->gathered[r].ensure(film->n_floats[r] * sizeof(float) * n_passes) != hipSuccess) return mfail(m, YK_ERR_OUT_OF_MEMORY, "gather buffer");
        }
        film->slab_passes = n_passes;
    }
    // The user's predicate is called by one thread at a time and never after it fired (CancelLatch).  A device's
    // render polls it while the GPU works only when the call is synchronous (stats given, yk_render.cpp), so a
    // cancellable frame always asks for per-device stats.
    CancelLatch latch;
    latch.fn = cancel;
    latch.user = user;
    yk_cancel_fn dev_cancel = cancel ? &CancelLatch::poll : nullptr;
    void* dev_user = cancel ? &latch : nullptr;
    // (1) every device renders its tiles; the host threads only enqueue (no stats) or wait for their own device
    std::vector<yk_render_stats> per(G);
    for (size_t r = 0; r < G; ++r) {
        yk_context* c = m->ctx[r];
        const yk_scene* sc = scene->per_device[r];
        const yk_tile_list* tl = film->lists[r];
        void* dst = film->slab[r].p;
        yk_render_stats* ps = (stats || cancel) ? &per[r] : nullptr;
        if (peer && exchange) {  // the slab's previous content has left for device 0 before this render overwrites it
            (void)hipSetDevice(m->devices[r]);
            (void)hipStreamWaitEvent((hipStream_t)yk_context_stream(c), m->ev_copied[r], 0);
        }
        if (accumulating)
            m->workers[r]->post([=] { return yk_render_tile_list_samples_device(c, sc, camera, sampler, integrator, tl, (uint32_t)first_sample, n_passes, dst, nullptr, ps, dev_cancel, dev_user); });
        else
            m->workers[r]->post([=] { return yk_render_tile_list_device(c, sc, camera, sampler, integrator, tl, dst, nullptr, ps, dev_cancel, dev_user); });
    }
    yk_status first = YK_OK;
    for (size_t r = 0; r < G; ++r) {
        yk_status w = m->workers[r]->wait();
        if (w != YK_OK && first == YK_OK) first = mfail(m, w, "device " + std::to_string(m->devices[r]) + ": " + m->ctx[r]->last_error);
    }
    if (first != YK_OK) return first;
    if (latch.hit.load()) return mfail(m, YK_ERR_CANCELLED, "cancelled by early_termination_predicate");
    // (2) slabs -> device 0, ordered after the renders without host synchronisation:
    //     RCCL: one group of point-to-point calls on the contexts' own streams;
    //     peer copy: hipMemcpyPeerAsync on device 0's stream behind an event of the sender's stream.
    const size_t passes = accumulating ? n_passes : 1;
    hipStream_t s0 = (hipStream_t)yk_context_stream(m->ctx[0]);
    if (exchange && peer) {
        for (size_t k = m->loopback ? 0 : 1; k < G; ++k) {
            hipStream_t sk = (hipStream_t)yk_context_stream(m->ctx[k]);
            const size_t bytes = film->n_floats[k] * sizeof(float) * passes;
            (void)hipSetDevice(m->devices[k]);
            if (hipEventRecord(m->ev_rendered[k], sk) != hipSuccess) return mfail(m, YK_ERR_DEVICE, "hipEventRecord (slab rendered)");
            (void)hipSetDevice(m->devices[0]);
            if (hipStreamWaitEvent(s0, m->ev_rendered[k], 0) != hipSuccess) return mfail(m, YK_ERR_DEVICE, "hipStreamWaitEvent (slab rendered)");
            const hipError_t e = m->devices[k] == m->devices[0] ? hipMemcpyAsync(film->gathered[k].p, film->slab[k].p, bytes, hipMemcpyDeviceToDevice, s0)
                                                                : hipMemcpyPeerAsync(film->gathered[k].p, m->devices[0], film->slab[k].p, m->devices[k], bytes, s0);
            if (e != hipSuccess) return mfail(m, YK_ERR_DEVICE, std::string("slab copy: ") + hipGetErrorString(e));
            if (hipEventRecord(m->ev_copied[k], s0) != hipSuccess) return mfail(m, YK_ERR_DEVICE, "hipEventRecord (slab copied)");
        }
    } else if (exchange) {
        Rccl* r = rccl();
        yk_status st = rccl_check(m, r->GroupStart(), "ncclGroupStart");
        if (st != YK_OK) return st;
        for (size_t k = m->loopback ? 0 : 1; k < G && st == YK_OK; ++k) {
            (void)hipSetDevice(m->devices[k]);
            st = rccl_check(m, r->Send(film->slab[k].p, film->n_floats[k] * passes, kNcclFloat, 0, m->comms[k], (hipStream_t)yk_context_stream(m->ctx[k])), "ncclSend");
            if (st != YK_OK) break;
            (void)hipSetDevice(m->devices[0]);
            st = rccl_check(m, r->Recv(film->gathered[k].p, film->n_floats[k] * passes, kNcclFloat, (int)k, m->comms[0], s0), "ncclRecv");
        }
        yk_status ge = rccl_check(m, r->GroupEnd(), "ncclGroupEnd");
        if (st != YK_OK) return st;
        if (ge != YK_OK) return ge;
    }
    // (3) Film::update_tile for every tile, on device 0, behind the receives
    (void)hipSetDevice(m->devices[0]);
    for (size_t k = 0; k < G; ++k) {
        const void* src = (k == 0 && !m->loopback) ? film->slab[0].p : film->gathered[k].p;
        yk_status st = accumulating ? yk_film_accumulate_tile_list_passes_device(m->ctx[0], film->lists0[k], src, n_passes, film->res_x, film->res_y, film->film.p, nullptr)
                                    : yk_film_update_tile_list_device(m->ctx[0], film->lists0[k], src, film->res_x, film->res_y, film->film.p, nullptr, 0);
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

yk_status yk_multi_render_film(yk_multi* m, const yk_multi_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                               const yk_integrator_desc* integrator, yk_multi_film* film, float* film_rgb, yk_render_stats* stats, yk_cancel_fn cancel,
                               void* user) {
    return render_frame(m, scene, camera, sampler, integrator, film, -1, 1, film_rgb, stats, cancel, user);
}

yk_status yk_multi_accumulate_film(yk_multi* m, const yk_multi_scene* scene, const yk_camera* camera, const yk_sampler_desc* sampler,
                                   const yk_integrator_desc* integrator, yk_multi_film* film, uint32_t first_sample, uint32_t n_passes, float* film_rgb,
                                   yk_render_stats* stats, yk_cancel_fn cancel, void* user) {
    return render_frame(m, scene, camera, sampler, integrator, film, (int64_t)first_sample, n_passes, film_rgb, stats, cancel, user);
}

yk_status yk_multi_film_clear(yk_multi* m, yk_multi_film* film) {
    if (!m || !film || film->owner != m) return YK_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> l(m->mu);
    DeviceGuard restore;
    (void)hipSetDevice(m->devices[0]);
    if (hipMemsetAsync(film->film.p, 0, (size_t)film->res_x * film->res_y * 3 * sizeof(float), (hipStream_t)yk_context_stream(m->ctx[0])) != hipSuccess)
        return mfail(m, YK_ERR_DEVICE, "film clear");
    return YK_OK;
}

// yk_context_interrupt for every rank: takes no lock (a render call holds it for its whole duration) and makes no HIP call
yk_status yk_multi_interrupt(yk_multi* m) {
    if (!m) return YK_ERR_INVALID_ARGUMENT;
    for (yk_context* c : m->ctx) (void)yk_context_interrupt(c);
    return YK_OK;
}

yk_status yk_multi_sync(yk_multi* m) {
    if (!m) return YK_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> l(m->mu);
    DeviceGuard restore;
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
    DeviceGuard restore;
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
    DeviceGuard restore;
    (void)hipSetDevice(d->ctx->device);
    (void)hipDeviceSynchronize();
    if (r && d->comm) (void)r->CommDestroy(d->comm);
    delete d;
}

yk_status yk_dist_gather(yk_dist* d, const void* d_send, void* d_recv, size_t count, void* stream) {
    if (!d || !d_send || (d->rank == 0 && !d_recv)) return YK_ERR_INVALID_ARGUMENT;
    Rccl* r = rccl();
    yk_context* ctx = d->ctx;
    if (!r) return fail(ctx, YK_ERR_UNSUPPORTED, "RCCL could not be loaded (librccl.so.1)");
    DeviceGuard restore;
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
