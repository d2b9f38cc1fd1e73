// yuki_hip.hpp — header-only C++17 mirror of the reference's interface for the Path
// hot path, over the C ABI of yuki_hip.h.  Names follow yuki's Rust types:
//
//   FilmSettings / FilmTile / film_tiles()     yuki/src/film.rs:14-65,409-475
//   CameraParameters / FoV / Camera            yuki/src/camera.rs:19-114
//   SamplerType::Uniform / ::Stratified        yuki/src/sampling/mod.rs:16-31
//   IntegratorType::{Path,Whitted,...}(params)  yuki/src/integrators/mod.rs:33-53
//   Integrator::render(scene, camera, sampler, tile, tile_pixels) -> ray count
//                                              yuki/src/integrators/mod.rs:120-185
//   Scene                                      yuki/src/scene/mod.rs:41-49
//
// Where the reference panics (assert!/unwrap) this wrapper throws yuki::Error
// carrying the yk_status.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "yuki_hip.h"

namespace yuki {

struct Error : std::runtime_error {
    yk_status status;
    Error(yk_status s, const std::string& what) : std::runtime_error(what), status(s) {}
};

inline void check(yk_status s, const yk_context* ctx = nullptr) {
    if (s == YK_OK) return;
    std::string msg = yk_status_string(s);
    if (ctx) {
        char buf[512] = {0};
        if (yk_last_error(ctx, buf, sizeof(buf)) == YK_OK && buf[0]) msg += std::string(": ") + buf;
    }
    throw Error(s, msg);
}

// film.rs:14-39
struct FilmSettings {
    uint16_t res_x = 640, res_y = 480;
    uint16_t tile_dim = 16;
    bool clear = true, accumulate = false, sixteenth_res = false;
};
// film.rs:43-65
using FilmTile = yk_tile;

// film.rs:409-475 — clipped tiles in outward-spiral order
inline std::vector<FilmTile> film_tiles(const FilmSettings& fs) {
    std::vector<FilmTile> t(yk_film_tiles(fs.res_x, fs.res_y, fs.tile_dim, nullptr, 0));
    yk_film_tiles(fs.res_x, fs.res_y, fs.tile_dim, t.data(), t.size());
    return t;
}

enum class FoV { X = 0, Y = 1 };
// camera.rs:24-41
struct CameraParameters {
    std::array<float, 3> position{0, 0, 0}, target{0, 0, 0}, up{0, 1, 0};
    FoV fov_axis = FoV::X;
    float fov_degrees = 0.0f;
};
// camera.rs:19-22,52-102
struct Camera {
    yk_camera matrices;
    Camera(const CameraParameters& p, const FilmSettings& fs) {
        yk_camera_params cp{};
        for (int k = 0; k < 3; ++k) {
            cp.position[k] = p.position[k];
            cp.target[k] = p.target[k];
            cp.up[k] = p.up[k];
        }
        cp.fov_axis = (uint32_t)p.fov_axis;
        cp.fov_degrees = p.fov_degrees;
        cp.res_x = fs.res_x;
        cp.res_y = fs.res_y;
        check(yk_camera_init(&cp, &matrices));
    }
};

// sampling/mod.rs:16-31; the seed is explicit (default: the reference's commented debug seed, uniform.rs:35)
struct SamplerType {
    static constexpr uint64_t DEBUG_SEED = 0x73B9642E74AC471CULL;
    static yk_sampler_desc Uniform(uint32_t pixel_samples, uint64_t seed = DEBUG_SEED) { return yk_sampler_desc{YK_SAMPLER_UNIFORM, pixel_samples, 1, 1, seed}; }
    static yk_sampler_desc Stratified(uint32_t nx, uint32_t ny, bool jitter_samples = true, uint64_t seed = DEBUG_SEED) {
        return yk_sampler_desc{YK_SAMPLER_STRATIFIED, nx, ny, jitter_samples ? 1u : 0u, seed};
    }
};

// integrators/path.rs:20-32
struct PathParams {
    uint32_t max_depth = 3;
    bool has_indirect_clamp = false;
    float indirect_clamp = 0.0f;
};
// integrators/whitted.rs:17-25
struct WhittedParams {
    uint32_t max_depth = 3;
};
struct IntegratorType {
    static yk_integrator_desc Whitted(const WhittedParams& p = WhittedParams()) { return yk_integrator_desc{YK_INTEGRATOR_WHITTED, p.max_depth, 0u, 0.0f}; }
    static yk_integrator_desc Path(const PathParams& p = PathParams()) { return yk_integrator_desc{YK_INTEGRATOR_PATH, p.max_depth, p.has_indirect_clamp ? 1u : 0u, p.indirect_clamp}; }
    static yk_integrator_desc BVHIntersections() { return yk_integrator_desc{YK_INTEGRATOR_BVH_INTERSECTIONS, 1, 0, 0.0f}; }
    static yk_integrator_desc GeometryNormals() { return yk_integrator_desc{YK_INTEGRATOR_GEOMETRY_NORMALS, 1, 0, 0.0f}; }
    static yk_integrator_desc ShadingNormals() { return yk_integrator_desc{YK_INTEGRATOR_SHADING_NORMALS, 1, 0, 0.0f}; }
};

class Context {
   public:
    explicit Context(int device = 0) { check(yk_context_create(device, &h_)); }
    ~Context() { yk_context_destroy(h_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    yk_context* handle() const { return h_; }
    void set_option(const char* key, int64_t v) { check(yk_context_set_option(h_, key, v), h_); }
    // stop what the context has enqueued (any thread; yk_cancel_fn in yuki_hip.h)
    void interrupt() { check(yk_context_interrupt(h_), h_); }

   private:
    yk_context* h_ = nullptr;
};

// scene/mod.rs:41-49.  ctx == nullptr: host-only (BVH build / export, no GPU).
class Scene {
   public:
    Scene(Context* ctx, const yk_scene_desc& desc) : ctx_(ctx) { check(yk_scene_create(ctx ? ctx->handle() : nullptr, &desc, &h_), ctx ? ctx->handle() : nullptr); }
    ~Scene() { yk_scene_destroy(h_); }
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;
    yk_scene* handle() const { return h_; }
    yk_scene_info info() const {
        yk_scene_info i;
        check(yk_scene_get_info(h_, &i));
        return i;
    }
    std::pair<std::vector<yk_bvh_node>, std::vector<uint32_t>> export_bvh() const {
        yk_scene_info i = info();
        std::vector<yk_bvh_node> nodes(i.n_nodes);
        std::vector<uint32_t> order(i.n_shapes);
        check(yk_scene_export_bvh(h_, nodes.data(), order.data()));
        return {std::move(nodes), std::move(order)};
    }

   private:
    Context* ctx_;
    yk_scene* h_ = nullptr;
};

// scene::pbrt::load / Scene::ply (scene/pbrt/mod.rs:94, scene/mod.rs:99): the parsed scene as a
// ready yk_scene_desc plus the CameraParameters and FilmSettings the reference's loaders return
class LoadedScene {
   public:
    enum class Format { Ply, Pbrt };
    LoadedScene(const std::string& path, Format format, uint32_t split_method = YK_SPLIT_SAH, uint32_t max_shapes_in_node = 1) {
        yk_status st = format == Format::Ply ? yk_load_ply(path.c_str(), split_method, max_shapes_in_node, &h_) : yk_load_pbrt(path.c_str(), split_method, max_shapes_in_node, &h_);
        if (st != YK_OK) throw Error(st, yk_loader_last_error());
        uint16_t tile_dim = 16;
        yk_camera_params cp;
        check(yk_loaded_scene_get(h_, &desc, &cp, &tile_dim));
        camera.position = {cp.position[0], cp.position[1], cp.position[2]};
        camera.target = {cp.target[0], cp.target[1], cp.target[2]};
        camera.up = {cp.up[0], cp.up[1], cp.up[2]};
        camera.fov_axis = cp.fov_axis == 0 ? FoV::X : FoV::Y;
        camera.fov_degrees = cp.fov_degrees;
        film.res_x = cp.res_x;
        film.res_y = cp.res_y;
        film.tile_dim = tile_dim;
    }
    ~LoadedScene() { yk_loaded_scene_destroy(h_); }
    LoadedScene(const LoadedScene&) = delete;
    LoadedScene& operator=(const LoadedScene&) = delete;
    yk_scene_desc desc{};  // valid while this object lives; pass to Scene(ctx, desc)
    CameraParameters camera;
    FilmSettings film;

   private:
    yk_loaded_scene* h_ = nullptr;
};

// app/util.rs:90-111
inline void write_exr(const std::string& path, uint32_t width, uint32_t height, const float* rgb) { check(yk_write_exr(path.c_str(), width, height, rgb)); }

// trait Integrator, integrators/mod.rs:92-186
class Integrator {
   public:
    Integrator(Context& ctx, yk_integrator_desc desc) : ctx_(ctx), desc_(desc) {}
    // render(): one tile; tile_pixels holds >= tile area RGB triples; returns the ray count
    size_t render(const Scene& scene, const Camera& camera, const yk_sampler_desc& sampler, const FilmTile& tile, float* tile_pixels) const {
        uint64_t rays = 0;
        check(yk_render_tile(ctx_.handle(), scene.handle(), &camera.matrices, &sampler, &desc_, &tile, tile_pixels, &rays), ctx_.handle());
        return (size_t)rays;
    }
    // all tiles of a GPU worker in one submission; out_rgb tile-major
    yk_render_stats render_tiles(const Scene& scene, const Camera& camera, const yk_sampler_desc& sampler, const std::vector<FilmTile>& tiles, float* out_rgb,
                                 yk_cancel_fn cancel = nullptr, void* user = nullptr) const {
        yk_render_stats st{};
        check(yk_render_tiles(ctx_.handle(), scene.handle(), &camera.matrices, &sampler, &desc_, tiles.data(), tiles.size(), out_rgb, &st, cancel, user), ctx_.handle());
        return st;
    }

    // Integrator::render(accumulating = true): one sample (global index tile_samples[t]) per pixel, raw value;
    // n_passes > 1 renders samples tile_samples[t] .. + n_passes - 1 at once (out_rgb: n_passes x pixels x RGB, pass-major)
    yk_render_stats render_tiles_accumulating(const Scene& scene, const Camera& camera, const yk_sampler_desc& sampler, const std::vector<FilmTile>& tiles,
                                              const std::vector<uint16_t>& tile_samples, float* out_rgb, uint32_t n_passes = 1) const {
        if (tile_samples.size() != tiles.size()) throw Error(YK_ERR_INVALID_ARGUMENT, "one sample index per tile");
        yk_render_stats st{};
        check(yk_render_tiles_accumulating_passes(ctx_.handle(), scene.handle(), &camera.matrices, &sampler, &desc_, tiles.data(), tile_samples.data(), tiles.size(), n_passes,
                                                  out_rgb, &st, nullptr, nullptr),
              ctx_.handle());
        return st;
    }

   private:
    Context& ctx_;
    yk_integrator_desc desc_;
};

// The render workers' shared door to the device (render_manager.rs:78-97: num_cpus - 1 threads, each calling Integrator::render for
// one tile): calls that wait at the same time share a submission on one of the lanes' contexts (yk_combiner in yuki_hip.h).
class Combiner {
   public:
    explicit Combiner(const std::vector<Context*>& lanes, uint32_t max_tiles = 0, uint32_t linger_us = 100) {
        std::vector<yk_context*> h;
        for (Context* c : lanes) h.push_back(c->handle());
        check(yk_combiner_create(h.data(), (uint32_t)h.size(), max_tiles, linger_us, &c_));
    }
    ~Combiner() { yk_combiner_destroy(c_); }
    Combiner(const Combiner&) = delete;
    Combiner& operator=(const Combiner&) = delete;
    // Integrator::render for one FilmTile from a worker thread; accumulating_sample < 0: all samples of the pixel, the mean stored.
    // Returns the ray count (the submission's, shared out by tile area: exact in sum).  Throws Error(YK_ERR_CANCELLED) when the
    // caller's own predicate fired.
    size_t render(const Scene& scene, const Camera& camera, const yk_sampler_desc& sampler, const yk_integrator_desc& integrator, const FilmTile& tile,
                  float* tile_pixels, int32_t accumulating_sample = -1, yk_cancel_fn cancel = nullptr, void* user = nullptr) {
        yk_render_stats st{};
        const yk_status rc = yk_combiner_render_tile(c_, scene.handle(), &camera.matrices, &sampler, &integrator, &tile, accumulating_sample, tile_pixels, &st, cancel, user);
        if (rc != YK_OK) {
            char buf[512] = {0};
            yk_combiner_last_error(c_, buf, sizeof buf);
            throw Error(rc, buf);
        }
        return (size_t)st.rays;
    }
    yk_combiner_info info() const {
        yk_combiner_info i{};
        check(yk_combiner_get_info(c_, &i));
        return i;
    }

   private:
    yk_combiner* c_ = nullptr;
};

// All GPUs of the process behind one object — RenderManager's role for GPU workers
// (renderer/render_manager.rs:78-97: one worker per device; :206-210: interleaved tiles;
// film.rs:210-282: the write-back).  devices[0] assembles the film.
class Node {
   public:
    // flags: YK_MULTI_SHARED_DEVICES (ranks may name the same device: G ranks on one GPU), YK_MULTI_PEER_COPY (hipMemcpyPeerAsync instead of RCCL)
    explicit Node(const std::vector<int>& devices, uint32_t flags = 0) { check(yk_multi_create_ex(devices.data(), (uint32_t)devices.size(), flags, &m_)); }
    // the tiles of rank `rank` among `n_ranks` (render_manager.rs:206-210: spiral tile i -> rank i mod n_ranks); needs no device
    static std::vector<yk_tile> deal(const FilmSettings& fs, uint32_t n_ranks, uint32_t rank, uint64_t* pixels = nullptr) {
        std::vector<yk_tile> t(yk_multi_deal(fs.res_x, fs.res_y, fs.tile_dim, n_ranks, rank, nullptr, 0, nullptr));
        yk_multi_deal(fs.res_x, fs.res_y, fs.tile_dim, n_ranks, rank, t.data(), t.size(), pixels);
        return t;
    }
    ~Node() {
        if (film_) yk_multi_film_destroy(film_);
        if (scene_) yk_multi_scene_destroy(scene_);
        if (m_) yk_multi_destroy(m_);
    }
    Node(const Node&) = delete;
    Node& operator=(const Node&) = delete;
    uint32_t device_count() const { return yk_multi_device_count(m_); }
    void set_option(const char* key, int64_t value) { mcheck(yk_multi_set_option(m_, key, value)); }
    // BoundingVolumeHierarchy::new once, one copy per device
    void set_scene(const yk_scene_desc& d) {
        if (scene_) yk_multi_scene_destroy(scene_);
        scene_ = nullptr;
        mcheck(yk_multi_scene_create(m_, &d, &scene_));
    }
    void set_film(const FilmSettings& fs) {
        if (film_) yk_multi_film_destroy(film_);
        film_ = nullptr;
        fs_ = fs;
        mcheck(yk_multi_film_create(m_, fs.res_x, fs.res_y, fs.tile_dim, &film_));
    }
    // every tile of the film on its device, the exchange, Film::update_tile: row-major RGB, res_x * res_y * 3 floats
    yk_render_stats render_film(const Camera& camera, const yk_sampler_desc& sampler, const yk_integrator_desc& integrator, float* film_rgb, yk_cancel_fn cancel = nullptr,
                                void* user = nullptr) {
        yk_render_stats st{};
        mcheck(yk_multi_render_film(m_, scene_, &camera.matrices, &sampler, &integrator, film_, film_rgb, &st, cancel, user));
        return st;
    }
    // the accumulating film over all devices (film.rs:260-272): passes first_sample .. + n_passes - 1 of every tile, added on device 0
    void clear_film() { mcheck(yk_multi_film_clear(m_, film_)); }
    yk_render_stats accumulate_film(const Camera& camera, const yk_sampler_desc& sampler, const yk_integrator_desc& integrator, uint32_t first_sample, uint32_t n_passes,
                                    float* film_rgb, yk_cancel_fn cancel = nullptr, void* user = nullptr) {
        yk_render_stats st{};
        mcheck(yk_multi_accumulate_film(m_, scene_, &camera.matrices, &sampler, &integrator, film_, first_sample, n_passes, film_rgb, &st, cancel, user));
        return st;
    }

   private:
    void mcheck(yk_status s) const {
        if (s == YK_OK) return;
        char buf[512] = {0};
        yk_multi_last_error(m_, buf, sizeof(buf));
        throw Error(s, std::string(yk_status_string(s)) + (buf[0] ? std::string(": ") + buf : std::string()));
    }
    yk_multi* m_ = nullptr;
    yk_multi_scene* scene_ = nullptr;
    yk_multi_film* film_ = nullptr;
    FilmSettings fs_;
};

}  // namespace yuki
