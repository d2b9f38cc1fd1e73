"""Host-side mirror of the reference's interface for the Path hot path.

Names, argument meaning and error behaviour follow yuki's Rust types so that the
parity tests read like the reference's call sites:

    FilmSettings / FilmTile / film_tiles      yuki/src/film.rs:14-65,409-475
    CameraParameters / FoV / Camera           yuki/src/camera.rs:19-114
    SamplerType.Uniform / .Stratified         yuki/src/sampling/mod.rs:16-31
    IntegratorType.Path(PathParams) ...       yuki/src/integrators/mod.rs:33-53
    Integrator.render(scene, camera, sampler, tile) -> (tile_pixels, ray_count)
                                              yuki/src/integrators/mod.rs:120-185
    Scene                                     yuki/src/scene/mod.rs:41-49

Everything computes through libyuki_hip.so (the C ABI of include/yuki_hip.h);
this file only marshals arguments.  The reference panics on contract
violations; here they surface as YukiError carrying the yk_status.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _ffi, abi
from ._ffi import RenderStats, SceneInfo, YukiError, check, lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# --------------------------------------------------------------------------- film
@dataclass
class FilmSettings:
    """film.rs:14-39 (defaults :28-38)."""

    res: tuple = (640, 480)
    tile_dim: int = 16
    clear: bool = True
    accumulate: bool = False
    sixteenth_res: bool = False


@dataclass
class FilmTile:
    """film.rs:43-65 — `bb` is (x0, y0, x1, y1), max exclusive."""

    bb: tuple
    sample: int = 0

    def as_struct(self):
        return abi.Tile(*[int(v) for v in self.bb])


def film_tiles(settings: FilmSettings):
    """film.rs:409-475: clipped tiles in outward-spiral order (numpy TILE_DTYPE)."""
    L = lib()
    n = L.yk_film_tiles(settings.res[0], settings.res[1], settings.tile_dim, None, 0)
    t = np.zeros(n, dtype=abi.TILE_DTYPE)
    L.yk_film_tiles(settings.res[0], settings.res[1], settings.tile_dim, _p(t), n)
    return t


def multi_deal(settings: FilmSettings, n_ranks, rank):
    """yk_multi_deal: the tiles of `rank` among `n_ranks` (spiral tile i -> rank i mod n_ranks, render_manager.rs:206-210)
    and the rank's pixel count; needs no device."""
    L = lib()
    px = C.c_uint64(0)
    n = L.yk_multi_deal(settings.res[0], settings.res[1], settings.tile_dim, n_ranks, rank, None, 0, C.byref(px))
    t = np.zeros(n, dtype=abi.TILE_DTYPE)
    L.yk_multi_deal(settings.res[0], settings.res[1], settings.tile_dim, n_ranks, rank, _p(t), n, C.byref(px))
    return t, int(px.value)


def update_tiles(tiles, tile_rgb, res):
    """Film::update_tile (film.rs:210-282) for a list of tiles: tile-major -> row-major."""
    tiles = np.ascontiguousarray(tiles, dtype=abi.TILE_DTYPE)
    tile_rgb = np.ascontiguousarray(tile_rgb, dtype=np.float32)
    film = np.zeros((res[1], res[0], 3), dtype=np.float32)
    check(lib().yk_film_update_tiles(_p(tiles), len(tiles), _p(tile_rgb), res[0], res[1], _p(film)))
    return film


def accumulate_tiles(tiles, tile_rgb, film, tile_sample_counts=None):
    """Film::update_tile with accumulation on (film.rs:260-272): film += tile pixels, in place;
    tile_sample_counts[t] += 1."""
    tiles = np.ascontiguousarray(tiles, dtype=abi.TILE_DTYPE)
    tile_rgb = np.ascontiguousarray(tile_rgb, dtype=np.float32)
    assert film.dtype == np.float32 and film.flags["C_CONTIGUOUS"]
    if tile_sample_counts is not None:
        assert tile_sample_counts.dtype == np.uint32 and len(tile_sample_counts) == len(tiles)
    check(lib().yk_film_accumulate_tiles(_p(tiles), len(tiles), _p(tile_rgb), film.shape[1], film.shape[0], _p(film), _p(tile_sample_counts)))
    return film


def write_exr(path, film):
    """app/util.rs:90-111 write_exr: (h, w, 3) float32 -> RGB OpenEXR file."""
    film = np.ascontiguousarray(film, dtype=np.float32)
    check(lib().yk_write_exr(str(path).encode(), film.shape[1], film.shape[0], _p(film)))


def write_pfm(path, film):
    film = np.ascontiguousarray(film, dtype=np.float32)
    check(lib().yk_write_pfm(str(path).encode(), film.shape[1], film.shape[0], _p(film)))


class TileList:
    """A tile list prepared once on the device (yk_tile_list): the GPU worker's tile queue."""

    def __init__(self, ctx, tiles, tile_samples=None):
        self.ctx = ctx
        self.tiles = np.ascontiguousarray(tiles, dtype=abi.TILE_DTYPE)
        ts = None if tile_samples is None else np.ascontiguousarray(tile_samples, dtype=np.uint16)
        h = C.c_void_p()
        check(lib().yk_tile_list_create(ctx.h, _p(self.tiles), _p(ts), len(self.tiles), C.byref(h)), ctx.h)
        self.h = h
        self.n_pixels = int(((self.tiles["x1"].astype(np.int64) - self.tiles["x0"]) * (self.tiles["y1"].astype(np.int64) - self.tiles["y0"])).sum())

    def update_film_device(self, d_tile_rgb_ptr, res, d_film_ptr, stream=None, accumulate=False, ctx=None, n_passes=1):
        """Film::update_tile for the whole list, device to device, enqueued on `stream` (default:
        the stream of `ctx`, any context on the list's device; default the one that made it).
        n_passes > 1: `d_tile_rgb_ptr` holds that many passes (pass-major), added one after the other."""
        c = ctx or self.ctx
        if n_passes != 1:
            if not accumulate:
                raise ValueError("several passes only make sense for the accumulating film")
            check(lib().yk_film_accumulate_tile_list_passes_device(c.h, self.h, C.c_void_p(d_tile_rgb_ptr), n_passes, res[0], res[1], C.c_void_p(d_film_ptr), C.c_void_p(stream) if stream else None), c.h)
            return
        check(lib().yk_film_update_tile_list_device(c.h, self.h, C.c_void_p(d_tile_rgb_ptr), res[0], res[1], C.c_void_p(d_film_ptr), C.c_void_p(stream) if stream else None, 1 if accumulate else 0), c.h)

    def close(self):
        if getattr(self, "h", None):
            lib().yk_tile_list_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# --------------------------------------------------------------------------- camera
class FoV:
    X, Y = abi.FOV_X, abi.FOV_Y


@dataclass
class CameraParameters:
    """camera.rs:24-41."""

    position: tuple = (0.0, 0.0, 0.0)
    target: tuple = (0.0, 0.0, 0.0)
    up: tuple = (0.0, 1.0, 0.0)
    fov_axis: int = FoV.X
    fov_degrees: float = 0.0


class Camera:
    """camera.rs:19-22,52-102: built on the host by yk_camera_init."""

    def __init__(self, params, film_settings):
        if isinstance(params, dict):
            params = CameraParameters(**params)
        p = abi.CameraParams()
        p.position = abi.f3(params.position)
        p.target = abi.f3(params.target)
        p.up = abi.f3(params.up)
        p.fov_axis = params.fov_axis
        p.fov_degrees = params.fov_degrees
        p.res_x, p.res_y = film_settings.res
        self.matrices = abi.CameraMatrices()
        check(lib().yk_camera_init(C.byref(p), C.byref(self.matrices)))


# --------------------------------------------------------------------------- sampler / integrator descriptions
class SamplerType:
    """sampling/mod.rs:16-31.  `seed` is explicit (the reference draws it from
    thread_rng(): uniform.rs:37); the default is the reference's commented debug seed."""

    DEBUG_SEED = 0x73B9642E74AC471C

    @staticmethod
    def Uniform(pixel_samples=1, seed=DEBUG_SEED):
        return abi.SamplerDesc(abi.SAMPLER_UNIFORM, pixel_samples, 1, 1, seed)

    @staticmethod
    def Stratified(pixel_samples=(1, 1), jitter_samples=True, seed=DEBUG_SEED):
        return abi.SamplerDesc(abi.SAMPLER_STRATIFIED, pixel_samples[0], pixel_samples[1], 1 if jitter_samples else 0, seed)


def samples_per_pixel(sampler):
    return sampler.nx if sampler.kind == abi.SAMPLER_UNIFORM else sampler.nx * sampler.ny


@dataclass
class PathParams:
    """integrators/path.rs:20-32."""

    max_depth: int = 3
    indirect_clamp: float = None


class LightFactory:
    """RectangularLight::new / SpotLight::new / PointLight::new on the host."""

    @staticmethod
    def make_rect_light(l2w, l2w_inv, L, size, out):
        check(lib().yk_make_rect_light(abi.f16(l2w), abi.f16(l2w_inv), abi.f3(L), (C.c_float * 2)(*[float(s) for s in size]), C.byref(out)))

    @staticmethod
    def make_spot_light(l2w, l2w_inv, I, total, falloff, out):
        check(lib().yk_make_spot_light(abi.f16(l2w), abi.f16(l2w_inv), abi.f3(I), float(total), float(falloff), C.byref(out)))

    @staticmethod
    def make_point_light(l2w, I, out):
        check(lib().yk_make_point_light(abi.f16(l2w), abi.f3(I), C.byref(out)))


# --------------------------------------------------------------------------- context / scene
class Context:
    """One HIP device + stream + work buffers (yk_context)."""

    def __init__(self, device=0, **options):
        h = C.c_void_p()
        check(lib().yk_context_create(device, C.byref(h)))
        self.h = h
        self.device = device
        for k, v in options.items():
            self.set_option(k, v)

    def set_option(self, key, value):
        check(lib().yk_context_set_option(self.h, key.encode(), int(value)), self.h)

    def interrupt(self):
        """yk_context_interrupt: stop what the context has enqueued (any thread)."""
        check(lib().yk_context_interrupt(self.h))

    @property
    def stream_handle(self):
        """The context's hipStream_t as an integer (yk_context_stream): wrap it, e.g. with
        torch.cuda.ExternalStream, to order other device work after a render."""
        return int(lib().yk_context_stream(self.h) or 0)

    def close(self):
        if getattr(self, "h", None):
            lib().yk_context_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scene:
    """scene/mod.rs:41-49: shapes + BVH + lights + background.  `ctx=None` builds
    the BVH on the host only (no GPU needed)."""

    def __init__(self, ctx, scene_data):
        self.ctx = ctx
        self.data = scene_data
        d, self._keep = scene_data.desc(LightFactory)
        h = C.c_void_p()
        check(lib().yk_scene_create(ctx.h if ctx else None, C.byref(d), C.byref(h)), ctx.h if ctx else None)
        self.h = h
        self._keep = None  # the library copied everything it needs

    def close(self):
        if getattr(self, "h", None):
            lib().yk_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        i = SceneInfo()
        check(lib().yk_scene_get_info(self.h, C.byref(i)))
        return i

    def export_bvh(self):
        i = self.info()
        nodes = np.zeros(i.n_nodes, dtype=abi.BVH_NODE_DTYPE)
        order = np.zeros(i.n_shapes, dtype=np.uint32)
        check(lib().yk_scene_export_bvh(self.h, _p(nodes), _p(order)))
        return nodes, order

    # per-stage entry points ---------------------------------------------------
    def intersect(self, o, d, t_max=None, counters=False):
        """BoundingVolumeHierarchy::intersect for n rays (bvh.rs:160-232)."""
        o = np.ascontiguousarray(o, dtype=np.float32)
        d = np.ascontiguousarray(d, dtype=np.float32)
        n = o.shape[0]
        tm = None if t_max is None else np.ascontiguousarray(t_max, dtype=np.float32)
        r = dict(shape=np.zeros(n, dtype=np.int32), t=np.zeros(n, dtype=np.float32), bary=np.zeros((n, 3), dtype=np.float32))
        if counters:
            r.update(node_tests=np.zeros(n, dtype=np.uint32), node_hits=np.zeros(n, dtype=np.uint32), shape_tests=np.zeros(n, dtype=np.uint32))
        check(
            lib().yk_trace_closest(self.ctx.h, self.h, n, _p(o), _p(d), _p(tm), _p(r["shape"]), _p(r["t"]), _p(r["bary"]), _p(r.get("node_tests")), _p(r.get("node_hits")), _p(r.get("shape_tests"))),
            self.ctx.h,
        )
        return r

    def any_intersect(self, o, d, t_max, area_light=None):
        """BoundingVolumeHierarchy::any_intersect (bvh.rs:235-302)."""
        o = np.ascontiguousarray(o, dtype=np.float32)
        d = np.ascontiguousarray(d, dtype=np.float32)
        tm = np.ascontiguousarray(t_max, dtype=np.float32)
        al = None if area_light is None else np.ascontiguousarray(area_light, dtype=np.int32)
        out = np.zeros(o.shape[0], dtype=np.uint8)
        check(lib().yk_trace_any(self.ctx.h, self.h, o.shape[0], _p(o), _p(d), _p(tm), _p(al), _p(out)), self.ctx.h)
        return out


# --------------------------------------------------------------------------- several GPUs
def _mcheck(status, m):
    if status != _ffi.YK_OK:
        buf = C.create_string_buffer(512)
        lib().yk_multi_last_error(m, buf, 512)
        raise YukiError(status, buf.value.decode(errors="replace") or lib().yk_status_string(status).decode())


class Multi:
    """The GPUs of one process (yk_multi): a context and a host thread per device; tiles of the
    film's spiral are dealt round-robin (render_manager.rs:206-210), the slabs meet on devices[0]
    through RCCL and Film::update_tile runs there."""

    SHARED_DEVICES, PEER_COPY = 1, 2  # yk_multi_create_ex flags

    def __init__(self, devices, flags=0, **options):
        """flags: Multi.SHARED_DEVICES — ranks may name the same device (G ranks on one GPU: what a one-GPU box can run of
        the G > 1 path; the slabs travel by device copies); Multi.PEER_COPY — hipMemcpyPeerAsync instead of RCCL."""
        devices = [int(d) for d in devices]
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        check(lib().yk_multi_create_ex(arr, len(devices), int(flags), C.byref(h)) if flags else lib().yk_multi_create(arr, len(devices), C.byref(h)))
        self.h = h
        self.devices = devices
        for k, v in options.items():
            self.set_option(k, v)

    def set_option(self, key, value):
        _mcheck(lib().yk_multi_set_option(self.h, key.encode(), int(value)), self.h)

    def scene(self, scene_data):
        return MultiScene(self, scene_data)

    def film(self, settings):
        return MultiFilm(self, settings)

    def render_film(self, scene, camera, sampler, integrator, film, want_host=True, want_stats=True, cancel=None):
        """Integrator::render for every tile of the film on its device, exchange, Film::update_tile.
        Returns (film (h, w, 3) float32 or None, RenderStats or None)."""
        out = np.zeros((film.res[1], film.res[0], 3), dtype=np.float32) if want_host else None
        stats = RenderStats() if want_stats else None
        cb = _ffi.CANCEL_FN(lambda _u: 1 if cancel() else 0) if cancel else None
        _mcheck(
            lib().yk_multi_render_film(self.h, scene.h, C.byref(camera.matrices), C.byref(sampler), C.byref(integrator), film.h, _p(out), C.byref(stats) if want_stats else None, C.cast(cb, C.c_void_p) if cb else None, None),
            self.h,
        )
        return out, stats

    def accumulate_film(self, scene, camera, sampler, integrator, film, first_sample, n_passes=1, want_host=True, want_stats=True, cancel=None):
        """The accumulating film over all devices (yk_multi_accumulate_film): passes first_sample .. + n_passes - 1 of every
        tile, each added to the film on device 0 (film.rs:260-272).  clear_film() starts a new accumulation."""
        out = np.zeros((film.res[1], film.res[0], 3), dtype=np.float32) if want_host else None
        stats = RenderStats() if want_stats else None
        cb = _ffi.CANCEL_FN(lambda _u: 1 if cancel() else 0) if cancel else None
        _mcheck(
            lib().yk_multi_accumulate_film(self.h, scene.h, C.byref(camera.matrices), C.byref(sampler), C.byref(integrator), film.h, int(first_sample), int(n_passes), _p(out), C.byref(stats) if want_stats else None, C.cast(cb, C.c_void_p) if cb else None, None),
            self.h,
        )
        return out, stats

    def clear_film(self, film):
        _mcheck(lib().yk_multi_film_clear(self.h, film.h), self.h)

    def interrupt(self):
        """yk_multi_interrupt: stop the frame in flight on every rank (any thread)."""
        check(lib().yk_multi_interrupt(self.h))

    def sync(self):
        _mcheck(lib().yk_multi_sync(self.h), self.h)

    def close(self):
        if getattr(self, "h", None):
            lib().yk_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiScene:
    def __init__(self, multi, scene_data):
        self.multi = multi
        d, keep = scene_data.desc(LightFactory)
        h = C.c_void_p()
        _mcheck(lib().yk_multi_scene_create(multi.h, C.byref(d), C.byref(h)), multi.h)
        self.h = h

    def info(self):
        i = SceneInfo()
        check(lib().yk_multi_scene_get_info(self.h, C.byref(i)))
        return i

    def close(self):
        if getattr(self, "h", None):
            lib().yk_multi_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiFilm:
    def __init__(self, multi, settings):
        self.multi = multi
        self.res = tuple(settings.res)
        h = C.c_void_p()
        _mcheck(lib().yk_multi_film_create(multi.h, settings.res[0], settings.res[1], settings.tile_dim, C.byref(h)), multi.h)
        self.h = h

    @property
    def device_ptr(self):
        return int(lib().yk_multi_film_device_ptr(self.h) or 0)

    def close(self):
        if getattr(self, "h", None):
            lib().yk_multi_film_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Dist:
    """One process per GPU (yk_dist): join an RCCL communicator with this rank's context; gather()
    moves every rank's slab into rank 0's buffer on the context's stream."""

    ID_BYTES = 128  # YK_DIST_ID_BYTES

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        check(lib().yk_dist_unique_id(buf))
        return bytes(buf)

    def __init__(self, ctx, unique_id, rank, world):
        self.ctx = ctx
        idb = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        h = C.c_void_p()
        check(lib().yk_dist_create(ctx.h, idb, rank, world, C.byref(h)), ctx.h)
        self.h, self.rank, self.world = h, rank, world

    def gather(self, d_send_ptr, d_recv_ptr, count_floats, stream=None):
        check(lib().yk_dist_gather(self.h, C.c_void_p(d_send_ptr), C.c_void_p(d_recv_ptr) if d_recv_ptr else None, count_floats, C.c_void_p(stream) if stream else None), self.ctx.h)

    def close(self):
        if getattr(self, "h", None):
            lib().yk_dist_destroy(self.h)
            self.h = None


# --------------------------------------------------------------------------- many workers, one device
class Combiner:
    """yk_combiner: Integrator::render called per tile from many worker threads (render_manager.rs:78-97), merged into shared
    submissions on `contexts` (one lane each).  `render` is what a worker thread calls; it blocks until its tile is done."""

    def __init__(self, contexts, max_tiles=0, linger_us=100):
        self.contexts = list(contexts)
        arr = (C.c_void_p * len(self.contexts))(*[c.h for c in self.contexts])
        h = C.c_void_p()
        check(lib().yk_combiner_create(arr, len(self.contexts), max_tiles, linger_us, C.byref(h)))
        self.h = h

    def _error(self, status):
        buf = C.create_string_buffer(512)
        lib().yk_combiner_last_error(self.h, buf, 512)
        return YukiError(status, buf.value.decode(errors="replace"))

    def render(self, integrator, scene, camera, sampler, tile, accumulating=False, cancel=None):
        """(tile_pixels[h*w,3], RenderStats) — integrators/mod.rs:120-185 for one FilmTile."""
        t = tile.as_struct() if isinstance(tile, FilmTile) else abi.Tile(*[int(v) for v in tile])
        w, h = t.x1 - t.x0, t.y1 - t.y0
        px = np.zeros((max(w, 0) * max(h, 0), 3), dtype=np.float32)
        stats = RenderStats()
        desc = integrator.desc if isinstance(integrator, Integrator) else integrator
        sample = (tile.sample if isinstance(tile, FilmTile) else 0) if accumulating else -1
        cb = _ffi.CANCEL_FN(lambda _u: 1 if cancel() else 0) if cancel else None
        st = lib().yk_combiner_render_tile(self.h, scene.h, C.byref(camera.matrices), C.byref(sampler), C.byref(desc), C.byref(t), sample, _p(px), C.byref(stats),
                                           C.cast(cb, C.c_void_p) if cb else None, None)
        if st != 0:
            raise self._error(st)
        return px, stats

    def info(self):
        out = _ffi.CombinerInfo()
        check(lib().yk_combiner_get_info(self.h, C.byref(out)))
        return out

    def close(self):
        if getattr(self, "h", None):
            lib().yk_combiner_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# --------------------------------------------------------------------------- integrators
class Integrator:
    """trait Integrator (integrators/mod.rs:92-186) over the HIP wavefront."""

    def __init__(self, ctx, desc):
        self.ctx = ctx
        self.desc = desc

    def render(self, scene, camera, sampler, tile, accumulating=False):
        """One tile; returns (tile_pixels[h*w,3], ray_count) — integrators/mod.rs:120-185."""
        t = tile.as_struct() if isinstance(tile, FilmTile) else abi.Tile(*[int(v) for v in tile])
        w, h = t.x1 - t.x0, t.y1 - t.y0
        if w <= 0 or h <= 0:
            raise YukiError(1, "Bounds2 with a dimension <= 0")
        if accumulating:  # one sample with global index tile.sample, raw value (integrators/mod.rs:146-161)
            sample = tile.sample if isinstance(tile, FilmTile) else 0
            tiles = np.array([(t.x0, t.y0, t.x1, t.y1)], dtype=abi.TILE_DTYPE)
            px, stats = self.render_tiles_accumulating(scene, camera, sampler, tiles, [sample])
            return px, stats.rays
        px = np.zeros((w * h, 3), dtype=np.float32)
        rays = C.c_uint64(0)
        check(lib().yk_render_tile(self.ctx.h, scene.h, C.byref(camera.matrices), C.byref(sampler), C.byref(self.desc), C.byref(t), _p(px), C.byref(rays)), self.ctx.h)
        return px, rays.value

    def render_tiles(self, scene, camera, sampler, tiles, cancel=None):
        """All tiles of one call as a single batch — the GPU-worker entry point.
        Returns (rgb tile-major [n_pixels,3], RenderStats)."""
        tiles = np.ascontiguousarray(tiles, dtype=abi.TILE_DTYPE)
        npx = int(((tiles["x1"].astype(np.int64) - tiles["x0"]) * (tiles["y1"].astype(np.int64) - tiles["y0"])).sum())
        out = np.zeros((npx, 3), dtype=np.float32)
        stats = RenderStats()
        cb = _ffi.CANCEL_FN(lambda _u: 1 if cancel() else 0) if cancel else None
        check(
            lib().yk_render_tiles(self.ctx.h, scene.h, C.byref(camera.matrices), C.byref(sampler), C.byref(self.desc), _p(tiles), len(tiles), _p(out), C.byref(stats), C.cast(cb, C.c_void_p) if cb else None, None),
            self.ctx.h,
        )
        return out, stats

    def render_tiles_accumulating(self, scene, camera, sampler, tiles, tile_samples, cancel=None, n_passes=1):
        """Integrator::render(accumulating=true) (integrators/mod.rs:146-161) for a list of
        (tile, FilmTile.sample) pairs: one sample per pixel, raw value.  Returns (rgb, stats);
        n_passes > 1 renders passes sample .. sample + n_passes - 1 at once: rgb is (n_passes, pixels, 3)."""
        tiles = np.ascontiguousarray(tiles, dtype=abi.TILE_DTYPE)
        ts = np.ascontiguousarray(tile_samples, dtype=np.uint16)
        if len(ts) != len(tiles):
            raise ValueError("one sample index per tile")
        npx = int(((tiles["x1"].astype(np.int64) - tiles["x0"]) * (tiles["y1"].astype(np.int64) - tiles["y0"])).sum())
        stats = RenderStats()
        cb = _ffi.CANCEL_FN(lambda _u: 1 if cancel() else 0) if cancel else None
        if n_passes != 1:
            out = np.zeros((n_passes, npx, 3), dtype=np.float32)
            check(
                lib().yk_render_tiles_accumulating_passes(self.ctx.h, scene.h, C.byref(camera.matrices), C.byref(sampler), C.byref(self.desc), _p(tiles), _p(ts), len(tiles), n_passes, _p(out), C.byref(stats), C.cast(cb, C.c_void_p) if cb else None, None),
                self.ctx.h,
            )
            return out, stats
        out = np.zeros((npx, 3), dtype=np.float32)
        check(
            lib().yk_render_tiles_accumulating(self.ctx.h, scene.h, C.byref(camera.matrices), C.byref(sampler), C.byref(self.desc), _p(tiles), _p(ts), len(tiles), _p(out), C.byref(stats), C.cast(cb, C.c_void_p) if cb else None, None),
            self.ctx.h,
        )
        return out, stats

    def render_tiles_device(self, scene, camera, sampler, tiles, d_out_ptr, stream=None, want_stats=True):
        """Radiance stays in HBM at `d_out_ptr` (e.g. a torch tensor's data_ptr())."""
        tiles = np.ascontiguousarray(tiles, dtype=abi.TILE_DTYPE)
        stats = RenderStats()
        check(
            lib().yk_render_tiles_device(self.ctx.h, scene.h, C.byref(camera.matrices), C.byref(sampler), C.byref(self.desc), _p(tiles), len(tiles), C.c_void_p(d_out_ptr), C.c_void_p(stream) if stream else None, C.byref(stats) if want_stats else None, None, None),
            self.ctx.h,
        )
        return stats

    def render_tile_list_samples_device(self, scene, camera, sampler, tile_list, first_sample, n_passes, d_out_ptr, stream=None, want_stats=False, cancel=None):
        """Passes first_sample .. first_sample + n_passes - 1 of every tile of a PLAIN tile list (the worker's accumulate loop,
        render_manager.rs:125-143), pass-major into HBM at `d_out_ptr`."""
        stats = RenderStats()
        cb = _ffi.CANCEL_FN(lambda _u: 1 if cancel() else 0) if cancel else None
        check(
            lib().yk_render_tile_list_samples_device(self.ctx.h, scene.h, C.byref(camera.matrices), C.byref(sampler), C.byref(self.desc), tile_list.h, int(first_sample), int(n_passes), C.c_void_p(d_out_ptr), C.c_void_p(stream) if stream else None, C.byref(stats) if want_stats else None, C.cast(cb, C.c_void_p) if cb else None, None),
            self.ctx.h,
        )
        return stats if want_stats else None

    def render_tile_list_device(self, scene, camera, sampler, tile_list, d_out_ptr, stream=None, want_stats=False, n_passes=1):
        """Render a prepared TileList into HBM at `d_out_ptr`; with want_stats=False the call only
        enqueues work on `stream` (no host synchronisation).  n_passes > 1 (accumulating list):
        that many passes at once, pass-major in `d_out_ptr`."""
        stats = RenderStats()
        if n_passes != 1:
            check(
                lib().yk_render_tile_list_passes_device(self.ctx.h, scene.h, C.byref(camera.matrices), C.byref(sampler), C.byref(self.desc), tile_list.h, n_passes, C.c_void_p(d_out_ptr), C.c_void_p(stream) if stream else None, C.byref(stats) if want_stats else None, None, None),
                self.ctx.h,
            )
            return stats if want_stats else None
        check(
            lib().yk_render_tile_list_device(self.ctx.h, scene.h, C.byref(camera.matrices), C.byref(sampler), C.byref(self.desc), tile_list.h, C.c_void_p(d_out_ptr), C.c_void_p(stream) if stream else None, C.byref(stats) if want_stats else None, None, None),
            self.ctx.h,
        )
        return stats if want_stats else None

    def li(self, scene, sampler, ray_o, ray_d, pixel_xy, sample_index, dimension=2):
        """Integrator::li for caller-supplied rays (integrators/mod.rs:94-101)."""
        o = np.ascontiguousarray(ray_o, dtype=np.float32)
        d = np.ascontiguousarray(ray_d, dtype=np.float32)
        pix = np.ascontiguousarray(pixel_xy, dtype=np.uint16)
        si = np.ascontiguousarray(sample_index, dtype=np.uint32)
        out = np.zeros((o.shape[0], 3), dtype=np.float32)
        check(lib().yk_li(self.ctx.h, scene.h, C.byref(sampler), C.byref(self.desc), o.shape[0], _p(o), _p(d), _p(pix), _p(si), dimension, _p(out), None), self.ctx.h)
        return out


class IntegratorType:
    """integrators/mod.rs:33-53."""

    @staticmethod
    def Path(params: PathParams = None):
        params = params or PathParams()
        return abi.IntegratorDesc(abi.INTEGRATOR_PATH, params.max_depth, 0 if params.indirect_clamp is None else 1, 0.0 if params.indirect_clamp is None else params.indirect_clamp)

    @staticmethod
    def Whitted(max_depth=3):
        return abi.IntegratorDesc(abi.INTEGRATOR_WHITTED, max_depth, 0, 0.0)

    BVHIntersections = abi.IntegratorDesc(abi.INTEGRATOR_BVH_INTERSECTIONS, 1, 0, 0.0)
    GeometryNormals = abi.IntegratorDesc(abi.INTEGRATOR_GEOMETRY_NORMALS, 1, 0, 0.0)
    ShadingNormals = abi.IntegratorDesc(abi.INTEGRATOR_SHADING_NORMALS, 1, 0, 0.0)

    @staticmethod
    def instantiate(ctx, desc):
        return Integrator(ctx, desc)


# --------------------------------------------------------------------------- misc stage hooks
def device_math(ctx, fn, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    bb = None if b is None else np.ascontiguousarray(b, dtype=np.float32)
    out = np.zeros_like(a)
    check(lib().yk_device_math(ctx.h, fn, a.size, _p(a), _p(bb), _p(out)), ctx.h)
    return out


def host_math(fn, a, b=None):
    """yk_host_math: the host instance of yk_libm.h (no device needed)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    bb = None if b is None else np.ascontiguousarray(b, dtype=np.float32)
    out = np.zeros_like(a)
    check(lib().yk_host_math(fn, a.size, _p(a), _p(bb), _p(out)))
    return out


def sampler_sequence(ctx, sampler, px, py, sample_index, dims):
    dims = np.ascontiguousarray(dims, dtype=np.uint8)
    out = np.zeros((len(dims), 2), dtype=np.float32)
    check(lib().yk_sampler_sequence(ctx.h, C.byref(sampler), px, py, sample_index, _p(dims), len(dims), _p(out)), ctx.h)
    return out


def camera_rays(ctx, camera, sampler, tile, sample_index):
    t = abi.Tile(*[int(v) for v in tile])
    n = (t.x1 - t.x0) * (t.y1 - t.y0)
    o = np.zeros((n, 3), dtype=np.float32)
    d = np.zeros((n, 3), dtype=np.float32)
    check(lib().yk_camera_rays(ctx.h, C.byref(camera.matrices), C.byref(sampler), C.byref(t), sample_index, _p(o), _p(d)), ctx.h)
    return o, d


def bsdf_eval(ctx, material, n_geom, n_shading, dpdu, wo, wi):
    arrs = [np.ascontiguousarray(x, dtype=np.float32) for x in (n_geom, n_shading, dpdu, wo, wi)]
    out = np.zeros((arrs[0].shape[0], 3), dtype=np.float32)
    check(lib().yk_bsdf_eval(ctx.h, C.byref(material), arrs[0].shape[0], *[_p(x) for x in arrs], _p(out)), ctx.h)
    return out


def light_sample(ctx, light, light_index, p, n_geom, u):
    """Light::sample_li + VisibilityTester::ray on the device: (n, 18) floats, see yk_light_sample"""
    arrs = [np.ascontiguousarray(x, dtype=np.float32) for x in (p, n_geom, u)]
    out = np.zeros((arrs[0].shape[0], 18), dtype=np.float32)
    check(lib().yk_light_sample(ctx.h, C.byref(light), int(light_index), arrs[0].shape[0], *[_p(x) for x in arrs], _p(out)), ctx.h)
    return out


def bsdf_sample(ctx, material, n_geom, n_shading, dpdu, wo, u):
    arrs = [np.ascontiguousarray(x, dtype=np.float32) for x in (n_geom, n_shading, dpdu, wo, u)]
    out = np.zeros((arrs[0].shape[0], 8), dtype=np.float32)
    check(lib().yk_bsdf_sample(ctx.h, C.byref(material), arrs[0].shape[0], *[_p(x) for x in arrs], _p(out)), ctx.h)
    return out
