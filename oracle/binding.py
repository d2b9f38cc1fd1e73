"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes binding of liboracle.so (the CPU restatement of the reference's Path
integrator hot path).  Imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg only; nothing under yuki_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from yuki_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}
_FLAVOUR = "default"
_FILES = {"default": "liboracle.so", "hostlibm": "liboracle_hostlibm.so", "nth": "liboracle_nth.so"}


def build(target="liboracle.so"):
    subprocess.check_call(["make", "-s", "-C", _HERE, target])


class flavour:
    """`with binding.flavour("hostlibm"):` — every oracle call inside goes to a sensitivity build
    (oracle/Makefile `flavours`).  Scenes must be created and closed inside the same block.  The
    default flavour is the parity oracle; the others only measure how far it could be from a
    glibc / pdqselect build of the reference (tools/libm_sensitivity.py)."""

    def __init__(self, name):
        assert name in _FILES, name
        self.name = name

    def __enter__(self):
        global _FLAVOUR
        self.prev, _FLAVOUR = _FLAVOUR, self.name
        return self

    def __exit__(self, *a):
        global _FLAVOUR
        _FLAVOUR = self.prev


def lib():
    if _FLAVOUR in _LIBS:
        return _LIBS[_FLAVOUR]
    path = os.path.join(_HERE, _FILES[_FLAVOUR])
    if not os.path.exists(path):
        build(_FILES[_FLAVOUR])
    L = C.CDLL(path)
    vp = C.c_void_p
    L.orc_scene_create.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(vp)]
    L.orc_scene_create.restype = C.c_int
    L.orc_scene_destroy.argtypes = [vp]
    L.orc_scene_destroy.restype = None
    L.orc_scene_node_count.argtypes = [vp]
    L.orc_scene_node_count.restype = C.c_size_t
    L.orc_scene_shape_count.argtypes = [vp]
    L.orc_scene_shape_count.restype = C.c_size_t
    L.orc_scene_export_bvh.argtypes = [vp, vp, vp]
    L.orc_camera_make.argtypes = [C.POINTER(abi.CameraParams), C.POINTER(abi.CameraMatrices)]
    L.orc_film_tiles.argtypes = [C.c_uint16, C.c_uint16, C.c_uint16, vp, C.c_size_t]
    L.orc_film_tiles.restype = C.c_size_t
    L.orc_make_rect_light.argtypes = [abi.f32p, abi.f32p, abi.f32p, abi.f32p, C.POINTER(abi.LightDesc)]
    L.orc_make_rect_light.restype = None
    L.orc_make_spot_light.argtypes = [abi.f32p, abi.f32p, abi.f32p, C.c_float, C.c_float, C.POINTER(abi.LightDesc)]
    L.orc_make_spot_light.restype = None
    L.orc_make_point_light.argtypes = [abi.f32p, abi.f32p, C.POINTER(abi.LightDesc)]
    L.orc_make_point_light.restype = None
    L.orc_render_tiles.argtypes = [vp, C.POINTER(abi.CameraMatrices), C.POINTER(abi.SamplerDesc), C.POINTER(abi.IntegratorDesc), vp, C.c_size_t, vp, C.POINTER(C.c_uint64), C.POINTER(abi.TraceStats), C.c_int, vp]
    L.orc_render_tiles.restype = C.c_int
    L.orc_render_tiles_accumulating.argtypes = [vp, C.POINTER(abi.CameraMatrices), C.POINTER(abi.SamplerDesc), C.POINTER(abi.IntegratorDesc), vp, vp, C.c_size_t, vp, C.POINTER(C.c_uint64), C.c_int]
    L.orc_render_tiles_accumulating.restype = C.c_int
    L.orc_camera_rays.argtypes = [C.POINTER(abi.CameraMatrices), C.POINTER(abi.SamplerDesc), C.POINTER(abi.Tile), C.c_uint32, vp, vp]
    L.orc_camera_rays.restype = None
    L.orc_intersect.argtypes = [vp, C.c_size_t] + [vp] * 11
    L.orc_intersect.restype = None
    L.orc_any_intersect.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, vp]
    L.orc_any_intersect.restype = None
    L.orc_siphash13.argtypes = [vp, C.c_size_t]
    L.orc_siphash13.restype = C.c_uint64
    L.orc_pcg32_sequence.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, vp, C.c_size_t]
    L.orc_pcg32_sequence.restype = None
    L.orc_permutation_element.argtypes = [C.c_uint32] * 3
    L.orc_permutation_element.restype = C.c_uint32
    L.orc_sampler_sequence.argtypes = [C.POINTER(abi.SamplerDesc), C.c_uint16, C.c_uint16, C.c_uint32, vp, C.c_size_t, vp]
    L.orc_sampler_sequence.restype = None
    for fn in ("sinf", "cosf", "tanf", "logf", "acosf"):
        f = getattr(L, "orc_" + fn)
        f.argtypes = [C.c_float]
        f.restype = C.c_float
    L.orc_libm_array.argtypes = [C.c_int, C.c_size_t, vp, vp, vp]
    L.orc_libm_array.restype = None
    L.orc_expf.argtypes = [C.c_float]
    L.orc_expf.restype = C.c_float
    L.orc_atan2f.argtypes = [C.c_float, C.c_float]
    L.orc_atan2f.restype = C.c_float
    L.orc_mat4_inverse_f32.argtypes = [vp, vp]
    L.orc_mat4_inverse_f64.argtypes = [vp, vp]
    L.orc_mat4_mul_f32.argtypes = [vp, vp, vp]
    L.orc_transform_apply_f32.argtypes = [vp, vp, C.c_int, vp, vp]
    L.orc_transform_bounds_f32.argtypes = [vp] * 5
    L.orc_look_at_f64.argtypes = [vp] * 5
    L.orc_look_at_f32.argtypes = [vp] * 5
    L.orc_rotation_f64.argtypes = [C.c_int, C.c_double, vp, vp, vp]
    L.orc_rotation_f32.argtypes = [C.c_int, C.c_float, vp, vp, vp]
    L.orc_vec3_ops_f32.argtypes = [vp] * 3
    L.orc_bounds_ops_f32.argtypes = [vp] * 4
    L.orc_coordinate_system_f32.argtypes = [vp] * 3
    L.orc_slab_test_f32.argtypes = [vp, vp, vp, vp, C.c_float, vp, vp]
    L.orc_slab_test_f32.restype = C.c_int
    L.orc_bsdf_eval.argtypes = [C.POINTER(abi.MaterialDesc)] + [vp] * 6
    L.orc_bsdf_eval.restype = None
    L.orc_bsdf_sample.argtypes = [C.POINTER(abi.MaterialDesc)] + [vp] * 6
    L.orc_bsdf_sample.restype = None
    L.orc_light_sample.argtypes = [C.POINTER(abi.LightDesc), C.c_int32, C.c_size_t] + [vp] * 4
    L.orc_light_sample.restype = None
    L.orc_texture_eval.argtypes = [C.POINTER(abi.TextureDesc), C.c_size_t, vp, vp]
    L.orc_texture_eval.restype = None
    L.orc_sizeof.argtypes = [C.c_int]
    L.orc_sizeof.restype = C.c_size_t
    _LIBS[_FLAVOUR] = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class LightFactory:
    """Light construction restated from lights/*.rs (oracle side)."""

    @staticmethod
    def make_rect_light(l2w, l2w_inv, L, size, out):
        lib().orc_make_rect_light(abi.f16(l2w), abi.f16(l2w_inv), abi.f3(L), (C.c_float * 2)(*[float(s) for s in size]), C.byref(out))

    @staticmethod
    def make_spot_light(l2w, l2w_inv, I, total, falloff, out):
        lib().orc_make_spot_light(abi.f16(l2w), abi.f16(l2w_inv), abi.f3(I), float(total), float(falloff), C.byref(out))

    @staticmethod
    def make_point_light(l2w, I, out):
        lib().orc_make_point_light(abi.f16(l2w), abi.f3(I), C.byref(out))


def make_camera(cam, res):
    p = abi.CameraParams()
    p.position = abi.f3(cam["position"])
    p.target = abi.f3(cam["target"])
    p.up = abi.f3(cam["up"])
    p.fov_axis = cam["fov_axis"]
    p.fov_degrees = cam["fov_degrees"]
    p.res_x, p.res_y = res
    out = abi.CameraMatrices()
    lib().orc_camera_make(C.byref(p), C.byref(out))
    return out


def film_tiles(res, tile_dim):
    n = lib().orc_film_tiles(res[0], res[1], tile_dim, None, 0)
    t = np.zeros(n, dtype=abi.TILE_DTYPE)
    lib().orc_film_tiles(res[0], res[1], tile_dim, _p(t), n)
    return t


class OracleScene:
    def __init__(self, scene_data):
        self.data = scene_data
        d, self._keep = scene_data.desc(LightFactory)
        h = C.c_void_p()
        rc = lib().orc_scene_create(C.byref(d), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"orc_scene_create failed: {rc}")
        self.h = h

    def close(self):
        if self.h:
            lib().orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def export_bvh(self):
        nn, ns = lib().orc_scene_node_count(self.h), lib().orc_scene_shape_count(self.h)
        nodes = np.zeros(nn, dtype=abi.BVH_NODE_DTYPE)
        order = np.zeros(ns, dtype=np.uint32)
        lib().orc_scene_export_bvh(self.h, _p(nodes), _p(order))
        return nodes, order

    def render_tiles(self, camera, sampler, integrator, tiles, n_threads=1, want_stats=False, per_sample=False):
        tiles = np.ascontiguousarray(tiles, dtype=abi.TILE_DTYPE)
        npx = int(((tiles["x1"].astype(np.int64) - tiles["x0"]) * (tiles["y1"].astype(np.int64) - tiles["y0"])).sum())
        out = np.zeros((npx, 3), dtype=np.float32)
        spp = sampler.nx if sampler.kind == abi.SAMPLER_UNIFORM else sampler.nx * sampler.ny
        ps = np.zeros((npx, spp, 3), dtype=np.float32) if per_sample else None
        rays = C.c_uint64(0)
        stats = abi.TraceStats()
        rc = lib().orc_render_tiles(self.h, C.byref(camera), C.byref(sampler), C.byref(integrator), _p(tiles), len(tiles), _p(out), C.byref(rays), C.byref(stats) if want_stats else None, n_threads, _p(ps))
        assert rc == 0
        res = [out, rays.value]
        if want_stats:
            res.append(stats)
        if per_sample:
            res.append(ps)
        return tuple(res)

    def render_tiles_accumulating(self, camera, sampler, integrator, tiles, tile_samples, n_threads=1):
        """Integrator::render(accumulating=true): one sample (index tile_samples[t]) per pixel."""
        tiles = np.ascontiguousarray(tiles, dtype=abi.TILE_DTYPE)
        ts = np.ascontiguousarray(tile_samples, dtype=np.uint16)
        assert len(ts) == len(tiles)
        assert int(ts.max()) < int(sampler.nx) * max(1, int(sampler.ny) if sampler.kind == 1 else 1), "sample index outside the sampler's domain (the permutation walk need not terminate)"
        npx = int(((tiles["x1"].astype(np.int64) - tiles["x0"]) * (tiles["y1"].astype(np.int64) - tiles["y0"])).sum())
        out = np.zeros((npx, 3), dtype=np.float32)
        rays = C.c_uint64(0)
        rc = lib().orc_render_tiles_accumulating(self.h, C.byref(camera), C.byref(sampler), C.byref(integrator), _p(tiles), _p(ts), len(tiles), _p(out), C.byref(rays), n_threads)
        assert rc == 0
        return out, rays.value

    def li(self, sampler, integrator, o, d, pixel_xy, sample_index, dimension=2):
        """Integrator::li for caller-supplied rays (integrators/mod.rs:94-101); the sampler is started at
        (pixel, sample_index) and `dimension` one-dimensional draws are consumed first (2 = after the camera sample).  -> (li (n, 3) f32, ray counts (n,) u64)"""
        o = np.ascontiguousarray(o, dtype=np.float32)
        d = np.ascontiguousarray(d, dtype=np.float32)
        pix = np.ascontiguousarray(pixel_xy, dtype=np.uint16)
        si = np.ascontiguousarray(sample_index, dtype=np.uint32)
        out = np.zeros((o.shape[0], 3), dtype=np.float32)
        rays = np.zeros(o.shape[0], dtype=np.uint64)
        f = lib().orc_li
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        rc = f(self.h, C.cast(C.byref(sampler), C.c_void_p), C.cast(C.byref(integrator), C.c_void_p), o.shape[0], _p(o), _p(d), _p(pix), _p(si), dimension, _p(out), _p(rays))
        assert rc == 0
        return out, rays

    def intersect(self, o, d, t_max=None):
        o = np.ascontiguousarray(o, dtype=np.float32)
        d = np.ascontiguousarray(d, dtype=np.float32)
        n = o.shape[0]
        tm = None if t_max is None else np.ascontiguousarray(t_max, dtype=np.float32)
        r = dict(
            shape=np.zeros(n, dtype=np.int32), t=np.zeros(n, dtype=np.float32), n=np.zeros((n, 3), dtype=np.float32), ns=np.zeros((n, 3), dtype=np.float32),
            p=np.zeros((n, 3), dtype=np.float32), node_tests=np.zeros(n, dtype=np.uint32), node_hits=np.zeros(n, dtype=np.uint32), shape_tests=np.zeros(n, dtype=np.uint32),
        )
        lib().orc_intersect(self.h, n, _p(o), _p(d), _p(tm), _p(r["shape"]), _p(r["t"]), _p(r["n"]), _p(r["ns"]), _p(r["p"]), _p(r["node_tests"]), _p(r["node_hits"]), _p(r["shape_tests"]))
        return r

    def any_intersect(self, o, d, t_max, area_light=None):
        o = np.ascontiguousarray(o, dtype=np.float32)
        d = np.ascontiguousarray(d, dtype=np.float32)
        tm = np.ascontiguousarray(t_max, dtype=np.float32)
        al = None if area_light is None else np.ascontiguousarray(area_light, dtype=np.int32)
        out = np.zeros(o.shape[0], dtype=np.uint8)
        lib().orc_any_intersect(self.h, o.shape[0], _p(o), _p(d), _p(tm), _p(al), _p(out))
        return out


def camera_rays(camera, sampler, tile, sample_index):
    t = abi.Tile(*[int(v) for v in tile])
    n = (t.x1 - t.x0) * (t.y1 - t.y0)
    o = np.zeros((n, 3), dtype=np.float32)
    d = np.zeros((n, 3), dtype=np.float32)
    lib().orc_camera_rays(C.byref(camera), C.byref(sampler), C.byref(t), sample_index, _p(o), _p(d))
    return o, d


def detile(tiles, rgb, res):
    """Film::update_tile (film.rs:210-282): tile-major -> row-major film."""
    film = np.zeros((res[1], res[0], 3), dtype=np.float32)
    off = 0
    for t in tiles:
        w, h = int(t["x1"]) - int(t["x0"]), int(t["y1"]) - int(t["y0"])
        film[t["y0"] : t["y1"], t["x0"] : t["x1"]] = rgb[off : off + w * h].reshape(h, w, 3)
        off += w * h
    return film


def libm_array(fn, x, y=None):
    """olibm.h over an array: fn 0 sin, 1 cos, 2 tan, 3 log, 4 acos, 5 atan2(x, y), 6 exp."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    yy = None if y is None else np.ascontiguousarray(y, dtype=np.float32)
    out = np.zeros_like(x)
    lib().orc_libm_array(fn, x.size, _p(x), _p(yy), _p(out))
    return out


def texture_eval(tex, uv):
    """ImageTexture::evaluate of the C++ restatement at (n, 2) uv pairs -> (n, 3)."""
    tex = np.ascontiguousarray(tex, dtype=np.float32)
    uv = np.ascontiguousarray(uv, dtype=np.float32)
    d = abi.TextureDesc(tex.shape[1], tex.shape[0], abi.ptr(tex, abi.f32p))
    out = np.zeros((uv.shape[0], 3), dtype=np.float32)
    lib().orc_texture_eval(C.byref(d), uv.shape[0], _p(uv), _p(out))
    return out
