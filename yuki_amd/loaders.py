"""Scene input, mirroring the reference's `scene` module.

    SceneLoadSettings                       yuki/src/scene/mod.rs:25-39
    Scene::ply(settings)                    yuki/src/scene/mod.rs:99-152  (+ scene/ply.rs)
    scene::pbrt::load(settings)             yuki/src/scene/pbrt/mod.rs:94-857

Both return what the reference returns — the scene, the `CameraParameters` and the
`FilmSettings` — with the scene as a `SceneData` ready for `core.Scene(ctx, data)`.
The parsing itself runs in libyuki_hip.so (`yk_load_ply` / `yk_load_pbrt`,
yuki_amd/csrc/yk_loaders.cpp); this file copies the result out of the library.
"""
import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from . import abi
from ._ffi import YukiError, lib
from .core import CameraParameters, FilmSettings
from .scenes import SceneData


@dataclass
class SceneLoadSettings:
    """scene/mod.rs:25-39 (defaults: SurfaceAreaHeuristic, max_shapes_in_node 1)."""

    path: str = ""
    split_method: int = abi.SPLIT_SAH
    max_shapes_in_node: int = 1


def _np(ptr, shape, dtype):
    n = int(np.prod(shape))
    if not ptr or n == 0:
        return None
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).reshape(shape).copy()


def _material(m):
    out = dict(kind=int(m.kind), a=tuple(m.a), b=tuple(m.b), c=float(m.c), remap=bool(m.flags & 1))
    if m.flags & 2:
        out["tex"] = int(m.a_texture)
    return out


def _unpack(handle, name, settings):
    L = lib()
    d = abi.SceneDesc()
    cp = abi.CameraParams()
    td = C.c_uint16(0)
    st = L.yk_loaded_scene_get(handle, C.byref(d), C.byref(cp), C.byref(td))
    if st != 0:
        raise YukiError(st, "yk_loaded_scene_get")
    nv, nt = d.n_vertices, d.n_triangles
    order = _np(d.shape_order, (nt + d.n_spheres,), np.uint32)
    data = SceneData(
        points=_np(d.points, (nv, 3), np.float32) if nv else np.zeros((0, 3), np.float32),
        indices=_np(d.indices, (nt, 3), np.uint32) if nt else np.zeros((0, 3), np.uint32),
        tri_mesh=_np(d.tri_mesh, (nt,), np.uint32) if nt else np.zeros(0, np.uint32),
        tri_material=_np(d.tri_material, (nt,), np.int32) if nt else np.zeros(0, np.int32),
        tri_area_light=_np(d.tri_area_light, (nt,), np.int32) if nt else np.zeros(0, np.int32),
        meshes=[(bool(d.meshes[k].has_normals), bool(d.meshes[k].has_uvs), bool(d.meshes[k].swaps_handedness)) for k in range(d.n_meshes)],
        materials=[_material(d.materials[k]) for k in range(d.n_materials)],
        textures=[_np(d.textures[k].rgb, (d.textures[k].height, d.textures[k].width, 3), np.float32) for k in range(d.n_textures)],
        lights=[],
        normals=_np(d.normals, (nv, 3), np.float32),
        uvs=_np(d.uvs, (nv, 2), np.float32),
        spheres=[
            dict(
                o2w=np.array(d.spheres[k].object_to_world, dtype=np.float32).reshape(4, 4),
                w2o=np.array(d.spheres[k].world_to_object, dtype=np.float32).reshape(4, 4),
                radius=float(d.spheres[k].radius),
                material=int(d.spheres[k].material),
            )
            for k in range(d.n_spheres)
        ],
        background=tuple(d.background),
        split_method=int(d.split_method),
        max_shapes_in_node=int(d.max_shapes_in_node),
        name=name,
        shape_order=order,
        film_res=(int(cp.res_x), int(cp.res_y)),
    )
    lights = []
    for k in range(d.n_lights):
        l = abi.LightDesc()
        C.memmove(C.byref(l), C.byref(d.lights[k]), C.sizeof(abi.LightDesc))
        lights.append(l)
    data.light_structs = lights
    cam = CameraParameters(position=tuple(cp.position), target=tuple(cp.target), up=tuple(cp.up), fov_axis=int(cp.fov_axis), fov_degrees=float(cp.fov_degrees))
    data.camera = dict(position=cam.position, target=cam.target, up=cam.up, fov_axis=cam.fov_axis, fov_degrees=cam.fov_degrees)
    film = FilmSettings(res=(int(cp.res_x), int(cp.res_y)), tile_dim=int(td.value))
    return data, cam, film


def _load(fn_name, settings):
    if isinstance(settings, (str, os.PathLike)):
        settings = SceneLoadSettings(path=os.fspath(settings))
    L = lib()
    h = C.c_void_p()
    st = getattr(L, fn_name)(os.fspath(settings.path).encode(), settings.split_method, settings.max_shapes_in_node, C.byref(h))
    if st != 0:
        raise YukiError(st, L.yk_loader_last_error().decode(errors="replace"))
    try:
        return _unpack(h, os.path.basename(settings.path), settings)
    finally:
        L.yk_loaded_scene_destroy(h)


def load_image_texture(path):
    """ImageTexture::new(path) (textures/image_texture.rs:66-70): (h, w, 3) float32, row 0 = top."""
    L = lib()
    t = abi.TextureDesc()
    st = L.yk_image_texture_load(os.fspath(path).encode(), C.byref(t))
    if st != 0:
        raise YukiError(st, L.yk_loader_last_error().decode(errors="replace"))
    try:
        return _np(t.rgb, (t.height, t.width, 3), np.float32)
    finally:
        L.yk_image_texture_free(C.byref(t))


def load_ply(settings):
    """Scene::ply: (SceneData, CameraParameters, FilmSettings)."""
    return _load("yk_load_ply", settings)


def load_pbrt(settings):
    """scene::pbrt::load: (SceneData, CameraParameters, FilmSettings)."""
    return _load("yk_load_pbrt", settings)
