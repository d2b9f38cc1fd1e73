"""ctypes mirrors of the POD structs declared in include/yuki_hip.h.

The same layouts are accepted by the parity oracle (oracle/oracle_api.h), so the
tests build one description of a scene / camera / sampler and hand it to both.
"""
import ctypes as C

import numpy as np

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)


class MeshDesc(C.Structure):
    _fields_ = [("has_normals", C.c_uint8), ("has_uvs", C.c_uint8), ("swaps_handedness", C.c_uint8), ("pad", C.c_uint8)]


class SphereDesc(C.Structure):
    _fields_ = [
        ("object_to_world", C.c_float * 16),
        ("world_to_object", C.c_float * 16),
        ("radius", C.c_float),
        ("material", C.c_int32),
    ]


class MaterialDesc(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("a", C.c_float * 3), ("b", C.c_float * 3), ("c", C.c_float), ("flags", C.c_uint32), ("a_texture", C.c_uint32)]


class TextureDesc(C.Structure):
    """textures/image_texture.rs:49-56: row-major RGB f32, row 0 = top row of the file."""

    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgb", C.POINTER(C.c_float))]


class LightDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32),
        ("p", C.c_float * 3),
        ("i", C.c_float * 3),
        ("cos_total_width", C.c_float),
        ("cos_falloff_start", C.c_float),
        ("world_to_light", C.c_float * 16),
        ("sample_to_world", C.c_float * 16),
        ("sample_to_world_inv", C.c_float * 16),
        ("area", C.c_float),
    ]


class SceneDesc(C.Structure):
    _fields_ = [
        ("n_vertices", C.c_uint32),
        ("points", f32p),
        ("normals", f32p),
        ("uvs", f32p),
        ("n_triangles", C.c_uint32),
        ("indices", u32p),
        ("tri_mesh", u32p),
        ("tri_material", i32p),
        ("tri_area_light", i32p),
        ("n_meshes", C.c_uint32),
        ("meshes", C.POINTER(MeshDesc)),
        ("n_spheres", C.c_uint32),
        ("spheres", C.POINTER(SphereDesc)),
        ("n_materials", C.c_uint32),
        ("materials", C.POINTER(MaterialDesc)),
        ("n_lights", C.c_uint32),
        ("lights", C.POINTER(LightDesc)),
        ("background", C.c_float * 3),
        ("split_method", C.c_uint32),
        ("max_shapes_in_node", C.c_uint32),
        ("shape_order", u32p),
        ("n_textures", C.c_uint32),
        ("textures", C.POINTER(TextureDesc)),
    ]


class CameraMatrices(C.Structure):
    """`Camera` of camera.rs:19-22: two Transforms (matrix + inverse each)."""

    _fields_ = [
        ("camera_to_world", C.c_float * 16),
        ("camera_to_world_inv", C.c_float * 16),
        ("raster_to_camera", C.c_float * 16),
        ("raster_to_camera_inv", C.c_float * 16),
    ]


class CameraParams(C.Structure):
    _fields_ = [
        ("position", C.c_float * 3),
        ("target", C.c_float * 3),
        ("up", C.c_float * 3),
        ("fov_axis", C.c_uint32),
        ("fov_degrees", C.c_float),
        ("res_x", C.c_uint16),
        ("res_y", C.c_uint16),
    ]


class SamplerDesc(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("nx", C.c_uint32), ("ny", C.c_uint32), ("jitter", C.c_uint32), ("seed", C.c_uint64)]


class IntegratorDesc(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("max_depth", C.c_uint32), ("has_clamp", C.c_uint32), ("indirect_clamp", C.c_float)]


class Tile(C.Structure):
    _fields_ = [("x0", C.c_uint16), ("y0", C.c_uint16), ("x1", C.c_uint16), ("y1", C.c_uint16)]


class BvhNode(C.Structure):
    _fields_ = [
        ("bmin", C.c_float * 3),
        ("bmax", C.c_float * 3),
        ("a", C.c_uint32),
        ("count", C.c_uint16),
        ("axis", C.c_uint8),
        ("is_leaf", C.c_uint8),
    ]


class TraceStats(C.Structure):
    _fields_ = [
        ("closest_rays", C.c_uint64),
        ("closest_node_tests", C.c_uint64),
        ("closest_shape_tests", C.c_uint64),
        ("shadow_rays", C.c_uint64),
        ("shadow_node_tests", C.c_uint64),
        ("shadow_shape_tests", C.c_uint64),
    ]


BVH_NODE_DTYPE = np.dtype(
    [("bmin", "<f4", 3), ("bmax", "<f4", 3), ("a", "<u4"), ("count", "<u2"), ("axis", "u1"), ("is_leaf", "u1")]
)
TILE_DTYPE = np.dtype([("x0", "<u2"), ("y0", "<u2"), ("x1", "<u2"), ("y1", "<u2")])

# enums (include/yuki_hip.h)
SPLIT_SAH, SPLIT_MIDDLE, SPLIT_EQUAL_COUNTS = 0, 1, 2
MAT_MATTE, MAT_GLASS, MAT_METAL, MAT_GLOSSY = 0, 1, 2, 3
LIGHT_POINT, LIGHT_SPOT, LIGHT_DISTANT, LIGHT_RECT = 0, 1, 2, 3
SAMPLER_UNIFORM, SAMPLER_STRATIFIED = 0, 1
INTEGRATOR_WHITTED, INTEGRATOR_PATH, INTEGRATOR_BVH_INTERSECTIONS, INTEGRATOR_GEOMETRY_NORMALS, INTEGRATOR_SHADING_NORMALS = 0, 1, 2, 3, 4
FOV_X, FOV_Y = 0, 1


def ptr(a, ty):
    """Pointer into a C-contiguous numpy array (None -> NULL)."""
    if a is None:
        return C.cast(None, ty)
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ty)


def f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def f16(m):
    return (C.c_float * 16)(*[float(x) for x in np.asarray(m, dtype=np.float32).reshape(16)])
