"""ctypes binding of libyuki_hip.so (include/yuki_hip.h).

The library is the product: if it is missing or cannot be loaded this module
raises — there is no CPU fallback anywhere in the package.
"""
import ctypes as C
import os
import subprocess

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("YK_LIB_PATH") or os.path.join(_HERE, "libyuki_hip.so")  # override: kernel-variant experiments
_LIB = None

YK_OK = 0
STATUS_NAMES = {
    0: "YK_OK",
    1: "YK_ERR_INVALID_ARGUMENT",
    2: "YK_ERR_NO_DEVICE",
    3: "YK_ERR_DEVICE",
    4: "YK_ERR_OUT_OF_MEMORY",
    5: "YK_ERR_UNSUPPORTED",
    6: "YK_ERR_BVH_BUILD",
    7: "YK_ERR_CANCELLED",
    8: "YK_ERR_STACK_OVERFLOW",
}


class YukiError(RuntimeError):
    def __init__(self, status, message=""):
        self.status = status
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")


class SceneInfo(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_uint64),
        ("n_interior", C.c_uint64),
        ("n_shapes", C.c_uint64),
        ("bounds_min", C.c_float * 3),
        ("bounds_max", C.c_float * 3),
        ("build_seconds", C.c_double),
        ("upload_seconds", C.c_double),
        ("device_bytes", C.c_uint64),
        ("max_leaf_shapes", C.c_uint32),
        ("tree_depth", C.c_uint32),
    ]


class RenderStats(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64),
        ("shadow_rays", C.c_uint64),
        ("samples", C.c_uint64),
        ("seconds_total", C.c_double),
        ("seconds_trace", C.c_double),
        ("seconds_shadow", C.c_double),
        ("seconds_shade", C.c_double),
        ("trace_launches", C.c_uint32),
        ("batches", C.c_uint32),
        ("shadow_launches", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class CombinerInfo(C.Structure):
    _fields_ = [("submissions", C.c_uint64), ("tiles", C.c_uint64), ("requeued", C.c_uint64), ("largest_submission", C.c_uint32), ("lanes", C.c_uint32)]


CANCEL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)

# name -> (restype, argtypes); every symbol include/yuki_hip.h declares
vp = C.c_void_p
SYMBOLS = {
    "yk_abi_version": (C.c_uint32, []),
    "yk_status_string": (C.c_char_p, [C.c_int]),
    "yk_context_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "yk_context_destroy": (None, [vp]),
    "yk_last_error": (C.c_int, [vp, C.c_char_p, C.c_size_t]),
    "yk_context_stream": (vp, [vp]),
    "yk_context_set_option": (C.c_int, [vp, C.c_char_p, C.c_int64]),
    "yk_camera_init": (C.c_int, [C.POINTER(abi.CameraParams), C.POINTER(abi.CameraMatrices)]),
    "yk_film_tiles": (C.c_size_t, [C.c_uint16, C.c_uint16, C.c_uint16, vp, C.c_size_t]),
    "yk_make_rect_light": (C.c_int, [abi.f32p, abi.f32p, abi.f32p, abi.f32p, C.POINTER(abi.LightDesc)]),
    "yk_make_spot_light": (C.c_int, [abi.f32p, abi.f32p, abi.f32p, C.c_float, C.c_float, C.POINTER(abi.LightDesc)]),
    "yk_make_point_light": (C.c_int, [abi.f32p, abi.f32p, C.POINTER(abi.LightDesc)]),
    "yk_film_update_tiles": (C.c_int, [vp, C.c_size_t, vp, C.c_uint16, C.c_uint16, vp]),
    "yk_scene_create": (C.c_int, [vp, C.POINTER(abi.SceneDesc), C.POINTER(vp)]),
    "yk_scene_destroy": (None, [vp]),
    "yk_scene_get_info": (C.c_int, [vp, C.POINTER(SceneInfo)]),
    "yk_scene_export_bvh": (C.c_int, [vp, vp, vp]),
    "yk_render_tiles": (C.c_int, [vp, vp, C.POINTER(abi.CameraMatrices), C.POINTER(abi.SamplerDesc), C.POINTER(abi.IntegratorDesc), vp, C.c_size_t, vp, C.POINTER(RenderStats), vp, vp]),
    "yk_render_tiles_device": (C.c_int, [vp, vp, C.POINTER(abi.CameraMatrices), C.POINTER(abi.SamplerDesc), C.POINTER(abi.IntegratorDesc), vp, C.c_size_t, vp, vp, C.POINTER(RenderStats), vp, vp]),
    "yk_render_tile": (C.c_int, [vp, vp, C.POINTER(abi.CameraMatrices), C.POINTER(abi.SamplerDesc), C.POINTER(abi.IntegratorDesc), C.POINTER(abi.Tile), vp, C.POINTER(C.c_uint64)]),
    "yk_film_update_tiles_device": (C.c_int, [vp, vp, C.c_size_t, vp, C.c_uint16, C.c_uint16, vp, vp]),
    "yk_li": (C.c_int, [vp, vp, C.POINTER(abi.SamplerDesc), C.POINTER(abi.IntegratorDesc), C.c_size_t, vp, vp, vp, vp, C.c_uint32, vp, vp]),
    "yk_trace_closest": (C.c_int, [vp, vp, C.c_size_t] + [vp] * 9),
    "yk_trace_any": (C.c_int, [vp, vp, C.c_size_t] + [vp] * 5),
    "yk_sampler_sequence": (C.c_int, [vp, C.POINTER(abi.SamplerDesc), C.c_uint16, C.c_uint16, C.c_uint32, vp, C.c_size_t, vp]),
    "yk_camera_rays": (C.c_int, [vp, C.POINTER(abi.CameraMatrices), C.POINTER(abi.SamplerDesc), C.POINTER(abi.Tile), C.c_uint32, vp, vp]),
    "yk_device_math": (C.c_int, [vp, C.c_int, C.c_size_t, vp, vp, vp]),
    "yk_host_math": (C.c_int, [C.c_int, C.c_size_t, vp, vp, vp]),
    "yk_bsdf_eval": (C.c_int, [vp, C.POINTER(abi.MaterialDesc), C.c_size_t] + [vp] * 6),
    "yk_bsdf_sample": (C.c_int, [vp, C.POINTER(abi.MaterialDesc), C.c_size_t] + [vp] * 6),
    "yk_light_sample": (C.c_int, [vp, C.POINTER(abi.LightDesc), C.c_int32, C.c_size_t] + [vp] * 4),
    "yk_sizeof": (C.c_size_t, [C.c_int]),
    "yk_load_ply": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(vp)]),
    "yk_load_pbrt": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(vp)]),
    "yk_loaded_scene_get": (C.c_int, [vp, C.POINTER(abi.SceneDesc), C.POINTER(abi.CameraParams), C.POINTER(C.c_uint16)]),
    "yk_loaded_scene_destroy": (None, [vp]),
    "yk_loader_last_error": (C.c_char_p, []),
    "yk_image_texture_load": (C.c_int, [C.c_char_p, C.POINTER(abi.TextureDesc)]),
    "yk_image_texture_free": (None, [C.POINTER(abi.TextureDesc)]),
    "yk_render_tiles_accumulating": (C.c_int, [vp, vp, C.POINTER(abi.CameraMatrices), C.POINTER(abi.SamplerDesc), C.POINTER(abi.IntegratorDesc), vp, vp, C.c_size_t, vp, C.POINTER(RenderStats), vp, vp]),
    "yk_render_tiles_accumulating_device": (C.c_int, [vp, vp, C.POINTER(abi.CameraMatrices), C.POINTER(abi.SamplerDesc), C.POINTER(abi.IntegratorDesc), vp, vp, C.c_size_t, vp, vp, C.POINTER(RenderStats), vp, vp]),
    "yk_film_accumulate_tiles": (C.c_int, [vp, C.c_size_t, vp, C.c_uint16, C.c_uint16, vp, vp]),
    "yk_film_accumulate_tiles_device": (C.c_int, [vp, vp, C.c_size_t, vp, C.c_uint16, C.c_uint16, vp, vp]),
    "yk_tile_list_create": (C.c_int, [vp, vp, vp, C.c_size_t, C.POINTER(vp)]),
    "yk_tile_list_destroy": (None, [vp]),
    "yk_render_tile_list_device": (C.c_int, [vp, vp, C.POINTER(abi.CameraMatrices), C.POINTER(abi.SamplerDesc), C.POINTER(abi.IntegratorDesc), vp, vp, vp, C.POINTER(RenderStats), vp, vp]),
    "yk_film_update_tile_list_device": (C.c_int, [vp, vp, vp, C.c_uint16, C.c_uint16, vp, vp, C.c_int]),
    "yk_render_tiles_accumulating_passes": (C.c_int, [vp, vp, C.POINTER(abi.CameraMatrices), C.POINTER(abi.SamplerDesc), C.POINTER(abi.IntegratorDesc), vp, vp, C.c_size_t, C.c_uint32, vp, C.POINTER(RenderStats), vp, vp]),
    "yk_render_tile_list_samples_device": (C.c_int, [vp, vp, C.POINTER(abi.CameraMatrices), C.POINTER(abi.SamplerDesc), C.POINTER(abi.IntegratorDesc), vp, C.c_uint32, C.c_uint32, vp, vp, C.POINTER(RenderStats), vp, vp]),
    "yk_render_tile_list_passes_device": (C.c_int, [vp, vp, C.POINTER(abi.CameraMatrices), C.POINTER(abi.SamplerDesc), C.POINTER(abi.IntegratorDesc), vp, C.c_uint32, vp, vp, C.POINTER(RenderStats), vp, vp]),
    "yk_film_accumulate_tile_list_passes_device": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_uint16, C.c_uint16, vp, vp]),
    "yk_write_exr": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, vp]),
    "yk_write_pfm": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, vp]),
    # several GPUs
    "yk_multi_create": (C.c_int, [C.POINTER(C.c_int), C.c_uint32, C.POINTER(vp)]),
    "yk_multi_create_ex": (C.c_int, [C.POINTER(C.c_int), C.c_uint32, C.c_uint32, C.POINTER(vp)]),
    "yk_multi_deal": (C.c_size_t, [C.c_uint16, C.c_uint16, C.c_uint16, C.c_uint32, C.c_uint32, vp, C.c_size_t, C.POINTER(C.c_uint64)]),
    "yk_multi_accumulate_film": (C.c_int, [vp, vp, C.POINTER(abi.CameraMatrices), C.POINTER(abi.SamplerDesc), C.POINTER(abi.IntegratorDesc), vp, C.c_uint32, C.c_uint32, vp, C.POINTER(RenderStats), vp, vp]),
    "yk_multi_film_clear": (C.c_int, [vp, vp]),
    "yk_multi_interrupt": (C.c_int, [vp]),
    "yk_context_interrupt": (C.c_int, [vp]),
    "yk_combiner_create": (C.c_int, [C.POINTER(vp), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(vp)]),
    "yk_combiner_destroy": (None, [vp]),
    "yk_combiner_render_tile": (C.c_int, [vp, vp, C.POINTER(abi.CameraMatrices), C.POINTER(abi.SamplerDesc), C.POINTER(abi.IntegratorDesc), C.POINTER(abi.Tile), C.c_int32, vp, C.POINTER(RenderStats), vp, vp]),
    "yk_combiner_get_info": (C.c_int, [vp, C.POINTER(CombinerInfo)]),
    "yk_combiner_last_error": (C.c_int, [vp, C.c_char_p, C.c_size_t]),
    "yk_multi_destroy": (None, [vp]),
    "yk_multi_device_count": (C.c_uint32, [vp]),
    "yk_multi_context": (vp, [vp, C.c_uint32]),
    "yk_multi_set_option": (C.c_int, [vp, C.c_char_p, C.c_int64]),
    "yk_multi_last_error": (C.c_int, [vp, C.c_char_p, C.c_size_t]),
    "yk_multi_scene_create": (C.c_int, [vp, C.POINTER(abi.SceneDesc), C.POINTER(vp)]),
    "yk_multi_scene_destroy": (None, [vp]),
    "yk_multi_scene_get_info": (C.c_int, [vp, C.POINTER(SceneInfo)]),
    "yk_multi_film_create": (C.c_int, [vp, C.c_uint16, C.c_uint16, C.c_uint16, C.POINTER(vp)]),
    "yk_multi_film_destroy": (None, [vp]),
    "yk_multi_film_device_ptr": (vp, [vp]),
    "yk_multi_render_film": (C.c_int, [vp, vp, C.POINTER(abi.CameraMatrices), C.POINTER(abi.SamplerDesc), C.POINTER(abi.IntegratorDesc), vp, vp, C.POINTER(RenderStats), vp, vp]),
    "yk_multi_sync": (C.c_int, [vp]),
    "yk_dist_unique_id": (C.c_int, [vp]),
    "yk_dist_create": (C.c_int, [vp, vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]),
    "yk_dist_destroy": (None, [vp]),
    "yk_dist_gather": (C.c_int, [vp, vp, vp, C.c_size_t, vp]),
}


def build():
    """Compile libyuki_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc")])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension is the product and has no fallback. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C yuki_amd/csrc`."
        )
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)  # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    if L.yk_abi_version() != 1:
        raise ImportError("libyuki_hip.so ABI version mismatch")
    _LIB = L
    return L


def check(status, ctx=None):
    if status != YK_OK:
        msg = ""
        if ctx:
            buf = C.create_string_buffer(512)
            lib().yk_last_error(ctx, buf, 512)
            msg = buf.value.decode(errors="replace")
        if not msg:
            msg = lib().yk_status_string(status).decode()
        raise YukiError(status, msg)
