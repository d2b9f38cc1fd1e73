"""The C-ABI library loads without a GPU and exports every symbol the header
declares; POD layouts agree between the product header, the ctypes mirror and the
oracle's copy."""
import ctypes as C
import os
import re

from yuki_amd import _ffi, abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported_and_bound():
    hdr = open(os.path.join(ROOT, "include", "yuki_hip.h")).read()
    declared = set(re.findall(r"^(?:yk_status|void\*?|size_t|uint32_t|const char\*|yk_context\*)\s+(yk_[a-z0-9_]+)\s*\(", hdr, re.M))
    assert len(declared) >= 25
    L = _ffi.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/yuki_hip.h but not exported"
    assert declared == set(_ffi.SYMBOLS), (declared ^ set(_ffi.SYMBOLS))
    assert L.yk_abi_version() == 1
    assert L.yk_status_string(0) == b"ok"


def test_struct_sizes_match_everywhere(oracle):
    L = _ffi.lib()
    O = oracle.lib()
    mirror = [abi.SceneDesc, abi.MaterialDesc, abi.LightDesc, abi.SphereDesc, abi.CameraMatrices, abi.CameraParams, abi.SamplerDesc, abi.IntegratorDesc, abi.Tile, abi.BvhNode, abi.MeshDesc]
    for what, ty in enumerate(mirror):
        assert L.yk_sizeof(what) == C.sizeof(ty) == O.orc_sizeof(what), ty.__name__
    assert L.yk_sizeof(11) == C.sizeof(_ffi.RenderStats)
    assert L.yk_sizeof(12) == C.sizeof(_ffi.SceneInfo)
    assert abi.BVH_NODE_DTYPE.itemsize == 32 == C.sizeof(abi.BvhNode)  # size_of::<BVHNode>() == 32, bvh.rs:556
    assert abi.TILE_DTYPE.itemsize == C.sizeof(abi.Tile)


def test_no_gpu_means_loud_failure_not_a_fallback():
    """Without a device the context cannot be created: status, no CPU path."""
    import torch

    if torch.cuda.is_available():
        return
    L = _ffi.lib()
    h = C.c_void_p()
    assert L.yk_context_create(0, C.byref(h)) == 2  # YK_ERR_NO_DEVICE
    assert not h.value


def test_product_never_references_the_oracle():
    """The oracle is test infrastructure: nothing under yuki_amd/ or include/ may
    import, include or link it."""
    bad = []
    for base in ("yuki_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".cpp", ".hip", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="replace").read()
                    if re.search(r"(from|import)\s+oracle|#include\s+\"[^\"]*oracle|liboracle|orc_[a-z]", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_multi_gpu_entry_points_fail_loudly_without_a_device():
    """yk_multi_* / yk_dist_* (include/yuki_hip.h, several GPUs): no device, no fallback."""
    import torch

    if torch.cuda.is_available():
        return
    L = _ffi.lib()
    devs = (C.c_int * 1)(0)
    h = C.c_void_p()
    assert L.yk_multi_create(devs, 1, C.byref(h)) == 2 and not h.value  # YK_ERR_NO_DEVICE
    assert L.yk_multi_create(devs, 0, C.byref(h)) == 1  # YK_ERR_INVALID_ARGUMENT
    two = (C.c_int * 2)(0, 0)
    assert L.yk_multi_create(two, 2, C.byref(h)) == 1  # one rank per GPU
    assert L.yk_multi_device_count(None) == 0 and not L.yk_multi_film_device_ptr(None)
    assert L.yk_multi_sync(None) == 1 and L.yk_dist_create(None, None, 0, 1, C.byref(h)) == 1
