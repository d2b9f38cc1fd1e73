"""The oracle's libm (oracle/olibm.h) against what the Rust reference calls on Linux, the
platform's glibc: all seven functions restate glibc 2.35's own algorithms (sinf / cosf / logf / expf in
the FMA variants every x86-64 host with FMA3 dispatches to) and must be bit-equal to them.
The exhaustive 2^32 comparison is tools/micro/glibc_libm_check.cpp
(profiles/r03_glibc_libm_check.txt); the strided one here runs in seconds.  Accuracy against
the correctly rounded value is bounded too, so that a host with another libm (where the
bit-equality test skips) still checks the functions are what they claim to be."""
import ctypes
import ctypes.util

import numpy as np
import pytest


def _host_libm():
    path = ctypes.util.find_library("m")
    if path is None:
        pytest.skip("no libm to compare with")
    L = ctypes.CDLL(path)
    for n in ("sinf", "cosf", "tanf", "logf", "acosf", "expf"):
        getattr(L, n).argtypes = [ctypes.c_float]
        getattr(L, n).restype = ctypes.c_float
    L.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]
    L.atan2f.restype = ctypes.c_float
    return L


def _host_has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return " fma " in line + " "
    except OSError:
        pass
    return False


def _glibc_version():
    try:
        f = ctypes.CDLL(None).gnu_get_libc_version
        f.restype = ctypes.c_char_p
        return tuple(int(v) for v in f().decode().split(".")[:2])
    except (AttributeError, ValueError):
        return None


def _ulp_diff(a, b):
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai)
    bi = np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    return np.abs(ai - bi)


def test_libm_accuracy_bounds(oracle):
    """None of glibc's binary32 functions is correctly rounded; the restatements inherit their error bounds
    (sinf / cosf / logf below 1 ulp, the fdlibm ones below 1 ulp on these ranges, tanf below 2)."""
    rng = np.random.default_rng(5)
    x = rng.uniform(-7, 7, 20000).astype(np.float32)
    x64 = x.astype(np.float64)
    for fn, ref64, bound in ((0, np.sin, 1), (1, np.cos, 1), (2, np.tan, 2)):
        assert _ulp_diff(oracle.libm_array(fn, x), ref64(x64).astype(np.float32)).max() <= bound, fn
    p = (np.abs(x) + 1e-6).astype(np.float32)
    assert _ulp_diff(oracle.libm_array(3, p), np.log(p.astype(np.float64)).astype(np.float32)).max() <= 1
    e = rng.uniform(-100, 80, 20000).astype(np.float32)
    assert _ulp_diff(oracle.libm_array(6, e), np.exp(e.astype(np.float64)).astype(np.float32)).max() <= 1
    c = rng.uniform(-1, 1, 20000).astype(np.float32)
    assert _ulp_diff(oracle.libm_array(4, c), np.arccos(c.astype(np.float64)).astype(np.float32)).max() <= 1
    y = rng.uniform(-3, 3, 20000).astype(np.float32)
    assert _ulp_diff(oracle.libm_array(5, y, x), np.arctan2(y.astype(np.float64), x64).astype(np.float32)).max() <= 1


def _same(a, b):
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


def test_the_libm_functions_are_the_platform_libms_bit_for_bit(oracle):
    """What `f32::{sin,cos,tan,ln,exp,acos,atan2}` lower to in the reference on Linux (sampling/mod.rs:62-87,
    trowbridge_reitz.rs:23-30, camera.rs:52-102, sphere.rs:38-119, pbrt/cie.rs:8-20): glibc's functions.  Every 65,521st float
    (65,548 arguments across all exponents, both signs, NaN / inf / denormals included) plus 100,000 random ones
    in the range the renderer uses; atan2f on 165,548 pairs."""
    if not _host_has_fma():
        pytest.skip("the restatement is of glibc's FMA variants; this host dispatches to others")
    v = _glibc_version()
    if v is None or v < (2, 35):
        pytest.skip("glibc older than 2.35 (tanf's reduction changed there; sinf / cosf / logf in 2.27-2.28)")
    host = _host_libm()
    bits = np.arange(0, 1 << 32, 65521, dtype=np.uint64).astype(np.uint32)
    rng = np.random.default_rng(77)
    x = np.concatenate([bits.view(np.float32), rng.uniform(-7, 7, 100000).astype(np.float32),
                        np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, 3.4e38, 0.785398185, 1.57079637, 1.0, -1.0, 0.5, -0.5], np.float32)])
    for fn, name in ((0, "sinf"), (1, "cosf"), (2, "tanf"), (3, "logf"), (4, "acosf"), (6, "expf")):
        got = oracle.libm_array(fn, x)
        f = getattr(host, name)
        want = np.array([f(float(a)) for a in x], dtype=np.float32)
        same = _same(got, want)
        assert same.all(), (name, x[~same][:8], got[~same][:8], want[~same][:8])
    y = np.concatenate([rng.permutation(bits).view(np.float32), rng.uniform(-3, 3, 100000).astype(np.float32),
                        np.array([0.0, -0.0, 0.0, -0.0, np.inf, -np.inf, np.inf, 1.0, -1.0, 1e-30, 1e30, 0.0, 1.0], np.float32)])
    got = oracle.libm_array(5, y, x)
    want = np.array([host.atan2f(float(a), float(b)) for a, b in zip(y, x)], dtype=np.float32)
    same = _same(got, want)
    assert same.all(), ("atan2f", y[~same][:8], x[~same][:8], got[~same][:8], want[~same][:8])


def test_libm_special_values(oracle):
    L = oracle.lib()
    assert L.orc_sinf(0.0) == 0.0 and L.orc_cosf(0.0) == 1.0 and L.orc_logf(1.0) == 0.0
    assert np.isnan(L.orc_sinf(float("inf"))) and np.isnan(L.orc_logf(-1.0))
    assert L.orc_logf(0.0) == float("-inf")
    assert L.orc_acosf(1.0) == 0.0 and abs(L.orc_acosf(-1.0) - np.float32(np.pi)) == 0.0
    assert L.orc_atan2f(0.0, -1.0) == np.float32(np.pi) and L.orc_atan2f(1.0, 0.0) == np.float32(np.pi / 2)


def test_the_products_host_instance_equals_the_oracle(oracle):
    """yk_libm.h as the HOST compiles it (the loaders' rotations and CIE fits, the camera's tan, the spot light's cosines, the
    roughness remap's log) against oracle/olibm.h: every 4,099th binary32 value plus the renderer's ranges, all functions, and the
    shared-reduction sin / cos pair against the two functions.  No device needed (yk_host_math)."""
    from yuki_amd import core as yk

    yk.lib()
    rng = np.random.default_rng(11)
    x = np.concatenate([np.arange(0, 1 << 32, 4099, dtype=np.uint64).astype(np.uint32).view(np.float32), rng.uniform(-7, 7, 100000).astype(np.float32),
                        np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, 0.785398185, 1.0, -1.0], np.float32)])
    y = rng.permutation(x)
    for host_fn, oracle_fn in ((0, 0), (1, 1), (2, 2), (3, 3), (4, 4), (5, 5), (28, 0), (29, 1), (30, 6)):
        got = yk.host_math(host_fn, x, y if host_fn == 5 else None)
        want = oracle.libm_array(oracle_fn, x, y if oracle_fn == 5 else None)
        same = _same(got, want)
        assert same.all(), (host_fn, int((~same).sum()), x[~same][:6])
