"""Pin the oracle's math layer with the known-answer values of the reference's own
unit tests (tests/src/{transform,matrix,vector,bounds,point,normal}.rs), stored
as data in tests/golden/reference_math_kats.json."""
import ctypes as C
import json
import os

import numpy as np
import pytest

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_math_kats.json")))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def f64(x):
    return np.ascontiguousarray(x, dtype=np.float64)


def inv32(L, m):
    out = np.zeros(16, dtype=np.float32)
    L.orc_mat4_inverse_f32(_p(f32(m).reshape(16)), _p(out))
    return out.reshape(4, 4)


def apply(L, m, mi, what, v):
    out = np.zeros(3, dtype=np.float32)
    L.orc_transform_apply_f32(_p(f32(m).reshape(16)), _p(f32(mi).reshape(16)), what, _p(f32(v)), _p(out))
    return out


def test_transform_mul(oracle):
    L = oracle.lib()
    k = KATS["transform_mul"]
    t, tp = f32(k["t"]), f32(k["tp"])
    ti, tpi = inv32(L, t), inv32(L, tp)
    assert np.array_equal(apply(L, t, ti, 0, k["v"]), f32(k["t_mul_v"]))
    assert np.array_equal(apply(L, t, ti, 1, k["p"]), f32(k["t_mul_p_num"]) / np.float32(k["t_mul_p_den"]))  # w != 1 -> divide
    assert np.array_equal(apply(L, tp, tpi, 1, k["p"]), f32(k["tp_mul_p"]))  # w == 1 -> no divide
    assert np.array_equal(apply(L, t, ti, 2, k["n"]), f32(k["t_mul_n"]))  # normals use the inverse transpose
    assert np.array_equal(apply(L, t, ti, 0, k["ray_d"]), f32(k["t_mul_ray_d"]))
    # Bounds3 transform == union of the transformed corners
    out6 = np.zeros(6, dtype=np.float32)
    L.orc_transform_bounds_f32(_p(t.reshape(16)), _p(ti.reshape(16)), _p(f32(k["bounds_min"])), _p(f32(k["bounds_max"])), _p(out6))
    lo, hi = k["bounds_min"], k["bounds_max"]
    corners = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])], dtype=np.float32)
    tc = np.array([apply(L, t, ti, 1, c) for c in corners])
    assert np.array_equal(out6[:3], tc.min(axis=0)) and np.array_equal(out6[3:], tc.max(axis=0))
    # (t*tp).m_inv = tp.m_inv * t.m_inv vs inverse of the product
    prod = np.zeros(16, dtype=np.float32)
    L.orc_mat4_mul_f32(_p(t.reshape(16)), _p(tp.reshape(16)), _p(prod))
    a = inv32(L, prod.reshape(4, 4))
    b = np.zeros(16, dtype=np.float32)
    L.orc_mat4_mul_f32(_p(f32(tpi).reshape(16)), _p(f32(ti).reshape(16)), _p(b))
    assert np.abs(a - b.reshape(4, 4)).max() <= k["inverse_product_eps"]


def test_matrix_inverted_and_mul(oracle):
    L = oracle.lib()
    k = KATS["matrix_inverted"]
    m = f32(k["m"])
    mi = inv32(L, m)
    assert np.abs(inv32(L, mi) - m).max() <= k["eps"]
    prod = np.zeros(16, dtype=np.float32)
    L.orc_mat4_mul_f32(_p(m.reshape(16)), _p(f32(mi).reshape(16)), _p(prod))
    assert np.abs(prod.reshape(4, 4) - np.eye(4)).max() <= k["eps"]
    k = KATS["matrix_mul"]
    L.orc_mat4_mul_f32(_p(f32(k["m"]).reshape(16)), _p(f32(k["m"]).reshape(16)), _p(prod))
    assert np.array_equal(prod.reshape(4, 4), f32(k["mm"]))


def test_translation_scale_inverse_exact(oracle):
    L = oracle.lib()
    d = KATS["translation"]["delta"]
    tm = np.eye(4, dtype=np.float32)
    tm[:3, 3] = d
    ti = np.eye(4, dtype=np.float32)
    ti[:3, 3] = [-x for x in d]
    assert np.array_equal(inv32(L, tm), ti)  # tests/src/transform.rs:170 expects exact equality
    s = KATS["scale"]["s"]
    assert np.array_equal(inv32(L, np.diag(s + [1]).astype(np.float32)), np.diag([1 / np.float32(x) for x in s] + [1]).astype(np.float32))


def test_rotations_f64(oracle):
    L = oracle.lib()
    for name, axis in (("rotation_x", 0), ("rotation_y", 1), ("rotation_z", 2), ("rotation", 3)):
        k = KATS[name]
        m, mi = np.zeros(16), np.zeros(16)
        av = f64(k.get("axis", [0, 0, 1]))
        L.orc_rotation_f64(axis, k["theta"], _p(av), _p(m), _p(mi))
        assert np.abs(m.reshape(4, 4) - f64(k["m"])).max() <= max(k["eps"], 1.3e-16), name
        assert np.abs(mi.reshape(4, 4) - f64(k["m"]).T).max() <= max(k["eps"], 1.3e-16), name


def test_look_at_f64(oracle):
    L = oracle.lib()
    k = KATS["look_at"]
    m, mi = np.zeros(16), np.zeros(16)
    L.orc_look_at_f64(_p(f64(k["pos"])), _p(f64(k["target"])), _p(f64(k["up"])), _p(m), _p(mi))
    assert np.abs(mi.reshape(4, 4) - f64(k["m_inv"])).max() <= k["eps"]


def test_vector_ops(oracle):
    L = oracle.lib()
    k = KATS["vector"]
    out = np.zeros(9, dtype=np.float32)
    L.orc_vec3_ops_f32(_p(f32(k["cross"]["a"])), _p(f32(k["cross"]["b"])), _p(out))
    assert np.array_equal(out[:3], f32(k["cross"]["expect"]))
    L.orc_vec3_ops_f32(_p(f32(k["dot"]["a"])), _p(f32(k["dot"]["b"])), _p(out))
    assert out[3] == k["dot"]["expect"]
    a = f32(k["len"]["a"])
    L.orc_vec3_ops_f32(_p(a), _p(a), _p(out))
    assert out[4] == np.sqrt(np.float32(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]))
    a = f32(k["normalized"]["a"])
    L.orc_vec3_ops_f32(_p(a), _p(a), _p(out))
    assert abs(np.linalg.norm(out[5:8].astype(np.float64)) - 1.0) < 1e-6
    a = f32(k["max_dimension"]["a"])
    L.orc_vec3_ops_f32(_p(a), _p(a), _p(out))
    assert out[8] == k["max_dimension"]["expect"]
    # tie rules of Vec3::max_dimension (vector.rs:181-195): x>y ? (x>z?0:2) : (y>z?1:2)
    for v, e in (([1, 1, 1], 2), ([2, 1, 2], 2), ([2, 2, 1], 1), ([3, 1, 2], 0)):
        L.orc_vec3_ops_f32(_p(f32(v)), _p(f32(v)), _p(out))
        assert out[8] == e


def test_bounds_ops(oracle):
    L = oracle.lib()
    k = KATS["bounds"]
    out = np.zeros(6, dtype=np.float32)
    o = k["offset"]
    L.orc_bounds_ops_f32(_p(f32(o["min"])), _p(f32(o["max"])), _p(f32(o["p"])), _p(out))
    assert np.array_equal(out[:3], f32(o["expect"]))
    for c in k["surface_area"]:
        L.orc_bounds_ops_f32(_p(f32(c["min"])), _p(f32(c["max"])), _p(f32(c["min"])), _p(out))
        assert out[3] == c["expect"]
    v = k["volume"]
    L.orc_bounds_ops_f32(_p(f32(v["min"])), _p(f32(v["max"])), _p(f32(v["min"])), _p(out))
    assert out[4] == v["expect"]
    for c in k["maximum_extent"]:
        L.orc_bounds_ops_f32(_p(f32(c["min"])), _p(f32(c["max"])), _p(f32(c["min"])), _p(out))
        assert out[5] == c["expect"]
    # offset with a degenerate axis leaves that component undivided (impl_bounds.rs:110-114)
    L.orc_bounds_ops_f32(_p(f32([1, 2, 3])), _p(f32([1, 5, 6])), _p(f32([3, 3.5, 4.5])), _p(out))
    assert np.array_equal(out[:3], f32([2, 0.5, 0.5]))
    assert np.float32(k["default_min"]) == np.finfo(np.float32).max and np.float32(k["default_max"]) == np.finfo(np.float32).min


def test_point_dist_and_normal_dot(oracle):
    L = oracle.lib()
    k = KATS["point_dist"]
    out = np.zeros(9, dtype=np.float32)
    d = f32(k["dir"])
    L.orc_vec3_ops_f32(_p(d), _p(d), _p(out))
    step = out[5:8] * np.float32(k["scale"])
    L.orc_vec3_ops_f32(_p(f32(step)), _p(f32(step)), _p(out))
    assert abs(out[4] - k["dist"]) < 1e-6 and abs(out[3] - k["dist_sqr"]) < 1e-5
    n = KATS["normal"]
    L.orc_vec3_ops_f32(_p(f32(n["n"])), _p(f32(n["v"])), _p(out))
    assert out[3] == n["dot_v"]


def test_slab_test_semantics(oracle):
    """Bounds3::slab_test (bounds.rs:176-195): clamped to [0, t_max], NaN lanes
    (0 * inf) dropped like Rust's f32::min/max."""
    L = oracle.lib()
    tmin, tmax = C.c_float(), C.c_float()

    def slab(lo, hi, o, d, t_max=np.inf):
        with np.errstate(all="ignore"):
            hit = L.orc_slab_test_f32(_p(f32(lo)), _p(f32(hi)), _p(f32(o)), _p(f32(d)), t_max, C.byref(tmin), C.byref(tmax))
        return hit, tmin.value, tmax.value

    assert slab([1, 1, 1], [2, 2, 2], [0, 0, 0], [1, 1, 1]) == (1, 1.0, 2.0)
    assert slab([1, 1, 1], [2, 2, 2], [0, 0, 0], [1, 1, 1], 0.5)[0] == 0  # beyond t_max
    assert slab([1, 1, 1], [2, 2, 2], [3, 3, 3], [1, 1, 1])[0] == 0  # behind the origin
    assert slab([-1, -1, -1], [1, 1, 1], [0, 0, 0], [0, 0, 1]) == (1, 0.0, 1.0)  # origin inside, tmin clamped to 0
    # ray in the plane x == lo.x with d.x == 0: (lo.x - o.x) * inf = NaN is dropped
    hit, a, b = slab([1, -1, -1], [2, 1, 1], [1, 0, -5], [0, 0, 1])
    assert hit == 0  # min(NaN, +inf) = +inf -> entry at +inf
    hit, a, b = slab([1, -1, -1], [2, 1, 1], [1.5, 0, -5], [0, 0, 1])
    assert (hit, a, b) == (1, 4.0, 6.0)
    # flat box is still hit (tmin <= tmax, not <)
    assert slab([0, 0, 1], [1, 1, 1], [0.5, 0.5, 0], [0, 0, 1])[0] == 1


def test_coordinate_system_quirk(oracle):
    """math/mod.rs:26-34: the |x| <= |y| branch divides by (y*y + z + z), no sqrt."""
    L = oracle.lib()
    v1, v2 = np.zeros(3, dtype=np.float32), np.zeros(3, dtype=np.float32)
    v = f32([0.0, 0.6, 0.8])
    L.orc_coordinate_system_f32(_p(v), _p(v1), _p(v2))
    den = np.float32(v[1] * v[1] + v[2] + v[2])
    assert np.array_equal(v1, f32([0.0, v[2] / den, -v[1] / den]))
    v = f32([0.8, 0.0, 0.6])
    L.orc_coordinate_system_f32(_p(v), _p(v1), _p(v2))
    den = np.sqrt(np.float32(v[0] * v[0] + v[2] * v[2]))
    assert np.array_equal(v1, f32([-v[2] / den, 0.0, v[0] / den]))


def _kat(L, op, a, b):
    out = np.full(16, np.nan, dtype=np.float32)
    aa = f32(a)
    bb = None if b is None else f32(b)
    L.orc_math_kat_f32(op, _p(aa), None if bb is None else _p(bb), _p(out))
    return out


def test_remaining_reference_math_tests(oracle):
    """The path-relevant remainder of tests/src/{bounds,point,normal,vector,ray}.rs (KATS["more"]), one operation per case:
    exact equality, as the reference's assert_eq! / assert_abs_diff_eq! (default epsilon) on these values demand."""
    L = oracle.lib()
    more = KATS["more"]
    assert len(more["cases"]) >= 50
    for c in more["cases"]:
        got = _kat(L, c["op"], c["a"], c["b"])
        want = f32(c["expect"])
        assert np.array_equal(got[: len(want)], want), (c["name"], got, want, c["source"])
    # assert_abs_diff_eq! with the default epsilon (f32::EPSILON)
    eps = np.finfo(np.float32).eps
    for c in more["abs_diff_cases"]:
        got = _kat(L, c["op"], c["a"], c["b"])
        if "expect_len" in c:
            n = np.sqrt(got[0] * got[0] + got[1] * got[1] + got[2] * got[2])  # Normal::len: f32 sum left to right, one sqrt
            assert abs(np.float32(n) - np.float32(c["expect_len"])) <= eps, c["name"]
        else:
            assert abs(got[0] - np.float32(c["expect"][0])) <= eps, (c["name"], got[0])
    out = np.zeros(6, dtype=np.float32)
    for c in more["surface_area_more"]:
        box = _kat(L, 19, c["min"], c["max"])  # Bounds3::new sorts the corners first
        L.orc_bounds_ops_f32(_p(f32(box[:3])), _p(f32(box[3:])), _p(f32(box[:3])), _p(out))
        assert out[3] == c["expect"]
    for c in more["volume_more"]:
        box = _kat(L, 19, c["min"], c["max"])
        L.orc_bounds_ops_f32(_p(f32(box[:3])), _p(f32(box[3:])), _p(f32(box[:3])), _p(out))
        assert out[4] == c["expect"]
    for c in more["maximum_extent_more"]:
        L.orc_bounds_ops_f32(_p(f32(c["min"])), _p(f32(c["max"])), _p(f32(c["min"])), _p(out))
        assert out[5] == c["expect"]
    # Bounds3::bounding_sphere (bounds.rs:156-169 of the crate): centre = (p_min + p_max) / 2, radius = dist(centre, p_max)
    bs = more["bounding_sphere"]
    centre = (f32(bs["min"]) + f32(bs["max"])) / np.float32(2)
    assert np.array_equal(centre, f32(bs["center"]))
    r = _kat(L, 16, centre, bs["max"])[0]
    assert r == np.sqrt(np.float32(3 * 1.5 * 1.5))


def test_reference_test_map_is_current():
    """tests/golden/REFERENCE_TESTS.md (which of the reference's 121 unit tests are replayed, by name and line range) is what
    tools/reference_test_map.py makes of the KAT file's citations today.  Needs the reference's test sources for the names and
    line ranges, so it runs in the build container only."""
    import importlib.util, os

    if not os.path.isdir("/root/reference/tests/src"):
        pytest.skip("reference sources not present on this machine")
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("reference_test_map", os.path.join(here, "tools", "reference_test_map.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    text, n_replayed, n_total = mod.render()
    assert n_total == 121 and n_replayed >= 61
    assert open(os.path.join(here, "tests", "golden", "REFERENCE_TESTS.md")).read() == text
