"""GPU parity, stage by stage: every kernel-level building block of the wavefront
against the CPU oracle on the same seeded inputs.  Bit-exact unless stated."""
import numpy as np
import pytest

from yuki_amd import abi, scenes

pytestmark = pytest.mark.gpu

SEED = 0x73B9642E74AC471C


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _same_or_both_nan(a, b):
    return (_bits(a) == _bits(b)) | (np.isnan(a) & np.isnan(b))


def test_device_libm_matches_oracle_bit_exact(ctx, yk, oracle):
    """yk_libm.h on gfx950 against oracle/olibm.h (= glibc 2.35, tests/test_oracle_libm.py): every 251st binary32
    value (17.1 M arguments over all exponents, signs, denormals, inf, NaN) plus the ranges the renderer feeds, for
    sinf, cosf, tanf, logf, acosf; atan2f on the same arguments against a shuffled copy.  NaN results compare as a
    class (x86 and gfx950 produce different default-NaN signs).  The full 2^32 sweep is tools/gpu_libm_exhaustive.py
    (profiles/r03_device_libm_exhaustive.txt)."""
    rng = np.random.default_rng(1)
    strided = np.arange(0, 1 << 32, 251, dtype=np.uint64).astype(np.uint32).view(np.float32)
    x = np.concatenate([strided, rng.uniform(-7, 7, 200000).astype(np.float32), rng.uniform(-1e4, 1e4, 20000).astype(np.float32),
                        np.array([0.0, -0.0, 1e-30, 3.1415927, 1.5707964, 0.7853982, 1.0, -1.0, 0.5, -0.5], np.float32)])
    for fn, name in [(0, "sinf"), (1, "cosf"), (2, "tanf"), (3, "logf"), (4, "acosf")]:
        got = yk.device_math(ctx, fn, x)
        want = oracle.libm_array(fn, x)
        ok = _same_or_both_nan(got, want)
        assert ok.all(), (name, int((~ok).sum()), x[~ok][:6], got[~ok][:6], want[~ok][:6])
    # the pair the shading code calls (one reduction, both results): the same bits as the two functions
    for fn, ofn in ((28, 0), (29, 1)):
        ok = _same_or_both_nan(yk.device_math(ctx, fn, x), oracle.libm_array(ofn, x))
        assert ok.all(), ("sincos pair", fn, int((~ok).sum()), x[~ok][:6])
    y = np.concatenate([rng.permutation(strided), rng.uniform(-3, 3, 220000).astype(np.float32), np.array([0.0, -0.0, 1.0, -1.0, 0.0, 1e-30, -0.0, 2.0, 0.0, 0.0], np.float32)])
    got = yk.device_math(ctx, 5, y, x)
    want = oracle.libm_array(5, y, x)
    ok = _same_or_both_nan(got, want)
    assert ok.all(), ("atan2f", int((~ok).sum()), y[~ok][:6], x[~ok][:6], got[~ok][:6], want[~ok][:6])


def test_oracle_libm_is_this_hosts_libm(oracle):
    """The same question as tests/test_oracle_libm.py, asked on the GPU box's own host (its glibc, its CPU's ifunc
    choice): the oracle's restatement against the `hostlibm` build of the oracle, which calls the platform's functions."""
    import ctypes

    try:
        f = ctypes.CDLL(None).gnu_get_libc_version
        f.restype = ctypes.c_char_p
        ver = tuple(int(v) for v in f().decode().split(".")[:2])
        with open("/proc/cpuinfo") as fh:
            fma = any(" fma " in line + " " for line in fh if line.startswith("flags"))
    except (AttributeError, OSError, ValueError):
        pytest.skip("cannot tell the platform")
    if ver < (2, 35) or not fma:
        pytest.skip("not the platform the oracle restates")
    rng = np.random.default_rng(3)
    x = np.concatenate([np.arange(0, 1 << 32, 1021, dtype=np.uint64).astype(np.uint32).view(np.float32), rng.uniform(-7, 7, 200000).astype(np.float32)])
    y = rng.permutation(x)
    for fn in range(7):  # 6 = expf (host side only: the pbrt loader's CIE fits)
        mine = oracle.libm_array(fn, x, y if fn == 5 else None)
        with oracle.flavour("hostlibm"):
            host = oracle.libm_array(fn, x, y if fn == 5 else None)
        assert _same_or_both_nan(mine, host).all(), fn


def test_device_sqrt_div_are_correctly_rounded(ctx, yk):
    """f32 sqrt and division on gfx950 must equal IEEE (numpy) results: the
    reference's f32::sqrt and `/` are correctly rounded."""
    rng = np.random.default_rng(2)
    a = np.concatenate([rng.uniform(0, 1e6, 300000), rng.uniform(0, 1e-30, 1000), 10.0 ** rng.uniform(-38, 38, 100000)]).astype(np.float32)
    b = np.concatenate([rng.uniform(-1e3, 1e3, 300000), rng.uniform(1e-20, 1e-10, 1000), 10.0 ** rng.uniform(-30, 30, 100000)]).astype(np.float32)
    assert np.array_equal(_bits(yk.device_math(ctx, 6, a)), _bits(np.sqrt(a)))
    assert np.array_equal(_bits(yk.device_math(ctx, 8, a)), _bits(np.sqrt(a.astype(np.float64)).astype(np.float32)))
    with np.errstate(all="ignore"):
        assert np.array_equal(_bits(yk.device_math(ctx, 7, a, b)), _bits(a / b))


@pytest.mark.parametrize("kind", ["uniform", "stratified", "stratified_nojitter"])
def test_sampler_sequence(ctx, yk, oracle, kind):
    if kind == "uniform":
        s = yk.SamplerType.Uniform(16, SEED)
    else:
        s = yk.SamplerType.Stratified((8, 8), kind == "stratified", SEED)
    dims = np.array([2, 2, 2, 2, 1, 2, 2, 1, 1, 2] * 4, dtype=np.uint8)
    import ctypes as C

    for px, py, idx in [(0, 0, 0), (17, 1033, 5), (1919, 1079, 15), (65535, 65535, 3)]:
        got = yk.sampler_sequence(ctx, s, px, py, idx, dims)
        want = np.zeros((len(dims), 2), dtype=np.float32)
        oracle.lib().orc_sampler_sequence(C.byref(s), px, py, idx, dims.ctypes.data_as(C.c_void_p), len(dims), want.ctypes.data_as(C.c_void_p))
        assert np.array_equal(_bits(got), _bits(want)), (kind, px, py, idx)


def test_camera_rays(ctx, yk, oracle):
    sd = scenes.by_name("cfg2")
    fs = yk.FilmSettings(res=(1920, 1080))
    cam = yk.Camera(sd.camera, fs)
    s = yk.SamplerType.Stratified((4, 4), True, SEED)
    for tile, idx in [((0, 0, 16, 16), 0), ((1904, 1072, 1920, 1080), 7), ((960, 528, 976, 544), 15)]:
        o1, d1 = yk.camera_rays(ctx, cam, s, tile, idx)
        o2, d2 = oracle.camera_rays(cam.matrices, s, tile, idx)
        assert np.array_equal(_bits(o1), _bits(o2))
        assert np.array_equal(_bits(d1), _bits(d2))


def _random_rays(sd, n, seed):
    rng = np.random.default_rng(seed)
    lo, hi = sd.points.min(axis=0), sd.points.max(axis=0)
    ext = hi - lo
    o = (lo - 0.3 * ext + rng.uniform(0, 1, (n, 3)) * 1.6 * ext).astype(np.float32)
    tgt = (lo + rng.uniform(0, 1, (n, 3)) * ext).astype(np.float32)
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # a share of axis-aligned and zero-component directions (0*inf NaN lanes in the slab test)
    k = n // 20
    d[:k] = 0
    d[np.arange(k), rng.integers(0, 3, k)] = rng.choice([-1.0, 1.0], k)
    d[k : 2 * k, rng.integers(0, 3)] = 0
    return o, d.astype(np.float32)


@pytest.mark.parametrize("name", ["cornell", "cornell-tris", "city-small", "cfg2"])
def test_trace_closest_bit_exact(ctx, yk, oracle, name):
    sd = scenes.by_name(name)
    sc = yk.Scene(ctx, sd)
    osc = oracle.OracleScene(sd)
    o, d = _random_rays(sd, 60000, 3)
    got = sc.intersect(o, d, counters=True)
    want = osc.intersect(o, d)
    assert np.array_equal(got["shape"], want["shape"])
    hit = want["shape"] >= 0
    assert hit.sum() > 1000
    assert np.array_equal(_bits(got["t"][hit]), _bits(want["t"][hit]))
    # IntersectionResult counters (bvh.rs:167-179) — the BVHIntersections integrator's output
    assert np.array_equal(got["node_tests"], want["node_tests"])
    assert np.array_equal(got["node_hits"], want["node_hits"])
    assert np.array_equal(got["shape_tests"], want["shape_tests"])
    # with a finite t_max
    tm = (np.abs(np.random.default_rng(4).normal(1.0, 0.7, o.shape[0])) + 0.01).astype(np.float32)
    got = sc.intersect(o, d, t_max=tm)
    want = osc.intersect(o, d, t_max=tm)
    assert np.array_equal(got["shape"], want["shape"])


@pytest.mark.parametrize("name", ["cornell", "cornell-tris", "city-small"])
def test_trace_any_bit_exact(ctx, yk, oracle, name):
    sd = scenes.by_name(name)
    sc = yk.Scene(ctx, sd)
    osc = oracle.OracleScene(sd)
    o, d = _random_rays(sd, 60000, 5)
    rng = np.random.default_rng(6)
    d = (d * rng.uniform(0.2, 3.0, (o.shape[0], 1))).astype(np.float32)  # shadow rays are not normalised
    tm = np.full(o.shape[0], 0.9999, dtype=np.float32)
    al = rng.integers(-1, max(1, len(sd.lights)), o.shape[0]).astype(np.int32)
    got = sc.any_intersect(o, d, tm, al)
    want = osc.any_intersect(o, d, tm, al)
    assert np.array_equal(got, want)
    assert 0.05 < want.mean() < 0.95


@pytest.mark.parametrize("option,value", [("wide_bvh", 1), ("top_nodes", 0)])
@pytest.mark.parametrize("name", ["cornell", "city-small"])
def test_trace_layout_options_bit_exact(yk, oracle, name, option, value):
    """4-wide collapsed nodes / no LDS-resident tree top: same hits (ids — hence exact-t tie
    winners — and t) for random rays incl. axis-aligned ones, closest and any-hit."""
    c = yk.Context(0, **{option: value})
    try:
        sd = scenes.by_name(name)
        sc = yk.Scene(c, sd)
        osc = oracle.OracleScene(sd)
        o, d = _random_rays(sd, 60000, 11)
        got, want = sc.intersect(o, d), osc.intersect(o, d)
        assert np.array_equal(got["shape"], want["shape"])
        hit = want["shape"] >= 0
        assert np.array_equal(_bits(got["t"][hit]), _bits(want["t"][hit]))
        tm = (np.abs(np.random.default_rng(4).normal(1.0, 0.7, o.shape[0])) + 0.01).astype(np.float32)
        assert np.array_equal(sc.intersect(o, d, t_max=tm)["shape"], osc.intersect(o, d, t_max=tm)["shape"])
        rng = np.random.default_rng(6)
        ds = (d * rng.uniform(0.2, 3.0, (o.shape[0], 1))).astype(np.float32)
        t1 = np.full(o.shape[0], 0.9999, dtype=np.float32)
        al = rng.integers(-1, max(1, len(sd.lights)), o.shape[0]).astype(np.int32)
        assert np.array_equal(sc.any_intersect(o, ds, t1, al), osc.any_intersect(o, ds, t1, al))
    finally:
        c.close()


MATERIALS = [
    dict(kind=abi.MAT_MATTE, a=(0.7, 0.5, 0.3), c=0.0),
    dict(kind=abi.MAT_MATTE, a=(0.7, 0.5, 0.3), c=0.349),
    dict(kind=abi.MAT_MATTE, a=(0, 0, 0), c=0.0),
    dict(kind=abi.MAT_GLASS, a=(1, 1, 1), b=(0.9, 0.95, 1.0), c=1.5),
    dict(kind=abi.MAT_METAL, a=(0.27105, 0.67693, 1.31640), b=(3.60920, 2.62480, 2.29210), c=0.01, remap=True),
    dict(kind=abi.MAT_METAL, a=(0.2, 0.9, 1.1), b=(3.9, 2.4, 2.2), c=0.2, remap=False),
    dict(kind=abi.MAT_GLOSSY, a=(0.8, 0.6, 0.2), c=0.3, remap=False),
    dict(kind=abi.MAT_GLOSSY, a=(0.8, 0.6, 0.2), c=0.4, remap=True),
]


def _mat(m):
    d = abi.MaterialDesc()
    d.kind = m["kind"]
    d.a = abi.f3(m.get("a", (0, 0, 0)))
    d.b = abi.f3(m.get("b", (0, 0, 0)))
    d.c = m.get("c", 0.0)
    d.flags = 1 if m.get("remap") else 0
    return d


@pytest.mark.parametrize("mi", range(len(MATERIALS)))
def test_bsdf_f_and_sample_f_bit_exact(ctx, yk, oracle, mi):
    import ctypes as C

    m = _mat(MATERIALS[mi])
    rng = np.random.default_rng(10 + mi)
    n = 4000

    def unit(k):
        v = rng.normal(size=(k, 3))
        return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)

    ns = unit(n)
    ng = (ns + 0.1 * unit(n)).astype(np.float32)
    ng /= np.linalg.norm(ng, axis=1, keepdims=True)
    dpdu = np.cross(ns, unit(n)).astype(np.float32)
    wo, wi = unit(n), unit(n)
    u = rng.uniform(0, 1, (n, 2)).astype(np.float32)
    got_f = yk.bsdf_eval(ctx, m, ng, ns, dpdu, wo, wi)
    got_s = yk.bsdf_sample(ctx, m, ng, ns, dpdu, wo, u)
    want_f = np.zeros((n, 3), dtype=np.float32)
    want_s = np.zeros((n, 8), dtype=np.float32)
    L = oracle.lib()
    p = lambda a, i: a[i : i + 1].ctypes.data_as(C.c_void_p)
    for i in range(n):
        L.orc_bsdf_eval(C.byref(m), p(ng, i), p(ns, i), p(dpdu, i), p(wo, i), p(wi, i), p(want_f, i))
        L.orc_bsdf_sample(C.byref(m), p(ng, i), p(ns, i), p(dpdu, i), p(wo, i), p(u, i), p(want_s, i))
    assert np.array_equal(_bits(got_f), _bits(want_f))
    assert np.array_equal(_bits(got_s), _bits(want_s))


def _rigid(rng):
    """random rotation + translation as row-major 4x4 float32 and its inverse"""
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    t = rng.uniform(-4, 4, 3)
    m = np.eye(4)
    m[:3, :3] = q
    m[:3, 3] = t
    return m.astype(np.float32), np.linalg.inv(m).astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["point", "spot", "distant", "rect"])
def test_light_sample_li_bit_exact(ctx, yk, oracle, kind):
    """Light::sample_li (lights/*.rs) and VisibilityTester::ray (visibility.rs:21-23, interaction.rs:44-59) as stages:
    l, li, pdf, the has-visibility flag, the area-light identity, p1 and the shadow ray of the device equal the oracle's
    bit for bit, for random lights of every kind against 4000 surface points each (points behind a rectangular light,
    outside / on the falloff of a spot cone and at distances from 1e-3 to 1e3 included)."""
    import ctypes as C

    rng = np.random.default_rng({"point": 1, "spot": 2, "distant": 3, "rect": 4}[kind])
    n = 4000
    L = oracle.lib()
    for trial in range(6):
        l2w, l2w_inv = _rigid(rng)
        d = abi.LightDesc()
        if kind == "point":
            yk.LightFactory.make_point_light(l2w, rng.uniform(0.1, 50, 3), d)
        elif kind == "spot":
            total = rng.uniform(10, 80)
            yk.LightFactory.make_spot_light(l2w, l2w_inv, rng.uniform(0.1, 50, 3), total, rng.uniform(1, total), d)
        elif kind == "rect":
            yk.LightFactory.make_rect_light(l2w, l2w_inv, rng.uniform(0.1, 20, 3), rng.uniform(0.05, 3, 2), d)
        else:
            w = rng.normal(size=3)
            d.kind = abi.LIGHT_DISTANT
            d.p = abi.f3(w / np.linalg.norm(w))
            d.i = abi.f3(rng.uniform(0.1, 5, 3))
        scale = 10.0 ** rng.uniform(-3, 3, (n, 1))
        p = (l2w[:3, 3] + rng.normal(size=(n, 3)) * scale).astype(np.float32)
        ng = rng.normal(size=(n, 3))
        ng = (ng / np.linalg.norm(ng, axis=1, keepdims=True)).astype(np.float32)
        u = rng.uniform(0, 1, (n, 2)).astype(np.float32)
        got = yk.light_sample(ctx, d, trial, p, ng, u)
        want = np.zeros((n, 18), dtype=np.float32)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        L.orc_light_sample(C.byref(d), trial, n, vp(p), vp(ng), vp(u), vp(want))
        assert np.array_equal(_bits(got), _bits(want)), (kind, trial)
        if kind == "spot":  # the three falloff branches all occur
            lum = want[:, 3:6].sum(axis=1)
            assert (lum == 0).any() and (want[:, 7] == 0).any() and (lum > 0).any()
        if kind == "rect":
            assert (want[:, 8] == trial).all() and (want[:, 3] == 0).any() and (want[:, 3] > 0).any()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(3000, 3021)))
def test_degenerate_rays_on_random_scenes(oracle, seed):
    """tools/stage_fuzz.py: intersect / any_intersect on random scenes for rays with exact zero
    direction components (NaN lanes in the slab test), origins on box faces, rays through vertices
    and along edges, |d| from 1e-6 to 1e6, t_max at the exact hit distance and one ulp around it:
    hit shape, t bits, the reference's three counters and the occlusion verdicts equal the oracle's."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import stage_fuzz

    assert stage_fuzz.check_seed(oracle, seed) == []


def test_device_vector_math_replays_the_reference_kats_and_matches_the_oracle(ctx, yk, oracle):
    """The reference's own vector known-answer values (tests/src/vector.rs, normal.rs — tests/golden/reference_math_kats.json)
    through the DEVICE versions of those functions (yk_math.h, used by every kernel), then 30,000 random triples against the
    oracle's, bit for bit: dot, the f64 cross product, len, normalized, max_dimension, abs, Normal::dot_v, min, max,
    faceforward, and the axis permutation Triangle::intersect derives from a ray direction (triangle.rs:60-66)."""
    import ctypes as C
    import json
    import os

    kats = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_math_kats.json")))
    f32 = lambda v: np.ascontiguousarray(v, dtype=np.float32)  # noqa: E731
    v = kats["vector"]
    assert yk.device_math(ctx, 11, f32(v["dot"]["a"]), f32(v["dot"]["b"]))[0] == v["dot"]["expect"]
    assert np.array_equal(yk.device_math(ctx, 12, f32(v["cross"]["a"]), f32(v["cross"]["b"])), f32(v["cross"]["expect"]))
    a = f32(v["len"]["a"])
    assert yk.device_math(ctx, 13, a)[0] == np.sqrt(np.float32(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]))
    assert abs(np.linalg.norm(yk.device_math(ctx, 14, f32(v["normalized"]["a"])).astype(np.float64)) - 1.0) < 1e-6
    assert yk.device_math(ctx, 15, f32(v["max_dimension"]["a"]))[0] == v["max_dimension"]["expect"]
    assert yk.device_math(ctx, 17, f32(kats["normal"]["n"]), f32(kats["normal"]["v"]))[0] == kats["normal"]["dot_v"]
    # orc_math_kat_f32 op -> yk_device_math fn, where the function exists on the device
    dev_of = {0: 19, 1: 20, 4: 15, 6: 16, 7: 26, 18: 27, 21: 22, 22: 23, 23: 24, 24: 25}
    replayed = 0
    for c in kats["more"]["cases"]:
        if c["op"] in dev_of:
            b = None if c["b"] is None else f32((list(c["b"]) + [0, 0])[:3])  # a scalar operand travels in b.x
            got = yk.device_math(ctx, dev_of[c["op"]], f32(c["a"]), b)
            assert np.array_equal(got[: len(c["expect"])], f32(c["expect"])), c["name"]
            replayed += 1
    assert replayed >= 22
    # tie rules of Vec3::max_dimension (vector.rs:181-195)
    for t, e in (([1, 1, 1], 2), ([2, 1, 2], 2), ([2, 2, 1], 1), ([3, 1, 2], 0)):
        assert yk.device_math(ctx, 15, f32(t))[0] == e
    # random triples against the oracle
    rng = np.random.default_rng(5)
    n = 30000
    A = (rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-3, 3, (n, 1))).astype(np.float32)
    B = (rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-3, 3, (n, 1))).astype(np.float32)
    A[:50, 0] = 0.0
    A[50:100] = np.abs(A[50:100, :1])  # equal components: the tie rules
    L = oracle.lib()
    want = np.zeros((n, 9), dtype=np.float32)
    for k in range(n):
        L.orc_vec3_ops_f32(A[k].ctypes.data_as(C.c_void_p), B[k].ctypes.data_as(C.c_void_p), want[k].ctypes.data_as(C.c_void_p))
    flatA, flatB = A.reshape(-1), B.reshape(-1)
    assert np.array_equal(_bits(yk.device_math(ctx, 12, flatA, flatB).reshape(n, 3)), _bits(want[:, 0:3]))
    assert np.array_equal(_bits(yk.device_math(ctx, 11, flatA, flatB).reshape(n, 3)[:, 0]), _bits(want[:, 3]))
    assert np.array_equal(_bits(yk.device_math(ctx, 13, flatA).reshape(n, 3)[:, 0]), _bits(want[:, 4]))
    assert np.array_equal(_bits(yk.device_math(ctx, 14, flatA).reshape(n, 3)), _bits(want[:, 5:8]))
    assert np.array_equal(yk.device_math(ctx, 15, flatA).reshape(n, 3)[:, 0], want[:, 8])
    kz = yk.device_math(ctx, 15, np.abs(flatA)).reshape(n, 3)[:, 0]
    perm = yk.device_math(ctx, 18, flatA).reshape(n, 3)
    assert np.array_equal(perm[:, 2], kz) and np.array_equal(perm[:, 0], (kz + 1) % 3) and np.array_equal(perm[:, 1], (kz + 2) % 3)
