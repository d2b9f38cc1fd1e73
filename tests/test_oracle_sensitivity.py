"""What "parity unpinned" leaves open, measured in image space (CPU only).

The oracle replaces two things the reference leaves to its platform — the libm behind Rust's
f32::{sin,cos,tan,ln,atan2,acos} and the order `select_nth_unstable_by` gives equal keys — by
fixed recipes (oracle/olibm.h, the select_nth spec in DESIGN.md §2).  Two more builds of the
oracle (oracle/Makefile `flavours`: glibc's functions; libstdc++'s nth_element) render the same
tiles; tools/libm_sensitivity.py writes the full table to profiles/r03_libm_sensitivity.txt.
Here a bounded subset is asserted against the north star's tolerance (RMSE < 1e-4):

* BASELINE-shaped workloads (cfg2's mesh, the city scenes behind cfg3 / cfg5) stay two to four
  orders of magnitude inside it;
* the Cornell box under the Path integrator does NOT: a handful of samples whose path forks on
  a last-bit difference (Russian roulette, a light sample landing on the other side of an edge)
  move pixels by 1e-2 under a 0.6 W/sr/m2 ceiling light — recorded, with its bound, so that a
  change in either recipe shows up.  Any two legitimate libms differ that way; no restatement
  can do better without the reference's own binary.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import libm_sensitivity as sens  # noqa: E402

from yuki_amd import abi, scenes  # noqa: E402

TOL = 1e-4  # BASELINE.json north_star: per-pixel radiance RMSE < 1e-4 vs the CPU reference


@pytest.fixture(scope="module")
def results():
    return sens.run(sens.QUICK, threads=4, log=lambda s: None)


def test_flavours_are_really_other_builds():
    from oracle import binding as oracle

    x = np.float32(0.1) * np.arange(1, 200000, dtype=np.float32)
    base = np.array([oracle.lib().orc_sinf(float(v)) for v in x[:20000]], dtype=np.float32)
    with oracle.flavour("hostlibm"):
        host = np.array([oracle.lib().orc_sinf(float(v)) for v in x[:20000]], dtype=np.float32)
    # the recipe is a correctly rounded sine almost everywhere, glibc's is within 1 ulp: they must
    # agree nearly always and never by more than an ulp, and the host build must be glibc's function
    ulp = np.abs(base.view(np.int32).astype(np.int64) - host.view(np.int32).astype(np.int64))
    assert ulp.max() <= 1
    import ctypes

    libm = ctypes.CDLL("libm.so.6")
    libm.sinf.restype = ctypes.c_float
    libm.sinf.argtypes = [ctypes.c_float]
    ref = np.array([libm.sinf(float(v)) for v in x[:20000]], dtype=np.float32)
    assert np.array_equal(host.view(np.uint32), ref.view(np.uint32))


def test_baseline_shaped_workloads_stay_inside_the_tolerance(results):
    for (name, fl), c in results.items():
        if name.startswith("cfg2") or "city" in name:
            assert c["rmse"] < TOL / 100, (name, fl, c)
            assert c["rays"][0] == c["rays"][1]


def test_cornell_path_miss_is_recorded(results):
    c = results[("golden cornell_path 32x32x4 (copper sphere: GGX + Sphere::intersect)", "hostlibm")]
    # measured 1.374e-03 with glibc 2.35 (3 of 4096 samples fork); the bound is loose on purpose
    assert c["samples_other_path"] <= 40
    assert c["rmse"] < 2e-2
    if c["rmse"] >= TOL:
        print(f"recorded: Cornell Path misses the 1e-4 tolerance under another libm by {c['rmse'] / TOL:.1f}x")


def test_selection_flavour_only_matters_where_the_selection_runs(results):
    # SAH / Middle trees never reach select_nth on these scenes: identical trees, identical images
    for (name, fl), c in results.items():
        if fl == "nth":
            assert c["nodes_same"] and c["order_diff"] == 0 and c["rmse"] == 0.0, (name, c)


def test_equal_counts_tree_differs_but_not_the_image():
    case = ("city-small", (96, 54), (abi.SAMPLER_STRATIFIED, 2, 2), (abi.INTEGRATOR_PATH, 8), None, abi.SPLIT_EQUAL_COUNTS)
    base = sens.render(case, "default", 4)
    other = sens.render(case, "nth", 4)
    c = sens.compare(base, other)
    assert not c["nodes_same"] or c["order_diff"] > 0  # the flavour does build another tree ...
    assert c["rmse"] == 0.0 and c["rays"][0] == c["rays"][1]  # ... whose image is the same bit for bit (no exact-t ties)


def test_committed_table_matches_the_code():
    """profiles/r03_libm_sensitivity.txt is the full run; its headline rows must be the ones DESIGN.md quotes."""
    txt = open(os.path.join(ROOT, "profiles", "r03_libm_sensitivity.txt")).read()
    assert "cfg3 (1,024,012 tri" in txt and "cfg2 (69,312 tri" in txt and "worst RMSE" in txt


def test_cornell_vertices_follow_transform_rs():
    """scene/mod.rs:177-185 + shapes/mesh.rs:27-29: the 4x4 applied as transform.rs:128-142 does — the twelve
    FRONT-plane vertices (z = 0) come out as +0.0, not -0.0, and the oracle's KAT-pinned
    Transform * Point3 gives the same bits for every vertex."""
    from oracle import binding as oracle
    import ctypes as C

    sd = scenes.cornell()
    z = sd.points[:, 2]
    assert int((z == 0).sum()) == 12 and not np.signbit(z[z == 0]).any()
    m = scenes._mat4_mul(np.diag(np.asarray([0.001, 0.001, 0.001, 1], dtype=np.float32)), np.diag(np.asarray([1, 1, -1, 1], dtype=np.float32)))
    rng = np.random.default_rng(5)
    p = np.concatenate([rng.uniform(-600, 600, (500, 3)), [[0, 0, 0], [0.0, 548.8, 0.0], [-0.0, 1.0, -0.0]]]).astype(np.float32)
    mine = scenes._transform_points(m, p)
    out = np.zeros_like(p)
    mi = np.linalg.inv(m.astype(np.float64)).astype(np.float32)
    for i in range(len(p)):
        o = np.zeros(3, dtype=np.float32)
        oracle.lib().orc_transform_apply_f32(m.ctypes.data_as(C.c_void_p), mi.ctypes.data_as(C.c_void_p), 1, p[i].ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p))
        out[i] = o
    assert np.array_equal(mine.view(np.uint32), out.view(np.uint32))
