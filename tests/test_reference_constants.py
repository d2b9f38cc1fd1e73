"""cfg1's scene against the reference's own text ('"constants verbatim" is unchecked').

tools/reference_cornell_check.py parses `Scene::cornell()` (scene/mod.rs:154-530) — constants, every mesh, material
order and constructors, sphere, light, camera, BVH parameters — and compares with yuki_amd.scenes.cornell().  Needs
/root/reference (build container only; nothing of it is copied: the script reads numbers out of the text at run time)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import reference_cornell_check as rc  # noqa: E402

from yuki_amd import scenes  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(rc.REF), reason="the reference's sources are only in the build container")


def test_f32_constant_folding():
    env = {}
    env["A"] = rc.f32_eval("555.0", env)
    env["B"] = rc.f32_eval("(A + 0.0) / 2.0", env)
    assert env["B"] == np.float32(277.5)
    assert rc.f32_eval("550.0 + 550.0 * 0.025", {}) == np.float32(550.0) + np.float32(550.0) * np.float32(0.025)
    assert rc.f32_eval("0.271_05", {}) == np.float32(0.27105)


def test_cornell_matches_the_references_text():
    ref, problems = rc.check()
    assert problems == []
    assert len(ref["meshes"]) == 14 and sum(len(p) for _, p, _ in ref["meshes"]) == 60


@pytest.mark.parametrize("what", ["vertex", "index", "material", "copper", "camera", "light", "sphere"])
def test_the_check_sees_a_changed_constant(monkeypatch, what):
    """The comparison is not vacuous: one changed constant of each kind is reported."""
    real = scenes.cornell

    def changed():
        sd = real()
        if what == "vertex":
            sd.points = sd.points.copy()
            sd.points[17, 2] = np.nextafter(sd.points[17, 2], np.float32(1))
        elif what == "index":
            sd.indices = sd.indices.copy()
            sd.indices[5] = sd.indices[5][::-1]
        elif what == "material":
            sd.tri_material = sd.tri_material.copy()
            sd.tri_material[20] = 3  # the right wall's green on the back wall
        elif what == "copper":
            sd.materials = [dict(m) for m in sd.materials]
            sd.materials[5]["b"] = (3.6092, 2.6248, 2.2922)
        elif what == "camera":
            sd.camera = dict(sd.camera, target=(0.278, 0.273, -0.26001))
        elif what == "light":
            sd.lights = [dict(sd.lights[0], L=tuple(np.nextafter(np.float32(v), np.float32(0)) for v in sd.lights[0]["L"]))]
        else:
            sd.spheres = [dict(sd.spheres[0], radius=0.0821)]
        return sd

    monkeypatch.setattr(scenes, "cornell", changed)
    _, problems = rc.check()
    assert problems, what


def _rust_f32_array(src, name):
    import re

    body = re.search(r"const %s: \[f32; \w+\] = \[(.*?)\];" % name, src, re.S).group(1)
    return np.asarray([float(v.replace("_", "")) for v in re.findall(r"-?\d[\d_]*\.?[\d_]*(?:e-?\d+)?", body)], dtype=np.float32)


def _cxx_f32_array(src, name):
    import re

    body = re.search(r"const float %s\[\d+\] = \{(.*?)\};" % name, src, re.S).group(1)
    return np.asarray([float(v) for v in re.findall(r"-?\d+\.?\d*(?:e-?\d+)?(?=f)", body)], dtype=np.float32)


def test_copper_tables_and_cie_fits_match_the_references_text():
    """The `metal` defaults (pbrt/mod.rs:1027-1105: 56 wavelengths, n, k) in the product loader's arrays and in the
    oracle loader's data file, and the Wyman-Sloan-Shirley fit constants and XYZ -> RGB matrix (cie.rs:4-21,
    pbrt/mod.rs:1011-1015) in both loaders, as binary32 values."""
    import json
    import re

    ref = open("/root/reference/yuki/src/scene/pbrt/mod.rs").read()
    cxx = open(os.path.join(ROOT, "yuki_amd", "csrc", "yk_loaders.cpp")).read()
    with open(os.path.join(ROOT, "tests", "golden", "copper_spd.json")) as f:
        data = json.load(f)
    for name, key in (("COPPER_WAVELENGTHS", "wavelengths"), ("COPPER_N", "n"), ("COPPER_K", "k")):
        want = _rust_f32_array(ref, name)
        assert want.size == 56
        assert np.array_equal(want, _cxx_f32_array(cxx, name)), name
        assert np.array_equal(want, np.asarray(data[key], dtype=np.float32)), key
    # every float literal of cie.rs, in order, must appear in the same order in both loaders' fit code
    cie = open("/root/reference/yuki/src/scene/pbrt/cie.rs").read()
    cie = cie[cie.index("pub fn x_fit_1931"):]
    want = [np.float32(v) for v in re.findall(r"(?<![\w.])\d+\.\d+", cie)]
    a = cxx.index("expf_once(float x)")
    got_cxx = [np.float32(v) for v in re.findall(r"(?<![\w.])(\d+\.\d+)f", cxx[a:cxx.index("void sampled_spectrum_into_rgb")])]
    assert got_cxx == want, (len(got_cxx), len(want))
    py_fit = open(os.path.join(ROOT, "oracle", "loaders.py")).read()
    py_fit = py_fit[py_fit.index("def _fit("):py_fit.index("F(3.240479)")]
    got_py_fit = {np.float32(v) for v in re.findall(r"(?<![\w.\[])\d+\.\d+", py_fit)} - {np.float32(0.5)}  # every centre once there, twice in cie.rs
    assert got_py_fit == set(want) - {np.float32(0.5)} and len(got_py_fit) == 28
    matrix = [np.float32(v.replace("_", "")) for v in re.findall(r"(?<![\w.])\d\.[\d_]+", ref[ref.index("3.240_479"):ref.index("fn is_sorted")])]
    py = open(os.path.join(ROOT, "oracle", "loaders.py")).read()
    got_py = [np.float32(v) for v in re.findall(r"F\((-?\d\.\d+)\)", py[py.index("F(3.240479)") - 4:py.index("def _copper")])]
    assert [abs(v) for v in got_py] == matrix
    b = cxx.index("rgb[0] = ")
    assert [np.float32(v) for v in re.findall(r"(\d\.\d+)f", cxx[b:cxx.index("measured copper")])] == matrix
