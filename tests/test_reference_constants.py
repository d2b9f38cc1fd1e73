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


def test_pbrt_loader_defaults_match_the_references_text(tmp_path):
    """Every `params.find_*("name", default)` of scene/pbrt/mod.rs: the default is read out of the reference's text and compared
    with what both loaders produce for the BARE directive (no parameters given)."""
    import re

    from oracle import loaders as ol
    from yuki_amd import abi, loaders

    ref = open("/root/reference/yuki/src/scene/pbrt/mod.rs").read()
    ref = re.sub(r"\s+", " ", ref)
    found = []
    for m in re.finditer(r'find_(f32|i32|bool|spectrum|point|string)\( ?"(\w+)", ?', ref):
        depth, j = 1, m.end()
        while depth:  # the default runs to the call's closing parenthesis
            depth += {"(": 1, ")": -1}.get(ref[j], 0)
            j += 1
        found.append((m.group(1), m.group(2), ref[m.end():j - 1].strip().rstrip(",").strip()))
    names = [n for _, n, _ in found]
    # the order of appearance fixes which directive a repeated name ("L", "from", "roughness", "eta", "Kd") belongs to
    assert names == ["fov", "xresolution", "yresolution", "L", "L", "from", "to", "I", "from", "radius", "filename", "filename",
                     "Kr", "Kt", "eta", "Rs", "roughness", "Kd", "Kd", "sigma", "eta", "k", "roughness", "remaproughness"], names
    d = [v for _, _, v in found]

    def spectrum(text, variables={"default_l": "Spectrum::ones()", "default_i": "Spectrum::ones()"}):
        text = variables.get(text, text)
        if text == "Spectrum::ones()":
            return (1.0, 1.0, 1.0)
        m = re.fullmatch(r"Spectrum::new\(([\d.]+), ([\d.]+), ([\d.]+)\)", text)
        return tuple(float(v) for v in m.groups())

    def point(text, variables={"default_pos": "Point3::zeros()"}):
        text = variables.get(text, text)
        if text == "Point3::zeros()":
            return (0.0, 0.0, 0.0)
        return tuple(float(v) for v in re.fullmatch(r"Point3::new\(([\d.]+), ([\d.]+), ([\d.]+)\)", text).groups())

    assert 'let default_l = Spectrum::ones();' in ref and 'let default_i = Spectrum::ones();' in ref and 'let default_pos = Point3::zeros();' in ref
    p = str(tmp_path / "bare.pbrt")
    with open(p, "w") as f:
        f.write('Camera "perspective"\nFilm "image"\nWorldBegin\nLightSource "infinite"\nLightSource "distant"\nLightSource "point"\n'
                'Material "glass"\nShape "sphere"\nMaterial "glossy"\nShape "sphere"\nMaterial "matte"\nShape "sphere"\nMaterial "metal"\nShape "sphere"\nWorldEnd\n')
    want, wcam, wres = ol.load_pbrt(p)
    got, cam, film = loaders.load_pbrt(p)
    # both loaders agree on the file (the full comparison is tests/test_loaders.py's)
    assert [bytes(x) for x in want.light_structs] == [bytes(x) for x in got.light_structs] and len(want.materials) == len(got.materials)
    f32 = np.float32
    assert f32(wcam["fov_degrees"]) == f32(d[0]) == f32(cam.fov_degrees) and wcam["fov_axis"] == abi.FOV_Y == cam.fov_axis  # FoV::Y(find_f32("fov", 45.0))
    assert tuple(wres) == (int(d[1]), int(d[2])) == tuple(film.res)
    assert tuple(f32(v) for v in want.background) == tuple(f32(v) for v in spectrum(d[3]))  # infinite: L
    distant, pnt = got.light_structs[0], got.light_structs[1]
    assert distant.kind == abi.LIGHT_DISTANT and tuple(distant.i) == spectrum(d[4])  # distant: L; direction (from - to).normalized()
    frm, to = np.asarray(point(d[5]), f32), np.asarray(point(d[6]), f32)
    w = frm - to
    assert np.allclose(np.asarray(tuple(distant.p), f32), w / np.linalg.norm(w))  # a distant light keeps its direction in `p`
    assert pnt.kind == abi.LIGHT_POINT and tuple(pnt.i) == spectrum(d[7]) and tuple(pnt.p) == point(d[8])
    assert all(f32(s["radius"]) == f32(d[9]) for s in got.spheres) and len(got.spheres) == 4
    assert d[10] == d[11] == '""'
    glass = next(m for m in want.materials if m["kind"] == abi.MAT_GLASS)
    assert tuple(glass["a"]) == spectrum(d[12]) and tuple(glass["b"]) == spectrum(d[13]) and f32(glass["c"]) == f32(d[14])
    glossy = next(m for m in want.materials if m["kind"] == abi.MAT_GLOSSY)
    assert tuple(f32(v) for v in glossy["a"]) == tuple(f32(v) for v in spectrum(d[15])) and f32(glossy["c"]) == f32(d[16]) and not glossy["remap"]
    matte = [m for m in want.materials if m["kind"] == abi.MAT_MATTE][-1]
    assert d[17] == '""' and tuple(f32(v) for v in matte["a"]) == tuple(f32(v) for v in spectrum(d[18])) and f32(matte["c"]) == f32(d[19]) == f32(0.0)
    metal = next(m for m in want.materials if m["kind"] == abi.MAT_METAL)
    assert "sampled_spectrum_into_rgb(&COPPER_WAVELENGTHS, &COPPER_N)" in d[20] and "sampled_spectrum_into_rgb(&COPPER_WAVELENGTHS, &COPPER_K)" in d[21]
    lam, n, k = ol._copper()
    assert tuple(f32(v) for v in metal["a"]) == tuple(f32(v) for v in ol.sampled_spectrum_into_rgb(lam, n))
    assert tuple(f32(v) for v in metal["b"]) == tuple(f32(v) for v in ol.sampled_spectrum_into_rgb(lam, k))
    assert f32(metal["c"]) == f32(d[22]) and bool(metal["remap"]) == (d[23] == "true")


def test_permutation_element_is_the_references_statement_by_statement():
    """stratified.rs:147-178 against oracle/osampler.h and yuki_amd/csrc/yk_rng.h: the loop body as a normalised sequence of
    (operator, constant-or-shift) steps — multipliers, shift amounts and their order — read out of all three texts."""
    import re

    def steps(text, start, stop):
        body = text[text.index(start):]
        body = body[:body.index(stop)]
        out = []
        for line in body.splitlines():
            line = line.strip().rstrip(";")
            m = re.match(r"i = i\.wrapping_mul\((.*)\)$", line) or re.match(r"i \*= (.*)$", line)
            if m:
                out.append(("mul", m.group(1).replace("u", "").replace(" ", "").lower()))
                continue
            m = re.match(r"i \^= (.*)$", line)
            if m:
                out.append(("xor", m.group(1).replace(" ", "")))
                continue
            m = re.match(r"i &= (.*)$", line)
            if m:
                out.append(("and", m.group(1).replace(" ", "")))
        return out

    ref = steps(open("/root/reference/yuki/src/sampling/stratified.rs").read(), "loop {", "if i < l")
    orc = steps(open(os.path.join(ROOT, "oracle", "osampler.h")).read().split("inline uint32_t permutation_element")[1], "do {", "} while")
    rng_h = open(os.path.join(ROOT, "yuki_amd", "csrc", "yk_rng.h")).read()
    dev = steps(rng_h[rng_h.index("0xe170893d") - 200:], "do {", "} while")  # the loop that holds the first multiplier
    assert len(ref) == 18 and [s for s in ref if s[0] == "mul"][0] == ("mul", "0xe170893d")
    assert orc == ref
    assert dev == ref


def test_every_numeric_literal_of_the_paths_files_is_in_the_oracle_and_in_the_product():
    """A net under "constants verbatim": every non-trivial floating-point literal of the reference files on the path (the
    roughness polynomial, Oren-Nayar's 0.33 / 0.09 / 0.45, the 0.001 ray offset, 0.9999, the roulette threshold, the camera's
    near / far, ...) must appear, as the same binary32 value, in the oracle file that restates it and in the product sources."""
    import re

    REF = "/root/reference/yuki/src/"

    def lits(paths):
        out = set()
        for path in paths:
            text = open(path).read()
            text = re.sub(r"//[^\n]*", "", text)
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            for m in re.finditer(r"(?<![\w.])(\d[\d_]*\.[\d_]*(?:e-?\d+)?|\d[\d_]*e-?\d+)(?:_?f32|f)?(?![\w.])", text):
                out.add(float(np.float32(float(m.group(1).replace("_", "")))))
        return out

    trivial = {float(np.float32(v)) for v in (0, 1, 2, 3, 4, 0.5, 10, 100, 180, 255, 1000)}
    o = lambda *names: [os.path.join(ROOT, "oracle", n) for n in names]
    c = lambda *names: [os.path.join(ROOT, "yuki_amd", "csrc", n) for n in names]
    bsdf = (o("obsdf.h"), c("yk_bsdf.h", "yk_scene.cpp"))
    geom = (o("oshapes.h", "olights.h"), c("yk_geom.h", "yk_shade.h", "yk_kernels.hip", "yk_trace.hip"))
    table = {
        "materials/bsdfs/trowbridge_reitz.rs": bsdf, "materials/bsdfs/oren_nayar.rs": bsdf, "materials/bsdfs/fresnel.rs": bsdf,
        "materials/bsdfs/microfacet.rs": bsdf, "materials/bsdfs/specular.rs": bsdf, "materials/bsdfs/lambertian.rs": bsdf, "materials/bsdfs/mod.rs": bsdf,
        "sampling/mod.rs": bsdf,
        "integrators/path.rs": (o("orender.h"), c("yk_shade.h", "yk_kernels.hip")), "integrators/whitted.rs": (o("orender.h"), c("yk_shade.h", "yk_trace.hip")),
        "interaction.rs": geom, "visibility.rs": geom, "shapes/triangle.rs": geom, "shapes/sphere.rs": geom,
        "lights/point_light.rs": geom, "lights/spot_light.rs": geom, "lights/distant_light.rs": geom, "lights/rectangular_light.rs": geom,
        "camera.rs": (o("orender.h"), c("yk_host.cpp")), "bvh.rs": (o("obvh.h"), c("yk_host.cpp")),
        "math/bounds.rs": (o("omath.h"), c("yk_math.h", "yk_host.cpp")),
    }
    checked = 0
    for ref_file, (oracle_files, product_files) in table.items():
        want = lits([REF + ref_file]) - trivial
        checked += len(want)
        assert want <= lits(oracle_files), (ref_file, "oracle", sorted(want - lits(oracle_files)))
        assert want <= lits(product_files), (ref_file, "product", sorted(want - lits(product_files)))
    assert checked >= 15  # 17 today: the files on the path hold few literals that are not 0, 1, 2 or 0.5
