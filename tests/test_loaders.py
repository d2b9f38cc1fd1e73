"""Scene input: yk_load_ply / yk_load_pbrt against the independent
restatement in oracle/loaders.py, bit for bit, on synthetic files.

Parity unpinned: the reference holds no tests or sample files for its loaders, so both
sides follow its source text (scene/ply.rs, scene/pbrt/*.rs); what these tests pin is
that the two independent implementations agree exactly, plus the behaviours spelled
out in the reference (quirks included)."""
import os

import numpy as np
import pytest

from yuki_amd import abi, loaders
from yuki_amd._ffi import YukiError

import scene_files as sf


def _eq(a, b):
    if a is None or b is None:
        return a is b
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def _f3(v):
    return np.array(v, dtype=np.float32).tobytes()


def assert_same_scene(want, got):
    for k in ("points", "normals", "uvs", "indices", "tri_mesh", "tri_material", "tri_area_light"):
        assert _eq(getattr(want, k), getattr(got, k)), k
    assert want.meshes == got.meshes
    assert len(want.materials) == len(got.materials)
    for a, b in zip(want.materials, got.materials):
        assert a["kind"] == b["kind"] and _f3(a["a"]) == _f3(b["a"]) and _f3(a["b"]) == _f3(b["b"]), (a, b)
        assert np.float32(a["c"]).tobytes() == np.float32(b["c"]).tobytes() and bool(a["remap"]) == bool(b["remap"]), (a, b)
        assert a.get("tex") == b.get("tex"), (a, b)
    assert len(want.textures) == len(got.textures)
    for a, b in zip(want.textures, got.textures):
        assert _eq(np.asarray(a, np.float32), np.asarray(b, np.float32))
    assert len(want.spheres) == len(got.spheres)
    for a, b in zip(want.spheres, got.spheres):
        assert _eq(np.asarray(a["o2w"], np.float32), b["o2w"]) and _eq(np.asarray(a["w2o"], np.float32), b["w2o"])
        assert np.float32(a["radius"]) == np.float32(b["radius"]) and a["material"] == b["material"]
    assert [bytes(x) for x in want.light_structs] == [bytes(x) for x in got.light_structs]
    assert _f3(want.background) == _f3(got.background)
    wo = want.shape_order if want.shape_order is not None else np.arange(want.n_triangles + len(want.spheres), dtype=np.uint32)
    go = got.shape_order if got.shape_order is not None else np.arange(got.n_triangles + len(got.spheres), dtype=np.uint32)
    assert _eq(wo, go)
    assert (want.split_method, want.max_shapes_in_node) == (got.split_method, got.max_shapes_in_node)


def assert_same_camera(want, got_cam, got_film, want_res):
    assert _f3(want["position"]) == _f3(got_cam.position) and _f3(want["target"]) == _f3(got_cam.target) and _f3(want["up"]) == _f3(got_cam.up)
    assert want["fov_axis"] == got_cam.fov_axis and np.float32(want["fov_degrees"]) == np.float32(got_cam.fov_degrees)
    assert tuple(want_res) == tuple(got_film.res) and got_film.tile_dim == 16


@pytest.fixture(scope="module")
def ol(oracle):
    from oracle import loaders as o

    return o


# ----------------------------------------------------------------------------- PLY
def test_ply_ascii_cube(tmp_path, ol):
    p = str(tmp_path / "cube.ply")
    sf.write_ascii_ply(p)
    want, wcam, wres = ol.load_ply(p)
    got, cam, film = loaders.load_ply(p)
    assert_same_scene(want, got)
    assert_same_camera(wcam, cam, film, wres)
    # Scene::ply (scene/mod.rs:99-152): fan-triangulated quads, fitted into the unit cube
    assert got.n_triangles == 12 and got.points.shape == (8, 3)
    assert np.array_equal(got.indices[0], [0, 3, 2]) and np.array_equal(got.indices[1], [0, 2, 1])
    ext = got.points.max(axis=0) - got.points.min(axis=0)
    assert np.float32(ext.max()) == np.float32(1.0) and np.allclose(got.points.min(axis=0) + ext / 2, 0, atol=1e-6)
    assert got.materials == [dict(kind=abi.MAT_MATTE, a=(1.0, 1.0, 1.0), b=(0.0, 0.0, 0.0), c=0.0, remap=False)]
    assert (cam.position, cam.target, cam.fov_axis, cam.fov_degrees) == ((2.0, 2.0, 2.0), (0.0, 0.0, 0.0), abi.FOV_X, 40.0)
    assert film.res == (640, 480)
    l = got.light_structs[0]
    assert l.kind == abi.LIGHT_POINT and tuple(l.p) == (5.0, 5.0, 0.0) and tuple(l.i) == (600.0, 600.0, 600.0)


@pytest.mark.parametrize("endian", ["<", ">"])
@pytest.mark.parametrize("normals,uvs,extra", [(True, True, True), (False, False, False), (True, False, True), (False, True, False)])
def test_ply_binary(tmp_path, ol, endian, normals, uvs, extra):
    p = str(tmp_path / "b.ply")
    sf.write_binary_ply(p, endian, normals=normals, uvs=uvs, extra=extra)
    want, _, _ = ol.load_ply(p, abi.SPLIT_MIDDLE, 4)
    got, _, _ = loaders.load_ply(loaders.SceneLoadSettings(path=p, split_method=abi.SPLIT_MIDDLE, max_shapes_in_node=4))
    assert_same_scene(want, got)
    assert got.meshes == [(normals, uvs, False)]
    assert (got.normals is not None) == normals and (got.uvs is not None) == uvs
    assert (got.split_method, got.max_shapes_in_node) == (abi.SPLIT_MIDDLE, 4)


def test_ply_ascii_decimal_parsing_and_ngons(tmp_path, ol):
    """ASCII floats are parsed straight to binary32 (str::parse::<f32>), and n-gons fan out."""
    rng = np.random.default_rng(5)
    v, f = sf.uv_sphere(9, 5, 1.0)
    v = [tuple(float(c) * 0.731 + float(rng.normal()) * 1e-3 for c in p) for p in v]
    f = f + [tuple(range(0, 7)), (1, 5, 9, 13, 17)]
    p = str(tmp_path / "s.ply")
    sf.write_ascii_ply(p, v, f, index_name="vertex_index", index_type="uint", fmt="{:.17g}")
    want, _, _ = ol.load_ply(p)
    got, _, _ = loaders.load_ply(p)
    assert_same_scene(want, got)
    assert got.n_triangles == sum(len(q) - 2 for q in f)


def test_ply_only_float32_properties_are_read(tmp_path, ol):
    """Property::Float only (ply.rs:237-256): double coordinates are ignored -> all zero points;
    the reference then divides by a zero extent; both sides must agree on the outcome."""
    p = str(tmp_path / "d.ply")
    with open(p, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 3\nproperty double x\nproperty float y\nproperty float z\nelement face 1\nproperty list uchar int vertex_indices\nend_header\n")
        f.write("5 0 0\n6 1 0\n7 0 1\n3 0 1 2\n")
    want, _, _ = ol.load_ply(p)
    got, _, _ = loaders.load_ply(p)
    assert_same_scene(want, got)
    assert np.all(got.points[:, 0] == got.points[0, 0])  # x never read


@pytest.mark.parametrize(
    "body,msg",
    [
        ("ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nproperty float y\nend_header\n0 0\n", "Unsupported content"),  # no z / no face
        ("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\nelement face 1\nproperty list uchar short vertex_indices\nend_header\n0 0 0\n1 0 0\n0 1 0\n3 0 1 2\n", "indices"),  # ListShort is not consumed
        ("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\nelement face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n0 1 0\n3 0 -1 2\n", "Negative PLY index"),
        ("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\nelement face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n", "truncated"),
        ("plx\n", "magic"),
    ],
)
def test_ply_errors(tmp_path, ol, body, msg):
    p = str(tmp_path / "bad.ply")
    with open(p, "w") as f:
        f.write(body)
    with pytest.raises(YukiError) as e:
        loaders.load_ply(p)
    assert msg in str(e.value)
    with pytest.raises(ol.LoadError):
        ol.load_ply(p)


def test_ply_missing_file(tmp_path):
    with pytest.raises(YukiError) as e:
        loaders.load_ply(str(tmp_path / "nope.ply"))
    assert "Could not open" in str(e.value)


# ----------------------------------------------------------------------------- pbrt
def test_pbrt_scene(tmp_path, ol):
    p = sf.write_scene(str(tmp_path))
    want, wcam, wres = ol.load_pbrt(p, abi.SPLIT_SAH, 2)
    got, cam, film = loaders.load_pbrt(loaders.SceneLoadSettings(path=p, max_shapes_in_node=2))
    assert_same_scene(want, got)
    assert_same_camera(wcam, cam, film, wres)
    # spot checks of behaviours spelled out in the reference
    assert film.res == (96, 64) and cam.fov_axis == abi.FOV_Y and cam.fov_degrees == 39.5  # res.y < res.x keeps FoV::Y
    assert _f3(got.background) == _f3((0.1, 0.2, 0.3))
    assert [l.kind for l in got.light_structs] == [abi.LIGHT_DISTANT, abi.LIGHT_POINT]  # black point light and "spot" are dropped
    assert tuple(got.light_structs[1].p) == (0.25, 0.5, 0.125)  # `from` only; the CTM is not applied (pbrt/mod.rs:571-577)
    kinds = [m["kind"] for m in got.materials]
    assert kinds == [abi.MAT_MATTE, abi.MAT_METAL, abi.MAT_METAL, abi.MAT_METAL, abi.MAT_GLASS, abi.MAT_MATTE, abi.MAT_GLOSSY, abi.MAT_MATTE]
    sigma = np.float32(20) * np.float32(np.float32(np.pi) / np.float32(180))
    assert np.float32(got.materials[5]["c"]) == sigma * np.float32(np.float32(np.pi) / np.float32(180))  # to_radians twice
    assert got.materials[1]["remap"] and not got.materials[2]["remap"]
    # shapes keep file order: floor (2 tris), sphere, sphere, the two included cubes, then the single triangles
    nt = got.n_triangles
    assert nt == 2 + 12 + 12 + 1 + 1 and len(got.spheres) == 2
    assert list(got.shape_order[:4]) == [0, 1, nt, nt + 1]
    # the included cube meshes carry the glossy material, then "plastic" -> default matte 0.5
    assert set(got.tri_material[2:14]) == {6} and set(got.tri_material[14:26]) == {7}
    assert got.tri_material[26] == 2 and got.tri_material[27] == 3  # NamedMaterial cu2 / cu3; "nope" -> default
    assert got.meshes[0] == (True, True, False)
    assert got.spheres[0]["material"] == 1 and got.spheres[1]["material"] == 4


def test_pbrt_bvh_of_loaded_scene_matches_oracle(tmp_path, ol, oracle, yk):
    """BoundingVolumeHierarchy::new over Scene.shapes in file order: the host builder behind
    yk_scene_create (ctx=None: no GPU) and the oracle's produce the same nodes and order."""
    p = sf.write_scene(str(tmp_path))
    for split in (abi.SPLIT_SAH, abi.SPLIT_MIDDLE, abi.SPLIT_EQUAL_COUNTS):
        got, _, _ = loaders.load_pbrt(loaders.SceneLoadSettings(path=p, split_method=split, max_shapes_in_node=1))
        want, _, _ = ol.load_pbrt(p, split, 1)
        gn, go = yk.Scene(None, got).export_bvh()
        wn, wo = oracle.OracleScene(want).export_bvh()
        assert gn.tobytes() == wn.tobytes() and np.array_equal(go, wo)
        natural = yk.Scene(None, _without_order(got)).export_bvh()
        assert sorted(go) == sorted(natural[1])


def _without_order(sd):
    import copy

    c = copy.copy(sd)
    c.shape_order = None
    return c


def test_pbrt_camera_axis_switch(tmp_path, ol):
    """res.y >= res.x turns the parsed FoV::Y into FoV::X with the same angle (pbrt/mod.rs:827-835)."""
    p = str(tmp_path / "t.pbrt")
    with open(p, "w") as f:
        f.write('Camera "perspective"\nFilm "image" "integer xresolution" [50] "integer yresolution" [50]\nShape "sphere"\nWorldEnd\n')
    got, cam, film = loaders.load_pbrt(p)
    want, wcam, wres = ol.load_pbrt(p)
    assert_same_scene(want, got)
    assert_same_camera(wcam, cam, film, wres)
    assert cam.fov_axis == abi.FOV_X and cam.fov_degrees == 45.0 and film.res == (50, 50)
    assert cam.position == (0.0, 0.0, 0.0) and cam.up == (0.0, 1.0, 0.0)  # CameraParameters::default


def test_pbrt_directive_at_end_of_file_is_dropped(tmp_path, ol):
    """get_next_token! breaks out of the parse loop on EndOfInput, so a directive whose
    parameter list runs into the end of the file never takes effect (pbrt/mod.rs:140-160)."""
    p = str(tmp_path / "t.pbrt")
    with open(p, "w") as f:
        f.write('Shape "sphere" "float radius" 2\nShape "sphere" "float radius" 3')
    got, _, _ = loaders.load_pbrt(p)
    want, _, _ = ol.load_pbrt(p)
    assert_same_scene(want, got)
    assert [s["radius"] for s in got.spheres] == [2.0]


def test_pbrt_lexer_quirks(tmp_path, ol):
    """`]` ends a number, `#` starts a comment anywhere, escapes are kept verbatim, first
    single-valued parameter of a name wins (param_set.rs:109-116)."""
    p = str(tmp_path / "t.pbrt")
    with open(p, "w") as f:
        f.write('Material "matte" "rgb Kd" [.1 .2 .3]#c\n"float sigma" [1 2] "float sigma" 3 "float sigma" 4\n')
        f.write('Shape "sphere" "float radius" [1.5e0]# comment\n Scale 2 2 2 Shape "sphere" "float radius" -.5\nWorldEnd\n')
    got, _, _ = loaders.load_pbrt(p)
    want, _, _ = ol.load_pbrt(p)
    assert_same_scene(want, got)
    rpd = np.float32(np.float32(np.pi) / np.float32(180))
    assert np.float32(got.materials[1]["c"]) == np.float32(3) * rpd * rpd
    assert [s["radius"] for s in got.spheres] == [1.5, -0.5]


@pytest.mark.parametrize(
    "text,msg",
    [
        ('Transform [1 0 0 0 0 1 0 0 0 0 1 0 0 0 0 1]\nShape "sphere"\n', "UnimplementedToken"),
        ('Camera "orthographic"\n', "Only perspective"),
        ('Frobnicate 1\nShape "sphere"\n', "UnknownIdentifier"),
        ('Shape "sphere" "float radius" [1 x]\n', "UnknownIdentifier"),
        ('Shape "sphere" "floaty radius" 1\nWorldEnd\n', "UnknownParamType"),
        ('Shape "sphere" "float" 1\nWorldEnd\n', "UnexpectedToken"),
        ('Translate 1 2 "x"\n', "UnexpectedToken"),
        ('Scale 1 2 1.2.3 \n', "InvalidNumber"),
        ('Scale 1 2 0x10 \n', "InvalidNumber"),
        ('Shape "sphere\n" \n', "UnterminatedString"),
        ('Material "matte" "texture Kd" "wood"\nShape "sphere"\nWorldEnd\n', "not found"),
        ('MakeNamedMaterial "a" "float type" "matte"\n', "UnknownParamType"),
        ('Shape "trianglemesh" "integer indices" [0 1 5] "point P" [0 0 0 1 0 0 0 1 0]\nWorldEnd\n', "out of range"),
        ('Shape "plymesh"\nWorldEnd\n', "Empty PLY filename"),
        ('Include "missing.pbrt"\n', "Could not open"),
        ("WorldBegin\nWorldEnd\n", "no shapes"),
    ],
)
def test_pbrt_errors(tmp_path, ol, text, msg):
    p = str(tmp_path / "bad.pbrt")
    with open(p, "w") as f:
        f.write(text)
    with pytest.raises(YukiError) as e:
        loaders.load_pbrt(p)
    assert msg in str(e.value), str(e.value)
    with pytest.raises((ol.LoadError, OSError)):
        ol.load_pbrt(p)


def test_render_scene_file_loads_identically(tmp_path, ol):
    p = sf.write_render_scene(str(tmp_path))
    got, cam, film = loaders.load_pbrt(p)
    want, wcam, wres = ol.load_pbrt(p)
    assert_same_scene(want, got)
    assert_same_camera(wcam, cam, film, wres)
    assert got.n_triangles == 2 + 2 * 2 * 24 * 12 and len(got.spheres) == 2 and film.res == (80, 60)


# ----------------------------------------------------------------------------- BASELINE-sized scenes through the loaders
# BASELINE.json configs[1] is "Stanford-bunny PLY (~70 k tri)", configs[2..4] "~1 M-tri pbrt-v3 scene": the same meshes the
# generators build (yuki_amd/scenes.py) written as the files the reference's loaders read (tests/scene_files.py) must come
# back as the generators' arrays, bit for bit — and equal the independent loader in oracle/loaders.py.
@pytest.fixture(scope="session")
def cfg2_ply(tmp_path_factory):
    return sf.write_cfg2_ply(str(tmp_path_factory.mktemp("cfg2") / "bunny_class.ply"))


@pytest.fixture(scope="session")
def cfg3_pbrt(tmp_path_factory, cfg3_scene):
    p, info = sf.write_scene_as_pbrt(str(tmp_path_factory.mktemp("cfg3")), cfg3_scene)
    assert info["ply_files"] == 802  # 800 displaced icospheres + the open box + the light's quad
    return p


@pytest.fixture(scope="session")
def cfg3_oracle_loaded(cfg3_pbrt, oracle):
    """oracle/loaders.py on the 1,024,012-triangle file set: per-vertex Python through the oracle's KAT-pinned transforms (~25 s)."""
    from oracle import loaders as o

    return o.load_pbrt(cfg3_pbrt)


def test_cfg2_ply_file_loads_to_the_generators_arrays(cfg2_ply, ol):
    from yuki_amd import scenes

    sd = scenes.by_name("cfg2")
    got, cam, film = loaders.load_ply(cfg2_ply)
    assert got.n_triangles == 69312
    # Scene::ply: fit to the unit cube (scene/ply.rs:99-108), white matte, one point light, camera at (2, 2, 2), 640 x 480
    assert _eq(got.points, sd.points) and _eq(got.indices, sd.indices)
    assert _eq(got.tri_material, sd.tri_material) and _eq(got.tri_area_light, sd.tri_area_light)
    assert got.materials[0]["kind"] == abi.MAT_MATTE and _f3(got.materials[0]["a"]) == _f3((1, 1, 1))
    assert tuple(cam.position) == (2.0, 2.0, 2.0) and film.res == (640, 480)
    want, wcam, wres = ol.load_ply(cfg2_ply)
    assert_same_scene(want, got)
    assert_same_camera(wcam, cam, film, wres)


def test_cfg3_pbrt_file_loads_to_the_generators_arrays(cfg3_pbrt, cfg3_scene, cfg3_oracle_loaded):
    import time

    sd = cfg3_scene
    t0 = time.time()
    got, cam, film = loaders.load_pbrt(cfg3_pbrt)
    dt = time.time() - t0
    print(f"yk_load_pbrt: 1,024,012 triangles from 802 PLY files + the pbrt file in {dt:.2f} s")
    assert got.n_triangles == 1024012 and film.res == (1920, 1080)
    assert _eq(got.points, sd.points) and _eq(got.indices, sd.indices) and _eq(got.tri_mesh, sd.tri_mesh)
    has_n = np.array([m[0] for m in sd.meshes])[sd.tri_mesh]  # per triangle: does its mesh carry shading normals
    v_n = np.zeros(len(sd.points), dtype=bool)
    v_n[sd.indices[has_n].ravel()] = True
    assert _eq(got.normals[v_n], sd.normals[v_n])
    has_uv = np.array([m[1] for m in sd.meshes])[sd.tri_mesh]
    v_uv = np.zeros(len(sd.points), dtype=bool)
    v_uv[sd.indices[has_uv].ravel()] = True
    assert v_uv.sum() == 8 and _eq(got.uvs[v_uv], sd.uvs[v_uv])
    assert [(m[0], m[1]) for m in got.meshes] == [(m[0], m[1]) for m in sd.meshes]
    # materials: one per mesh, constants as written (text round trip of nine digits is exact for float32); Oren-Nayar's sigma
    # goes through the loader's two to_radians (pbrt/mod.rs:906-910) and comes back within an ulp or two of the generator's
    assert len(got.materials) == len(sd.meshes) + 1  # the graphics state's default matte first (pbrt/mod.rs:119-131), then one per Material directive
    assert len(got.light_structs) == 2 and _f3(got.background) == _f3(sd.background)  # two point lights; "infinite" is the background
    for k in (0, 1, 7, 123, 799, 800):
        a, b = got.materials[int(got.tri_material[np.searchsorted(sd.tri_mesh, k)])], sd.materials[int(sd.tri_material[np.searchsorted(sd.tri_mesh, k)])]
        assert a["kind"] == b["kind"] and _f3(a["a"]) == _f3(b["a"])
        assert abs(np.float32(a["c"]) - np.float32(b["c"])) <= 4 * np.spacing(np.float32(b["c"])) + 0
    # camera as written; the film's aspect decides the FoV axis (pbrt/mod.rs:807-816)
    assert _f3(cam.position) == _f3(sd.camera["position"]) and _f3(cam.target) == _f3(sd.camera["target"])
    assert np.float32(cam.fov_degrees) == np.float32(sd.camera["fov_degrees"])
    # the independent loader reads the same files to the same scene
    want, wcam, wres = cfg3_oracle_loaded
    assert_same_scene(want, got)
    assert_same_camera(wcam, cam, film, wres)
    assert dt < 5.0
    # the PLY files are read in parallel after the parse (pbrt/mod.rs:786-800); one thread gives the same scene
    os.environ["YK_LOADER_THREADS"] = "1"
    try:
        serial, _, _ = loaders.load_pbrt(cfg3_pbrt)
    finally:
        del os.environ["YK_LOADER_THREADS"]
    assert_same_scene(serial, got)


def test_pbrt_parse_error_after_a_broken_ply_wins(tmp_path, ol):
    """The reference reads `plymesh` files AFTER the parse (pbrt/mod.rs:786-800), so an unknown directive further down the file is
    what the load reports; a missing file ends the load where it is named (canonicalize(), :689-699)."""
    (tmp_path / "bad.ply").write_bytes(b"ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nend_header\n0\n")
    p = tmp_path / "s.pbrt"
    p.write_text('WorldBegin\nShape "plymesh" "string filename" "bad.ply"\nObjectBegin "x"\nWorldEnd\n')
    with pytest.raises(YukiError) as e:
        loaders.load_pbrt(str(p))
    assert "UnimplementedToken" in str(e.value) or "ObjectBegin" in str(e.value)
    with pytest.raises(ol.LoadError) as oe:  # the independent loader orders its errors the same way
        ol.load_pbrt(str(p))
    assert "PLY" not in str(oe.value)
    p.write_text('WorldBegin\nShape "plymesh" "string filename" "bad.ply"\nWorldEnd\n')
    with pytest.raises(YukiError) as e:
        loaders.load_pbrt(str(p))
    assert "PLY" in str(e.value)
    p.write_text('WorldBegin\nShape "plymesh" "string filename" "missing.ply"\nObjectBegin "x"\nWorldEnd\n')
    with pytest.raises(YukiError) as e:
        loaders.load_pbrt(str(p))
    assert "Could not open" in str(e.value)


# ----------------------------------------------------------------------------- GPU: load -> render
@pytest.mark.gpu
def test_loaded_pbrt_scene_renders_like_the_oracle(tmp_path, ol, oracle, yk, ctx):
    """File -> yk_load_pbrt -> yk_scene_create -> Path render on the device, against the
    oracle's loader + renderer on the same file (bit-identical; stated bar RMSE < 1e-4)."""
    p = sf.write_render_scene(str(tmp_path))
    got_sd, cam_p, film = loaders.load_pbrt(p)
    want_sd, _, _ = ol.load_pbrt(p)
    fs = yk.FilmSettings(res=film.res, tile_dim=film.tile_dim)
    cam = yk.Camera(cam_p, fs)
    tiles = yk.film_tiles(fs)
    sampler = yk.SamplerType.Stratified((2, 2), True, 0x73B9642E74AC471C)
    integ = yk.IntegratorType.Path(yk.PathParams(max_depth=6))
    it = yk.IntegratorType.instantiate(ctx, integ)
    got, stats = it.render_tiles(yk.Scene(ctx, got_sd), cam, sampler, tiles)
    want, rays = oracle.OracleScene(want_sd).render_tiles(cam.matrices, sampler, integ, tiles, n_threads=0)
    assert stats.rays == rays
    assert float(np.sqrt(np.mean((got.astype(np.float64) - want) ** 2))) < 1e-4
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert got.mean() > 0.01


@pytest.mark.gpu
def test_loaded_ply_scene_renders_like_the_oracle(tmp_path, ol, oracle, yk, ctx):
    v, f = sf.uv_sphere(32, 16, 1.0)
    p = str(tmp_path / "ball.ply")
    sf.write_ascii_ply(p, v, f)
    got_sd, cam_p, film = loaders.load_ply(p)
    want_sd, _, _ = ol.load_ply(p)
    fs = yk.FilmSettings(res=(160, 120), tile_dim=film.tile_dim)
    cam = yk.Camera(cam_p, fs)
    tiles = yk.film_tiles(fs)
    sampler = yk.SamplerType.Uniform(4, 0x73B9642E74AC471C)
    integ = yk.IntegratorType.Path(yk.PathParams(max_depth=3))
    it = yk.IntegratorType.instantiate(ctx, integ)
    got, stats = it.render_tiles(yk.Scene(ctx, got_sd), cam, sampler, tiles)
    want, rays = oracle.OracleScene(want_sd).render_tiles(cam.matrices, sampler, integ, tiles, n_threads=0)
    assert stats.rays == rays and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert got.max() > 0.05


@pytest.mark.gpu
def test_cfg2_ply_file_renders_like_the_oracle(cfg2_ply, ol, oracle, yk, ctx):
    """BASELINE configs[1] end to end from its file: yk_load_ply -> yk_scene_create -> Path 8, Uniform 16 spp, 1920 x 1080:
    the first 96 spiral tiles equal the oracle's render of the scene oracle/loaders.py reads from the same file."""
    got_sd, cam_p, film = loaders.load_ply(cfg2_ply)
    want_sd, _, _ = ol.load_ply(cfg2_ply)
    fs = yk.FilmSettings(res=(1920, 1080), tile_dim=film.tile_dim)
    cam = yk.Camera(cam_p, fs)
    tiles = yk.film_tiles(fs)[:96]
    sampler = yk.SamplerType.Uniform(16, 0x73B9642E74AC471C)
    integ = yk.IntegratorType.Path(yk.PathParams(max_depth=8))
    sc = yk.Scene(ctx, got_sd)
    got, stats = yk.IntegratorType.instantiate(ctx, integ).render_tiles(sc, cam, sampler, tiles)
    want, rays = oracle.OracleScene(want_sd).render_tiles(cam.matrices, sampler, integ, tiles, n_threads=0)
    sc.close()
    assert stats.rays == rays and np.array_equal(got.view(np.uint32), want.view(np.uint32)) and got.max() > 0.05


@pytest.mark.gpu
def test_cfg3_pbrt_file_renders_like_the_oracle(cfg3_pbrt, cfg3_oracle_loaded, oracle, yk, ctx):
    """BASELINE configs[2]'s shape end to end from files: yk_load_pbrt (802 PLY meshes) -> SAH BVH -> Path 8, Stratified 8 x 8,
    1920 x 1080: the first 48 spiral tiles against the oracle's render of what oracle/loaders.py read, bit for bit."""
    got_sd, cam_p, film = loaders.load_pbrt(cfg3_pbrt)
    want_sd = cfg3_oracle_loaded[0]
    fs = yk.FilmSettings(res=film.res, tile_dim=film.tile_dim)
    cam = yk.Camera(cam_p, fs)
    tiles = yk.film_tiles(fs)[:48]
    sampler = yk.SamplerType.Stratified((8, 8), True, 0x73B9642E74AC471C)
    integ = yk.IntegratorType.Path(yk.PathParams(max_depth=8))
    sc = yk.Scene(ctx, got_sd)
    got, stats = yk.IntegratorType.instantiate(ctx, integ).render_tiles(sc, cam, sampler, tiles)
    osc = oracle.OracleScene(want_sd)
    want, rays = osc.render_tiles(cam.matrices, sampler, integ, tiles, n_threads=0)
    osc.close()
    sc.close()
    assert stats.rays == rays and np.array_equal(got.view(np.uint32), want.view(np.uint32)) and got.mean() > 0.01
