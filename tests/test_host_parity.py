"""Host-side logic of the product (no GPU): BVH builder, Camera::new, film tiles,
light construction, Film::update_tile — against the oracle, bit for bit."""
import ctypes as C

import numpy as np
import pytest

from yuki_amd import abi, scenes


@pytest.mark.parametrize("name", ["cornell-tris", "city-tiny", "city-small", "cfg2"])
@pytest.mark.parametrize("method", [abi.SPLIT_SAH, abi.SPLIT_MIDDLE, abi.SPLIT_EQUAL_COUNTS])
@pytest.mark.parametrize("max_shapes", [1, 4])
def test_bvh_identical_to_oracle(yk, oracle, name, method, max_shapes):
    """BoundingVolumeHierarchy::new (bvh.rs:39-115): same 32-byte nodes in the same
    depth-first order, same leaf-order shape permutation."""
    sd = scenes.by_name(name)
    sd.split_method, sd.max_shapes_in_node = method, max_shapes
    hs = yk.Scene(None, sd)
    os_ = oracle.OracleScene(sd)
    n1, o1 = hs.export_bvh()
    n2, o2 = os_.export_bvh()
    assert n1.tobytes() == n2.tobytes()
    assert np.array_equal(o1, o2)
    assert sorted(o1.tolist()) == list(range(sd.n_triangles))  # a permutation of the shapes
    # structural invariants of flatten_tree (bvh.rs:396-420)
    leaves = n1[n1["is_leaf"] == 1]
    assert int(leaves["count"].sum()) == sd.n_triangles
    inter = np.nonzero(n1["is_leaf"] == 0)[0]
    assert (n1["a"][inter] > inter + 1).all()  # second child after the first child's subtree
    info = hs.info()
    assert info.n_nodes == len(n1) and info.n_interior == len(inter)


def test_bvh_with_spheres_host_only(yk, oracle):
    """The full built-in Cornell box (triangles + the copper sphere) builds the
    same tree on the host even though the sphere has no device kernel."""
    sd = scenes.cornell()
    n1, o1 = yk.Scene(None, sd).export_bvh()
    n2, o2 = oracle.OracleScene(sd).export_bvh()
    assert n1.tobytes() == n2.tobytes() and np.array_equal(o1, o2)


def test_degenerate_scenes(yk, oracle):
    """Single triangle (root is a leaf), duplicated triangles (zero centroid
    extent -> multi-shape leaf, bvh.rs:334-345)."""
    base = scenes.by_name("city-tiny")
    one = scenes.SceneData(points=base.points[:3].copy(), indices=np.array([[0, 1, 2]], dtype=np.uint32), tri_mesh=np.zeros(1, np.uint32),
                           tri_material=np.zeros(1, np.int32), tri_area_light=np.full(1, -1, np.int32), meshes=[(False, False, False)],
                           materials=base.materials[:1], lights=base.lights, camera=base.camera)
    n1, _ = yk.Scene(None, one).export_bvh()
    assert len(n1) == 1 and n1[0]["is_leaf"] == 1 and n1[0]["count"] == 1
    dup = scenes.SceneData(points=base.points[:3].copy(), indices=np.array([[0, 1, 2]] * 7, dtype=np.uint32), tri_mesh=np.zeros(7, np.uint32),
                           tri_material=np.zeros(7, np.int32), tri_area_light=np.full(7, -1, np.int32), meshes=[(False, False, False)],
                           materials=base.materials[:1], lights=base.lights, camera=base.camera)
    n2, o2 = yk.Scene(None, dup).export_bvh()
    n3, o3 = oracle.OracleScene(dup).export_bvh()
    assert len(n2) == 1 and n2[0]["count"] == 7 and n2.tobytes() == n3.tobytes() and np.array_equal(o2, o3)


def test_scene_validation_errors(yk):
    base = scenes.by_name("city-tiny")
    bad = scenes.by_name("city-tiny")
    bad.indices = bad.indices.copy()
    bad.indices[0, 0] = 10**8
    with pytest.raises(yk.YukiError) as e:
        yk.Scene(None, bad)
    assert e.value.status == 1
    bad = scenes.by_name("city-tiny")
    bad.max_shapes_in_node = 0
    with pytest.raises(yk.YukiError):
        yk.Scene(None, bad)
    assert base.n_triangles > 0


@pytest.mark.parametrize("res", [(1920, 1080), (640, 480), (480, 640), (3840, 2160), (100, 100)])
@pytest.mark.parametrize("axis", [abi.FOV_X, abi.FOV_Y])
def test_camera_identical_to_oracle(yk, oracle, res, axis):
    for sd in (scenes.by_name("cfg2"), scenes.by_name("city-tiny"), scenes.cornell()):
        cam = dict(sd.camera)
        cam["fov_axis"] = axis
        c1 = yk.Camera(cam, yk.FilmSettings(res=res)).matrices
        c2 = oracle.make_camera(cam, res)
        assert bytes(c1) == bytes(c2)


def test_camera_maps_film_corners_to_the_fov(yk):
    """camera.rs:78-93: the screen window is +-1 along the FoV axis."""
    cam = yk.Camera(dict(position=(0, 0, 0), target=(0, 0, 1), up=(0, 1, 0), fov_axis=abi.FOV_X, fov_degrees=90.0), yk.FilmSettings(res=(200, 100)))
    m = np.array(list(cam.matrices.raster_to_camera), dtype=np.float64).reshape(4, 4)

    def unproject(x, y):
        p = m @ np.array([x, y, 0, 1.0])
        return p[:3] / p[3]

    left, right = unproject(0, 50), unproject(200, 50)
    assert abs(left[0] / left[2] + 1.0) < 1e-5 and abs(right[0] / right[2] - 1.0) < 1e-5  # tan(45 deg) = 1
    top = unproject(100, 0)
    assert top[1] > 0 and abs(top[1] / top[2] - 0.5) < 1e-5  # raster y points down; aspect 2:1


@pytest.mark.parametrize("res,td", [((1920, 1080), 16), ((640, 480), 16), ((100, 70), 16), ((33, 17), 8), ((16, 16), 16), ((17, 16), 16), ((3840, 2160), 64), ((5, 3), 16)])
def test_film_tiles_spiral(yk, oracle, res, td):
    t1 = yk.film_tiles(yk.FilmSettings(res=res, tile_dim=td))
    t2 = oracle.film_tiles(res, td)
    assert np.array_equal(t1, t2)
    # every pixel exactly once, tiles clipped to the film (film.rs:299-331)
    cover = np.zeros((res[1], res[0]), dtype=np.int32)
    for t in t1:
        assert t["x1"] <= res[0] and t["y1"] <= res[1] and t["x0"] < t["x1"] and t["y0"] < t["y1"]
        cover[t["y0"] : t["y1"], t["x0"] : t["x1"]] += 1
    assert (cover == 1).all()
    # starts at the centre tile (film.rs:343-346)
    h, v = -(-res[0] // td), -(-res[1] // td)
    cx, cy = (h // 2) - (1 - h % 2), (v // 2) - (1 - v % 2)
    assert (t1[0]["x0"], t1[0]["y0"]) == (cx * td, cy * td)


def test_lights_identical_to_oracle(yk, oracle):
    rng = np.random.default_rng(3)
    for k in range(20):
        ang = rng.uniform(0, 6.28)
        c, s = np.cos(ang), np.sin(ang)
        l2w = np.eye(4, dtype=np.float32)
        l2w[:3, :3] = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float32)
        l2w[:3, 3] = rng.uniform(-5, 5, 3)
        l2w_inv = np.linalg.inv(l2w.astype(np.float64)).astype(np.float32)
        a, b = abi.LightDesc(), abi.LightDesc()
        yk.LightFactory.make_rect_light(l2w, l2w_inv, (1, 2, 3), (0.5 + k, 1.5), a)
        oracle.LightFactory.make_rect_light(l2w, l2w_inv, (1, 2, 3), (0.5 + k, 1.5), b)
        assert bytes(a) == bytes(b)
        yk.LightFactory.make_spot_light(l2w, l2w_inv, (4, 5, 6), 30.0 + k, 20.0 + k, a)
        oracle.LightFactory.make_spot_light(l2w, l2w_inv, (4, 5, 6), 30.0 + k, 20.0 + k, b)
        assert bytes(a) == bytes(b)
        yk.LightFactory.make_point_light(l2w, (7, 8, 9), a)
        oracle.LightFactory.make_point_light(l2w, (7, 8, 9), b)
        assert bytes(a) == bytes(b)


def test_update_tiles_roundtrip(yk, oracle):
    """Film::update_tile (film.rs:236-278): tile-major -> row-major, ragged edges."""
    res = (70, 41)
    tiles = yk.film_tiles(yk.FilmSettings(res=res, tile_dim=16))
    n = sum((int(t["x1"]) - int(t["x0"])) * (int(t["y1"]) - int(t["y0"])) for t in tiles)
    assert n == res[0] * res[1]
    rgb = np.arange(n * 3, dtype=np.float32).reshape(n, 3)
    film = yk.update_tiles(tiles, rgb, res)
    assert np.array_equal(film, oracle.detile(tiles, rgb, res))
    with pytest.raises(yk.YukiError):  # "Tile doesn't fit film" (film.rs:227-234)
        yk.update_tiles(tiles, rgb, (64, 41))


def test_scene_generators_are_deterministic():
    a, b = scenes.by_name("city-small"), scenes.by_name("city-small")
    assert a.points.tobytes() == b.points.tobytes() and a.indices.tobytes() == b.indices.tobytes()
    c = scenes.by_name("cfg2")
    assert c.n_triangles == 69312
    assert abs(np.abs(c.points).max() - 0.5) < 1e-6  # fitted to the unit cube like scene/ply.rs:99-108


@pytest.mark.parametrize("method,max_shapes", [(abi.SPLIT_SAH, 1), (abi.SPLIT_SAH, 4), (abi.SPLIT_MIDDLE, 1), (abi.SPLIT_EQUAL_COUNTS, 2)])
def test_parallel_bvh_build_is_the_sequential_one(yk, oracle, method, max_shapes, monkeypatch):
    """Inputs of 2^16 shapes or more are built by worker threads (subtrees cut off the top, spliced
    back in depth-first order): the tree equals the single-threaded build and the oracle's
    sequential builder, node for node, for every thread count."""
    sd = scenes.city((12, 10), 3, 1, "mixed")  # 153,610 triangles
    sd.split_method, sd.max_shapes_in_node = method, max_shapes
    assert sd.n_triangles >= 1 << 16
    n_ref, o_ref = oracle.OracleScene(sd).export_bvh()
    for threads in ("1", "2", "5", "16"):
        monkeypatch.setenv("YK_BVH_THREADS", threads)
        hs = yk.Scene(None, sd)
        n, o = hs.export_bvh()
        assert n.tobytes() == n_ref.tobytes(), threads
        assert np.array_equal(o, o_ref), threads
        assert hs.info().n_nodes == len(n_ref)
