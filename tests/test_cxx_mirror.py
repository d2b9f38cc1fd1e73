"""include/yuki_hip.hpp (the C++ mirror of the reference's interface) compiles with a
plain host compiler against the C ABI and behaves: host parts on CPU, rendering on GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "mirror_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "mirror_test.cpp"),
                           "-o", exe, "-L", os.path.join(ROOT, "yuki_amd"), "-lyuki_hip", "-Wl,-rpath," + os.path.join(ROOT, "yuki_amd")])
    return exe


def _kv(out):
    d = {}
    for tok in out.split():
        if "=" in tok:
            k, v = tok.split("=", 1)
            d[k] = v
    return d


def test_cxx_mirror_host_side(tmp_path, yk):
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scene_files as sf

    pbrt = sf.write_scene(str(tmp_path / "scene"))
    out = subprocess.check_output([_build(tmp_path), "cpu", pbrt], text=True)
    kv = _kv(out)
    # LoadedScene (scene::pbrt::load through the C++ mirror) agrees with the Python binding
    from yuki_amd import loaders

    sd, cam_p, film = loaders.load_pbrt(pbrt)
    assert int(kv["pbrt_triangles"]) == sd.n_triangles and int(kv["pbrt_spheres"]) == len(sd.spheres) and int(kv["pbrt_lights"]) == len(sd.light_structs)
    assert kv["pbrt_res"] == "%dx%d" % film.res and float(kv["pbrt_fov"]) == cam_p.fov_degrees
    assert int(kv["pbrt_nodes"]) == yk.Scene(None, sd).info().n_nodes
    assert kv["missing_scene"] == "status1"
    assert kv["tiles"] == "6" and kv["first"] == "16,0"  # 3x2 tiles, spiral starts at the centre tile (film.rs:343-346)
    assert kv["nodes"] == "3" and kv["shapes"] == "4"  # each quad = 2 triangles with identical bounds -> one 2-shape leaf (bvh.rs:334-345)
    assert kv["bad_scene"] == "status1"
    cam = yk.Camera(dict(position=(0, 2, -3), target=(0, 0.3, 0), up=(0, 1, 0), fov_axis=0, fov_degrees=50.0), yk.FilmSettings(res=(40, 24)))
    assert abs(float(kv["camera_c2w_03"]) - cam.matrices.camera_to_world[3]) < 1e-6


@pytest.mark.gpu
def test_cxx_mirror_renders(tmp_path):
    out = subprocess.check_output([_build(tmp_path), "gpu"], text=True)
    kv = _kv(out)
    assert int(kv["samples"]) == 40 * 24 * 4 and int(kv["rays"]) >= int(kv["samples"])
    assert float(kv["mean"]) > 0.05
    assert kv["tile_matches_batch"] == "1" and int(kv["tile_rays"]) > 0
    assert kv["bad_tile"] == "status1"
    assert kv["accumulate_matches_plain"] == "1"
    # six worker threads through the Combiner: every tile is its slab of the batched film
    assert kv["combiner_matches_batch"] == "1" and kv["combiner_rays_match"] == "1" and kv["combiner_merged"] == "1"
    assert kv["node_rays_match"] == "1" and kv["node_film_matches"] == "1" and kv["node_devices"] == "1" and kv["node_dup"] == "status1"
    # three virtual ranks on one device, the accumulating film over them, the device-free deal
    assert kv["virtual_ranks_match"] == "1" and kv["virtual_ranks_accumulate"] == "1" and kv["deal_covers_film"] == "1"
