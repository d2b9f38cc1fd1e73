"""bench.py's launch contract: `--gpus N` (N > 1) without a launcher starts N ranks itself — child processes, created
before the parent has imported torch or touched HIP — relays rank 0's ONE JSON line and exits with the ranks' status.
It never falls through to a one-GPU run that prints n_gpus 1."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)


def test_more_ranks_than_devices_is_an_error_not_a_one_gpu_run():
    """Here (no GPU) and on the one-GPU box alike: two ranks cannot each have a device."""
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices visible")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--workload", "smoke"])
    assert r.returncode != 0
    assert r.stdout.strip() == "", r.stdout  # no record, in particular none that says n_gpus 1
    assert "launching 2 ranks" in r.stderr and "device(s) visible" in r.stderr


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and r.stdout.strip() == "" and "WORLD_SIZE=3" in r.stderr


@pytest.mark.gpu
def test_self_launched_ranks_render_the_same_film():
    """Two self-launched ranks rehearsed on ONE GPU (both use cuda:0, the slabs travel through gloo: RCCL refuses two
    ranks on a device): the N > 1 control flow — shard, render, gather, scatter, max over ranks — end to end; the film
    equals the single-rank one, the record says n_gpus 2 and stdout holds nothing but the record."""
    one = _run(["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--workload", "smoke"])
    assert one.returncode == 0, one.stderr[-2000:]
    two = _run(["--gpus", "2", "--rehearse-on-one-gpu", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--workload", "smoke"])
    assert two.returncode == 0, two.stderr[-2000:]
    assert len(one.stdout.strip().splitlines()) == 1 and len(two.stdout.strip().splitlines()) == 1
    a, b = json.loads(one.stdout), json.loads(two.stdout)
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2
    assert b["extra"]["film_mean_rgb"] == a["extra"]["film_mean_rgb"]
    assert b["extra"]["rays_per_step"] == a["extra"]["rays_per_step"]
    assert a["roofline"] is None or a["roofline"]["frac"] is None or a["roofline"]["frac"] <= 1.0


@pytest.mark.gpu
def test_scene_file_workload(tmp_path):
    """`bench.py --scene-file`: BASELINE configs[1] from the PLY file its text names (tests/scene_files.py writes the generated
    mesh as Scene::ply reads it) — the same triangles, ray count and film as the generated workload, the file's load time
    reported, and a roofline without the stored PMC entries (they belong to the generated scene's launches)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scene_files as sf

    p = sf.write_cfg2_ply(str(tmp_path / "bunny_class.ply"))
    gen = _run(["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--workload", "cfg2"])
    assert gen.returncode == 0, gen.stderr[-2000:]
    got = _run(["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--workload", "cfg2", "--scene-file", p])
    assert got.returncode == 0, got.stderr[-2000:]
    a, b = json.loads(gen.stdout), json.loads(got.stdout)
    assert b["config"]["triangles"] == a["config"]["triangles"] == 69312
    assert "scene file bunny_class.ply" in b["config"]["workload"] and b["extra"]["scene_load_s"] is not None
    assert b["extra"]["rays_per_step"] == a["extra"]["rays_per_step"] and b["extra"]["film_mean_rgb"] == a["extra"]["film_mean_rgb"]
    assert b["roofline"] is None or b["roofline"]["traffic"] is None
    if a["roofline"] is not None:  # the generated workload quotes a tracked PMC set and names its content
        assert a["roofline"]["traffic_source"].startswith("profiles/") and len(a["roofline"]["traffic_source_git_blob"]) == 40
        assert a["roofline"]["binding"]["unit"].startswith("vector-memory")
