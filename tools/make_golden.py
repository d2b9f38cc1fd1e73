#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz from the CPU oracle.

The Rust reference cannot be built or run here (no toolchain),
so these vectors are produced by the oracle — the quirk-faithful restatement of
the reference — not by the reference itself.  They freeze the oracle's behaviour
(any later edit of oracle/ that changes a bit is caught on CPU) and give the GPU
tests fixtures that do not need the oracle at run time.
Run:  python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as oracle  # noqa: E402
from yuki_amd import abi, scenes  # noqa: E402

SEED = 0x73B9642E74AC471C
OUT = os.path.join(ROOT, "tests", "golden")

CASES = {
    # name: (scene, res, sampler, integrator)
    "cornell_whitted": ("cornell", (48, 48), abi.SamplerDesc(abi.SAMPLER_UNIFORM, 1, 1, 1, SEED), abi.IntegratorDesc(abi.INTEGRATOR_WHITTED, 3, 0, 0.0)),
    "cornell_path": ("cornell", (32, 32), abi.SamplerDesc(abi.SAMPLER_UNIFORM, 4, 1, 1, SEED), abi.IntegratorDesc(abi.INTEGRATOR_PATH, 8, 0, 0.0)),
    "cornell_tris_path_strat": ("cornell-tris", (40, 40), abi.SamplerDesc(abi.SAMPLER_STRATIFIED, 2, 2, 1, SEED), abi.IntegratorDesc(abi.INTEGRATOR_PATH, 8, 0, 0.0)),
    "city_tiny_path_uniform": ("city-tiny", (64, 36), abi.SamplerDesc(abi.SAMPLER_UNIFORM, 4, 1, 1, SEED), abi.IntegratorDesc(abi.INTEGRATOR_PATH, 8, 0, 0.0)),
    "city_small_path_strat": ("city-small", (64, 36), abi.SamplerDesc(abi.SAMPLER_STRATIFIED, 2, 2, 1, SEED), abi.IntegratorDesc(abi.INTEGRATOR_PATH, 8, 0, 0.0)),
    "city_small_path_clamp": ("city-small", (48, 27), abi.SamplerDesc(abi.SAMPLER_UNIFORM, 2, 1, 1, SEED), abi.IntegratorDesc(abi.INTEGRATOR_PATH, 6, 1, 0.5)),
    "city_small_geometry_normals": ("city-small", (48, 27), abi.SamplerDesc(abi.SAMPLER_UNIFORM, 1, 1, 1, SEED), abi.IntegratorDesc(abi.INTEGRATOR_GEOMETRY_NORMALS, 1, 0, 0.0)),
    "city_small_bvh_intersections": ("city-small", (48, 27), abi.SamplerDesc(abi.SAMPLER_UNIFORM, 1, 1, 1, SEED), abi.IntegratorDesc(abi.INTEGRATOR_BVH_INTERSECTIONS, 1, 0, 0.0)),
}


def render_case(name):
    scene, res, smp, integ = CASES[name]
    sd = scenes.by_name(scene)
    cam = oracle.make_camera(sd.camera, res)
    tiles = oracle.film_tiles(res, 16)
    rgb, rays = oracle.OracleScene(sd).render_tiles(cam, smp, integ, tiles, n_threads=1)
    return rgb, rays


def main():
    os.makedirs(OUT, exist_ok=True)
    for name in CASES:
        rgb, rays = render_case(name)
        np.savez_compressed(os.path.join(OUT, f"render_{name}.npz"), rgb=rgb, rays=np.uint64(rays))
        print(name, rgb.shape, rays, float(rgb.mean()))
    # traversal vectors
    sd = scenes.by_name("city-small")
    rng = np.random.default_rng(42)
    lo, hi = sd.points.min(axis=0), sd.points.max(axis=0)
    o = (lo + rng.uniform(-0.2, 1.2, (4000, 3)) * (hi - lo)).astype(np.float32)
    d = (lo + rng.uniform(0, 1, (4000, 3)) * (hi - lo)).astype(np.float32) - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    osc = oracle.OracleScene(sd)
    r = osc.intersect(o, d)
    tm = np.full(4000, 0.9999, dtype=np.float32)
    d2 = (d * rng.uniform(0.3, 4.0, (4000, 1))).astype(np.float32)
    occ = osc.any_intersect(o, d2, tm, np.full(4000, 0, dtype=np.int32))
    np.savez_compressed(os.path.join(OUT, "trace_city_small.npz"), o=o, d=d, shape=r["shape"], t=r["t"], node_tests=r["node_tests"], node_hits=r["node_hits"],
                        shape_tests=r["shape_tests"], d_shadow=d2, occluded=occ)
    print("trace", int((r["shape"] >= 0).sum()), int(occ.sum()))


if __name__ == "__main__":
    main()
