#!/usr/bin/env python3
"""The oracle's platform-dependent choices in image space.  (CPU only.)

The reference leaves two things to its platform: Rust's f32::{sin,cos,tan,ln,atan2,acos} are the
platform libm (call sites sampling/mod.rs:62-87, trowbridge_reitz.rs:23-30,60-74, camera.rs:52-102,
sphere.rs:38-119), and `select_nth_unstable_by` (bvh.rs:430) orders equal keys as std's pdqselect
happens to.  The oracle (and the HIP kernels) restate glibc 2.35's functions bit for bit (olibm.h)
and fix the selection by a spec.  This script renders the same tiles with two more oracle builds
(oracle/Makefile `flavours`):

  hostlibm   the six functions call the platform's sinf ... — what a build of the reference links here
  nth        std::nth_element in place of the select_nth spec — another legitimate selection

and reports, against the default oracle, per-pixel RMSE (the north-star tolerance is 1e-4), the
largest absolute difference, how many pixels / camera samples differ in any bit and how many
samples took a different path (a sample whose radiance moves by more than 1e-3 relative is counted
as "another path").  On glibc >= 2.35 / x86-64 with FMA every hostlibm row must read 0: the
restatement IS the platform's libm.  On another platform the rows are that platform's distance from
the Linux images; profiles/r03_libm_sensitivity_f64_recipe.txt keeps the table of the fixed f64
recipe the oracle used before (1.7e-5 on cfg3, 1.4e-3 on a Cornell tile, all of it sinf / cosf).

    python tools/libm_sensitivity.py [--quick] [--out profiles/r03_libm_sensitivity.txt]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as oracle  # noqa: E402
from yuki_amd import abi, scenes  # noqa: E402

SEED = 0x73B9642E74AC471C
U, S = abi.SAMPLER_UNIFORM, abi.SAMPLER_STRATIFIED
PATH, WHITTED = abi.INTEGRATOR_PATH, abi.INTEGRATOR_WHITTED

# name: (scene, res, sampler, integrator, first K spiral tiles or None = whole film)
CASES = {
    "cfg2 (69,312 tri, Path 8, Uniform 16, 1080p): first 192 tiles": ("cfg2", (1920, 1080), (U, 16, 1), (PATH, 8), 192),
    "cfg3 (1,024,012 tri, Path 8, Stratified 8x8, 1080p): first 96 tiles": ("cfg3", (1920, 1080), (S, 8, 8), (PATH, 8), 96),
    "golden cornell_whitted 48x48x1": ("cornell", (48, 48), (U, 1, 1), (WHITTED, 3), None),
    "golden cornell_path 32x32x4 (copper sphere: GGX + Sphere::intersect)": ("cornell", (32, 32), (U, 4, 1), (PATH, 8), None),
    "cornell_path 64x64x64": ("cornell", (64, 64), (S, 8, 8), (PATH, 8), None),
    "golden cornell_tris_path_strat 40x40x4": ("cornell-tris", (40, 40), (S, 2, 2), (PATH, 8), None),
    "golden city_tiny_path_uniform 64x36x4": ("city-tiny", (64, 36), (U, 4, 1), (PATH, 8), None),
    "golden city_small_path_strat 64x36x4": ("city-small", (64, 36), (S, 2, 2), (PATH, 8), None),
    "city_small 128x72x64": ("city-small", (128, 72), (S, 8, 8), (PATH, 8), None),
    "glass_balls whitted 64x64x1 depth 8": ("glass-balls", (64, 64), (U, 1, 1), (WHITTED, 8), None),
    # the selection is only reached through SplitMethod::EqualCounts or the fallback chain (bvh.rs:352-388): force it
    "city_small, EqualCounts split, 128x72x16": ("city-small", (128, 72), (S, 4, 4), (PATH, 8), None, abi.SPLIT_EQUAL_COUNTS),
    "coplanar_slabs (duplicated centroids), EqualCounts split, 64x64x16": ("coplanar-slabs", (64, 64), (S, 4, 4), (PATH, 8), None, abi.SPLIT_EQUAL_COUNTS),
    "cfg2, EqualCounts split: first 48 tiles": ("cfg2", (1920, 1080), (U, 16, 1), (PATH, 8), 48, abi.SPLIT_EQUAL_COUNTS),
}
QUICK = {
    "cfg2 (69,312 tri, Path 8, Uniform 16, 1080p): first 24 tiles": ("cfg2", (1920, 1080), (U, 16, 1), (PATH, 8), 24),
    "golden cornell_path 32x32x4 (copper sphere: GGX + Sphere::intersect)": ("cornell", (32, 32), (U, 4, 1), (PATH, 8), None),
    "golden city_small_path_strat 64x36x4": ("city-small", (64, 36), (S, 2, 2), (PATH, 8), None),
}


def render(case, flavour, threads):
    scene, res, (sk, nx, ny), (ik, depth), k = case[:5]
    smp = abi.SamplerDesc(sk, nx, ny, 1, SEED)
    integ = abi.IntegratorDesc(ik, depth, 0, 0.0)
    with oracle.flavour(flavour):
        sd = scenes.by_name(scene)
        if len(case) > 5:
            sd.split_method = case[5]
        cam = oracle.make_camera(sd.camera, res)
        tiles = oracle.film_tiles(res, 16)
        if k is not None:
            tiles = tiles[:k]
        osc = oracle.OracleScene(sd)
        nodes, order = osc.export_bvh()
        rgb, rays, ps = osc.render_tiles(cam, smp, integ, tiles, n_threads=threads, per_sample=True)
        osc.close()
    return dict(rgb=rgb, rays=rays, ps=ps, nodes=nodes, order=order)


def compare(base, other):
    a, b = base["rgb"].astype(np.float64), other["rgb"].astype(np.float64)
    pa, pb = base["ps"], other["ps"]
    d = b - a
    pix_diff = int((base["rgb"].view(np.uint32) != other["rgb"].view(np.uint32)).any(axis=1).sum())
    smp_diff = (pa.view(np.uint32) != pb.view(np.uint32)).any(axis=2)
    scale = np.maximum(np.abs(pa).max(axis=2), 1e-3)
    smp_path = (np.abs(pb.astype(np.float64) - pa).max(axis=2) > 1e-3 * scale)
    nodes_same = base["nodes"].shape == other["nodes"].shape and bool((base["nodes"].tobytes() == other["nodes"].tobytes()))
    order_diff = int((base["order"] != other["order"]).sum()) if base["order"].shape == other["order"].shape else -1
    return dict(
        rmse=float(np.sqrt(np.mean(d * d))), max_abs=float(np.abs(d).max()), mean=float(a.mean()),
        pixels=a.shape[0], pixels_differ=pix_diff, samples=int(smp_diff.size), samples_differ=int(smp_diff.sum()),
        samples_other_path=int(smp_path.sum()), rays=(base["rays"], other["rays"]), nodes_same=nodes_same, order_diff=order_diff,
    )


def fmt(name, fl, c):
    return (f"{name}\n    {fl:9s} rmse {c['rmse']:.3e}  max|d| {c['max_abs']:.3e}  (mean radiance {c['mean']:.3f})  "
            f"pixels differing {c['pixels_differ']}/{c['pixels']}  samples differing {c['samples_differ']}/{c['samples']} "
            f"({c['samples_other_path']} on another path)  rays {c['rays'][0]} -> {c['rays'][1]}  "
            f"BVH nodes identical: {c['nodes_same']}, leaf-order slots moved: {c['order_diff']}")


def run(cases, threads, log=print):
    results = {}
    for name, case in cases.items():
        t0 = time.time()
        base = render(case, "default", threads)
        for fl in ("hostlibm", "nth"):
            c = compare(base, render(case, fl, threads))
            results[(name, fl)] = c
            log(fmt(name, fl, c))
        log(f"    ({time.time() - t0:.1f} s)")
    return results


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--threads", type=int, default=max(1, (os.cpu_count() or 2) - 1))
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    lines = []

    def log(s):
        print(s, flush=True)
        lines.append(s)

    log("# oracle flavours against the default oracle (tools/libm_sensitivity.py); tolerance of the north star: RMSE < 1e-4")
    log(f"# glibc: {os.confstr('CS_GNU_LIBC_VERSION')}; RMSE over all pixels and channels of the rendered tiles at full spp")
    res = run(QUICK if a.quick else CASES, a.threads, log)
    worst = {fl: max(c["rmse"] for (n, f), c in res.items() if f == fl) for fl in ("hostlibm", "nth")}
    log(f"# worst RMSE: hostlibm {worst['hostlibm']:.3e}, nth {worst['nth']:.3e}")
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
