"""What a hipGraph returns for the reference's per-tile calls ("hipGraph was rejected from a timeline estimate, not from a
measurement").  Timing build: tools/build_variant.sh graph -DYK_EXPERIMENT_GRAPH; run on the GPU box with
YK_LIB_PATH=yuki_amd/libyuki_hip_graph.so python tools/graph_tile_bench.py
One 16x16 tile of the cfg3 frame at 64 spp (16 K paths, 8 bounces, ~40 dependent launches on two streams):
  plain    N calls of Integrator::render from one thread (each: enqueue, wait, read back)
  graph    the same job captured once and launched N times back to back (device time per launch, host time per launch)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time; sys.path.insert(0, sys.argv[1])
import numpy as np
from yuki_amd import scenes, core as yk
sd = scenes.by_name("cfg3")
fs = yk.FilmSettings(res=(1920, 1080)); tiles = yk.film_tiles(fs)
ctx = yk.Context(0); sc = yk.Scene(ctx, sd); cam = yk.Camera(sd.camera, fs)
smp = yk.SamplerType.Stratified((8, 8), True)
it = yk.IntegratorType.instantiate(ctx, yk.IntegratorType.Path(yk.PathParams(max_depth=8)))
import os
GRAPH = os.environ.pop('YK_GRAPH_REPLAY_LATER', None)
N = 2 if GRAPH else 200  # a graph-mode call replays its captured job that many times itself
for t in (0, 1500, 8000):
    tile = yk.FilmTile(tuple(int(v) for v in tiles[t]))
    os.environ.pop('YK_GRAPH_REPLAY', None)
    it.render(sc, cam, smp, tile)  # plain: the buffers exist afterwards
    if GRAPH:
        os.environ['YK_GRAPH_REPLAY'] = GRAPH
    t0 = time.perf_counter()
    for _ in range(N):
        px, rays = it.render(sc, cam, smp, tile)
    dt = (time.perf_counter() - t0) / N
    print(f"tile {t}: {rays} rays, {dt * 1e3:.3f} ms per call", flush=True)
'''
for mode, env in (("plain", {}), ("graph", {"YK_GRAPH_REPLAY_LATER": "200"})):
    print(f"== {mode}")
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT], capture_output=True, text=True, env=dict(os.environ, **env))
    print(r.stdout.strip())
    print("\n".join(l for l in r.stderr.split("\n") if "graph replay" in l or "rror" in l))
