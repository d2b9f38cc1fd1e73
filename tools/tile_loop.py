"""N per-tile renders (Integrator::render, one 16x16 tile each) from one thread: the workload of tools/gpu_tile_timeline.sh."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yuki_amd import scenes, core as yk

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
sd = scenes.by_name("cfg3")
fs = yk.FilmSettings(res=(1920, 1080))
tiles = yk.film_tiles(fs)[:N]
smp = yk.SamplerType.Stratified((8, 8), True)
ctx = yk.Context(0)
sc = yk.Scene(ctx, sd)
cam = yk.Camera(sd.camera, fs)
it = yk.IntegratorType.instantiate(ctx, yk.IntegratorType.Path(yk.PathParams(max_depth=8)))
it.render(sc, cam, smp, yk.FilmTile(tuple(int(v) for v in tiles[0])))
t0 = time.perf_counter()
for t in tiles:
    it.render(sc, cam, smp, yk.FilmTile(tuple(int(v) for v in t)))
dt = time.perf_counter() - t0
print(f"{N} tiles, {dt / N * 1e3:.3f} ms per tile (host wall clock)")
