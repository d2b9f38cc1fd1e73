"""Time the per-rank share of a strong-scaling run on ONE GPU: render tiles[r::G] of the
cfg3 frame for G in (1, 2, 4, 8) — what each rank of bench.py does before the gather."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from yuki_amd import scenes, core as yk, dist as ydist

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
sd = scenes.by_name(name)
ctx = yk.Context(0)
sc = yk.Scene(ctx, sd)
fs = yk.FilmSettings(res=(1920, 1080))
cam = yk.Camera(sd.camera, fs)
tiles = yk.film_tiles(fs)
smp = yk.SamplerType.Stratified((8, 8), True)
it = yk.IntegratorType.instantiate(ctx, yk.IntegratorType.Path(yk.PathParams(max_depth=8)))
t1 = None
for G in (1, 2, 4, 8):
    mine = ydist.shard_tiles(tiles, 0, G)
    best = 1e9
    for rep in range(4):
        out, st = it.render_tiles(sc, cam, smp, mine)
        best = min(best, st.seconds_total)
    if G == 1:
        t1 = best
    print(f"G={G}: {len(mine)} tiles, device {best*1e3:.2f} ms, rays {st.rays}, ideal-speedup {t1/best:.2f} of {G}, batches {st.batches}")
