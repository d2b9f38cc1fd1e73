"""Device time and ray count of every rank's share of the cfg3 frame for G ranks (run one after
the other on one GPU): the slowest share bounds the strong-scaling step."""
import sys
sys.path.insert(0, ".")
import numpy as np
from yuki_amd import scenes, core as yk, dist as ydist

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sd = scenes.by_name("cfg3")
ctx = yk.Context(0)
sc = yk.Scene(ctx, sd)
fs = yk.FilmSettings(res=(1920, 1080))
cam = yk.Camera(sd.camera, fs)
tiles = yk.film_tiles(fs)
smp = yk.SamplerType.Stratified((8, 8), True)
it = yk.IntegratorType.instantiate(ctx, yk.IntegratorType.Path(yk.PathParams(max_depth=8)))
ms, rays = [], []
for r in range(G):
    mine = ydist.shard_tiles(tiles, r, G)
    best = 1e9
    for _ in range(3):
        _, st = it.render_tiles(sc, cam, smp, mine)
        best = min(best, st.seconds_total)
    ms.append(best * 1e3)
    rays.append(st.rays)
ms, rays = np.array(ms), np.array(rays)
print("ms per share:", np.round(ms, 2), f"max/mean {ms.max()/ms.mean():.3f}")
print("rays per share (M):", np.round(rays / 1e6, 2), f"max/mean {rays.max()/rays.mean():.3f}")
