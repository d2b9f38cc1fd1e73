"""Load balance of the round-robin tile deal: device time and ray count of EVERY rank's share of the cfg3 frame for G = 8 (and 4),
one after the other on one GPU.  The frame time of a G-GPU run is the slowest share's."""
import sys
sys.path.insert(0, ".")
import numpy as np
from yuki_amd import scenes, core as yk, dist as ydist

sd = scenes.by_name(sys.argv[1] if len(sys.argv) > 1 else "cfg3")
ctx = yk.Context(0)
sc = yk.Scene(ctx, sd)
fs = yk.FilmSettings(res=(1920, 1080))
cam = yk.Camera(sd.camera, fs)
tiles = yk.film_tiles(fs)
smp = yk.SamplerType.Stratified((8, 8), True)
it = yk.IntegratorType.instantiate(ctx, yk.IntegratorType.Path(yk.PathParams(max_depth=8)))
for G in (4, 8):
    ms, rays = [], []
    for r in range(G):
        mine = ydist.shard_tiles(tiles, r, G)
        best = 1e9
        for rep in range(3):
            out, st = it.render_tiles(sc, cam, smp, mine)
            best = min(best, st.seconds_total)
        ms.append(best * 1e3)
        rays.append(st.rays)
    ms, rays = np.array(ms), np.array(rays, dtype=np.float64)
    print(f"G={G}: share times {np.round(ms, 2).tolist()} ms; max / mean = {ms.max() / ms.mean():.3f}; rays max / mean = {rays.max() / rays.mean():.3f}")
