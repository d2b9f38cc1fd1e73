import sys; sys.path.insert(0, ".")
import numpy as np, copy
from yuki_amd import scenes, core as yk, abi
sd = scenes.by_name("cfg3")
if sys.argv[1] == "lambert":
    sd.materials = [dict(kind=abi.MAT_MATTE, a=(0.6, 0.55, 0.5), b=(0, 0, 0), c=0.0, remap=False) for m in sd.materials]
ctx = yk.Context(0); sc = yk.Scene(ctx, sd)
fs = yk.FilmSettings(res=(1920, 1080)); cam = yk.Camera(sd.camera, fs); tiles = yk.film_tiles(fs)
smp = yk.SamplerType.Stratified((8, 8), True)
it = yk.IntegratorType.instantiate(ctx, yk.IntegratorType.Path(yk.PathParams(max_depth=8)))
for rep in range(2):
    out, st = it.render_tiles(sc, cam, smp, tiles)
