import sys; sys.path.insert(0, ".")
import numpy as np
from yuki_amd import scenes, core as yk, dist as ydist
sd = scenes.by_name("cfg3"); ctx = yk.Context(0, streams=int(sys.argv[1])); sc = yk.Scene(ctx, sd)
fs = yk.FilmSettings(res=(1920, 1080)); cam = yk.Camera(sd.camera, fs); tiles = yk.film_tiles(fs)
smp = yk.SamplerType.Stratified((8, 8), True)
it = yk.IntegratorType.instantiate(ctx, yk.IntegratorType.Path(yk.PathParams(max_depth=8)))
mine = ydist.shard_tiles(tiles, 0, 8)
for rep in range(3):
    out, st = it.render_tiles(sc, cam, smp, mine)
    print("device %.2f ms trace %.2f shadow %.2f shade %.2f batches %d" % (st.seconds_total*1e3, st.seconds_trace*1e3, st.seconds_shadow*1e3, st.seconds_shade*1e3, st.batches))
