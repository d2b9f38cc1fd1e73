"""A/B one context option on the cfg3 frame and on the 1/8-frame shard:  opt_bench.py key v1 v2 ..."""
import sys
sys.path.insert(0, ".")
from yuki_amd import scenes, core as yk, dist as ydist

key, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
sd = scenes.by_name("cfg3")
fs = yk.FilmSettings(res=(1920, 1080))
smp = yk.SamplerType.Stratified((8, 8), True)
for v in vals:
    ctx = yk.Context(0, **{key: v})
    sc = yk.Scene(ctx, sd)
    cam = yk.Camera(sd.camera, fs)
    tiles = yk.film_tiles(fs)
    it = yk.IntegratorType.instantiate(ctx, yk.IntegratorType.Path(yk.PathParams(max_depth=8)))
    out = []
    for G in (1, 8):
        mine = ydist.shard_tiles(tiles, 0, G)
        best = min(it.render_tiles(sc, cam, smp, mine)[1].seconds_total for _ in range(4))
        out.append(best * 1e3)
    print(f"{key}={v}: full {out[0]:.2f} ms  1/8 {out[1]:.2f} ms  (x{out[0]/out[1]:.2f})")
    del sc
    ctx.close()
