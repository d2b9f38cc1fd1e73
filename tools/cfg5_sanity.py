"""cfg5-scale sanity: ~10 M triangles, 3840x2160, Path 16 — one frame at reduced spp, timings + memory."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from yuki_amd import scenes, core as yk

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
t0 = time.time(); sd = scenes.by_name("cfg5"); print("generate %.1f s, %d triangles" % (time.time() - t0, sd.n_triangles), flush=True)
ctx = yk.Context(0)
t0 = time.time(); sc = yk.Scene(ctx, sd); i = sc.info()
print("scene create %.1f s (BVH build %.1f s, upload %.2f s), %d nodes, depth %d, %.2f GB on device" % (time.time() - t0, i.build_seconds, i.upload_seconds, i.n_nodes, i.tree_depth, i.device_bytes / 1e9), flush=True)
fs = yk.FilmSettings(res=(3840, 2160)); cam = yk.Camera(sd.camera, fs); tiles = yk.film_tiles(fs)
n = int(round(spp ** 0.5))
smp = yk.SamplerType.Stratified((n, n), True)
it = yk.IntegratorType.instantiate(ctx, yk.IntegratorType.Path(yk.PathParams(max_depth=16)))
for rep in range(2):
    out, st = it.render_tiles(sc, cam, smp, tiles)
    print("render %dx%d spp %d: %.3f s device, %d rays, %.1f Mray/s, batches %d, finite %s, mean %s" % (fs.res[0], fs.res[1], n * n, st.seconds_total, st.rays, st.rays / st.seconds_total * 1e-6, st.batches, bool(np.isfinite(out).all()), out.mean(axis=0)), flush=True)
