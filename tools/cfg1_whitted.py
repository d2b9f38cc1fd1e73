"""BASELINE configs[0]: built-in Cornell box, Whitted max_depth 3, 512x512, Uniform 1 spp — the
reference's own CPU-runnable case, on the device (k_whitted) and on the CPU restatement."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from yuki_amd import scenes, core as yk
from oracle import binding as ob

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sd = scenes.by_name("cornell")
ctx = yk.Context(0)
sc = yk.Scene(ctx, sd)
fs = yk.FilmSettings(res=(512, 512))
cam = yk.Camera(sd.camera, fs)
tiles = yk.film_tiles(fs)
smp = yk.SamplerType.Uniform(spp, 0x73B9642E74AC471C)
integ = yk.IntegratorType.Whitted(3)
it = yk.IntegratorType.instantiate(ctx, integ)
for _ in range(3):
    img, st = it.render_tiles(sc, cam, smp, tiles)
t0 = time.time()
want, rays = ob.OracleScene(sd).render_tiles(cam.matrices, smp, integ, tiles, n_threads=15)
dt = time.time() - t0
print(f"GPU: {st.rays} rays, {st.shadow_rays} shadow rays in {st.seconds_total*1e3:.3f} ms = {st.rays/st.seconds_total*1e-6:.1f} Mray/s; "
      f"oracle (15 threads): {rays} rays in {dt*1e3:.1f} ms = {rays/dt*1e-6:.2f} Mray/s; identical: {img.tobytes() == want.tobytes()}")
