"""The reference's interactive mode (accumulating film, render_manager.rs:125-143): passes of ONE
sample per pixel over the whole 1080p tile queue, accumulated into the film on the device.
Rays per second when a submission renders 1, 2, 4, 8, 16 passes, on one context and alternating
between two contexts (submissions in flight)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from yuki_amd import scenes, core as yk
import os
OPTS = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("YK_OPTS", "").split(",") if kv)}  # e.g. YK_OPTS=overlap_shadow=0

TOTAL = int(sys.argv[1]) if len(sys.argv) > 1 else 32  # passes per measurement
sd = scenes.by_name("cfg3")
fs = yk.FilmSettings(res=(1920, 1080), accumulate=True)
tiles = yk.film_tiles(fs)
smp = yk.SamplerType.Stratified((8, 8), True)
dev = torch.device("cuda:0")
integ = yk.IntegratorType.Path(yk.PathParams(max_depth=8))
ctxs = [yk.Context(0, **OPTS), yk.Context(0, **OPTS)]
sc = yk.Scene(ctxs[0], sd)
cam = yk.Camera(sd.camera, fs)
its = [yk.IntegratorType.instantiate(c, integ) for c in ctxs]
lists = {s: yk.TileList(ctxs[0], tiles, np.full(len(tiles), s, dtype=np.uint16)) for s in range(TOTAL)}
npx = lists[0].n_pixels
film = torch.zeros(1080 * 1920 * 3, dtype=torch.float32, device=dev)
films = {}
for n in (1, 2, 4, 8, 16):
    slabs = [torch.zeros(n * npx * 3, dtype=torch.float32, device=dev) for _ in range(2)]
    st = its[0].render_tile_list_device(sc, cam, smp, lists[0], slabs[0].data_ptr(), want_stats=True, n_passes=n)
    rays_per_pass = st.rays / n
    line = f"{n:2d} passes per submission: {st.seconds_total*1e3/n:6.2f} ms per pass alone;"
    for n_ctx in (1, 2):
        for rep in range(2):
            film.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for j, s in enumerate(range(0, TOTAL, n)):
                k = j % n_ctx
                its[k].render_tile_list_device(sc, cam, smp, lists[s], slabs[k].data_ptr(), n_passes=n)
                lists[s].update_film_device(slabs[k].data_ptr(), fs.res, film.data_ptr(), accumulate=True, ctx=ctxs[k], n_passes=n)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / TOTAL
        line += f"  {n_ctx} context(s): {dt*1e3:5.2f} ms per pass = {rays_per_pass/dt*1e-6:5.0f} Mray/s;"
        if n_ctx == 1:
            films[n] = film.clone()
    print(line)
print("films of all groupings identical:", all(torch.equal(films[1], f) for f in films.values()))
