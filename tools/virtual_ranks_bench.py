"""What the G-rank machinery costs when the wire is free: the cfg3 frame rendered by yk_multi with G = 1 / 2 / 4 / 8 VIRTUAL ranks on
one GPU (YK_MULTI_SHARED_DEVICES: every rank a context, a host thread, a scene copy, a tile list and a slab of its own; slabs moved
by device copies, scattered on rank 0) against the plain single-context render.  The G ranks share one GPU, so the frame cannot get
faster — the figure of interest is how little slower it gets (deal, G x fewer rays per launch, exchange, scatter).
Run on the GPU box: python tools/virtual_ranks_bench.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch  # noqa: F401  (one HIP runtime per process: first)
from yuki_amd import scenes, core as yk

sd = scenes.by_name("cfg3")
fs = yk.FilmSettings(res=(1920, 1080), tile_dim=16)
smp = yk.SamplerType.Stratified((8, 8), True)
integ = yk.IntegratorType.Path(yk.PathParams(max_depth=8))
cam = yk.Camera(sd.camera, fs)
ctx = yk.Context(0)
sc = yk.Scene(ctx, sd)
tiles = yk.film_tiles(fs)
it = yk.IntegratorType.instantiate(ctx, integ)
best = 1e9
for _ in range(4):
    t0 = time.perf_counter()
    rgb, st = it.render_tiles(sc, cam, smp, tiles)
    best = min(best, time.perf_counter() - t0)
want = yk.update_tiles(tiles, rgb, fs.res)
print(f"single context: {best * 1e3:.1f} ms per frame (host clock, film read back), {st.rays} rays")
sc.close(); ctx.close()
for G in (1, 2, 4, 8):
    m = yk.Multi([0] * G, flags=yk.Multi.SHARED_DEVICES if G > 1 else 0)
    msc = m.scene(sd)
    film = m.film(fs)
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        got, st = m.render_film(msc, cam, smp, integ, film)
        best = min(best, time.perf_counter() - t0)
    same = np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # asynchronous frames back to back (no read-back, no statistics): the steady state of a display loop
    K = 6
    m.render_film(msc, cam, smp, integ, film, want_host=False, want_stats=False)
    m.sync()
    t0 = time.perf_counter()
    for _ in range(K):
        m.render_film(msc, cam, smp, integ, film, want_host=False, want_stats=False)
    m.sync()
    dt = (time.perf_counter() - t0) / K
    print(f"G = {G} virtual ranks: {best * 1e3:.1f} ms per frame synchronous, {dt * 1e3:.1f} ms per frame enqueued back to back; film identical: {same}; rays {st.rays}")
    film.close(); msc.close(); m.close()
