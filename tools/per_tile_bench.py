"""The literal drop-in: Integrator::render called once per 16x16 tile from T worker threads
(render_manager.rs:78-97) — each thread with its own context on the shared scene, and all threads through
one yk_combiner (calls waiting at the same time share a submission) with 1 / 2 / 3 lanes.
Rays per second on the first N tiles of the cfg3 frame."""
import sys, time, threading
sys.path.insert(0, ".")
import numpy as np
from yuki_amd import scenes, core as yk
import os
OPTS = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("YK_OPTS", "").split(",") if kv)}  # e.g. YK_OPTS=overlap_shadow=0

N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
sd = scenes.by_name("cfg3")
fs = yk.FilmSettings(res=(1920, 1080))
tiles = yk.film_tiles(fs)[:N]
smp = yk.SamplerType.Stratified((8, 8), True)
integ = yk.IntegratorType.Path(yk.PathParams(max_depth=8))
ctx0 = yk.Context(0, **OPTS)
sc = yk.Scene(ctx0, sd)
cam = yk.Camera(sd.camera, fs)
ref, st = yk.IntegratorType.instantiate(ctx0, integ).render_tiles(sc, cam, smp, tiles)
print(f"one batched call: {st.rays} rays in {st.seconds_total*1e3:.1f} ms = {st.rays/st.seconds_total*1e-6:.0f} Mray/s")
offs = np.concatenate([[0], np.cumsum((tiles["x1"].astype(int) - tiles["x0"]) * (tiles["y1"].astype(int) - tiles["y0"]))])
for T in (1, 4, 15):
    ctxs = [yk.Context(0, **OPTS) for _ in range(T)]
    its = [yk.IntegratorType.instantiate(c, integ) for c in ctxs]
    out = np.zeros_like(ref)
    rays = [0] * T
    nxt = [0]
    lock = threading.Lock()

    def worker(k):
        while True:
            with lock:  # the tile queue (render_worker.rs:172-180)
                t = nxt[0]
                nxt[0] += 1
            if t >= len(tiles):
                return
            px, n = its[k].render(sc, cam, smp, yk.FilmTile(tuple(int(v) for v in tiles[t])))
            out[offs[t] : offs[t + 1]] = px
            rays[k] += n

    for k in range(T):  # warm up every context's work buffers
        its[k].render(sc, cam, smp, yk.FilmTile(tuple(int(v) for v in tiles[0])))
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(k,)) for k in range(T)]
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    print(f"{T:2d} threads x own context: {len(tiles)} tiles in {dt*1e3:.0f} ms = {dt/len(tiles)*1e3:.2f} ms per tile, {sum(rays)/dt*1e-6:.1f} Mray/s, identical: {out.tobytes() == ref.tobytes()}")
    for c in ctxs:
        c.close()

# ---- the same workers through the combiner
for T, lanes in ((4, 1), (15, 1), (15, 2), (15, 3), (32, 2), (64, 2)):
    ctxs = [yk.Context(0, **OPTS) for _ in range(lanes)]
    it = yk.IntegratorType.instantiate(ctxs[0], integ)
    for c in ctxs:  # warm up every lane's work buffers with a batch of the size it will see
        yk.IntegratorType.instantiate(c, integ).render_tiles(sc, cam, smp, tiles[: max(1, T // lanes)])
    comb = yk.Combiner(ctxs, linger_us=100)
    out = np.zeros_like(ref)
    rays = [0] * T
    nxt = [0]
    lock = threading.Lock()

    def cworker(k):
        while True:
            with lock:
                t = nxt[0]
                nxt[0] += 1
            if t >= len(tiles):
                return
            px, s = comb.render(it, sc, cam, smp, yk.FilmTile(tuple(int(v) for v in tiles[t])))
            out[offs[t] : offs[t + 1]] = px
            rays[k] += s.rays

    t0 = time.perf_counter()
    th = [threading.Thread(target=cworker, args=(k,)) for k in range(T)]
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    info = comb.info()
    print(f"{T:2d} threads, combiner with {lanes} lane(s): {len(tiles)} tiles in {dt*1e3:.0f} ms = {dt/len(tiles)*1e3:.3f} ms per tile, {sum(rays)/dt*1e-6:.1f} Mray/s, "
          f"{info.submissions} submissions (largest {info.largest_submission}), identical: {out.tobytes() == ref.tobytes()}, rays exact in sum: {sum(rays) == st.rays}")
    comb.close()
    for c in ctxs:
        c.close()
