"""Per render of the 1/G frame share: host wall time of the synchronous call vs the device time
between the events the library records around it."""
import sys, time
sys.path.insert(0, ".")
import torch
from yuki_amd import scenes, core as yk, dist as ydist

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sd = scenes.by_name("cfg3")
fs = yk.FilmSettings(res=(1920, 1080))
tiles = yk.film_tiles(fs)
mine = ydist.shard_tiles(tiles, 0, G)
smp = yk.SamplerType.Stratified((8, 8), True)
dev = torch.device("cuda:0")
ctx = yk.Context(0)
sc = yk.Scene(ctx, sd)
cam = yk.Camera(sd.camera, fs)
it = yk.IntegratorType.instantiate(ctx, yk.IntegratorType.Path(yk.PathParams(max_depth=8)))
tl = yk.TileList(ctx, mine)
slab = torch.zeros(ydist.slab_pixels(tiles, G) * 3, dtype=torch.float32, device=dev)
for mode in ("stats", "nostats"):
    for tk in (1, 0):
        ctx.set_option("time_kernels", tk)
        rows = []
        for k in range(8):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = it.render_tile_list_device(sc, cam, smp, tl, slab.data_ptr(), stream=None, want_stats=(mode == "stats"))
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            rows.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3, st.seconds_total * 1e3 if st else 0.0))
        r = rows[-1]
        print(f"{mode} time_kernels={tk}: call returns after {r[0]:.2f} ms, synchronised after {r[1]:.2f} ms, device events {r[2]:.2f} ms")
