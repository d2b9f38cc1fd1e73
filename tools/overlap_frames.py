"""Experiment: K renders of the 1/G share of the cfg3 frame, enqueued (a) on one context and
stream, (b) alternately on two contexts with their own streams, so that the latency tail of
render k overlaps the bulk of render k+1.  Prints ms per render for both."""
import sys, time
sys.path.insert(0, ".")
import torch
from yuki_amd import scenes, core as yk, dist as ydist

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sd = scenes.by_name("cfg3")
fs = yk.FilmSettings(res=(1920, 1080))
tiles = yk.film_tiles(fs)
mine = ydist.shard_tiles(tiles, 0, G)
smp = yk.SamplerType.Stratified((8, 8), True)
dev = torch.device("cuda:0")
sets = []
NCTX = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 2
for i in range(NCTX):
    ctx = yk.Context(0)
    sc = yk.Scene(ctx, sd)
    cam = yk.Camera(sd.camera, fs)
    it = yk.IntegratorType.instantiate(ctx, yk.IntegratorType.Path(yk.PathParams(max_depth=8)))
    tl = yk.TileList(ctx, mine)
    slab = torch.zeros(ydist.slab_pixels(tiles, G) * 3, dtype=torch.float32, device=dev)
    sets.append((ctx, sc, cam, it, tl, slab, torch.cuda.Stream(dev) if "torchstreams" in sys.argv else None))

def run(n_sets):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            ctx, sc, cam, it, tl, slab, s = sets[k % n_sets]
            it.render_tile_list_device(sc, cam, smp, tl, slab.data_ptr(), stream=s.cuda_stream if s else None, want_stats=False)
            if "sync" in sys.argv:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return dt / K * 1e3

a = run(1)
b = a if "one" in sys.argv else run(NCTX)
print(f"G={G}: one context {a:.2f} ms/render, {NCTX} alternating contexts {b:.2f} ms/render")
assert all(torch.equal(sets[0][5], x[5]) for x in sets[1:])
