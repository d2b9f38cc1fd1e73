"""Details of the mismatches parity_fuzz.py reports for one seed."""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import numpy as np
import parity_fuzz as pf
from oracle import binding as oracle
from yuki_amd import core as yk

seed = int(sys.argv[1])
res = (48, 32)
ctx = yk.Context(0)
r = np.random.default_rng(seed ^ 0x5EED)
sd = pf.random_scene(seed)
fs = yk.FilmSettings(res=res, tile_dim=16)
cam = yk.Camera(sd.camera, fs)
tiles = yk.film_tiles(fs)
sc = yk.Scene(ctx, sd)
osc = oracle.OracleScene(sd)
smp = yk.SamplerType.Uniform(int(r.integers(1, 5)), int(r.integers(1, 2**62))) if r.random() < 0.5 else yk.SamplerType.Stratified((int(r.integers(1, 4)), int(r.integers(1, 4))), bool(r.integers(0, 2)), int(r.integers(1, 2**62)))
clamp = None if r.random() < 0.6 else float(r.uniform(0.1, 5))
integs = {"path": yk.IntegratorType.Path(yk.PathParams(max_depth=int(r.integers(0, 9)), indirect_clamp=clamp)), "whitted": yk.IntegratorType.Whitted(int(r.integers(0, 7))),
          "geometry_normals": yk.IntegratorType.GeometryNormals, "shading_normals": yk.IntegratorType.ShadingNormals, "bvh": yk.IntegratorType.BVHIntersections}
print("scene:", sd.n_triangles, "tris", len(sd.spheres), "spheres", len(sd.lights), "lights", "split", sd.split_method, "max_shapes", sd.max_shapes_in_node, "sampler", smp.kind, smp.nx, smp.ny, "camera", sd.camera)
offs = np.concatenate([[0], np.cumsum((tiles["x1"].astype(int) - tiles["x0"]) * (tiles["y1"].astype(int) - tiles["y0"]))])
for name, integ in integs.items():
    got, st = yk.IntegratorType.instantiate(ctx, integ).render_tiles(sc, cam, smp, tiles)
    want, rays = osc.render_tiles(cam.matrices, smp, integ, tiles, n_threads=8)
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    px = np.nonzero(~same.all(axis=1))[0]
    print(name, "max_depth", integ.max_depth, "mismatching pixels", len(px), "rays", st.rays, rays)
    for p in px[:6]:
        t = int(np.searchsorted(offs, p, side="right") - 1)
        loc = p - offs[t]
        w = int(tiles[t]["x1"]) - int(tiles[t]["x0"])
        print("   pixel", (int(tiles[t]["x0"]) + loc % w, int(tiles[t]["y0"]) + loc // w), "got", got[p], "want", want[p])
# closest-hit stage on the camera rays of sample 0
o, d = yk.camera_rays(ctx, cam, smp, (0, 0, res[0], res[1]), 0)
g = sc.intersect(o, d, counters=True)
w = osc.intersect(o, d)
for k in ("shape", "node_tests", "node_hits", "shape_tests"):
    diff = np.nonzero(g[k] != w[k])[0]
    print("stage", k, "differs at", len(diff), "rays", diff[:8], "got", g[k][diff[:8]], "want", w[k][diff[:8]])
hit = w["shape"] >= 0
dt = np.nonzero(hit & (g["t"].view(np.uint32) != w["t"].view(np.uint32)))[0]
print("stage t differs at", len(dt))
for i in np.unique(np.concatenate([np.nonzero(g["shape"] != w["shape"])[0][:3], np.nonzero(g["node_tests"] != w["node_tests"])[0][:3]])).astype(int):
    print("  ray", i, "o", o[i], "d", d[i], "got shape/t", g["shape"][i], g["t"][i], "want", w["shape"][i], w["t"][i])
