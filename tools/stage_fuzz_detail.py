import sys
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import numpy as np
import parity_fuzz as pf, stage_fuzz as sf
from oracle import binding as oracle
from yuki_amd import core as yk
seed = int(sys.argv[1])
ctx = pf.variant_context(seed)
sd = pf.random_scene(seed)
r = np.random.default_rng(seed ^ 0xABCDEF)
o, d = sf.rays_for(sd, r)
sc = yk.Scene(ctx, sd); osc = oracle.OracleScene(sd)
g = sc.intersect(o, d, counters=True); w = osc.intersect(o, d)
k = len(o) // 6
names = ["random", "axis-parallel", "through vertices", "along edges", "zeros+scale", "inside"]
bad = (g["shape"] != w["shape"]) | (g["node_tests"] != w["node_tests"]) | (g["node_hits"] != w["node_hits"]) | (g["shape_tests"] != w["shape_tests"])
for c in range(6):
    b = bad[c * k:(c + 1) * k]
    print(names[c], int(b.sum()), "of", k)
idx = np.nonzero(bad)[0][:12]
for i in idx:
    print("ray", i, names[i // k], "o", o[i], "d", d[i], "| got", g["shape"][i], g["t"][i], g["node_tests"][i], g["node_hits"][i], g["shape_tests"][i], "| want", w["shape"][i], w["t"][i], w["node_tests"][i], w["node_hits"][i], w["shape_tests"][i])
