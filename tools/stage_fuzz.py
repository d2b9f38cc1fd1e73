"""Randomised parity of the traversal stages: BoundingVolumeHierarchy::intersect / any_intersect
for degenerate rays — directions with exact zero components (0 * inf = NaN lanes in the slab
test), origins on box faces and on vertices, rays along triangle edges and through vertices,
small and large directions, finite t_max at exact hit distances — on the random scenes of
parity_fuzz.py.  Hit shape, t bits and (binary layout) the three counters must equal the oracle's.
    stage_fuzz.py [first_seed] [count]"""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import numpy as np
import parity_fuzz as pf
from yuki_amd import core as yk

F = np.float32


def rays_for(sd, r, n=1500):
    p = sd.points
    k = n // 6
    o, d = [], []
    # random
    o.append(r.uniform(-3, 3, (k, 3))); d.append(r.normal(size=(k, 3)))
    # axis-parallel, exact zeros
    ax = np.eye(3)[r.integers(0, 3, k)] * r.choice([-1.0, 1.0], (k, 1))
    o.append(r.uniform(-3, 3, (k, 3))); d.append(ax)
    # from outside through vertices (grazing edges / vertices), origins snapped to vertex coordinates (on box faces)
    v = p[r.integers(0, len(p), k)]
    oo = r.uniform(-3, 3, (k, 3))
    snap = r.random((k, 3)) < 0.3
    oo[snap] = v[snap]
    o.append(oo); d.append(v - oo + (r.random((k, 1)) < 0.5) * r.normal(scale=1e-6, size=(k, 3)))
    # along edges
    t = sd.indices[r.integers(0, len(sd.indices), k)]
    a, b = p[t[:, 0]], p[t[:, 1]]
    o.append(a - (b - a) * r.uniform(0, 2, (k, 1))); d.append(b - a)
    # zero components + small / large scale (the renderer's rays are unit vectors or point-to-point segments;
    # |d| ~ 1e20 overflows the triangle test to NaN hits, whose ordering the reference leaves to NaN-dropping min/max)
    dd = r.normal(size=(k, 3)) * r.choice([1e-6, 1e-3, 1.0, 1e3, 1e6], (k, 1))
    dd[r.random((k, 3)) < 0.4] = 0.0
    o.append(r.uniform(-2, 2, (k, 3))); d.append(dd)
    # inside the scene
    o.append(r.uniform(-1, 1, (k, 3))); d.append(r.normal(size=(k, 3)))
    o, d = np.concatenate(o).astype(F), np.concatenate(d).astype(F)
    zero = ~d.any(axis=1)  # a ray needs a direction
    d[zero] = (1.0, 0.0, 0.0)
    return o, d


def check_seed(oracle, seed):
    ctx = pf.variant_context(seed)
    sd = pf.random_scene(seed)
    r = np.random.default_rng(seed ^ 0xABCDEF)
    o, d = rays_for(sd, r)
    sc = yk.Scene(ctx, sd)
    osc = oracle.OracleScene(sd)
    bad = []
    binary = pf.VARIANTS[seed % len(pf.VARIANTS)].get("wide_bvh", 2) == 0
    g = sc.intersect(o, d, counters=True)
    w = osc.intersect(o, d)
    hit = w["shape"] >= 0
    if not np.array_equal(g["shape"], w["shape"]):
        bad.append(("shape", int((g["shape"] != w["shape"]).sum())))
    if not np.array_equal(g["t"][hit].view(np.uint32), w["t"][hit].view(np.uint32)):
        bad.append(("t", int((g["t"][hit].view(np.uint32) != w["t"][hit].view(np.uint32)).sum())))
    for k in ("node_tests", "node_hits", "shape_tests"):  # the counting kernel always walks the binary nodes
        if not np.array_equal(g[k], w[k]):
            bad.append((k, int((g[k] != w[k]).sum())))
    # finite t_max: exactly the hit distance, one ulp below / above it, and random
    tm = np.where(hit, w["t"], F(1.0)).astype(F)
    for name, tmax in (("t_max = t", tm), ("t_max = t-", np.nextafter(tm, F(0))), ("t_max = t+", np.nextafter(tm, F(np.inf))), ("random t_max", r.uniform(0, 4, len(o)).astype(F))):
        g2 = sc.intersect(o, d, t_max=tmax)
        w2 = osc.intersect(o, d, tmax)
        if not np.array_equal(g2["shape"], w2["shape"]):
            bad.append((name + " shape", int((g2["shape"] != w2["shape"]).sum())))
        al = r.integers(-1, max(1, len(sd.lights)), len(o)).astype(np.int32)
        ga = sc.any_intersect(o, d, tmax, al)
        wa = osc.any_intersect(o, d, tmax, al)
        if not np.array_equal(ga, wa):
            bad.append((name + " any", int((ga != wa).sum())))
    sc.close()
    return bad


if __name__ == "__main__":
    from oracle import binding as oracle

    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    failures = 0
    for seed in range(first, first + count):
        bad = check_seed(oracle, seed)
        if bad:
            failures += 1
            print(f"seed {seed} (variant {pf.VARIANTS[seed % len(pf.VARIANTS)]}): MISMATCH {bad}", flush=True)
        elif seed % 25 == 0:
            print(f"seed {seed}: ok", flush=True)
    print(f"{count} seeds, {failures} with mismatches")
