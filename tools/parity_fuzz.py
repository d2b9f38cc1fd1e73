"""Randomised parity: seeded random scenes — triangle soups with degenerate and duplicated
triangles, axis-aligned slabs (rays parallel to box faces), vertices at +-0, all five material
kinds with extreme parameters, every light kind, spheres, normals / uvs on some meshes — rendered
by the device and by the oracle with every integrator; any differing bit is a failure.
    parity_fuzz.py [first_seed] [count]"""
import sys
sys.path.insert(0, ".")
import numpy as np
from yuki_amd import abi, scenes, core as yk

F = np.float32


def _translation(v):
    m = np.eye(4, dtype=F)
    m[:3, 3] = v
    inv = np.eye(4, dtype=F)
    inv[:3, 3] = -np.asarray(v, dtype=F)
    return m, inv


def random_scene(seed):
    r = np.random.default_rng(seed)
    pts, idx, tmesh, tmat, tal, meshes, normals, uvs = [], [], [], [], [], [], [], []
    base = 0
    n_mesh = int(r.integers(1, 5))
    n_mat = int(r.integers(1, 7))
    mats = []
    for _ in range(n_mat):
        k = int(r.integers(0, 5))
        if k == 0:
            mats.append(dict(kind=abi.MAT_MATTE, a=tuple(r.choice([0.0, 0.2, 0.9, 1.0], 3)), c=float(r.choice([0.0, 0.0, 0.35, 20.0]))))
        elif k == 1:
            mats.append(dict(kind=abi.MAT_GLASS, a=tuple(r.uniform(0.5, 1, 3)), b=tuple(r.uniform(0.5, 1, 3)), c=float(r.choice([1.0, 1.33, 1.5, 2.4]))))
        elif k == 2:
            mats.append(dict(kind=abi.MAT_METAL, a=tuple(r.uniform(0.1, 2, 3)), b=tuple(r.uniform(1, 4, 3)), c=float(r.choice([0.0, 0.001, 0.05, 0.5, 1.0])), remap=bool(r.integers(0, 2))))
        elif k == 3:
            mats.append(dict(kind=abi.MAT_GLOSSY, a=tuple(r.uniform(0, 1, 3)), c=float(r.choice([0.0, 0.01, 0.3, 1.0])), remap=bool(r.integers(0, 2))))
        else:
            mats.append(dict(kind=abi.MAT_MATTE, a=(0.0, 0.0, 0.0), c=0.0))  # black: no lobes
    textures = []
    for m in mats:  # image textures on some matte materials (point-sampled, repeat, uv far outside [0, 1])
        if m["kind"] == abi.MAT_MATTE and r.random() < 0.4:
            th, tw = int(r.integers(1, 9)), int(r.integers(1, 9))
            t = r.uniform(0, 1, (th, tw, 3)).astype(F)
            t[r.random((th, tw)) < 0.2] = 0  # black texels: no lobe there (matte.rs:31)
            m["tex"] = len(textures)
            textures.append(t)
    lights = []
    for _ in range(int(r.choice([0, 1, 2, 3, 3, 4, 6, 9]))):  # above 3 lights k_shade stages shadow rays light by light
        k = int(r.integers(0, 4))
        pos = r.uniform(-2, 2, 3).astype(F)
        if k == 0:
            l2w, _ = _translation(pos)
            lights.append(dict(kind="point", l2w=l2w, I=tuple(r.uniform(0, 20, 3))))
        elif k == 1:
            lights.append(dict(kind="distant", w=tuple(r.normal(size=3)), L=tuple(r.uniform(0, 2, 3))))
        elif k == 2:
            l2w, inv = _translation(pos)
            lights.append(dict(kind="spot", l2w=l2w, l2w_inv=inv, I=tuple(r.uniform(0, 30, 3)), total_width=float(r.uniform(10, 80)), falloff_start=float(r.uniform(1, 10))))
        else:
            l2w, inv = _translation(pos)
            lights.append(dict(kind="rect", l2w=l2w, l2w_inv=inv, L=tuple(r.uniform(0, 10, 3)), size=(float(r.uniform(0.1, 1)), float(r.uniform(0.1, 1)))))
    rect_ids = [i for i, l in enumerate(lights) if l["kind"] == "rect"]
    for mi in range(n_mesh):
        nv = int(r.integers(3, 40))
        p = r.uniform(-1.5, 1.5, (nv, 3)).astype(F)
        style = int(r.integers(0, 5))
        if style == 1:  # axis-aligned slab: many vertices share a coordinate, some exactly +-0
            p[:, int(r.integers(0, 3))] = F(r.choice([0.0, -0.0, 0.5, -1.0]))
        elif style == 2:  # snapped to a coarse grid: coincident vertices, degenerate and duplicated triangles
            p = (np.round(p * 2) / 2).astype(F)
        nt = int(r.integers(1, 60))
        ii = r.integers(0, nv, (nt, 3)).astype(np.uint32)
        if style == 3:
            ii[: nt // 2] = ii[nt // 2 : nt // 2 * 2]  # duplicated triangles: equal t, later candidate wins
        hn, hu = bool(r.integers(0, 2)), bool(r.integers(0, 2))
        pts.append(p)
        nn = r.normal(size=(nv, 3)).astype(F)
        nn /= np.maximum(np.linalg.norm(nn, axis=1, keepdims=True), 1e-3).astype(F)
        normals.append(nn.astype(F))
        uvs.append(r.uniform(-2, 2, (nv, 2)).astype(F))
        idx.append(ii + np.uint32(base))
        tmesh += [mi] * nt
        tmat += list(r.integers(0, n_mat, nt))
        tal += [int(r.choice(rect_ids)) if (rect_ids and r.random() < 0.1) else -1 for _ in range(nt)]
        meshes.append((hn, hu, bool(r.integers(0, 2))))
        base += nv
    if r.random() < 0.25:  # a larger mesh: deeper tree, multi-level LDS top, long traversal stacks
        g = int(r.integers(8, 60))
        gx, gy = np.meshgrid(np.arange(g + 1), np.arange(g + 1), indexing="xy")
        p = np.stack([gx.ravel() / g * 3 - 1.5, r.normal(scale=float(r.choice([0.0, 0.02, 0.3])), size=(g + 1) ** 2), gy.ravel() / g * 3 - 1.5], axis=1).astype(F)
        p = p[:, r.permutation(3)]
        a = (gy[:-1, :-1] * (g + 1) + gx[:-1, :-1]).ravel()
        ii = np.concatenate([np.stack([a, a + 1, a + g + 2], axis=1), np.stack([a, a + g + 2, a + g + 1], axis=1)]).astype(np.uint32)
        nv, nt = len(p), len(ii)
        pts.append(p)
        nn = r.normal(size=(nv, 3)).astype(F)
        nn /= np.maximum(np.linalg.norm(nn, axis=1, keepdims=True), 1e-3).astype(F)
        normals.append(nn.astype(F))
        uvs.append(r.uniform(-2, 2, (nv, 2)).astype(F))
        idx.append(ii + np.uint32(base))
        tmesh += [len(meshes)] * nt
        tmat += list(r.integers(0, n_mat, nt))
        tal += [-1] * nt
        meshes.append((bool(r.integers(0, 2)), bool(r.integers(0, 2)), bool(r.integers(0, 2))))
        base += nv
    spheres = []
    for _ in range(int(r.integers(0, 3))):
        o2w, w2o = _translation(r.uniform(-1, 1, 3).astype(F))
        spheres.append(dict(o2w=o2w, w2o=w2o, radius=float(r.uniform(0.1, 0.6)), material=int(r.integers(0, n_mat))))
    cam_pos = tuple(float(v) for v in r.choice([-3.0, 0.0, 3.0, 2.5], 3)) if r.random() < 0.5 else tuple(r.uniform(-3, 3, 3))
    if np.allclose(cam_pos, 0):
        cam_pos = (0.0, 0.0, 3.0)
    up = (0, 1, 0) if abs(cam_pos[0]) + abs(cam_pos[2]) > 1e-3 else (0, 0, 1)
    return scenes.SceneData(
        points=np.concatenate(pts), normals=np.concatenate(normals), uvs=np.concatenate(uvs), indices=np.concatenate(idx),
        tri_mesh=np.asarray(tmesh, dtype=np.uint32), tri_material=np.asarray(tmat, dtype=np.int32), tri_area_light=np.asarray(tal, dtype=np.int32),
        meshes=meshes, materials=mats, lights=lights, spheres=spheres, background=tuple(r.choice([0.0, 0.1, 1.0], 3)),
        split_method=int(r.integers(0, 3)), max_shapes_in_node=int(r.choice([1, 1, 2, 4, 255])),
        camera=dict(position=cam_pos, target=(0.0, 0.0, 0.0), up=up, fov_axis=int(r.integers(0, 2)), fov_degrees=float(r.uniform(20, 100))),
        textures=textures, name=f"fuzz-{seed}")


# context options cycled by seed: both node layouts, with and without the LDS tree top, packets on deeper bounces
VARIANTS = [{}, {"wide_bvh": 0}, {"wide_bvh": 1}, {"wide_bvh": 0, "top_nodes": 0}, {"wide_bvh": 0, "packet_bounces": 3, "packet_shadow_bounces": 3},
            {"wide_bvh": 0, "overlap_shadow": 0, "shade_reorder": 0}, {"wide_bvh": 0, "batch_paths": 1500, "streams": 2}]
_variant_ctx = {}


def variant_context(seed):
    k = seed % len(VARIANTS)
    if k not in _variant_ctx:
        _variant_ctx[k] = yk.Context(0, **VARIANTS[k])
    return _variant_ctx[k]


def check_seed(ctx, oracle, seed, res=(48, 32)):
    """-> list of (integrator name, mismatching values); ctx = None: a context with the seed's option variant"""
    ctx = ctx or variant_context(seed)
    r = np.random.default_rng(seed ^ 0x5EED)
    sd = random_scene(seed)
    fs = yk.FilmSettings(res=res, tile_dim=16)
    cam = yk.Camera(sd.camera, fs)
    tiles = yk.film_tiles(fs)
    sc = yk.Scene(ctx, sd)
    osc = oracle.OracleScene(sd)
    smp = yk.SamplerType.Uniform(int(r.integers(1, 5)), int(r.integers(1, 2**62))) if r.random() < 0.5 else yk.SamplerType.Stratified((int(r.integers(1, 4)), int(r.integers(1, 4))), bool(r.integers(0, 2)), int(r.integers(1, 2**62)))
    clamp = None if r.random() < 0.6 else float(r.uniform(0.1, 5))
    integs = {"path": yk.IntegratorType.Path(yk.PathParams(max_depth=int(r.integers(0, 9)), indirect_clamp=clamp)), "whitted": yk.IntegratorType.Whitted(int(r.integers(0, 7))),
              "geometry_normals": yk.IntegratorType.GeometryNormals, "shading_normals": yk.IntegratorType.ShadingNormals, "bvh": yk.IntegratorType.BVHIntersections}
    bad = []
    for name, integ in integs.items():
        got, st = yk.IntegratorType.instantiate(ctx, integ).render_tiles(sc, cam, smp, tiles)
        want, rays = osc.render_tiles(cam.matrices, smp, integ, tiles, n_threads=8)
        gb, wb = got.view(np.uint32), want.view(np.uint32)
        same = (gb == wb) | (np.isnan(got) & np.isnan(want))  # a NaN is a NaN (payload bits are not defined by the reference)
        if not same.all() or st.rays != rays:
            bad.append((name, int((~same).sum()), st.rays, rays))
    # the accumulating film: passes s0 .. s0 + 2 in one submission against the oracle's single passes
    spp = yk.samples_per_pixel(smp)
    n_passes = min(3, spp)
    first = r.integers(0, spp - n_passes + 1, len(tiles)).astype(np.uint16)  # sample + passes <= spp (render_manager.rs:135-143)
    it = yk.IntegratorType.instantiate(ctx, integs["path"])
    got, st = it.render_tiles_accumulating(sc, cam, smp, tiles, first, n_passes=n_passes)
    got = got.reshape(n_passes, -1, 3)
    for k in range(n_passes):
        want, _ = osc.render_tiles_accumulating(cam.matrices, smp, integs["path"], tiles, first + k)
        same = (got[k].view(np.uint32) == want.view(np.uint32)) | (np.isnan(got[k]) & np.isnan(want))
        if not same.all():
            bad.append((f"accumulating pass {k}", int((~same).sum()), 0, 0))
    sc.close()
    return bad


if __name__ == "__main__":
    from oracle import binding as oracle

    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    failures = 0
    verbose = len(sys.argv) > 3
    for seed in range(first, first + count):
        if verbose:
            print(f"seed {seed} (variant {VARIANTS[seed % len(VARIANTS)]}) ...", flush=True)
        try:
            bad = check_seed(None, oracle, seed)
        except yk.YukiError as e:
            print(f"seed {seed}: rejected by the library: {e}", flush=True)
            continue
        if bad:
            failures += 1
            print(f"seed {seed}: MISMATCH {bad}", flush=True)
        elif seed % 25 == 0:
            print(f"seed {seed}: ok", flush=True)
    print(f"{count} seeds, {failures} with mismatches")
