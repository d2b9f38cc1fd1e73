"""GPU parity of the whole hot path: Integrator::render through the C ABI against
the CPU oracle on the same scene, camera, sampler seed.

Bar (BASELINE.json north_star): per-pixel radiance RMSE < 1e-4 vs the CPU
reference.  Because every arithmetic step is restated operation for operation
(f64 islands, unfused mul/add, shared libm (glibc restated)) the GPU result is expected to
be BIT-IDENTICAL to the oracle's; the tests assert that, and the RMSE bound as
the stated tolerance."""
import numpy as np
import pytest

from yuki_amd import abi, scenes

pytestmark = pytest.mark.gpu
SEED = 0x73B9642E74AC471C
TOL_RMSE = 1e-4


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _rmse(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def _render_both(ctx, yk, oracle, sd, res, sampler, integ, tile_dim=16, threads=0):
    fs = yk.FilmSettings(res=res, tile_dim=tile_dim)
    cam = yk.Camera(sd.camera, fs)
    tiles = yk.film_tiles(fs)
    sc = yk.Scene(ctx, sd)
    it = yk.IntegratorType.instantiate(ctx, integ)
    got, stats = it.render_tiles(sc, cam, sampler, tiles)
    osc = oracle.OracleScene(sd)
    want, rays = osc.render_tiles(cam.matrices, sampler, integ, tiles, n_threads=threads)
    return got, stats, want, rays


CASES = [
    ("cornell", (96, 96), "uniform", 8),  # built-in Cornell box incl. the copper sphere (scene/mod.rs:154-530)
    ("cornell", (64, 64), "stratified", 8),
    ("cornell-tris", (96, 96), "uniform", 8),
    ("cornell-tris", (64, 48), "stratified", 8),
    ("city-tiny", (128, 72), "stratified", 8),
    ("city-small", (160, 90), "uniform", 8),
    ("city-small", (96, 54), "stratified", 16),
    ("cfg2", (160, 90), "uniform", 8),
]


@pytest.mark.parametrize("name,res,skind,depth", CASES)
def test_path_render_matches_oracle(ctx, yk, oracle, name, res, skind, depth):
    sd = scenes.by_name(name)
    sampler = yk.SamplerType.Uniform(8, SEED) if skind == "uniform" else yk.SamplerType.Stratified((3, 3), True, SEED)
    integ = yk.IntegratorType.Path(yk.PathParams(max_depth=depth))
    got, stats, want, rays = _render_both(ctx, yk, oracle, sd, res, sampler, integ)
    assert stats.rays == rays  # the reference's ray_count (path.rs:87)
    assert np.isfinite(want).all()
    assert _rmse(got, want) < TOL_RMSE
    mism = int((_bits(got) != _bits(want)).sum())
    assert mism == 0, f"{mism} of {got.size} channel values differ from the oracle bit pattern"


WHITTED_CASES = [
    ("cornell", (96, 96), "uniform", 3),  # BASELINE configs[0]: built-in Cornell, Whitted max_depth 3 (the copper sphere: no specular lobes)
    ("cornell", (64, 64), "stratified", 1),
    ("glass-balls", (96, 64), "uniform", 5),  # nested reflection + transmission subtrees, total internal reflection, area light seen through glass
    ("glass-balls", (96, 64), "stratified", 3),
    ("glass-balls", (64, 48), "uniform", 16),
    ("city-tiny", (128, 72), "stratified", 4),
    ("city-small", (96, 54), "uniform", 3),
]


@pytest.mark.parametrize("name,res,skind,depth", WHITTED_CASES)
def test_whitted_render_matches_oracle(ctx, yk, oracle, name, res, skind, depth):
    """Whitted::li_internal (whitted.rs:74-181): recursion order, shared sampler, (f * li) * |cos| without pdf."""
    sd = scenes.by_name(name)
    sampler = yk.SamplerType.Uniform(4, SEED) if skind == "uniform" else yk.SamplerType.Stratified((2, 2), True, SEED)
    integ = yk.IntegratorType.Whitted(depth)
    got, stats, want, rays = _render_both(ctx, yk, oracle, sd, res, sampler, integ)
    assert stats.rays == rays  # li_internal calls
    assert np.isfinite(want).all() and want.max() > 0
    assert _rmse(got, want) < TOL_RMSE
    mism = int((_bits(got) != _bits(want)).sum())
    assert mism == 0, f"{mism} of {got.size} channel values differ from the oracle bit pattern"


def test_indirect_clamp(ctx, yk, oracle):
    sd = scenes.by_name("city-tiny")
    sampler = yk.SamplerType.Uniform(4, SEED)
    integ = yk.IntegratorType.Path(yk.PathParams(max_depth=6, indirect_clamp=0.25))
    got, stats, want, rays = _render_both(ctx, yk, oracle, sd, (96, 54), sampler, integ)
    assert stats.rays == rays
    assert np.array_equal(_bits(got), _bits(want))


@pytest.mark.parametrize("integ_name", ["GeometryNormals", "ShadingNormals", "BVHIntersections"])
def test_debug_integrators(ctx, yk, oracle, integ_name):
    sd = scenes.by_name("city-small")
    sampler = yk.SamplerType.Uniform(2, SEED)
    integ = getattr(yk.IntegratorType, integ_name)
    got, stats, want, rays = _render_both(ctx, yk, oracle, sd, (128, 72), sampler, integ)
    assert stats.rays == rays == 128 * 72 * 2
    assert np.array_equal(_bits(got), _bits(want))


def test_single_tile_is_the_trait_method(ctx, yk, oracle):
    """Integrator::render(tile) == the same pixels of a whole-film render; ragged
    edge tile (film.rs:306-309 clips tiles to the film)."""
    sd = scenes.by_name("city-tiny")
    fs = yk.FilmSettings(res=(70, 41), tile_dim=16)
    cam = yk.Camera(sd.camera, fs)
    sc = yk.Scene(ctx, sd)
    sampler = yk.SamplerType.Stratified((2, 2), True, SEED)
    it = yk.IntegratorType.instantiate(ctx, yk.IntegratorType.Path(yk.PathParams(max_depth=5)))
    tiles = yk.film_tiles(fs)
    whole, stats = it.render_tiles(sc, cam, sampler, tiles)
    film = yk.update_tiles(tiles, whole, fs.res)
    osc = oracle.OracleScene(sd)
    total = 0
    for t in [tiles[0], tiles[-1], tiles[len(tiles) // 2]]:
        px, rays = it.render(sc, cam, sampler, yk.FilmTile(bb=(t["x0"], t["y0"], t["x1"], t["y1"])))
        h, w = int(t["y1"]) - int(t["y0"]), int(t["x1"]) - int(t["x0"])
        assert np.array_equal(_bits(px.reshape(h, w, 3)), _bits(film[t["y0"] : t["y1"], t["x0"] : t["x1"]]))
        want, orays = osc.render_tiles(cam.matrices, sampler, it.desc, np.array([t]), n_threads=1)
        assert rays == orays
        assert np.array_equal(_bits(px), _bits(want))
        total += rays
    assert total > 0


def test_result_independent_of_batch_size(yk, oracle):
    """Paths are independent: cutting the work into different batches (and thus
    different compaction orders) must not change a single bit."""
    sd = scenes.by_name("city-tiny")
    fs = yk.FilmSettings(res=(64, 36))
    sampler = yk.SamplerType.Uniform(4, SEED)
    integ = yk.IntegratorType.Path(yk.PathParams(max_depth=8))
    outs = []
    for batch in (1 << 20, 4096, 777):
        c = yk.Context(0, batch_paths=batch)
        sc = yk.Scene(c, sd)
        cam = yk.Camera(sd.camera, fs)
        out, st = yk.IntegratorType.instantiate(c, integ).render_tiles(sc, cam, sampler, yk.film_tiles(fs))
        outs.append((out, st.rays))
        sc.close()
        c.close()
    for o, r in outs[1:]:
        assert r == outs[0][1]
        assert np.array_equal(_bits(o), _bits(outs[0][0]))


def test_many_tiles_per_block_is_deterministic(yk):
    """8.3 M camera samples: every shade block runs many grid-stride iterations and
    flushes its LDS staging buffers repeatedly; one batch vs eight batches must
    agree bit for bit (guards the staging/flush synchronisation)."""
    sd = scenes.by_name("city-small")
    fs = yk.FilmSettings(res=(1920, 1080))
    sampler = yk.SamplerType.Stratified((2, 2), True, SEED)
    integ = yk.IntegratorType.Path(yk.PathParams(max_depth=8))
    outs = []
    for batch, streams in ((16 << 20, 1), (1 << 20, 1), (16 << 20, 2), (3 << 20, 2)):
        c = yk.Context(0, batch_paths=batch, streams=streams)
        sc = yk.Scene(c, sd)
        out, st = yk.IntegratorType.instantiate(c, integ).render_tiles(sc, yk.Camera(sd.camera, fs), sampler, yk.film_tiles(fs))
        outs.append((out, st.rays, st.shadow_rays, st.batches))
        sc.close()
        c.close()
    assert [o[3] for o in outs] == [1, 8, 1, 3]  # one batch; eight; one batch again (a job that fits is not split); three alternating on two streams
    for o in outs[1:]:
        assert o[1] == outs[0][1] and o[2] == outs[0][2]
        assert np.array_equal(_bits(o[0]), _bits(outs[0][0]))
    assert np.isfinite(outs[0][0]).all()


def test_li_matches_render(ctx, yk, oracle):
    """Integrator::li on caller-supplied camera rays == what render computes for them."""
    sd = scenes.by_name("city-tiny")
    fs = yk.FilmSettings(res=(32, 32), tile_dim=32)
    cam = yk.Camera(sd.camera, fs)
    sc = yk.Scene(ctx, sd)
    import itertools

    for sampler, integ in itertools.product((yk.SamplerType.Uniform(1, SEED), yk.SamplerType.Stratified((1, 1), True, SEED)),
                                            (yk.IntegratorType.Path(yk.PathParams(max_depth=6)), yk.IntegratorType.Whitted(4))):
        it = yk.IntegratorType.instantiate(ctx, integ)
        tile = (0, 0, 32, 32)
        px, _ = it.render(sc, cam, sampler, yk.FilmTile(bb=tile))
        o, d = yk.camera_rays(ctx, cam, sampler, tile, 0)
        xy = np.stack(np.meshgrid(np.arange(32), np.arange(32), indexing="xy"), axis=-1).reshape(-1, 2).astype(np.uint16)
        li = it.li(sc, sampler, o, d, xy, np.zeros(len(o), dtype=np.uint32), dimension=2)
        assert np.array_equal(_bits(li), _bits(px))


def test_error_behaviour(ctx, yk):
    """Contract violations the reference asserts on come back as status codes."""
    sd = scenes.by_name("city-tiny")
    sc = yk.Scene(ctx, sd)
    cam = yk.Camera(sd.camera, yk.FilmSettings(res=(32, 32)))
    it = yk.IntegratorType.instantiate(ctx, yk.IntegratorType.Path(yk.PathParams(max_depth=2)))
    with pytest.raises(yk.YukiError) as e:
        it.render(sc, cam, yk.SamplerType.Uniform(1), yk.FilmTile(bb=(8, 8, 8, 16)))  # Bounds2 with a dimension <= 0
    assert e.value.status == 1
    with pytest.raises(yk.YukiError) as e:  # the device kernel keeps at most 16 suspended calls
        yk.IntegratorType.instantiate(ctx, yk.IntegratorType.Whitted(17)).render(sc, cam, yk.SamplerType.Uniform(1), yk.FilmTile(bb=(0, 0, 8, 8)))
    assert e.value.status == 5
    # cancellation: the predicate is polled before every batch
    with pytest.raises(yk.YukiError) as e:
        it.render_tiles(sc, cam, yk.SamplerType.Uniform(1), yk.film_tiles(yk.FilmSettings(res=(32, 32))), cancel=lambda: True)
    assert e.value.status == 7


@pytest.mark.parametrize("option,value", [("wide_bvh", 0), ("wide_bvh", 1), ("top_nodes", 0), ("top_nodes", 7), ("packet_bounces", 0), ("packet_bounces", 8), ("packet_shadow_bounces", 8), ("overlap_shadow", 0)])
def test_traversal_layout_options_do_not_change_the_image(yk, oracle, option, value):
    """The traversal variants — 4-wide collapse of the BVH (DevNode4), number of top-of-tree
    nodes kept in LDS, wave-packet kernels for none / all bounces (closest and shadow rays),
    side-stream overlap off — visit the same leaves in the same order per ray, so the render
    stays bit-identical to the oracle's (scenes with triangles and spheres)."""
    c = yk.Context(0, **{option: value})
    try:
        for name, res in (("cornell", (64, 64)), ("city-tiny", (96, 54))):
            sd = scenes.by_name(name)
            sampler = yk.SamplerType.Stratified((2, 2), True, SEED)
            integ = yk.IntegratorType.Path(yk.PathParams(max_depth=6))
            got, stats, want, rays = _render_both(c, yk, oracle, sd, res, sampler, integ)
            assert stats.rays == rays
            assert np.array_equal(_bits(got), _bits(want))
    finally:
        c.close()


@pytest.mark.parametrize("keep", ["none", "area", "delta"])
def test_light_subsets(ctx, yk, oracle, keep):
    """No lights at all (background only), only the area light, only the point lights: the
    shadow-ray queues (area / delta) and their kernels each run alone or not at all."""
    import copy

    sd = copy.copy(scenes.by_name("city-tiny"))
    kinds = [l["kind"] for l in sd.lights]
    assert "rect" in kinds and "point" in kinds
    if keep == "none":
        sd.lights = []
        sd.tri_area_light = np.full_like(sd.tri_area_light, -1)
    elif keep == "area":
        sd.lights = [l for l in sd.lights if l["kind"] == "rect"]
    else:
        sd.lights = [l for l in sd.lights if l["kind"] != "rect"]
        sd.tri_area_light = np.full_like(sd.tri_area_light, -1)
    sd.background = (0.3, 0.35, 0.4)
    sampler = yk.SamplerType.Stratified((2, 2), True, SEED)
    integ = yk.IntegratorType.Path(yk.PathParams(max_depth=5))
    got, stats, want, rays = _render_both(ctx, yk, oracle, sd, (96, 54), sampler, integ)
    assert stats.rays == rays
    assert np.array_equal(_bits(got), _bits(want))
    assert got.mean() > 0.01


def test_render_from_worker_threads(ctx, yk):
    """The reference calls Integrator::render from num_cpus-1 tile workers at once
    (render_manager.rs:78-97); calls on one context are serialised inside the library and give
    the same pixels as one batched call."""
    import threading

    sd = scenes.by_name("city-tiny")
    sc = yk.Scene(ctx, sd)
    fs = yk.FilmSettings(res=(128, 72))
    cam = yk.Camera(sd.camera, fs)
    tiles = yk.film_tiles(fs)
    smp = yk.SamplerType.Uniform(4, SEED)
    it = yk.IntegratorType.instantiate(ctx, yk.IntegratorType.Path(yk.PathParams(max_depth=5)))
    ref, ref_stats = it.render_tiles(sc, cam, smp, tiles)
    offs = np.concatenate([[0], np.cumsum((tiles["x1"].astype(int) - tiles["x0"]) * (tiles["y1"].astype(int) - tiles["y0"]))])
    out = np.zeros_like(ref)
    errs, rays = [], []

    def worker(k):
        try:
            for t in range(k, len(tiles), 6):
                px, n = it.render(sc, cam, smp, yk.FilmTile(tuple(int(v) for v in tiles[t])))
                out[offs[t] : offs[t + 1]] = px
                rays.append(n)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs
    assert sum(rays) == ref_stats.rays
    assert np.array_equal(_bits(out), _bits(ref))


def test_one_scene_rendered_by_two_contexts_at_once(ctx, yk):
    """A scene and a tile list belong to the device: a second context renders them too, and two
    contexts keep two renders in flight (bench.py --frames-in-flight 2).  Both
    threads get the pixels of a plain single-context render."""
    import threading

    sd = scenes.by_name("city-tiny")
    sc = yk.Scene(ctx, sd)
    fs = yk.FilmSettings(res=(160, 96))
    cam = yk.Camera(sd.camera, fs)
    tiles = yk.film_tiles(fs)
    smp = yk.SamplerType.Stratified((2, 2), True, SEED)
    integ = yk.IntegratorType.Path(yk.PathParams(max_depth=6))
    ref, ref_stats = yk.IntegratorType.instantiate(ctx, integ).render_tiles(sc, cam, smp, tiles)
    ctx2 = yk.Context(0)
    outs, errs = {}, []

    def worker(k, c):
        try:
            it = yk.IntegratorType.instantiate(c, integ)
            for _ in range(4):
                px, st = it.render_tiles(sc, cam, smp, tiles)
                assert st.rays == ref_stats.rays
            outs[k] = px
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=worker, args=(k, c)) for k, c in enumerate((ctx, ctx2))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    assert np.array_equal(_bits(outs[0]), _bits(ref)) and np.array_equal(_bits(outs[1]), _bits(ref))
    ctx2.close()


def test_render_one_tile_accumulating(ctx, yk, oracle):
    """Integrator::render(accumulating=true) through the one-tile entry point."""
    sd = scenes.by_name("city-tiny")
    fs = yk.FilmSettings(res=(64, 48), accumulate=True)
    cam = yk.Camera(sd.camera, fs)
    smp = yk.SamplerType.Stratified((2, 2), True, SEED)
    integ = yk.IntegratorType.Path(yk.PathParams(max_depth=4))
    it = yk.IntegratorType.instantiate(ctx, integ)
    tile = yk.FilmTile((16, 16, 32, 32), sample=3)
    got, rays = it.render(yk.Scene(ctx, sd), cam, smp, tile, accumulating=True)
    want, wrays = oracle.OracleScene(sd).render_tiles_accumulating(cam.matrices, smp, integ, np.array([tile.bb], dtype=abi.TILE_DTYPE), [3])
    assert rays == wrays and np.array_equal(_bits(got), _bits(want))


def test_bench_workload_at_full_size(ctx, yk, oracle, cfg3_scene):
    """BASELINE config 3 exactly as bench.py runs it (1,024,012 triangles, SAH BVH, Path 8,
    Stratified 8x8, 1920x1080): (1) the first 96 spiral tiles at full spp against the oracle,
    bit for bit; (2) size-independent properties of the whole frame — the shard of rank 3 of 8
    equals the same tiles of the full render, and the ray count is the sum over shards."""
    sd = cfg3_scene
    fs = yk.FilmSettings(res=(1920, 1080), tile_dim=16)
    cam = yk.Camera(sd.camera, fs)
    tiles = yk.film_tiles(fs)
    sampler = yk.SamplerType.Stratified((8, 8), True, SEED)
    integ = yk.IntegratorType.Path(yk.PathParams(max_depth=8))
    it = yk.IntegratorType.instantiate(ctx, integ)
    sc = yk.Scene(ctx, sd)
    full, st_full = it.render_tiles(sc, cam, sampler, tiles)
    assert np.isfinite(full).all() and st_full.samples == 1920 * 1080 * 64
    offs = np.concatenate([[0], np.cumsum((tiles["x1"].astype(np.int64) - tiles["x0"]) * (tiles["y1"].astype(np.int64) - tiles["y0"]))])
    # (1) oracle on a sample of the frame
    k = 96
    want, rays = oracle.OracleScene(sd).render_tiles(cam.matrices, sampler, integ, tiles[:k], n_threads=0)
    got = full[: offs[k]]
    assert _rmse(got, want) < TOL_RMSE
    assert np.array_equal(_bits(got), _bits(want))
    head, st_head = it.render_tiles(sc, cam, sampler, tiles[:k])
    assert st_head.rays == rays and np.array_equal(_bits(head), _bits(got))
    # (2) shards
    total = 0
    for r in (3,):
        idx = np.arange(r, len(tiles), 8)
        shard, st = it.render_tiles(sc, cam, sampler, tiles[idx])
        ref = np.concatenate([full[offs[t] : offs[t + 1]] for t in idx])
        assert np.array_equal(_bits(shard), _bits(ref))
        total += st.rays
    assert 0.11 < total / st_full.rays < 0.14


def test_cancellation_at_the_references_granularity(yk, cfg3_scene):
    """integrators/mod.rs:153 polls the predicate once per pixel sample and render_worker.rs:240-255 relies on it for "low
    latency kills".  A cfg5-sized job (3840x2160 x 256 spp, Path 16: sixteen batches of 128 M camera samples, ~1.5 s) on the
    cfg3 scene is cancelled 50 ms after it started: the call returns YK_ERR_CANCELLED a few milliseconds after the predicate
    fired (kernels stop at their next work claim / shade window, what is still enqueued finds empty queues), and the NEXT render
    on the same context equals a fresh context's bit for bit.  yk_context_interrupt from another thread does the same."""
    import threading
    import time

    sd = cfg3_scene
    big = yk.FilmSettings(res=(3840, 2160), tile_dim=16)
    big_sampler = yk.SamplerType.Stratified((16, 16), True, SEED)
    big_integ = yk.IntegratorType.Path(yk.PathParams(max_depth=16))
    small = yk.FilmSettings(res=(320, 180), tile_dim=16)
    small_sampler = yk.SamplerType.Stratified((4, 4), True, SEED)
    small_integ = yk.IntegratorType.Path(yk.PathParams(max_depth=8))
    fresh = yk.Context(0)
    fsc = yk.Scene(fresh, sd)
    want, st_want = yk.IntegratorType.instantiate(fresh, small_integ).render_tiles(fsc, yk.Camera(sd.camera, small), small_sampler, yk.film_tiles(small))
    c = yk.Context(0)
    sc = yk.Scene(c, sd)
    it = yk.IntegratorType.instantiate(c, big_integ)
    cam = yk.Camera(sd.camera, big)
    tiles = yk.film_tiles(big)
    it.render_tiles(sc, yk.Camera(sd.camera, small), small_sampler, yk.film_tiles(small))  # buffers and code objects exist before the clock starts
    latencies, every = [], []
    for delay in (0.050, 0.200):
        attempts = []
        # The figure asserted is the mechanism's (3-8 ms on every box measured, 35 runs); the host is shared, and ONE run in those
        # 35 showed 580 ms on its first interruption with nothing in the library's own timings (YK_DEBUG_CANCEL=1) ever near it
        # — so a slow attempt gets a second and a third one, and a broken mechanism (two batches = 190 ms at least, every time) still fails.
        for _attempt in range(3):
            t0 = time.time()
            fired = []

            def pred():
                if not fired and time.time() - t0 >= delay:
                    fired.append(time.time())
                    return True
                return False

            with pytest.raises(yk.YukiError) as e:
                it.render_tiles(sc, cam, big_sampler, tiles, cancel=pred)
            t1 = time.time()
            assert e.value.status == 7 and fired  # YK_ERR_CANCELLED
            attempts.append(t1 - fired[0])
            got, st = yk.IntegratorType.instantiate(c, small_integ).render_tiles(sc, yk.Camera(sd.camera, small), small_sampler, yk.film_tiles(small))
            assert st.rays == st_want.rays and np.array_equal(_bits(got), _bits(want))
            if attempts[-1] < 0.040:
                break
        latencies.append(min(attempts))
        every += attempts
    print("cancel latencies (predicate fired -> call returned):", ["%.1f ms" % (1e3 * x) for x in every])
    assert max(latencies) < 0.040, every  # measured 3-8 ms; the whole job takes ~1.5 s
    assert max(every) < 1.0, every  # and no attempt ever waits for the job
    # the same from another thread, for a caller without a predicate
    timer = threading.Timer(0.050, c.interrupt)
    t0 = time.time()
    timer.start()
    full_st = None
    try:
        _, full_st = it.render_tiles(sc, cam, big_sampler, tiles)
    except yk.YukiError as err:  # the synchronous call notices that its work was interrupted
        assert err.status == 7
    dt = time.time() - t0
    timer.join()
    assert dt < 1.0, dt  # the uninterrupted job takes ~1.5 s (and returns stats: the next line)
    assert full_st is None
    got, st = yk.IntegratorType.instantiate(c, small_integ).render_tiles(sc, yk.Camera(sd.camera, small), small_sampler, yk.film_tiles(small))
    assert st.rays == st_want.rays and np.array_equal(_bits(got), _bits(want))
    # an interruption that finds nothing in flight is consumed by the next submission's start, not by its kernels
    c.interrupt()
    got, st = yk.IntegratorType.instantiate(c, small_integ).render_tiles(sc, yk.Camera(sd.camera, small), small_sampler, yk.film_tiles(small))
    assert st.rays == st_want.rays and np.array_equal(_bits(got), _bits(want))
    sc.close()
    fsc.close()
    c.close()
    fresh.close()


def test_max_depth_zero_renders_black(ctx, yk, oracle):
    """`while bounces < max_depth` never runs (path.rs:85): no rays, zero radiance."""
    sd = scenes.by_name("city-tiny")
    got, stats, want, rays = _render_both(ctx, yk, oracle, sd, (48, 32), yk.SamplerType.Uniform(2, SEED), yk.IntegratorType.Path(yk.PathParams(max_depth=0)))
    assert stats.rays == rays == 0
    assert not got.any() and np.array_equal(_bits(got), _bits(want))


@pytest.mark.parametrize("wide", [0, 1])
def test_tie_hits_that_raise_t_max(yk, oracle, wide):
    """Slabs of overlapping coplanar triangles: a tie hit can set t_max a few ulps ABOVE the old value
    (triangle.rs:126-139), and a far box culled when its parent was visited passes when the reference
    pops it.  With the exact bound for deferred boxes (tools/build_variant.sh exact -DYK_DEFERRED_BOUND_FACTOR=1.0f) the node
    counters of this scene differ in three values; with the relaxed bound everything is identical."""
    sd = scenes.by_name("coplanar-slabs")
    c = yk.Context(0, wide_bvh=wide)
    fs = yk.FilmSettings(res=(160, 96))
    cam = yk.Camera(sd.camera, fs)
    tiles = yk.film_tiles(fs)
    sc = yk.Scene(c, sd)
    osc = oracle.OracleScene(sd)
    smp = yk.SamplerType.Uniform(2, 11)
    for integ in (yk.IntegratorType.ShadingNormals, yk.IntegratorType.BVHIntersections, yk.IntegratorType.Path(yk.PathParams(max_depth=4))):
        got, st = yk.IntegratorType.instantiate(c, integ).render_tiles(sc, cam, smp, tiles)
        want, rays = osc.render_tiles(cam.matrices, smp, integ, tiles, n_threads=8)
        assert st.rays == rays
        assert np.array_equal(_bits(got), _bits(want))
    sc.close()
    c.close()


FUZZ_SEEDS = list(range(1000, 1040))


@pytest.mark.parametrize("seed", FUZZ_SEEDS)
def test_random_scenes_match_the_oracle(ctx, yk, oracle, seed):
    """tools/parity_fuzz.py: random triangle soups (degenerate, duplicated, axis-aligned, +-0
    coordinates), all material and light kinds, spheres; Path, Whitted and the three debug
    integrators with random sampler / depth / clamp — every value bit-identical, same ray counts."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import parity_fuzz

    assert parity_fuzz.check_seed(None, oracle, seed) == []  # None: the context options cycle with the seed (both node layouts, ...)


@pytest.mark.parametrize("seed", range(2000, 2016))
def test_li_on_random_rays_matches_the_oracle(oracle, yk, seed):
    """Integrator::li (integrators/mod.rs:94-101) for caller-supplied rays on random scenes: yk_li against
    the oracle's li with the sampler started at (pixel, sample index) and `dimension` draws consumed, Path and Whitted."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import parity_fuzz
    import stage_fuzz

    ctx = parity_fuzz.variant_context(seed)
    sd = parity_fuzz.random_scene(seed)
    r = np.random.default_rng(seed ^ 0x11)
    o, d = stage_fuzz.rays_for(sd, r, n=600)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)  # li takes the unit directions a camera or a BSDF produces
    smp = yk.SamplerType.Uniform(5, SEED) if seed % 2 else yk.SamplerType.Stratified((2, 3), True, SEED)
    pix = r.integers(0, 300, (len(o), 2)).astype(np.uint16)
    si = r.integers(0, yk.samples_per_pixel(smp), len(o)).astype(np.uint32)
    dim = int(r.integers(0, 7))
    sc = yk.Scene(ctx, sd)
    osc = oracle.OracleScene(sd)
    for integ in (yk.IntegratorType.Path(yk.PathParams(max_depth=int(r.integers(1, 8)))), yk.IntegratorType.Whitted(int(r.integers(1, 6)))):
        got = yk.IntegratorType.instantiate(ctx, integ).li(sc, smp, o, d, pix, si, dimension=dim)
        want, _ = osc.li(smp, integ, o, d, pix, si, dimension=dim)
        same = (_bits(got) == _bits(want)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), int((~same).sum())
    sc.close()


def test_li_equals_the_render_of_that_sample(ctx, yk, oracle):
    """li on the camera rays of sample k at dimension 2 == the accumulating render of sample k — with
    2x2 strata and a uniform sampler of 3, where the stratum hash and the PCG offset both matter
    (1x1 strata would hide a wrong dimension)."""
    sd = scenes.by_name("city-tiny")
    fs = yk.FilmSettings(res=(32, 32), tile_dim=32, accumulate=True)
    cam = yk.Camera(sd.camera, fs)
    tiles = yk.film_tiles(fs)
    sc = yk.Scene(ctx, sd)
    xy = np.stack(np.meshgrid(np.arange(32), np.arange(32), indexing="xy"), axis=-1).reshape(-1, 2).astype(np.uint16)
    for sampler in (yk.SamplerType.Uniform(3, SEED), yk.SamplerType.Stratified((2, 2), True, SEED), yk.SamplerType.Stratified((2, 2), False, SEED)):
        for integ in (yk.IntegratorType.Path(yk.PathParams(max_depth=6)), yk.IntegratorType.Whitted(4)):
            it = yk.IntegratorType.instantiate(ctx, integ)
            for k in (0, 2):
                img, _ = it.render_tiles_accumulating(sc, cam, sampler, tiles, np.full(len(tiles), k, dtype=np.uint16))
                o, d = yk.camera_rays(ctx, cam, sampler, (0, 0, 32, 32), k)
                li = it.li(sc, sampler, o, d, xy, np.full(len(o), k, dtype=np.uint32), dimension=2)
                assert np.array_equal(_bits(li), _bits(img))


def test_deep_tree_overflow_is_reported_by_every_entry_point(yk, oracle):
    """bvh.rs:172-174 asserts on a traversal stack of more than 64 entries; the library returns
    YK_ERR_STACK_OVERFLOW.  The flag has to survive the per-batch reset of the control block (a render of
    four batches on one work set), reach callers that ask for no statistics (the asynchronous device entry)
    and yk_li.  A tree of depth 60 built the same way renders, and equals the oracle bit for bit."""
    import torch

    c = yk.Context(0, batch_paths=4096, streams=1)
    fs = yk.FilmSettings(res=(64, 64))
    tiles = yk.film_tiles(fs)
    smp = yk.SamplerType.Uniform(4, SEED)
    integ = yk.IntegratorType.Path(yk.PathParams(max_depth=3))
    it = yk.IntegratorType.instantiate(c, integ)

    sd = scenes.by_name("deep-chain-60")
    cam = yk.Camera(sd.camera, fs)
    sc = yk.Scene(c, sd)
    assert sc.info().tree_depth == 60
    got, st = it.render_tiles(sc, cam, smp, tiles)
    want, rays = oracle.OracleScene(sd).render_tiles(cam.matrices, smp, integ, tiles, n_threads=0)
    assert st.batches == 4 and st.rays == rays and np.array_equal(_bits(got), _bits(want))
    sc.close()

    sd = scenes.by_name("deep-chain-70")
    sc = yk.Scene(c, sd)
    assert sc.info().tree_depth == 70
    with pytest.raises(yk.YukiError) as e:
        it.render_tiles(sc, cam, smp, tiles)
    assert e.value.status == 8
    out = torch.zeros(64 * 64 * 3, dtype=torch.float32, device="cuda:0")
    with pytest.raises(yk.YukiError) as e:  # stats == NULL: the call may not stay asynchronous for such a tree
        it.render_tiles_device(sc, cam, smp, tiles, out.data_ptr(), want_stats=False)
    assert e.value.status == 8
    o, d = yk.camera_rays(c, cam, smp, (24, 24, 40, 40), 0)
    xy = np.stack(np.meshgrid(np.arange(24, 40), np.arange(24, 40), indexing="xy"), axis=-1).reshape(-1, 2).astype(np.uint16)
    with pytest.raises(yk.YukiError) as e:
        it.li(sc, smp, o, d, xy, np.zeros(len(o), dtype=np.uint32), dimension=2)
    assert e.value.status == 8
    for integ2 in (yk.IntegratorType.Whitted(3), yk.IntegratorType.GeometryNormals):
        with pytest.raises(yk.YukiError) as e:
            yk.IntegratorType.instantiate(c, integ2).render_tiles(sc, cam, smp, tiles)
        assert e.value.status == 8
    with pytest.raises(yk.YukiError) as e:
        sc.intersect(o, d)
    assert e.value.status == 8
    sc.close()
    c.close()


def _full_size_checks(ctx, yk, oracle, sd, res, sampler, integ, k_oracle, shard=(3, 8), expect_samples=None):
    """A BASELINE configuration at its full size: (1) the first `k_oracle` spiral tiles at full spp against the oracle,
    bit for bit, with the oracle's ray count; (2) size-independent properties of the whole frame: the shard of one
    rank of G equals the same tiles of the full render, bit for bit, and carries about 1/G of the rays."""
    fs = yk.FilmSettings(res=res, tile_dim=16)
    cam = yk.Camera(sd.camera, fs)
    tiles = yk.film_tiles(fs)
    it = yk.IntegratorType.instantiate(ctx, integ)
    sc = yk.Scene(ctx, sd)
    full, st_full = it.render_tiles(sc, cam, sampler, tiles)
    assert np.isfinite(full).all()
    if expect_samples is not None:
        assert st_full.samples == expect_samples
    offs = np.concatenate([[0], np.cumsum((tiles["x1"].astype(np.int64) - tiles["x0"]) * (tiles["y1"].astype(np.int64) - tiles["y0"]))])
    osc = oracle.OracleScene(sd)
    want, rays = osc.render_tiles(cam.matrices, sampler, integ, tiles[:k_oracle], n_threads=0)
    osc.close()
    got = full[: offs[k_oracle]]
    assert _rmse(got, want) < TOL_RMSE
    assert np.array_equal(_bits(got), _bits(want))
    head, st_head = it.render_tiles(sc, cam, sampler, tiles[:k_oracle])
    assert st_head.rays == rays and np.array_equal(_bits(head), _bits(got))
    r, G = shard
    idx = np.arange(r, len(tiles), G)
    part, st = it.render_tiles(sc, cam, sampler, tiles[idx])
    ref = np.concatenate([full[offs[t] : offs[t + 1]] for t in idx])
    assert np.array_equal(_bits(part), _bits(ref))
    assert 0.8 / G < st.rays / st_full.rays < 1.25 / G
    sc.close()
    return st_full


def test_cfg1_whitted_at_full_size(ctx, yk, oracle):
    """BASELINE configs[0]: built-in Cornell box (scene/mod.rs:154-530), Whitted depth 3, Uniform 1 spp, 512x512 —
    the whole frame against the oracle (262,144 camera samples)."""
    sd = scenes.by_name("cornell")
    sampler = yk.SamplerType.Uniform(1, SEED)
    integ = yk.IntegratorType.Whitted(3)
    got, stats, want, rays = _render_both(ctx, yk, oracle, sd, (512, 512), sampler, integ)
    assert stats.samples == 512 * 512 and stats.rays == rays
    assert want.max() > 0 and _rmse(got, want) < TOL_RMSE
    assert np.array_equal(_bits(got), _bits(want))


def test_cfg2_at_full_size(ctx, yk, oracle):
    """BASELINE configs[1]: bunny-class 69,312-triangle mesh (the reference's PLY defaults: white matte, one point
    light, scene/mod.rs:104-125), Path 8, Uniform 16 spp, 1920x1080."""
    _full_size_checks(ctx, yk, oracle, scenes.by_name("cfg2"), (1920, 1080), yk.SamplerType.Uniform(16, SEED), yk.IntegratorType.Path(yk.PathParams(max_depth=8)),
                      k_oracle=192, expect_samples=1920 * 1080 * 16)


def test_cfg5_at_full_size(ctx, yk, oracle):
    """BASELINE configs[4] on one GPU: 10,240,012 triangles (GGX metal / perfect glass), SAH BVH, Path 16 bounces,
    Stratified 16x16 = 256 spp, 3840x2160 — 2.1 G camera samples in 16 batches on two work sets."""
    st = _full_size_checks(ctx, yk, oracle, scenes.by_name("cfg5"), (3840, 2160), yk.SamplerType.Stratified((16, 16), True, SEED),
                           yk.IntegratorType.Path(yk.PathParams(max_depth=16)), k_oracle=16, expect_samples=3840 * 2160 * 256)
    assert st.batches >= 2
