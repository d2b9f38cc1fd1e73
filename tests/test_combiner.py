"""yk_combiner: the reference's calling pattern — Integrator::render per 16x16 tile from many worker threads
(render_manager.rs:78-97, render_worker.rs:205-256) — merged into shared submissions.

CPU: the host logic (group commit, lanes, per-caller predicates, re-queueing after an interruption, error fan-out, counts exact
in sum) with stand-ins for the render calls (tests/cpp/combiner_test.cpp).  GPU: real worker threads against one batched call."""
import os
import subprocess
import threading

import numpy as np
import pytest

from yuki_amd import abi, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0x73B9642E74AC471C


def test_combiner_host_logic(tmp_path):
    exe = str(tmp_path / "combiner_test")
    subprocess.check_call(["hipcc", "-x", "hip", "--cuda-host-only", "-O1", "-std=c++17", "-w", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "combiner_test.cpp"), os.path.join(ROOT, "yuki_amd", "csrc", "yk_combiner.cpp"), "-o", exe, "-lpthread"])
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "combiner_test: ok" in out.stdout, out.stdout + out.stderr


def test_combiner_rejects_bad_arguments(yk):
    import ctypes as C

    L = yk.lib()
    h = C.c_void_p()
    assert L.yk_combiner_create(None, 1, 0, 0, C.byref(h)) == 1
    arr = (C.c_void_p * 1)(None)
    assert L.yk_combiner_create(arr, 1, 0, 0, C.byref(h)) == 1
    assert L.yk_combiner_render_tile(None, None, None, None, None, None, -1, None, None, None, None) == 1


def _workers(n_threads, tiles, fn):
    nxt = [0]
    lock = threading.Lock()
    errors = []

    def worker(k):
        try:
            while True:
                with lock:  # the tile queue (render_worker.rs:172-180)
                    t = nxt[0]
                    nxt[0] += 1
                if t >= len(tiles):
                    return
                fn(k, t)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=worker, args=(k,)) for k in range(n_threads)]
    [t.start() for t in th]
    [t.join() for t in th]
    return errors


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [1, 2])
def test_worker_threads_through_the_combiner_render_the_batched_film(yk, lanes):
    """15 worker threads (num_cpus - 1 on a 16-core host) each rendering one tile at a time: every tile bit-identical to
    the one-call film, ray counts exact in sum, and the calls really were merged."""
    sd = scenes.by_name("city-small")
    fs = yk.FilmSettings(res=(320, 180))
    tiles = yk.film_tiles(fs)
    smp = yk.SamplerType.Stratified((2, 2), True, SEED)
    integ = yk.IntegratorType.Path(yk.PathParams(max_depth=5))
    ctxs = [yk.Context(0) for _ in range(lanes)]
    sc = yk.Scene(ctxs[0], sd)
    cam = yk.Camera(sd.camera, fs)
    it = yk.IntegratorType.instantiate(ctxs[0], integ)
    ref, st = it.render_tiles(sc, cam, smp, tiles)
    areas = (tiles["x1"].astype(int) - tiles["x0"]) * (tiles["y1"].astype(int) - tiles["y0"])
    offs = np.concatenate([[0], np.cumsum(areas)])
    comb = yk.Combiner(ctxs, linger_us=200)
    out = np.full_like(ref, -1.0)
    rays = [0] * len(tiles)

    def one(k, t):
        px, s = comb.render(it, sc, cam, smp, yk.FilmTile(tuple(int(v) for v in tiles[t])))
        out[offs[t]:offs[t + 1]] = px
        rays[t] = s.rays

    assert _workers(15, tiles, one) == []
    assert out.tobytes() == ref.tobytes()
    assert sum(rays) == st.rays
    info = comb.info()
    assert info.tiles == len(tiles) and info.lanes == lanes and info.submissions < len(tiles) / 2 and info.largest_submission >= 4
    comb.close()


@pytest.mark.gpu
def test_combiner_accumulating_mode_and_mixed_jobs(yk, oracle):
    """accumulating = true (one sample with FilmTile.sample, raw value) and a different integrator from other threads at the
    same time: calls for different jobs never share a submission, each gets its own job's pixels."""
    sd = scenes.by_name("cornell-tris")
    fs = yk.FilmSettings(res=(96, 64))
    tiles = yk.film_tiles(fs)
    smp = yk.SamplerType.Uniform(4, SEED)
    ctx = yk.Context(0)
    sc = yk.Scene(ctx, sd)
    cam = yk.Camera(sd.camera, fs)
    path = yk.IntegratorType.instantiate(ctx, yk.IntegratorType.Path(yk.PathParams(max_depth=4)))
    normals = yk.IntegratorType.instantiate(ctx, yk.IntegratorType.GeometryNormals)
    want_acc, _ = path.render_tiles_accumulating(sc, cam, smp, tiles, [2] * len(tiles))
    want_nrm, _ = normals.render_tiles(sc, cam, smp, tiles)
    areas = (tiles["x1"].astype(int) - tiles["x0"]) * (tiles["y1"].astype(int) - tiles["y0"])
    offs = np.concatenate([[0], np.cumsum(areas)])
    comb = yk.Combiner([ctx], linger_us=300)
    got_acc, got_nrm = np.full_like(want_acc, -1.0), np.full_like(want_nrm, -1.0)
    jobs = [(t, m) for t in range(len(tiles)) for m in (0, 1)]

    def one(k, j):
        t, mode = jobs[j]
        ft = yk.FilmTile(tuple(int(v) for v in tiles[t]), sample=2)
        if mode == 0:
            got_acc[offs[t]:offs[t + 1]], _ = comb.render(path, sc, cam, smp, ft, accumulating=True)
        else:
            got_nrm[offs[t]:offs[t + 1]], _ = comb.render(normals, sc, cam, smp, ft)

    assert _workers(8, jobs, one) == []
    assert got_acc.tobytes() == want_acc.tobytes() and got_nrm.tobytes() == want_nrm.tobytes()
    comb.close()


@pytest.mark.gpu
def test_combiner_interruption_reaches_only_its_caller(yk):
    """One worker's predicate fires while its tile is inside a running submission (a long job: 4096 spp): that call returns
    CANCELLED; the other workers' tiles are queued again and come back complete and identical to a plain render."""
    sd = scenes.by_name("city-small")
    fs = yk.FilmSettings(res=(128, 64))
    tiles = yk.film_tiles(fs)  # 32 tiles x 256 pixels x 4096 spp: ~100 ms, three times what the assertion below needs
    smp = yk.SamplerType.Stratified((64, 64), True, SEED)
    integ = yk.IntegratorType.Path(yk.PathParams(max_depth=8))
    ctx = yk.Context(0)
    sc = yk.Scene(ctx, sd)
    cam = yk.Camera(sd.camera, fs)
    it = yk.IntegratorType.instantiate(ctx, integ)
    ref, st = it.render_tiles(sc, cam, smp, tiles)
    assert st.seconds_total > 0.040, "the job must be long enough for the interruption to land inside it"
    fire_after = 0.015
    areas = (tiles["x1"].astype(int) - tiles["x0"]) * (tiles["y1"].astype(int) - tiles["y0"])
    offs = np.concatenate([[0], np.cumsum(areas)])
    # max_tiles = all tiles with a long linger: the leader submits the moment every worker has arrived, so ONE submission holds
    # them all however the host schedules the threads; tile 3's predicate fires 15 ms after the last worker entered: inside the job
    comb = yk.Combiner([ctx], max_tiles=len(tiles), linger_us=5_000_000)
    polls = [0] * len(tiles)
    results = {}
    import time

    entered = []
    lock = threading.Lock()

    def worker(t):
        def pred():
            polls[t] += 1
            return t == 3 and len(entered) == len(tiles) and time.perf_counter() - entered[-1] > fire_after

        with lock:
            entered.append(time.perf_counter())
        try:
            px, _ = comb.render(it, sc, cam, smp, yk.FilmTile(tuple(int(v) for v in tiles[t])), cancel=pred)
            results[t] = px
        except yk.YukiError as e:
            results[t] = e

    th = [threading.Thread(target=worker, args=(t,)) for t in range(len(tiles))]
    [x.start() for x in th]
    [x.join() for x in th]
    assert isinstance(results[3], yk.YukiError) and results[3].status == 7
    for t in range(len(tiles)):
        if t != 3:
            assert not isinstance(results[t], Exception), results[t]
            assert results[t].tobytes() == ref[offs[t]:offs[t + 1]].tobytes()
        assert polls[t] > 0
    info = comb.info()
    assert info.requeued >= 1
    comb.close()
