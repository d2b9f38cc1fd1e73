"""Several GPUs behind the C ABI (include/yuki_hip.h, yk_multi_* / yk_dist_*): what a single Rust process needs in
place of render_manager.rs:78-97,206-210 + film.rs:210-282.  On the one-GPU test box:
  * G = 2 / 3 / 4 / 8 VIRTUAL ranks on device 0 (YK_MULTI_SHARED_DEVICES): the deal, the per-rank tile lists, slab layout,
    exchange ordering, device 0's scatter of foreign slabs, stats — everything but the wire (device copies stand in for RCCL,
    which refuses two ranks on one device);
  * the RCCL calls themselves through `rccl_loopback` (rank 0's slab takes ncclSend/ncclRecv to itself inside the same group
    call the G > 1 case uses) and a one-rank yk_dist communicator;
  * where two or more GPUs are visible, the real thing in a child process (fails, not xfails, when the film differs or it hangs)."""
import ctypes as C

import numpy as np
import pytest
import torch  # noqa: F401  first: this process then holds ONE RCCL, PyTorch's — libyuki_hip.so binds the loaded copy (yk_multi.cpp)

from yuki_amd import scenes

pytestmark = pytest.mark.gpu
SEED = 0x73B9642E74AC471C


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _single(yk, ctx, sd, fs, sampler, integ):
    cam = yk.Camera(sd.camera, fs)
    tiles = yk.film_tiles(fs)
    sc = yk.Scene(ctx, sd)
    rgb, st = yk.IntegratorType.instantiate(ctx, integ).render_tiles(sc, cam, sampler, tiles)
    sc.close()
    return yk.update_tiles(tiles, rgb, fs.res), st, cam


@pytest.mark.parametrize("loopback", [0, 1])
def test_multi_film_equals_the_single_device_render(ctx, yk, oracle, loopback):
    sd = scenes.by_name("city-small")
    fs = yk.FilmSettings(res=(200, 120), tile_dim=16)  # ragged right and bottom tiles
    sampler = yk.SamplerType.Stratified((2, 2), True, SEED)
    integ = yk.IntegratorType.Path(yk.PathParams(max_depth=6))
    want, st1, cam = _single(yk, ctx, sd, fs, sampler, integ)
    m = yk.Multi([0], rccl_loopback=loopback)
    msc = m.scene(sd)
    film = m.film(fs)
    got, st = m.render_film(msc, cam, sampler, integ, film)
    assert st.rays == st1.rays and st.samples == 200 * 120 * 4
    assert np.array_equal(_bits(got), _bits(want))
    # the oracle's render of the same film (whole frame, host-side Film::update_tile)
    owant, orays = oracle.OracleScene(sd).render_tiles(cam.matrices, sampler, integ, yk.film_tiles(fs), n_threads=0)
    assert orays == st.rays and np.array_equal(_bits(got), _bits(yk.update_tiles(yk.film_tiles(fs), owant, fs.res)))
    # asynchronous submission: nothing read back by the call; the frame is in device 0's film after yk_multi_sync
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemset(C.c_void_p(film.device_ptr), 0, C.c_size_t(got.nbytes))
    hip.hipDeviceSynchronize()
    none, nost = m.render_film(msc, cam, sampler, integ, film, want_host=False, want_stats=False)
    assert none is None and nost is None
    m.sync()
    back = np.zeros_like(got)
    assert hip.hipMemcpy(back.ctypes.data_as(C.c_void_p), C.c_void_p(film.device_ptr), C.c_size_t(back.nbytes), 2) == 0
    assert np.array_equal(_bits(back), _bits(want))
    film.close()
    msc.close()
    m.close()


@pytest.mark.parametrize("G", [2, 3, 4, 8])
def test_virtual_ranks_render_one_film(ctx, yk, G):
    """G ranks on ONE device: spiral tile i rendered by rank i mod G into its slab, the slabs of ranks 1 .. G-1 moved into
    device 0's gather buffers behind an event of the sender's stream, scattered through the lists0 tile lists — the film and
    the ray count equal the single-device render bit for bit (ragged film: partial tiles at the right and bottom edge)."""
    sd = scenes.by_name("city-small")
    fs = yk.FilmSettings(res=(200, 120), tile_dim=16)
    sampler = yk.SamplerType.Stratified((2, 2), True, SEED)
    integ = yk.IntegratorType.Path(yk.PathParams(max_depth=6))
    want, st1, cam = _single(yk, ctx, sd, fs, sampler, integ)
    m = yk.Multi([0] * G, flags=yk.Multi.SHARED_DEVICES)
    msc = m.scene(sd)
    film = m.film(fs)
    for loopback in (0, 1, 0):  # frame after frame on the same slabs; rank 0's slab through the exchange too
        m.set_option("rccl_loopback", loopback)
        got, st = m.render_film(msc, cam, sampler, integ, film)
        assert st.rays == st1.rays and st.shadow_rays == st1.shadow_rays and st.samples == 200 * 120 * 4
        assert np.array_equal(_bits(got), _bits(want))
    # asynchronous frames back to back: the next render of a rank must not overwrite a slab that is still being copied
    hip = C.CDLL("libamdhip64.so")
    for _ in range(3):
        m.render_film(msc, cam, sampler, integ, film, want_host=False, want_stats=False)
    m.sync()
    back = np.zeros_like(want)
    assert hip.hipMemcpy(back.ctypes.data_as(C.c_void_p), C.c_void_p(film.device_ptr), C.c_size_t(back.nbytes), 2) == 0
    assert np.array_equal(_bits(back), _bits(want))
    with pytest.raises(yk.YukiError):
        m.set_option("peer_copy", 0)  # ranks sharing a device cannot use RCCL
    film.close()
    msc.close()
    m.close()


def test_virtual_ranks_cfg3_at_full_size(ctx, yk, cfg3_scene):
    """BASELINE configs[3]'s shape on one GPU: the cfg3 frame (1,024,012 triangles, Path 8, Stratified 8x8, 1920x1080) rendered
    by 8 virtual ranks — 1020 tiles each, seven foreign slabs of ~3.1 MB scattered on device 0 — equals the single-device
    film bit for bit, and the ranks' ray counts add up to the single-device count."""
    sd = cfg3_scene
    fs = yk.FilmSettings(res=(1920, 1080), tile_dim=16)
    sampler = yk.SamplerType.Stratified((8, 8), True, SEED)
    integ = yk.IntegratorType.Path(yk.PathParams(max_depth=8))
    want, st1, cam = _single(yk, ctx, sd, fs, sampler, integ)
    m = yk.Multi([0] * 8, flags=yk.Multi.SHARED_DEVICES)
    msc = m.scene(sd)
    film = m.film(fs)
    got, st = m.render_film(msc, cam, sampler, integ, film)
    assert st.samples == 1920 * 1080 * 64 and st.rays == st1.rays and st.shadow_rays == st1.shadow_rays
    assert np.array_equal(_bits(got), _bits(want))
    film.close()
    msc.close()
    m.close()


def test_virtual_ranks_accumulating_film(ctx, yk):
    """yk_multi_accumulate_film: passes first .. first + n - 1 of every tile on its rank, added to device 0's film pass after
    pass (film.rs:260-272) — equal to the single-device accumulation (one pass per submission) bit for bit, and, divided by
    spp, to the plain film."""
    sd = scenes.by_name("city-tiny")
    fs = yk.FilmSettings(res=(96, 54), tile_dim=16, accumulate=True)
    sampler = yk.SamplerType.Stratified((2, 2), True, SEED)
    integ = yk.IntegratorType.Path(yk.PathParams(max_depth=5))
    cam = yk.Camera(sd.camera, fs)
    tiles = yk.film_tiles(fs)
    it = yk.IntegratorType.instantiate(ctx, integ)
    sc = yk.Scene(ctx, sd)
    want = np.zeros((54, 96, 3), dtype=np.float32)
    rays = 0
    for k in range(4):  # the reference's loop: every tile once per sample index, Film::update_tile adds
        px, st = it.render_tiles_accumulating(sc, cam, sampler, tiles, np.full(len(tiles), k, dtype=np.uint16))
        yk.accumulate_tiles(tiles, px, want)
        rays += st.rays
    plain, _ = it.render_tiles(sc, cam, sampler, tiles)
    sc.close()
    m = yk.Multi([0, 0, 0], flags=yk.Multi.SHARED_DEVICES)
    msc = m.scene(sd)
    film = m.film(fs)
    m.clear_film(film)
    _, s01 = m.accumulate_film(msc, cam, sampler, integ, film, 0, 1, want_host=False)
    got, s13 = m.accumulate_film(msc, cam, sampler, integ, film, 1, 3)
    assert s01.rays + s13.rays == rays
    assert np.array_equal(_bits(got), _bits(want))
    assert np.array_equal(_bits(got / np.float32(4)), _bits(yk.update_tiles(tiles, plain, fs.res)))
    m.clear_film(film)
    got2, _ = m.accumulate_film(msc, cam, sampler, integ, film, 0, 4)  # all four passes in one submission
    assert np.array_equal(_bits(got2), _bits(want))
    with pytest.raises(yk.YukiError):
        m.accumulate_film(msc, cam, sampler, integ, film, 3, 2)  # beyond the sampler's samples per pixel
    film.close()
    msc.close()
    m.close()


@pytest.mark.parametrize("G", [1, 4])
def test_one_shot_predicate_stops_every_rank(ctx, yk, G):
    """The reference's predicate is a consuming FnMut polled by one thread (render_worker.rs:240-249, `try_recv`): it answers
    "stop" ONCE.  G device threads poll through a latch — the user's function is never entered concurrently, never called
    again after it fired, and all ranks stop (YK_ERR_CANCELLED), not only the one whose poll consumed the message."""
    import threading
    import time

    sd = scenes.by_name("city-small")
    fs = yk.FilmSettings(res=(640, 360), tile_dim=16)
    sampler = yk.SamplerType.Stratified((8, 8), True, SEED)
    integ = yk.IntegratorType.Path(yk.PathParams(max_depth=8))
    cam = yk.Camera(sd.camera, fs)
    m = yk.Multi([0] * G, flags=yk.Multi.SHARED_DEVICES if G > 1 else 0)
    msc = m.scene(sd)
    film = m.film(fs)
    want, st_full = m.render_film(msc, cam, sampler, integ, film)
    state = dict(calls=0, inside=0, overlap=0, fired=False, after=0, lock=threading.Lock())

    def one_shot():
        with state["lock"]:
            state["inside"] += 1
            state["overlap"] = max(state["overlap"], state["inside"])
            state["calls"] += 1
            if state["fired"]:
                state["after"] += 1
        time.sleep(0.0002)  # releases the GIL: a second thread inside the predicate would be seen
        with state["lock"]:
            state["inside"] -= 1
            if not state["fired"] and state["calls"] >= 3:
                state["fired"] = True
                return True
        return False

    t0 = time.time()
    with pytest.raises(yk.YukiError) as e:
        m.render_film(msc, cam, sampler, integ, film, cancel=one_shot)
    dt = time.time() - t0
    assert e.value.status == 7  # YK_ERR_CANCELLED
    assert state["fired"] and state["after"] == 0 and state["overlap"] == 1
    assert dt < st_full.seconds_total + 2.0  # (a sanity bound: a shared host can stall a thread for hundreds of ms; that every rank stopped is the status above)
    # the next frame on the same object is unaffected
    got, st = m.render_film(msc, cam, sampler, integ, film)
    assert st.rays == st_full.rays and np.array_equal(_bits(got), _bits(want))
    # yk_multi_interrupt from another thread while a frame renders: every rank's call comes back cancelled (or the frame had
    # already finished); the frame after it is whole again
    timer = threading.Timer(0.002, m.interrupt)
    timer.start()
    try:
        m.render_film(msc, cam, sampler, integ, film)
    except yk.YukiError as err:
        assert err.status == 7
    timer.join()
    got, st = m.render_film(msc, cam, sampler, integ, film)
    assert st.rays == st_full.rays and np.array_equal(_bits(got), _bits(want))
    film.close()
    msc.close()
    m.close()


def test_multi_calls_leave_the_callers_device_alone(yk):
    """A host such as PyTorch allocates on ITS current device: every yk_multi entry point puts it back."""
    hip = C.CDLL("libamdhip64.so")
    cur = C.c_int(-1)
    assert hip.hipGetDevice(C.byref(cur)) == 0
    before = cur.value
    m = yk.Multi([0, 0], flags=yk.Multi.SHARED_DEVICES)
    film = m.film(yk.FilmSettings(res=(64, 64), tile_dim=16))
    m.sync()
    film.close()
    m.close()
    assert hip.hipGetDevice(C.byref(cur)) == 0 and cur.value == before


def test_multi_argument_errors(yk):
    with pytest.raises(yk.YukiError):
        yk.Multi([])
    with pytest.raises(yk.YukiError):
        yk.Multi([0, 0])  # one rank per GPU
    with pytest.raises(yk.YukiError):
        yk.Multi([97])
    m = yk.Multi([0])
    with pytest.raises(yk.YukiError):
        m.set_option("no_such_option", 1)
    with pytest.raises(yk.YukiError):
        m.film(yk.FilmSettings(res=(0, 16)))
    m.close()


def test_dist_one_rank_gather(ctx, yk):
    """yk_dist_*: ncclGetUniqueId -> ncclCommInitRank -> grouped ncclSend/ncclRecv on the context's stream."""
    import torch

    d = yk.Dist(ctx, yk.Dist.unique_id(), 0, 1)
    src = torch.arange(4096, dtype=torch.float32, device="cuda:0") * 0.5
    dst = torch.zeros_like(src)
    torch.cuda.synchronize()
    d.gather(src.data_ptr(), dst.data_ptr(), src.numel())
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamSynchronize(C.c_void_p(ctx.stream_handle))
    assert torch.equal(src, dst)
    d.close()


_SEVERAL_DEVICES = r"""
import sys
import numpy as np
import torch  # first: one RCCL / HIP runtime per process (tests/conftest.py)
sys.path.insert(0, sys.argv[1])
from yuki_amd import scenes, core as yk

G = int(sys.argv[2])
sd = scenes.by_name("city-small")
fs = yk.FilmSettings(res=(200, 120), tile_dim=16)
sampler = yk.SamplerType.Stratified((2, 2), True, 0x73B9642E74AC471C)
integ = yk.IntegratorType.Path(yk.PathParams(max_depth=6))
ctx = yk.Context(0)
cam = yk.Camera(sd.camera, fs)
tiles = yk.film_tiles(fs)
sc = yk.Scene(ctx, sd)
rgb, st1 = yk.IntegratorType.instantiate(ctx, integ).render_tiles(sc, cam, sampler, tiles)
want = yk.update_tiles(tiles, rgb, fs.res)
for flags in (0, yk.Multi.PEER_COPY):  # RCCL send / recv, then hipMemcpyPeerAsync
    m = yk.Multi(list(range(G)), flags=flags)
    msc = m.scene(sd)
    film = m.film(fs)
    for _ in range(3):  # communicators are made on the first frame and reused
        got, st = m.render_film(msc, cam, sampler, integ, film)
        assert st.rays == st1.rays
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    film.close(); msc.close(); m.close()
print("identical", G)
"""


def test_several_devices_render_one_film():
    """cfg4's shape on real hardware, when the box has it: yk_multi over 2 .. 4 GPUs — spiral tiles dealt round-robin, slabs to
    device 0 over RCCL send / recv, Film::update_tile there — equals the single-device film bit for bit.  Runs in a child process
    under a time limit: a wrong film, a stuck collective (timeout) or a fault FAILS the test with the child's stderr; skipped
    where only one GPU is visible (the virtual-rank tests above cover everything but the wire there)."""
    import os
    import subprocess
    import sys

    n = torch.cuda.device_count()  # counting does not initialise the GPU
    if n < 2:
        pytest.skip("one GPU visible")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _SEVERAL_DEVICES, root, str(min(n, 4))], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "identical" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
