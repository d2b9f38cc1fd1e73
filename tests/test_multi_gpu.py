"""Several GPUs behind the C ABI (include/yuki_hip.h, yk_multi_* / yk_dist_*): what a single Rust process needs in
place of render_manager.rs:78-97,206-210 + film.rs:210-282.  On the one-GPU test box the device list is [0]; the RCCL
exchange itself is exercised by `rccl_loopback` (rank 0's slab takes ncclSend/ncclRecv to itself inside the same group
call the G > 1 case uses) and by a one-rank yk_dist communicator.  The N-rank case runs in the driver's scaling bench."""
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
m = yk.Multi(list(range(G)))
msc = m.scene(sd)
film = m.film(fs)
for _ in range(3):  # communicators are made on the first frame and reused
    got, st = m.render_film(msc, cam, sampler, integ, film)
    assert st.rays == st1.rays
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
print("identical", G)
"""


@pytest.mark.xfail(reason="first run on hardware: the development boxes have one GPU, this path had never executed when it was written", strict=False)
def test_several_devices_render_one_film():
    """cfg4's shape on real hardware, when the box has it: yk_multi over 2 .. 4 GPUs — spiral tiles dealt round-robin, slabs to
    device 0 over RCCL send / recv, Film::update_tile there — equals the single-device film bit for bit.  Runs in a child process
    under a time limit (a stuck collective fails the test instead of stalling the suite); skipped on a one-GPU box."""
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
