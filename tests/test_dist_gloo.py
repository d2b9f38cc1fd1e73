"""N > 1 path on CPU: two processes over the gloo backend exercise the same
partition + gather + scatter code bench.py runs on RCCL (yuki_amd/dist.py).  The
per-rank 'renderer' here is the CPU oracle (this is a test), so the assembled film
must equal a single-process render bit for bit — tiles are independent units."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from yuki_amd import abi, dist as ydist, scenes

SEED = 0x73B9642E74AC471C
RES = (70, 41)  # ragged right/bottom tiles -> unequal slabs


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding as oracle
    from yuki_amd import core as yk

    sd = scenes.by_name("city-tiny")
    fs = yk.FilmSettings(res=RES, tile_dim=16)
    tiles = yk.film_tiles(fs)
    cam = yk.Camera(sd.camera, fs)
    smp = abi.SamplerDesc(abi.SAMPLER_STRATIFIED, 2, 2, 1, SEED)
    integ = abi.IntegratorDesc(abi.INTEGRATOR_PATH, 6, 0, 0.0)
    mine, my_px = yk.multi_deal(fs, world, rank)  # the C ABI's deal (yk_multi_deal: what yk_multi_film_create uses) ...
    assert np.array_equal(mine, ydist.shard_tiles(tiles, rank, world)) and my_px == ydist.tile_pixels(mine)  # ... and bench.py's are one
    rgb, rays = oracle.OracleScene(sd).render_tiles(cam.matrices, smp, integ, mine, n_threads=1)
    slab = torch.zeros(ydist.slab_pixels(tiles, world) * 3, dtype=torch.float32)
    slab[: rgb.size] = torch.from_numpy(rgb.reshape(-1))
    gathered = ydist.gather_slabs(slab, world, rank, dist)
    total = torch.tensor([rays], dtype=torch.int64)
    dist.all_reduce(total, op=dist.ReduceOp.SUM)
    if rank == 0:
        film = ydist.assemble_film_host(tiles, [g.numpy() for g in gathered], RES, yk.update_tiles)
        np.savez(out_path, film=film, rays=np.int64(total.item()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_render_equals_single_process(tmp_path, oracle, yk):
    out = str(tmp_path / "film.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    sd = scenes.by_name("city-tiny")
    fs = yk.FilmSettings(res=RES, tile_dim=16)
    tiles = yk.film_tiles(fs)
    cam = yk.Camera(sd.camera, fs)
    smp = abi.SamplerDesc(abi.SAMPLER_STRATIFIED, 2, 2, 1, SEED)
    integ = abi.IntegratorDesc(abi.INTEGRATOR_PATH, 6, 0, 0.0)
    rgb, rays = oracle.OracleScene(sd).render_tiles(cam.matrices, smp, integ, tiles, n_threads=1)
    want = yk.update_tiles(tiles, rgb, RES)
    assert int(got["rays"]) == rays
    assert np.array_equal(got["film"].view(np.uint32), want.view(np.uint32))


def test_shard_tiles_partition():
    import itertools

    tiles = np.zeros(37, dtype=abi.TILE_DTYPE)
    tiles["x0"] = np.arange(37)
    tiles["x1"] = tiles["x0"] + 1
    tiles["y1"] = 1
    for world in (1, 2, 4, 8):
        parts = [ydist.shard_tiles(tiles, r, world) for r in range(world)]
        allx = sorted(itertools.chain.from_iterable(p["x0"].tolist() for p in parts))
        assert allx == list(range(37))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
        assert ydist.slab_pixels(tiles, world) == max(len(p) for p in parts)
