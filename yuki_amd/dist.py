"""Multi-GPU partition of the film.

The path shards naturally: every pixel-sample re-derives its RNG from
(seed, pixel, sample_index) (uniform.rs:72-84) and tiles share nothing but the
read-only scene.  The spiral-ordered tile list (film.rs:333-376) is dealt
round-robin to the ranks (the reference's own TODO suggests interleaving,
render_manager.rs:206-210), each rank renders its tiles into a dense tile-major
slab, and ONE gather moves the slabs to rank 0, which scatters them into the
row-major film (the role of Film::update_tile, film.rs:210-282).  No reduction
is needed because tiles are disjoint.

`backend` is whatever torch.distributed was initialised with: "nccl" (= RCCL over
xGMI) on the GPUs, "gloo" in the CPU tests.
"""
import numpy as np

from . import abi


def shard_tiles(tiles, rank, world):
    """Tile i of the spiral order goes to rank i mod world (for a whole film: yk_multi_deal, core.multi_deal — the same
    deal behind the C ABI, which yk_multi_film_create uses; tests/test_multi_deal.py holds the two together)."""
    return np.ascontiguousarray(np.asarray(tiles, dtype=abi.TILE_DTYPE)[rank::world])


def tile_pixels(tiles):
    t = np.asarray(tiles, dtype=abi.TILE_DTYPE)
    return int(((t["x1"].astype(np.int64) - t["x0"]) * (t["y1"].astype(np.int64) - t["y0"])).sum())


def slab_pixels(tiles, world):
    """Pixels of the largest per-rank slab (edge tiles are clipped, so slabs differ)."""
    return max(tile_pixels(shard_tiles(tiles, r, world)) for r in range(world))


def gather_slabs(slab, world, rank, dist):
    """torch.distributed gather of equally sized (padded) slabs to rank 0."""
    import torch

    gathered = [torch.empty_like(slab) for _ in range(world)] if rank == 0 else None
    dist.gather(slab, gathered, dst=0)
    return gathered


def assemble_film_host(tiles, slabs, res, update_tiles):
    """Rank 0, host memory: scatter every rank's slab into the film.
    `update_tiles(tiles, rgb, res) -> film` is Film::update_tile for a tile list."""
    world = len(slabs)
    film = np.zeros((res[1], res[0], 3), dtype=np.float32)
    for r in range(world):
        tr = shard_tiles(tiles, r, world)
        n = tile_pixels(tr)
        part = update_tiles(tr, np.ascontiguousarray(slabs[r][: n * 3]).reshape(n, 3), res)
        mask = np.zeros((res[1], res[0]), dtype=bool)
        for t in tr:
            mask[t["y0"] : t["y1"], t["x0"] : t["x1"]] = True
        film[mask] = part[mask]
    return film
