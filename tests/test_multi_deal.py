"""The multi-GPU tile deal without a device (yk_multi_deal, the function yk_multi_film_create and bench.py's ranks use):
spiral tile i of film_tiles(res, tile_dim) (film.rs:333-376, 409-475) belongs to rank i mod G (render_manager.rs:206-210),
for G = 1 / 2 / 4 / 8 and the film shapes of the BASELINE configurations plus ragged ones.  The spiral itself is checked
against the oracle's independent restatement of film.rs."""
import numpy as np
import pytest

from yuki_amd import abi

FILMS = [((1920, 1080), 16), ((3840, 2160), 16), ((200, 120), 16), ((33, 17), 16), ((64, 64), 64), ((640, 480), 16), ((70, 41), 8), ((512, 512), 16)]


def _px(t):
    return (t["x1"].astype(np.int64) - t["x0"]) * (t["y1"].astype(np.int64) - t["y0"])


@pytest.mark.parametrize("res,tile_dim", FILMS)
@pytest.mark.parametrize("G", [1, 2, 3, 4, 8])
def test_deal_is_the_spiral_taken_i_mod_g(yk, oracle, res, tile_dim, G):
    fs = yk.FilmSettings(res=res, tile_dim=tile_dim)
    spiral = oracle.film_tiles(res, tile_dim)  # film.rs restated independently (oracle/orender.h)
    assert np.array_equal(yk.film_tiles(fs), spiral)
    if len(spiral) < G:
        pytest.skip("fewer tiles than ranks")
    cover = np.zeros((res[1], res[0]), dtype=np.uint8)
    total_px = 0
    for r in range(G):
        mine, px = yk.multi_deal(fs, G, r)
        assert np.array_equal(mine, spiral[r::G])  # tile i -> rank i mod G, spiral order kept inside the rank
        assert px == int(_px(mine).sum())  # the slab of the rank: 3 * px floats, tile after tile
        total_px += px
        for t in mine:
            cover[t["y0"] : t["y1"], t["x0"] : t["x1"]] += 1
    assert total_px == res[0] * res[1] and np.all(cover == 1)  # the ranks partition the film exactly


def test_deal_sizes_are_balanced_at_1080p(yk):
    """The G = 8 shares of the 1080p film: 1020 tiles each; the half-height tiles of the bottom row (film.rs:409-475: 1080 = 67.5 x 16)
    do not fall evenly, so pixel counts differ by up to 1.4 % (largest share 0.35 % above the mean)."""
    fs = yk.FilmSettings(res=(1920, 1080), tile_dim=16)
    deals = [yk.multi_deal(fs, 8, r) for r in range(8)]
    px = [d[1] for d in deals]
    assert all(len(d[0]) == 1020 for d in deals)
    assert sum(px) == 1920 * 1080 and max(px) / (sum(px) / 8) < 1.004 and min(px) / (sum(px) / 8) > 0.99


def test_deal_argument_errors(yk):
    fs = yk.FilmSettings(res=(64, 64), tile_dim=16)
    L = yk.lib()
    assert L.yk_multi_deal(64, 64, 16, 0, 0, None, 0, None) == 0
    assert L.yk_multi_deal(64, 64, 16, 2, 2, None, 0, None) == 0
    assert L.yk_multi_deal(0, 64, 16, 2, 0, None, 0, None) == 0
    assert L.yk_multi_deal(64, 64, 0, 2, 0, None, 0, None) == 0
    # a short output buffer is filled up to its capacity and the full count still returned
    t = np.zeros(3, dtype=abi.TILE_DTYPE)
    n = L.yk_multi_deal(64, 64, 16, 2, 1, t.ctypes.data, 3, None)
    assert n == 8 and np.array_equal(t, yk.multi_deal(fs, 2, 1)[0][:3])
