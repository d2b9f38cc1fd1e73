"""Film output and the accumulating film.

  * yk_write_exr / yk_write_pfm: the file is parsed back here by an independent reader and
    must return the film bit for bit (the reference writes through the `exr` crate, absent
    from /root/reference; byte-identical files are not claimed — parity unpinned — only a
    standard-conforming RGB float EXR that round-trips exactly).
  * accumulate mode: Integrator::render(accumulating=true) renders ONE sample per pixel with
    global index FilmTile.sample and Film::update_tile sums them (integrators/mod.rs:146-161,
    film.rs:260-272, render_manager.rs:135-143).  GPU == oracle bit for bit; and the sum of
    the per-sample passes equals spp x the plain render up to float summation order."""
import struct

import numpy as np
import pytest

from yuki_amd import abi, scenes


def read_exr(path):
    """Minimal reader for uncompressed scan-line OpenEXR files (independent of the writer)."""
    d = open(path, "rb").read()
    magic, version = struct.unpack_from("<II", d, 0)
    assert magic == 20000630 and version & 0xFF == 2 and version >> 8 == 0  # single-part scan line
    pos, attrs = 8, {}
    while d[pos] != 0:
        e = d.index(b"\0", pos)
        name = d[pos:e].decode()
        pos = e + 1
        e = d.index(b"\0", pos)
        typ = d[pos:e].decode()
        pos = e + 1
        (size,) = struct.unpack_from("<i", d, pos)
        attrs[name] = (typ, d[pos + 4 : pos + 4 + size])
        pos += 4 + size
    pos += 1
    assert attrs["compression"] == ("compression", b"\0") and attrs["lineOrder"] == ("lineOrder", b"\0")
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    assert attrs["displayWindow"][1] == attrs["dataWindow"][1] and (x0, y0) == (0, 0)
    w, h = x1 + 1, y1 + 1
    chans, c, body = [], 0, attrs["channels"][1]
    while body[c] != 0:
        e = body.index(b"\0", c)
        ptype, plin, xs, ys = struct.unpack_from("<iB3xii", body, e + 1)
        assert (ptype, xs, ys) == (2, 1, 1)
        chans.append(body[c:e].decode())
        c = e + 1 + 16
    assert chans == sorted(chans) == ["B", "G", "R"]
    offs = struct.unpack_from("<%dQ" % h, d, pos)
    img = np.zeros((h, w, 3), dtype=np.float32)
    for y in range(h):
        yy, size = struct.unpack_from("<ii", d, offs[y])
        assert yy == y and size == 12 * w
        row = np.frombuffer(d, dtype="<f4", count=3 * w, offset=offs[y] + 8).reshape(3, w)
        img[y, :, 2], img[y, :, 1], img[y, :, 0] = row[0], row[1], row[2]
    assert offs[-1] + 8 + 12 * w == len(d)
    return img


def read_pfm(path):
    d = open(path, "rb").read()
    head = d.split(b"\n", 3)
    assert head[0] == b"PF"
    w, h = map(int, head[1].split())
    assert float(head[2]) < 0  # little endian
    return np.frombuffer(head[3], dtype="<f4").reshape(h, w, 3)[::-1]


@pytest.mark.parametrize("res", [(1, 1), (7, 3), (64, 48)])
def test_exr_and_pfm_round_trip(tmp_path, yk, res):
    rng = np.random.default_rng(res[0])
    film = rng.normal(size=(res[1], res[0], 3)).astype(np.float32) * 50
    film[0, 0] = [np.inf, -0.0, 1e-42]  # specials survive: the file stores raw binary32
    yk.write_exr(tmp_path / "f.exr", film)
    yk.write_pfm(tmp_path / "f.pfm", film)
    assert read_exr(tmp_path / "f.exr").tobytes() == film.tobytes()
    assert read_pfm(tmp_path / "f.pfm").tobytes() == film.tobytes()


def test_write_errors(tmp_path, yk):
    from yuki_amd._ffi import YukiError

    with pytest.raises(YukiError):
        yk.write_exr(tmp_path / "no_such_dir" / "f.exr", np.zeros((2, 2, 3), np.float32))


def test_film_accumulate_host(yk, oracle):
    """Film::update_tile, accumulating branch: += per pixel and samples[tile] += 1."""
    fs = yk.FilmSettings(res=(40, 24), tile_dim=16)
    tiles = yk.film_tiles(fs)
    npx = 40 * 24
    rng = np.random.default_rng(1)
    film = np.zeros((24, 40, 3), dtype=np.float32)
    counts = np.zeros(len(tiles), dtype=np.uint32)
    want = np.zeros_like(film)
    for k in range(3):
        rgb = rng.uniform(size=(npx, 3)).astype(np.float32)
        yk.accumulate_tiles(tiles, rgb, film, counts)
        want = want + yk.update_tiles(tiles, rgb, fs.res)  # same float op per pixel: a += b
    assert film.tobytes() == want.tobytes() and np.all(counts == 3)
    from yuki_amd._ffi import YukiError

    bad = tiles.copy()
    bad["x1"][0] = 100
    with pytest.raises(YukiError):  # "Tile doesn't fit film", film.rs:227-234
        yk.accumulate_tiles(bad, np.zeros((npx + 2000, 3), np.float32), film)


@pytest.mark.gpu
def test_one_tile_per_call_like_the_reference(ctx, yk, oracle):
    """Integrator::render as the reference's workers call it — ONE tile per call (render_worker.rs:230-250), plain and
    accumulating (FilmTile.sample = 0 .. spp - 1): a one-tile chunk passes its tile as a kernel argument (k_pixel_table_one);
    pixels and ray counts equal the oracle's for ragged edge tiles too, and the samples fold into the plain result."""
    sd = scenes.by_name("city-tiny")
    fs = yk.FilmSettings(res=(70, 41), tile_dim=16)  # 6-wide and 9-high tiles at the right / bottom edge
    cam = yk.Camera(sd.camera, fs)
    tiles = yk.film_tiles(fs)
    sampler = yk.SamplerType.Stratified((2, 2), True, 0x51C0FFEE)
    integ = yk.IntegratorType.Path(yk.PathParams(max_depth=6))
    it = yk.IntegratorType.instantiate(ctx, integ)
    sc = yk.Scene(ctx, sd)
    osc = oracle.OracleScene(sd)
    picked = [tiles[0], tiles[len(tiles) // 2], tiles[-1], max(tiles, key=lambda t: int(t["x0"])), max(tiles, key=lambda t: int(t["y0"]))]
    for t in picked:
        one = np.array([tuple(int(v) for v in t)], dtype=abi.TILE_DTYPE)
        got, rays = it.render(sc, cam, sampler, yk.FilmTile(tuple(int(v) for v in t)))
        want, orays = osc.render_tiles(cam.matrices, sampler, integ, one, n_threads=0)
        assert rays == orays and np.array_equal(got.view(np.uint32), want.view(np.uint32))
        acc = np.zeros_like(got)
        for k in range(4):
            px, r = it.render(sc, cam, sampler, yk.FilmTile(tuple(int(v) for v in t), sample=k), accumulating=True)
            owant, _ = osc.render_tiles_accumulating(cam.matrices, sampler, integ, one, np.array([k], dtype=np.uint16), n_threads=0)
            assert np.array_equal(px.view(np.uint32), owant.view(np.uint32))
            acc = acc + px  # Film::update_tile's `+=` in sample order (film.rs:260-272)
        assert np.array_equal((acc / np.float32(4)).view(np.uint32), got.view(np.uint32))
    sc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("skind", ["uniform", "stratified"])
def test_accumulating_render_matches_oracle(ctx, yk, oracle, skind):
    sd = scenes.by_name("city-tiny")
    fs = yk.FilmSettings(res=(96, 54), tile_dim=16, accumulate=True)
    cam = yk.Camera(sd.camera, fs)
    tiles = yk.film_tiles(fs)
    seed = 0x73B9642E74AC471C
    sampler = yk.SamplerType.Uniform(4, seed) if skind == "uniform" else yk.SamplerType.Stratified((2, 2), True, seed)
    spp = yk.samples_per_pixel(sampler)
    integ = yk.IntegratorType.Path(yk.PathParams(max_depth=5))
    it = yk.IntegratorType.instantiate(ctx, integ)
    sc = yk.Scene(ctx, sd)
    osc = oracle.OracleScene(sd)
    # the manager's queue (render_manager.rs:135-143): every tile once per sample index; here
    # in ONE call, tiles of different sample indices mixed
    all_tiles = np.concatenate([tiles] * spp)
    all_samples = np.repeat(np.arange(spp, dtype=np.uint16), len(tiles))
    perm = np.random.default_rng(3).permutation(len(all_tiles))
    got, stats = it.render_tiles_accumulating(sc, cam, sampler, all_tiles[perm], all_samples[perm])
    want, rays = osc.render_tiles_accumulating(cam.matrices, sampler, integ, all_tiles[perm], all_samples[perm], n_threads=0)
    assert stats.rays == rays and stats.samples == 96 * 54 * spp
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # fold into the film sample by sample (deterministic order) and compare with the plain render
    film = np.zeros((54, 96, 3), dtype=np.float32)
    counts = np.zeros(len(tiles), dtype=np.uint32)
    npx = 96 * 54
    inv = np.argsort(perm)
    for s in range(spp):
        sel = inv[s * len(tiles) : (s + 1) * len(tiles)]  # positions of sample s's tiles in the permuted call
        offs = np.concatenate([[0], np.cumsum((all_tiles[perm]["x1"].astype(int) - all_tiles[perm]["x0"]) * (all_tiles[perm]["y1"].astype(int) - all_tiles[perm]["y0"]))])
        rgb = np.concatenate([got[offs[p] : offs[p + 1]] for p in sel])
        yk.accumulate_tiles(tiles, rgb, film, counts)
    assert np.all(counts == spp)
    plain, pstats = it.render_tiles(sc, cam, sampler, tiles)
    plain_film = yk.update_tiles(tiles, plain, fs.res)
    assert pstats.rays == stats.rays
    # same samples, same order of addition (sample 0..spp-1) -> identical sums; the plain film divides once
    assert np.array_equal((film / np.float32(spp)).view(np.uint32), plain_film.view(np.uint32))


@pytest.mark.gpu
def test_film_accumulate_device(ctx, yk):
    import torch

    fs = yk.FilmSettings(res=(50, 30), tile_dim=16)
    tiles = yk.film_tiles(fs)
    rng = np.random.default_rng(2)
    film_h = np.zeros((30, 50, 3), dtype=np.float32)
    film_d = torch.zeros((30, 50, 3), dtype=torch.float32, device="cuda:0")
    for k in range(3):
        rgb = rng.uniform(size=(50 * 30, 3)).astype(np.float32)
        yk.accumulate_tiles(tiles, rgb, film_h)
        t = torch.from_numpy(rgb).to("cuda:0")
        torch.cuda.synchronize()
        yk.check(yk.lib().yk_film_accumulate_tiles_device(ctx.h, tiles.ctypes.data, len(tiles), t.data_ptr(), 50, 30, film_d.data_ptr(), None), ctx.h)
    assert film_d.cpu().numpy().tobytes() == film_h.tobytes()


@pytest.mark.gpu
def test_prepared_tile_list_matches_the_array_entry_points(ctx, yk):
    """yk_tile_list: render + film update from a device-resident tile list, enqueued on a caller
    stream without host synchronisation, equals yk_render_tiles + yk_film_update_tiles."""
    import torch

    sd = scenes.by_name("city-tiny")
    fs = yk.FilmSettings(res=(100, 60), tile_dim=16)
    cam = yk.Camera(sd.camera, fs)
    tiles = yk.film_tiles(fs)[::2]  # a shard: every other tile
    smp = yk.SamplerType.Stratified((2, 2), True, 0x73B9642E74AC471C)
    it = yk.IntegratorType.instantiate(ctx, yk.IntegratorType.Path(yk.PathParams(max_depth=5)))
    sc = yk.Scene(ctx, sd)
    want, _ = it.render_tiles(sc, cam, smp, tiles)
    want_film = yk.update_tiles(tiles, want, fs.res)
    tl = yk.TileList(ctx, tiles)
    assert tl.n_pixels == want.shape[0]
    stream = torch.cuda.Stream()
    slab = torch.zeros(tl.n_pixels * 3, dtype=torch.float32, device="cuda:0")
    film = torch.zeros(60 * 100 * 3, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    for _ in range(3):  # back-to-back frames, nothing waits on the host in between
        it.render_tile_list_device(sc, cam, smp, tl, slab.data_ptr(), stream=stream.cuda_stream)
        tl.update_film_device(slab.data_ptr(), fs.res, film.data_ptr(), stream=stream.cuda_stream)
    stream.synchronize()
    assert slab.cpu().numpy().tobytes() == want.tobytes()
    assert film.cpu().numpy().tobytes() == want_film.tobytes()
    st = it.render_tile_list_device(sc, cam, smp, tl, slab.data_ptr(), want_stats=True)
    assert st.samples == tl.n_pixels * 4 and st.rays > 0
    # accumulating list: one sample per pixel with the list's per-tile sample index
    acc = yk.TileList(ctx, tiles, np.full(len(tiles), 2, dtype=np.uint16))
    it.render_tile_list_device(sc, cam, smp, acc, slab.data_ptr(), want_stats=True)
    got_acc = slab.cpu().numpy().reshape(-1, 3)
    want_acc, _ = it.render_tiles_accumulating(sc, cam, smp, tiles, np.full(len(tiles), 2, dtype=np.uint16))
    assert got_acc.tobytes() == want_acc.tobytes()


@pytest.mark.gpu
def test_context_stream_orders_foreign_work_after_a_render(ctx, yk):
    """yk_context_stream: with no caller stream everything a context does is ordered on its own
    stream, which a caller may wrap (torch.cuda.ExternalStream) to queue other device work behind
    a render — what bench.py does with the RCCL gather.  A second context on the device renders
    the same scene and tile list and scatters with the list made by the first."""
    import torch

    sd = scenes.by_name("city-tiny")
    fs = yk.FilmSettings(res=(96, 64), tile_dim=16)
    cam = yk.Camera(sd.camera, fs)
    tiles = yk.film_tiles(fs)
    smp = yk.SamplerType.Stratified((2, 2), True, 0x73B9642E74AC471C)
    integ = yk.IntegratorType.Path(yk.PathParams(max_depth=5))
    sc = yk.Scene(ctx, sd)
    want, _ = yk.IntegratorType.instantiate(ctx, integ).render_tiles(sc, cam, smp, tiles)
    want_film = yk.update_tiles(tiles, want, fs.res)
    tl = yk.TileList(ctx, tiles)
    ctx2 = yk.Context(0)
    assert ctx.stream_handle and ctx2.stream_handle and ctx.stream_handle != ctx2.stream_handle
    outs = []
    for c in (ctx, ctx2):
        it = yk.IntegratorType.instantiate(c, integ)
        slab = torch.zeros(tl.n_pixels * 3, dtype=torch.float32, device="cuda:0")
        film = torch.zeros(64 * 96 * 3, dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        ext = torch.cuda.ExternalStream(c.stream_handle, device=torch.device("cuda:0"))
        with torch.cuda.stream(ext):
            it.render_tile_list_device(sc, cam, smp, tl, slab.data_ptr())  # stream=None: the context's own
            copy = slab.clone()  # torch work queued on the same stream: sees the finished render
            tl.update_film_device(slab.data_ptr(), fs.res, film.data_ptr(), ctx=c)
        outs.append((copy, film))
    torch.cuda.synchronize()
    for copy, film in outs:
        assert copy.cpu().numpy().tobytes() == want.tobytes()
        assert film.cpu().numpy().tobytes() == want_film.tobytes()
    ctx2.close()


@pytest.mark.gpu
@pytest.mark.parametrize("integ_name,skind", [("path", "stratified"), ("path", "uniform"), ("whitted", "stratified")])
def test_several_accumulating_passes_in_one_submission(ctx, yk, oracle, integ_name, skind):
    """yk_render_tiles_accumulating_passes: passes s, s+1, ... rendered at once are bit for bit the
    single-pass renders with those sample indices (and the oracle's), and the film accumulated from
    them pass after pass equals the film of n separate submissions (film.rs:260-272)."""
    import torch

    sd = scenes.by_name("city-tiny" if integ_name == "path" else "glass-balls")
    fs = yk.FilmSettings(res=(80, 48), tile_dim=16, accumulate=True)
    cam = yk.Camera(sd.camera, fs)
    tiles = yk.film_tiles(fs)
    smp = yk.SamplerType.Stratified((3, 3), True, 0x73B9642E74AC471C) if skind == "stratified" else yk.SamplerType.Uniform(9, 0x73B9642E74AC471C)
    integ = yk.IntegratorType.Path(yk.PathParams(max_depth=5)) if integ_name == "path" else yk.IntegratorType.Whitted(4)
    it = yk.IntegratorType.instantiate(ctx, integ)
    sc = yk.Scene(ctx, sd)
    first = (np.arange(len(tiles)) % 3).astype(np.uint16)  # tiles at different sample counts, like a queue mid-way through a frame
    n = 5
    got, st = it.render_tiles_accumulating(sc, cam, smp, tiles, first, n_passes=n)
    assert got.shape == (n, 80 * 48, 3)
    osc = oracle.OracleScene(sd)
    rays = 0
    for k in range(n):
        one, st1 = it.render_tiles_accumulating(sc, cam, smp, tiles, first + k)
        assert got[k].tobytes() == one.tobytes()
        rays += st1.rays
        if k in (0, n - 1):
            want, _ = osc.render_tiles_accumulating(cam.matrices, smp, integ, tiles, first + k)
            assert got[k].tobytes() == want.tobytes()
    assert st.rays == rays and st.samples == n * 80 * 48
    # device film: one fused accumulate of n passes == n accumulates
    tl = yk.TileList(ctx, tiles, first)
    slab = torch.zeros(n * tl.n_pixels * 3, dtype=torch.float32, device="cuda:0")
    film_a = torch.full((48 * 80 * 3,), 0.25, dtype=torch.float32, device="cuda:0")
    film_b = film_a.clone()
    it.render_tile_list_device(sc, cam, smp, tl, slab.data_ptr(), n_passes=n)
    tl.update_film_device(slab.data_ptr(), fs.res, film_a.data_ptr(), accumulate=True, n_passes=n)
    for k in range(n):
        tl.update_film_device(slab.data_ptr() + 4 * 3 * tl.n_pixels * k, fs.res, film_b.data_ptr(), accumulate=True)
    torch.cuda.synchronize()
    assert slab.cpu().numpy().tobytes() == got.tobytes()
    assert torch.equal(film_a, film_b)
    host = np.full((48, 80, 3), 0.25, dtype=np.float32)
    for k in range(n):
        yk.accumulate_tiles(tiles, got[k], host)  # the host restatement of Film::update_tile, in place
    assert film_a.cpu().numpy().tobytes() == host.tobytes()
    with pytest.raises(yk.YukiError):
        it.render_tiles_accumulating(sc, cam, smp, tiles, first, n_passes=70000)
    # the reference queues samples 0 .. spp-1 only (render_manager.rs:135-143); beyond that the stratified
    # permutation walk need not terminate, so the library refuses instead of launching
    with pytest.raises(yk.YukiError, match="beyond the sampler"):
        it.render_tiles_accumulating(sc, cam, smp, tiles, first + 8, n_passes=2)
    with pytest.raises(yk.YukiError, match="beyond the sampler"):
        it.render_tiles_accumulating(sc, cam, smp, tiles, np.full(len(tiles), 9, dtype=np.uint16))
