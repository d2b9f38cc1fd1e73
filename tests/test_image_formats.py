"""ImageTexture::new for the containers besides PNG (textures/image_texture.rs:66-70,114-141):
the decoder is chosen from the file extension (image::io::Reader::open) and 8-bit samples become
c/255, 16-bit c/65535, float samples stay.  Files are produced by the writers in scene_files.py
from a known pixel array, which is the expected result bit for bit; the oracle's numpy decoders
(oracle/images.py) and the library's (yk_image_formats.cpp) must both reproduce it.
Parity unpinned: the reference ships no image fixtures and the `image` crate is absent."""
import os

import numpy as np
import pytest

import scene_files as sf
from oracle import images as oi
from yuki_amd import loaders as yl
from yuki_amd._ffi import YukiError


def _u8(img):
    return img.astype(np.float32) / np.float32(255.0)


def _both(path):
    got = yl.load_image_texture(path)
    want = oi.load_image(path)
    assert got.shape == want.shape
    assert got.tobytes() == want.tobytes()
    return got


@pytest.mark.parametrize("bits,kw", [(24, {}), (24, {"top_down": True}), (32, {}), (24, {"header": 124}), (24, {"header": 12}), (32, {"bitfields": (0xFF0000, 0xFF00, 0xFF)}),
                                     (32, {"bitfields": (0xFF, 0xFF00, 0xFF0000), "header": 108}), (32, {"bitfields": (0xFF000000, 0xFF0000, 0xFF00), "header": 56})])
def test_bmp_true_colour(tmp_path, bits, kw):
    img = sf.test_pattern(37, 21).astype(np.uint8)  # 37 * 3 bytes: rows need padding
    p = str(tmp_path / "t.bmp")
    sf.write_bmp(p, img, bits=bits, **kw)
    assert _both(p).tobytes() == _u8(img).tobytes()


@pytest.mark.parametrize("bits,header", [(8, 40), (4, 40), (1, 40), (8, 12)])
def test_bmp_palette(tmp_path, bits, header):
    rng = np.random.default_rng(bits)
    n = 1 << bits
    pal = rng.integers(0, 256, size=(n if (bits < 8 or header == 12) else 200, 3), dtype=np.uint8)  # a core header has no colour count
    idx = rng.integers(0, len(pal), size=(19, 45), dtype=np.uint8)
    p = str(tmp_path / "t.bmp")
    sf.write_bmp(p, idx, bits=bits, header=header, palette=pal)
    assert _both(p).tobytes() == _u8(pal[idx]).tobytes()


@pytest.mark.parametrize("kw", [{}, {"alpha": True}, {"rle": True}, {"rle": True, "alpha": True, "top_left": True}, {"top_left": True, "id_field": b"hello"}])
def test_tga_true_colour(tmp_path, kw):
    img = sf.test_pattern(40, 23).astype(np.uint8)
    img[5:9, :] = img[5, 0]  # long runs for the RLE packets
    p = str(tmp_path / "t.tga")
    sf.write_tga(p, img, **kw)
    assert _both(p).tobytes() == _u8(img).tobytes()


@pytest.mark.parametrize("rle", [False, True])
def test_tga_colour_mapped_and_grey(tmp_path, rle):
    rng = np.random.default_rng(5)
    cmap = rng.integers(0, 256, size=(77, 3), dtype=np.uint8)
    idx = rng.integers(0, 77, size=(16, 31), dtype=np.uint8)
    idx[3, :] = 9
    p = str(tmp_path / "t.tga")
    sf.write_tga(p, idx, cmap=cmap, rle=rle)
    assert _both(p).tobytes() == _u8(cmap[idx]).tobytes()
    # a grey file (type 3) decodes to Luma8 in the `image` crate: the reference's "Unsupported image format"
    raw = bytearray(open(p, "rb").read())
    grey = bytes([0, 0, 11 if rle else 3]) + bytes(raw[3:12]) + bytes(raw[12:18])
    g = str(tmp_path / "g.tga")
    open(g, "wb").write(grey + bytes(16 * 31))
    with pytest.raises(YukiError, match="Unsupported image format"):
        yl.load_image_texture(g)
    with pytest.raises(oi.ImageError, match="Unsupported image format"):
        oi.load_image(g)


@pytest.mark.parametrize("maxval,ascii", [(255, False), (65535, False), (255, True), (65535, True)])
def test_ppm(tmp_path, maxval, ascii):
    img = sf.test_pattern(29, 17, depth=8 if maxval == 255 else 16)
    p = str(tmp_path / ("t.ppm" if not ascii else "t.pnm"))
    sf.write_ppm(p, img, maxval=maxval, ascii=ascii)
    want = img.astype(np.float32) / np.float32(maxval)
    assert _both(p).tobytes() == want.tobytes()


def test_pgm_is_unsupported_like_luma(tmp_path):
    p = str(tmp_path / "t.pgm")
    open(p, "wb").write(b"P5\n4 4\n255\n" + bytes(16))
    with pytest.raises(YukiError, match="Unsupported image format"):
        yl.load_image_texture(p)
    with pytest.raises(oi.ImageError, match="Unsupported image format"):
        oi.load_image(p)


@pytest.mark.parametrize("alpha", [False, True])
def test_qoi(tmp_path, alpha):
    img = sf.test_pattern(48, 30).astype(np.uint8)
    img[4:12, :] = img[4, 0]      # runs (longer than 62)
    img[12:16] = img[12:16] // 64 * 64  # few colours: index hits
    img[16:20, 1:] = img[16:20, :-1] + 1  # small differences
    a = (sf.test_pattern(48, 30, seed=9)[:, :, 2] | 128).astype(np.uint8) if alpha else None
    p = str(tmp_path / "t.qoi")
    sf.write_qoi(p, img, alpha=a)
    assert _both(p).tobytes() == _u8(img).tobytes()


def test_farbfeld(tmp_path):
    img = sf.test_pattern(21, 13, depth=16)
    p = str(tmp_path / "t.ff")
    sf.write_farbfeld(p, img)
    assert _both(p).tobytes() == (img.astype(np.float32) / np.float32(65535.0)).tobytes()


@pytest.mark.parametrize("compression,half,extra,origin", [(0, False, (), (0, 0)), (2, False, (), (0, 0)), (3, False, ("A",), (0, 0)), (3, True, ("A", "Z"), (-7, 5)), (0, True, (), (3, 3))])
def test_exr(tmp_path, compression, half, extra, origin):
    rng = np.random.default_rng(11)
    img = (rng.random((37, 29, 3), dtype=np.float32) * np.float32(4.0)).astype(np.float32)
    img[0, 0] = (0.0, 1e-6, 65000.0)
    img[3:20, :, :] = np.float32(0.5)  # compressible
    if half:
        img = img.astype(np.float16).astype(np.float32)
    p = str(tmp_path / "t.exr")
    sf.write_exr(p, img, compression=compression, half=half, extra_channels=extra, data_origin=origin)
    assert _both(p).tobytes() == img.tobytes()


def test_exr_written_by_the_library_reads_back(tmp_path):
    """yk_write_exr output (the film writer) is a valid texture input."""
    from yuki_amd import core as yk

    rng = np.random.default_rng(2)
    img = rng.random((9, 14, 3), dtype=np.float32)
    p = str(tmp_path / "film.exr")
    yk.write_exr(p, img)
    assert _both(p).tobytes() == img.tobytes()


@pytest.mark.parametrize("name", ["x.jpg", "x.gif", "x.hdr", "x.webp", "x", "x.dat"])
def test_formats_that_are_not_implemented_fail_loudly(tmp_path, name):
    p = str(tmp_path / name)
    open(p, "wb").write(b"\xff\xd8\xff\xe0" + bytes(64))
    with pytest.raises(YukiError, match="not implemented|could not be determined"):
        yl.load_image_texture(p)
    with pytest.raises(oi.ImageError):
        oi.load_image(p)


@pytest.mark.parametrize("ext,writer", [("bmp", lambda p, i: sf.write_bmp(p, i)), ("tga", lambda p, i: sf.write_tga(p, i, rle=True)), ("ppm", lambda p, i: sf.write_ppm(p, i)),
                                        ("qoi", lambda p, i: sf.write_qoi(p, i)), ("ff", lambda p, i: sf.write_farbfeld(p, i.astype(np.int64) * 257)),
                                        ("exr", lambda p, i: sf.write_exr(p, i.astype(np.float32), compression=3))])
def test_truncated_and_corrupt_files_are_errors_not_crashes(tmp_path, ext, writer):
    img = sf.test_pattern(33, 19).astype(np.uint8)
    p = str(tmp_path / f"t.{ext}")
    writer(p, img)
    data = open(p, "rb").read()
    rng = np.random.default_rng(7)
    q = str(tmp_path / f"bad.{ext}")
    for cut in (0, 1, 7, 13, 17, 25, 40, len(data) // 2, len(data) - 9):
        open(q, "wb").write(data[:cut])
        with pytest.raises(YukiError):
            yl.load_image_texture(q)
    for _ in range(60):  # byte flips: either an error or some image, never a crash
        b = bytearray(data)
        for k in rng.integers(0, min(len(b), 96), size=3):
            b[k] = int(rng.integers(0, 256))
        open(q, "wb").write(bytes(b))
        try:
            yl.load_image_texture(q)
        except YukiError:
            pass


def test_pbrt_scene_with_textures_in_other_containers(tmp_path, oracle):
    """`Texture "imagemap"` files are decoded by extension: the same scene with its three PNGs
    re-encoded as TGA (RLE), 16-bit PPM and BMP (palette) loads to the same textures."""
    from oracle import loaders as ol
    from test_loaders import assert_same_scene

    p = sf.write_textured_scene(str(tmp_path))
    base, _, _ = yl.load_pbrt(p)
    tex = os.path.join(str(tmp_path), "tex")
    checks = (base.textures[0] * np.float32(255.0)).round().astype(np.uint8)
    sf.write_tga(os.path.join(tex, "checks.tga"), checks, rle=True, alpha=True)
    deep = (base.textures[1] * np.float32(65535.0)).round().astype(np.int64)
    sf.write_ppm(os.path.join(tex, "deep16.ppm"), deep, maxval=65535)
    pal = np.array([[255, 40, 30], [20, 200, 60], [0, 0, 0], [250, 250, 240], [30, 60, 220]], dtype=np.uint8)
    idx = (np.add.outer(np.arange(11), np.arange(13)) % 5).astype(np.uint8)
    sf.write_bmp(os.path.join(tex, "pal.bmp"), idx, bits=4, palette=pal)
    q = os.path.join(str(tmp_path), "other.pbrt")
    open(q, "w").write(open(p).read().replace("checks.png", "checks.tga").replace("deep16.png", "deep16.ppm").replace("pal.png", "pal.bmp"))
    got, _, _ = yl.load_pbrt(q)
    want, _, _ = ol.load_pbrt(q)
    assert_same_scene(want, got)
    assert len(got.textures) == 3
    for a, b in zip(got.textures, base.textures):
        assert a.tobytes() == b.tobytes()


@pytest.mark.parametrize("seed", range(60))
def test_random_images_in_every_container(tmp_path, seed):
    """Random sizes (1 x 1 up), contents and encoder options for every container: both decoders return the
    source pixels exactly."""
    r = np.random.default_rng(seed)
    w, h = int(r.integers(1, 41)), int(r.integers(1, 31))
    img = r.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if r.random() < 0.5:
        img[:, : w // 2] = img[0, 0]  # runs
    img16 = r.integers(0, 65536, (h, w, 3)).astype(np.int64)
    d = str(tmp_path)
    cases = []
    p = os.path.join(d, "a.bmp")
    bits = int(r.choice([24, 32]))
    sf.write_bmp(p, img, bits=bits, top_down=bool(r.integers(0, 2)), header=int(r.choice([40, 52, 56, 108, 124])) if bits == 24 else 40)
    cases.append((p, _u8(img)))
    pal = r.integers(0, 256, (int(r.integers(2, 17)), 3), dtype=np.uint8)
    idx = r.integers(0, len(pal), (h, w), dtype=np.uint8)
    p = os.path.join(d, "b.bmp")
    sf.write_bmp(p, idx, bits=int(r.choice([4, 8])), palette=pal, top_down=bool(r.integers(0, 2)))
    cases.append((p, _u8(pal[idx])))
    p = os.path.join(d, "a.tga")
    sf.write_tga(p, img, alpha=bool(r.integers(0, 2)), rle=bool(r.integers(0, 2)), top_left=bool(r.integers(0, 2)), id_field=bytes(r.integers(0, 256, int(r.integers(0, 9)), dtype=np.uint8)))
    cases.append((p, _u8(img)))
    p = os.path.join(d, "b.tga")
    sf.write_tga(p, idx, cmap=pal, rle=bool(r.integers(0, 2)), top_left=bool(r.integers(0, 2)))
    cases.append((p, _u8(pal[idx])))
    p = os.path.join(d, "a.ppm")
    deep = bool(r.integers(0, 2))
    sf.write_ppm(p, img16 if deep else img, maxval=65535 if deep else 255, ascii=bool(r.integers(0, 2)), comment=bool(r.integers(0, 2)))
    cases.append((p, (img16 if deep else img).astype(np.float32) / np.float32(65535 if deep else 255)))
    p = os.path.join(d, "a.qoi")
    sf.write_qoi(p, img, alpha=r.integers(0, 256, (h, w), dtype=np.uint8) if r.random() < 0.5 else None)
    cases.append((p, _u8(img)))
    p = os.path.join(d, "a.ff")
    sf.write_farbfeld(p, img16)
    cases.append((p, img16.astype(np.float32) / np.float32(65535.0)))
    f = (r.random((h, w, 3), dtype=np.float32) * np.float32(r.choice([1.0, 100.0, 1e-3]))).astype(np.float32)
    half = bool(r.integers(0, 2))
    if half:
        f = f.astype(np.float16).astype(np.float32)
    p = os.path.join(d, "a.exr")
    sf.write_exr(p, f, compression=int(r.choice([0, 2, 3])), half=half, extra_channels=tuple(r.choice(["A", "Z", "Y"], int(r.integers(0, 3)), replace=False)),
                 data_origin=(int(r.integers(-9, 9)), int(r.integers(-9, 9))))
    cases.append((p, f))
    p = os.path.join(d, "a.png")
    sf.write_png(p, img.astype(np.int64), alpha=bool(r.integers(0, 2)), interlace=bool(r.integers(0, 2)))
    cases.append((p, _u8(img)))
    for path, want in cases:
        assert _both(path).tobytes() == want.tobytes(), path
