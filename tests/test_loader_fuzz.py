"""Malformed input never takes the parsers down: mutated PLY / pbrt-v3 / PNG / BMP / TGA / PPM / QOI / farbfeld / EXR files are either
loaded or rejected with an error (the reference returns LoadError or panics; a C ABI must not
crash).  The same corpus is run under AddressSanitizer by tools/asan/run.sh (CPU build)."""
import os
import random

import numpy as np

from yuki_amd import loaders
from yuki_amd._ffi import YukiError

import scene_files as sf


IMAGE_EXTS = ["bmp", "tga", "ppm", "qoi", "ff", "exr"]


def _seeds(d):
    sf.write_ascii_ply(os.path.join(d, "a.ply"))
    sf.write_binary_ply(os.path.join(d, "b.ply"))
    sf.write_binary_ply(os.path.join(d, "c.ply"), ">")
    sf.write_png(os.path.join(d, "p.png"), sf.test_pattern(9, 7))
    sf.write_png(os.path.join(d, "q.png"), sf.test_pattern(9, 7, 16), depth=16, interlace=True, alpha=True)
    pal = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9], [10, 11, 12]], dtype=np.uint8)
    sf.write_png(os.path.join(d, "r.png"), (np.arange(35).reshape(5, 7) % 4).astype(np.uint8), depth=2, palette=pal)
    pb = sf.write_scene(d)
    tx = sf.write_textured_scene(d)
    img = sf.test_pattern(9, 7).astype(np.uint8)
    sf.write_bmp(os.path.join(d, "s.bmp"), img)
    sf.write_bmp(os.path.join(d, "t.bmp"), (np.arange(35).reshape(5, 7) % 4).astype(np.uint8), bits=4, palette=pal)
    sf.write_bmp(os.path.join(d, "u.bmp"), img, bits=32, bitfields=(0xFF, 0xFF00, 0xFF0000), header=108)
    sf.write_tga(os.path.join(d, "s.tga"), img, rle=True, alpha=True)
    sf.write_tga(os.path.join(d, "t.tga"), (np.arange(35).reshape(5, 7) % 4).astype(np.uint8), cmap=pal, rle=True)
    sf.write_ppm(os.path.join(d, "s.ppm"), img)
    sf.write_ppm(os.path.join(d, "t.ppm"), img, ascii=True)
    sf.write_qoi(os.path.join(d, "s.qoi"), img)
    sf.write_farbfeld(os.path.join(d, "s.ff"), img.astype(np.int64) * 257)
    sf.write_exr(os.path.join(d, "s.exr"), img.astype(np.float32), compression=3, half=True, extra_channels=("A",))
    sf.write_exr(os.path.join(d, "t.exr"), img.astype(np.float32))
    rd = lambda p: open(p, "rb").read()
    out = {
        "ply": [rd(os.path.join(d, n)) for n in ("a.ply", "b.ply", "c.ply")],
        "png": [rd(os.path.join(d, n)) for n in ("p.png", "q.png", "r.png")],
        "pbrt": [rd(pb), rd(tx)],
    }
    for ext, names in (("bmp", "stu"), ("tga", "st"), ("ppm", "st"), ("qoi", "s"), ("ff", "s"), ("exr", "st")):
        out[ext] = [rd(os.path.join(d, f"{n}.{ext}")) for n in names]
    return out


def _mutate(rng, b):
    b = bytearray(b)
    for _ in range(rng.randint(1, 6)):
        k = rng.random()
        if k < 0.4 and b:
            b[rng.randrange(len(b))] = rng.randrange(256)
        elif k < 0.6 and b:
            p = rng.randrange(len(b))
            del b[p : p + rng.randint(1, 40)]
        elif k < 0.8:
            p = rng.randrange(len(b) + 1)
            b[p:p] = bytes(rng.randrange(256) for _ in range(rng.randint(1, 20)))
        elif b:
            p = rng.randrange(len(b))
            b[p : p + 4] = rng.choice([b"\xff\xff\xff\xff", b"\x00\x00\x00\x00", b"\x7f\xff\xff\xff", b"9999", b"-1  "])
    return bytes(b)


def write_corpus(d, seed=1, count=1000):
    """Mutated files fzNNNNN.{ply,png,pbrt} next to the seed files (so Include / plymesh /
    imagemap references inside mutated pbrt files still resolve)."""
    rng = random.Random(seed)
    seeds = _seeds(d)
    paths = []
    for i in range(count):
        kind = rng.choice(["ply", "png", "pbrt", "ply", "png", "pbrt"] + IMAGE_EXTS)
        p = os.path.join(d, f"fz{i:05d}.{kind}")
        with open(p, "wb") as f:
            f.write(_mutate(rng, rng.choice(seeds[kind])))
        paths.append(p)
    return paths


def test_mutated_files_are_loaded_or_rejected(tmp_path):
    loaded = rejected = 0
    for p in write_corpus(str(tmp_path), seed=3, count=900):
        try:
            if p.endswith(".ply"):
                loaders.load_ply(p)
            elif p.rsplit(".", 1)[-1] in ["png"] + IMAGE_EXTS:
                loaders.load_image_texture(p)
            else:
                loaders.load_pbrt(p)
            loaded += 1
        except YukiError as e:
            assert e.status in (1, 5), (p, e)  # INVALID_ARGUMENT or UNSUPPORTED, with a message
            assert str(e)
            rejected += 1
    assert loaded > 5 and rejected > 300


def test_directories_and_empty_files_are_rejected(tmp_path):
    (tmp_path / "empty.ply").write_bytes(b"")
    (tmp_path / "empty.pbrt").write_bytes(b"")
    (tmp_path / "empty.png").write_bytes(b"")
    for fn, p in ((loaders.load_ply, tmp_path), (loaders.load_pbrt, tmp_path), (loaders.load_image_texture, tmp_path), (loaders.load_ply, tmp_path / "empty.ply"),
                  (loaders.load_pbrt, tmp_path / "empty.pbrt"), (loaders.load_image_texture, tmp_path / "empty.png")):
        try:
            fn(str(p))
            raise AssertionError("accepted " + str(p))
        except YukiError:
            pass


def test_random_valid_pbrt_files_load_identically(tmp_path, oracle):
    """tools/loader_fuzz.py: random scene files in the reference's pbrt dialect (number formats,
    comments, bare single values, nested blocks, named materials, every material / light / shape
    kind, plymesh in three encodings): the product loader and the independent restatement agree on
    every array, constant and camera field, or both reject the file."""
    import sys
    import warnings

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import loader_fuzz

    warnings.filterwarnings("ignore", category=RuntimeWarning)  # 1/0 in the oracle's Scale(…, 0) inverse, as in the reference
    for seed in range(7000, 7120):
        assert loader_fuzz.check_seed(seed, str(tmp_path)) is None, seed


def test_random_valid_ply_files_load_identically(tmp_path, oracle):
    """tools/ply_fuzz.py: random PLY files (three encodings, property orders, extra properties and
    elements of every type, type-name aliases, list count / index types, polygons): both loaders
    return the same mesh or both reject."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import ply_fuzz

    for seed in range(9000, 9200):
        assert ply_fuzz.check_seed(seed, str(tmp_path)) is None, seed
