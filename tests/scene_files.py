"""Generators for the PLY / pbrt-v3 files the loader tests read (the reference ships
no sample scenes; these are synthetic inputs written at test time)."""
import os
import struct

CUBE_V = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 2), (1, 0, 2), (1, 1, 2), (0, 1, 2)]
CUBE_F = [(0, 3, 2, 1), (4, 5, 6, 7), (0, 1, 5, 4), (1, 2, 6, 5), (2, 3, 7, 6), (3, 0, 4, 7)]


def write_ascii_ply(path, verts=CUBE_V, faces=CUBE_F, index_name="vertex_indices", index_type="int", fmt="{:.9g}", scale=1.0):
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment synthetic\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n" % len(verts))
        f.write("element face %d\nproperty list uchar %s %s\nend_header\n" % (len(faces), index_type, index_name))
        for v in verts:
            f.write(" ".join(fmt.format(c * scale) for c in v) + "\n")
        for fc in faces:
            f.write("%d %s\n" % (len(fc), " ".join(map(str, fc))))


def write_binary_ply(path, endian="<", verts=CUBE_V, faces=CUBE_F, normals=True, uvs=True, extra=True):
    """binary cube with optional normals / uv and properties the loader must skip
    (a double, a uchar, a per-face int)."""
    name = "binary_little_endian" if endian == "<" else "binary_big_endian"
    h = "ply\nformat %s 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n" % (name, len(verts))
    if normals:
        h += "property float nx\nproperty float ny\nproperty float nz\n"
    if extra:
        h += "property double quality\n"
    if uvs:
        h += "property float u\nproperty float v\n"
    if extra:
        h += "property uchar red\n"
    h += "element face %d\nproperty list uchar uint vertex_index\n" % len(faces)
    if extra:
        h += "property int flags\n"
    h += "end_header\n"
    b = h.encode()
    for i, v in enumerate(verts):
        b += struct.pack(endian + "3f", *[c * 0.3 + 0.01 * i for c in v])
        if normals:
            n = [c * 2 - 1 for c in v]
            l = sum(c * c for c in n) ** 0.5
            b += struct.pack(endian + "3f", *[c / l for c in n])
        if extra:
            b += struct.pack(endian + "d", i * 0.5)
        if uvs:
            b += struct.pack(endian + "2f", v[0] * 0.5, v[1] * 0.25)
        if extra:
            b += struct.pack(endian + "B", i)
    for fc in faces:
        b += struct.pack(endian + "B%dI" % len(fc), len(fc), *fc)
        if extra:
            b += struct.pack(endian + "i", 7)
    with open(path, "wb") as f:
        f.write(b)


def uv_sphere(nu, nv, r=1.0):
    import math

    verts, faces = [], []
    for j in range(nv + 1):
        th = math.pi * j / nv
        for i in range(nu):
            ph = 2 * math.pi * i / nu
            verts.append((r * math.sin(th) * math.cos(ph), r * math.sin(th) * math.sin(ph), r * math.cos(th)))
    for j in range(nv):
        for i in range(nu):
            a, b = j * nu + i, j * nu + (i + 1) % nu
            faces.append((a, a + nu, b + nu, b))  # outward winding
    return verts, faces


SCENE_PBRT = """# synthetic test scene: every directive the reference's loader implements
LookAt 3 4 1.5  .5 .5 0  0 0 1
Camera "perspective" "float fov" [ 39.5 ]
Sampler "halton" "integer pixelsamples" 16
Integrator "path" "integer maxdepth" [5]
Film "image" "integer xresolution" [96] "integer yresolution" [64] "string filename" "out.exr"
WorldBegin
LightSource "infinite" "rgb L" [0.1 0.2 0.3]
LightSource "distant" "point from" [1 2 3] "point to" [0 0 0] "rgb L" [3 3 2.5]
AttributeBegin
  Translate 0 0 4.5
  AreaLightSource "diffuse" "rgb L" [1 1 1]
  LightSource "point" "color I" [40 41 42] "point from" [0.25 0.5 0.125]
  LightSource "point" "color I" [0 0 0]
  LightSource "spot" "color I" [1 1 1]
AttributeEnd
MakeNamedMaterial "shiny" "string type" "metal" "float roughness" 0.05
MakeNamedMaterial "cu2" "string type" "metal" "spectrum eta" [400 1.1 500 1.2 600 0.9 700 0.3] "rgb k" [3 2.5 2] "bool remaproughness" "false"
MakeNamedMaterial "cu3" "string type" "metal" "spectrum eta" "geo/eta.spd" "blackbody k" [6500 1]
MakeNamedMaterial "glassy" "string type" "glass" "float eta" 1.33 "rgb Kt" [.9 .95 1]
Material "matte" "rgb Kd" [.6 .5 .4] "float sigma" 20
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-5 -5 0 5 -5 0 5 5 0 -5 5 0]
   "normal N" [0 0 1 0 0 1 0 0 1 0 0 1] "float uv" [0 0 1 0 1 1 0 1]
AttributeBegin
  NamedMaterial "shiny"
  Translate 1 0.5 0.75
  Rotate 30 0 1 1
  Scale 0.75 0.75 -0.75
  Shape "sphere" "float radius" 1.0
AttributeEnd
AttributeBegin
  NamedMaterial "glassy"
  TransformBegin
  Translate -1.5 0.25 0.5
  Shape "sphere" "float radius" 0.5
  TransformEnd
AttributeEnd
AttributeBegin
  Material "glossy" "rgb Rs" [.3 .7 .2] "float roughness" 0.2
  Translate -0.5 -1.5 0.0
  Rotate -40 0 0 1
  Include "geo/inc.pbrt"
AttributeEnd
NamedMaterial "nope"
Shape "cone"
Shape "trianglemesh" "integer indices" [0 1] "point P" [0 0 0 1 1 1 2 2 2]
AttributeBegin
NamedMaterial "cu2"
Translate 2 -2 0
Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 1 ]
NamedMaterial "cu3"
Shape "trianglemesh" "integer indices" [0 2 1] "point P" [0 0 1 1 0 1 0 1 2 ]
AttributeEnd
WorldEnd
"""

INC_PBRT = """# included from scene.pbrt
Shape "plymesh" "string filename" "cube_n.ply"
Material "plastic"
Shape "plymesh" "string filename" "../cube.ply"
ActiveTransform All
"""

ETA_SPD = "# wavelength value\n400 1.1\n500 1.25 # trailing\n\n600 0.85\n700 0.35\n"


def write_scene(dirname):
    """scene.pbrt + geo/{inc.pbrt,cube_n.ply,eta.spd} + cube.ply ; returns the .pbrt path."""
    os.makedirs(os.path.join(dirname, "geo"), exist_ok=True)
    write_ascii_ply(os.path.join(dirname, "cube.ply"))
    write_binary_ply(os.path.join(dirname, "geo", "cube_n.ply"))
    for name, text in (("scene.pbrt", SCENE_PBRT), ("geo/inc.pbrt", INC_PBRT), ("geo/eta.spd", ETA_SPD)):
        with open(os.path.join(dirname, name), "w") as f:
            f.write(text)
    return os.path.join(dirname, "scene.pbrt")


def write_render_scene(dirname, nu=24, nv=12):
    """A small closed scene for render parity: floor + tessellated ply spheres + analytic
    spheres, all material kinds, point + distant lights + background."""
    os.makedirs(dirname, exist_ok=True)
    v, f = uv_sphere(nu, nv, 0.8)
    write_ascii_ply(os.path.join(dirname, "ball.ply"), v, f, index_name="vertex_index", index_type="uint")
    text = """LookAt 0 -6 2.5  0 0 0.8  0 0 1
Camera "perspective" "float fov" 42
Film "image" "integer xresolution" [80] "integer yresolution" [60]
WorldBegin
LightSource "infinite" "rgb L" [0.25 0.3 0.4]
LightSource "distant" "point from" [2 -3 4] "point to" [0 0 0] "rgb L" [2 2 1.8]
LightSource "point" "rgb I" [30 25 20] "point from" [-2 -1 4]
Material "matte" "rgb Kd" [.55 .5 .45]
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-6 -6 0 6 -6 0 6 6 0 -6 6 0]
AttributeBegin
  Material "matte" "rgb Kd" [.7 .3 .3] "float sigma" 2000
  Translate -1.8 0.5 0.8
  Shape "plymesh" "string filename" "ball.ply"
AttributeEnd
AttributeBegin
  Material "metal" "float roughness" 0.2 "rgb eta" [0.2 0.9 1.1] "rgb k" [3.9 2.4 2.2]
  Translate 0 1.5 0.8
  Rotate 25 0 0 1
  Scale 1 1 1.2
  Shape "plymesh" "string filename" "ball.ply"
AttributeEnd
AttributeBegin
  Material "glass" "float eta" 1.5
  Translate 1.7 -0.5 0.7
  Shape "sphere" "float radius" 0.7
AttributeEnd
AttributeBegin
  Material "glossy" "rgb Rs" [.4 .6 .3] "float roughness" 0.3
  Translate -0.2 -1.6 0.45
  Shape "sphere" "float radius" 0.45
AttributeEnd
WorldEnd
"""
    p = os.path.join(dirname, "render.pbrt")
    with open(p, "w") as fh:
        fh.write(text)
    return p


# ----------------------------------------------------------------------------- PNG
def write_png(path, img, depth=8, alpha=False, palette=None, interlace=False, filters=(0, 1, 2, 3, 4), level=6, fixed=False, gray=False, trns=False):
    """Minimal PNG encoder for test inputs.  img: (h, w, 3) integer array (values < 2**depth),
    or (h, w) palette indices when `palette` ((n,3) uint8) is given.  Scanline filter types
    cycle through `filters`."""
    import zlib

    import numpy as np

    img = np.asarray(img)
    h, w = img.shape[:2]
    if palette is not None:
        ctype, channels = 3, 1
    elif gray:
        ctype, channels = (4 if alpha else 0), (2 if alpha else 1)
        img = img[..., :1]
    else:
        ctype, channels = (6 if alpha else 2), (4 if alpha else 3)
    if alpha:
        a = np.full(img.shape[:2] + (1,), (1 << depth) - 1 - 3, dtype=img.dtype)
        img = np.concatenate([img.reshape(h, w, -1), a], axis=2)

    def pack_rows(sub):
        sh, sw = sub.shape[:2]
        if palette is not None:
            bits = ((sub.reshape(sh, sw, 1).astype(np.uint8) >> np.arange(depth - 1, -1, -1)) & 1).reshape(sh, sw * depth)
            pad = (-bits.shape[1]) % 8
            bits = np.pad(bits, ((0, 0), (0, pad)))
            return np.packbits(bits.astype(np.uint8), axis=1)
        if depth == 8:
            return sub.reshape(sh, sw * channels).astype(np.uint8)
        v = sub.reshape(sh, sw * channels).astype(np.uint16)
        return np.stack([(v >> 8).astype(np.uint8), (v & 0xFF).astype(np.uint8)], axis=2).reshape(sh, -1)

    def paeth(a, b, c):
        p = a + b - c
        pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
        return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)

    bpp = max(1, channels * depth // 8)
    raw = bytearray()
    fcount = 0

    def emit(sub):
        nonlocal fcount
        rows = pack_rows(sub).astype(np.int64)
        prev = np.zeros(rows.shape[1], dtype=np.int64)
        for line in rows:
            ft = filters[fcount % len(filters)]
            fcount += 1
            out = np.zeros_like(line)
            for x in range(len(line)):
                a = line[x - bpp] if x >= bpp else 0
                b = prev[x]
                c = prev[x - bpp] if x >= bpp else 0
                pred = [0, a, b, (a + b) >> 1, paeth(a, b, c)][ft]
                out[x] = (line[x] - pred) & 0xFF
            raw.append(ft)
            raw.extend(out.astype(np.uint8).tobytes())
            prev = line

    if interlace:
        for x0, y0, dx, dy in [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]:
            sub = img[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                emit(sub)
    else:
        emit(img)
    if fixed:
        co = zlib.compressobj(level, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
        comp = co.compress(bytes(raw)) + co.flush()
    else:
        comp = zlib.compress(bytes(raw), level)

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body))

    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    out += chunk(b"gAMA", struct.pack(">I", 45455))  # must be ignored
    if palette is not None:
        out += chunk(b"PLTE", np.asarray(palette, dtype=np.uint8).tobytes())
    if trns:
        out += chunk(b"tRNS", bytes([0, 128]) if palette is not None else struct.pack(">3H", 1, 2, 3))
    half = len(comp) // 2
    out += chunk(b"IDAT", comp[:half]) + chunk(b"IDAT", comp[half:]) + chunk(b"tEXt", b"Comment\0synthetic") + chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(out)


def test_pattern(w, h, depth=8, seed=3):
    """Deterministic colourful (h, w, 3) integer image with smooth and noisy regions."""
    import numpy as np

    y, x = np.mgrid[0:h, 0:w]
    m = (1 << depth) - 1
    r = (x * m) // max(1, w - 1)
    g = (y * m) // max(1, h - 1)
    b = ((x * 2654435761 + y * 40503 + seed * 97) >> 3) & m
    chk = (((x // 4) + (y // 4)) & 1) * (m // 3)
    return np.stack([r, (g + chk) & m, b], axis=2).astype(np.int64)


TEXTURED_PBRT = """LookAt 0 -5 3  0 0 0.6  0 0 1
Camera "perspective" "float fov" 45
Film "image" "integer xresolution" [72] "integer yresolution" [54]
WorldBegin
LightSource "infinite" "rgb L" [0.35 0.35 0.4]
LightSource "point" "rgb I" [40 38 35] "point from" [1 -2 5]
Texture "checks" "spectrum" "imagemap" "string filename" "tex/checks.png"
Texture "deep" "spectrum" "imagemap" "string filename" "tex/deep16.png"
Texture "unused" "float" "scale" "float tex1" 2
Material "matte" "texture Kd" "checks"
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-4 -4 0 4 -4 0 4 4 0 -4 4 0]
   "float uv" [-1.25 -0.5 2.75 -0.5 2.75 3.5 -1.25 3.5]
AttributeBegin
  Material "matte" "texture Kd" "deep" "float sigma" 1500
  Translate -1.2 0.2 0.8
  Rotate 35 0 0 1
  Shape "sphere" "float radius" 0.8
AttributeEnd
AttributeBegin
  Material "matte" "texture Kd" "checks"
  Translate 1.3 0.4 0.7
  Shape "plymesh" "string filename" "ball.ply"
AttributeEnd
AttributeBegin
  Texture "checks" "spectrum" "imagemap" "string filename" "tex/pal.png"
  Material "matte" "texture Kd" "checks"
  Translate 0.2 -1.6 0.5
  Scale 0.5 0.5 -0.5
  Shape "plymesh" "string filename" "ball.ply"
AttributeEnd
WorldEnd
"""


def write_textured_scene(dirname):
    import numpy as np

    os.makedirs(os.path.join(dirname, "tex"), exist_ok=True)
    v, f = uv_sphere(20, 10, 0.7)
    write_ascii_ply(os.path.join(dirname, "ball.ply"), v, f)
    img = test_pattern(37, 23)
    img[5:9, 7:20] = 0  # a black patch: matte adds no lobe there (matte.rs:31)
    write_png(os.path.join(dirname, "tex", "checks.png"), img, alpha=True, filters=(4, 1, 3, 2, 0))
    write_png(os.path.join(dirname, "tex", "deep16.png"), test_pattern(16, 16, depth=16), depth=16, interlace=True)
    pal = np.array([[255, 40, 30], [20, 200, 60], [0, 0, 0], [250, 250, 240], [30, 60, 220]], dtype=np.uint8)
    idx = (np.add.outer(np.arange(11), np.arange(13)) % 5).astype(np.uint8)
    write_png(os.path.join(dirname, "tex", "pal.png"), idx, depth=4, palette=pal)
    p = os.path.join(dirname, "textured.pbrt")
    with open(p, "w") as fh:
        fh.write(TEXTURED_PBRT)
    return p


# --------------------------------------------------------------------------- other image containers
# Writers for the formats yk_image_formats.cpp reads, written from the format specifications
# (independently of both decoders): the source pixel array is the ground truth of the tests.
def write_bmp(path, img, bits=24, top_down=False, header=40, bitfields=None, palette=None):
    """img: (h, w, 3) u8 (bits 24/32) or (h, w) palette indices (bits 1/4/8, palette (n, 3) rgb).
    bitfields = (rmask, gmask, bmask) makes a 32-bit BI_BITFIELDS file."""
    import struct

    import numpy as np

    h, w = img.shape[:2]
    stride = (w * bits + 31) // 32 * 4
    rows = np.zeros((h, stride), dtype=np.uint8)
    if bits <= 8:
        b = np.zeros((h, stride * 8), dtype=np.uint8)
        for k in range(bits):
            b[:, k : w * bits : bits] = (img >> (bits - 1 - k)) & 1
        rows[:] = np.packbits(b, axis=1)
    elif bits == 24:
        rows[:, : w * 3] = img[:, :, ::-1].reshape(h, w * 3)
    else:
        if bitfields:
            sh = [(m & -m).bit_length() - 1 for m in bitfields]
            v = sum(img[:, :, k].astype(np.uint32) << sh[k] for k in range(3)) | np.uint32(0x5A << [s for s in (0, 8, 16, 24) if s not in sh][0])
        else:
            v = (img[:, :, 0].astype(np.uint32) << 16) | (img[:, :, 1].astype(np.uint32) << 8) | img[:, :, 2].astype(np.uint32) | np.uint32(0x7F000000)
        rows[:, : w * 4] = v.astype("<u4").view(np.uint8).reshape(h, w * 4)
    if not top_down:
        rows = rows[::-1]
    pal = b""
    if bits <= 8:
        esz = 3 if header == 12 else 4
        pal = b"".join(bytes([p[2], p[1], p[0]]) + (b"\0" if esz == 4 else b"") for p in palette)
    comp = 3 if bitfields else 0
    if header == 12:
        hd = struct.pack("<IHHHH", 12, w, h, 1, bits)
    else:
        hd = struct.pack("<IiiHHIIiiII", header, w, -h if top_down else h, 1, bits, comp, stride * h, 2835, 2835, len(palette) if (bits <= 8 and len(palette) != 1 << bits) else 0, 0)
        extra = header - 40
        masks = struct.pack("<III", *bitfields) if bitfields else b""
        if header == 40:
            hd += masks
        else:
            hd += (masks + b"\0" * extra)[:extra] if bitfields else b"\0" * extra
    off = 14 + len(hd) + len(pal)
    with open(path, "wb") as f:
        f.write(b"BM" + struct.pack("<IHHI", off + stride * h, 0, 0, off) + hd + pal + rows.tobytes())


def write_tga(path, img, alpha=False, rle=False, top_left=False, cmap=None, id_field=b""):
    """img (h, w, 3) u8 true colour, or (h, w) indices with cmap (n, 3) rgb."""
    import struct

    import numpy as np

    h, w = img.shape[:2]
    if cmap is not None:
        px = img.astype(np.uint8).reshape(h, w, 1)
        cm = b"".join(bytes([c[2], c[1], c[0]]) for c in cmap)
        typ, cmt, cm_len, cm_bits, depth = 1, 1, len(cmap), 24, 8
    else:
        px = img[:, :, ::-1].astype(np.uint8)
        if alpha:
            px = np.concatenate([px, np.full((h, w, 1), 200, dtype=np.uint8)], axis=2)
        cm, typ, cmt, cm_len, cm_bits, depth = b"", 2, 0, 0, 0, 32 if alpha else 24
    if not top_left:
        px = px[::-1]
    pb = px.shape[2]
    flat = px.reshape(-1, pb)
    if rle:
        typ += 8
        out, i, n = bytearray(), 0, len(flat)
        while i < n:
            run = 1
            while i + run < n and run < 128 and (flat[i + run] == flat[i]).all():
                run += 1
            if run > 1:
                out += bytes([0x80 | (run - 1)]) + flat[i].tobytes()
                i += run
            else:
                lit = 1
                while i + lit < n and lit < 128 and not (i + lit + 1 < n and (flat[i + lit] == flat[i + lit + 1]).all()):
                    lit += 1
                out += bytes([lit - 1]) + flat[i : i + lit].tobytes()
                i += lit
        body = bytes(out)
    else:
        body = flat.tobytes()
    hd = struct.pack("<BBBHHBHHHHBB", len(id_field), cmt, typ, 0, cm_len, cm_bits, 0, 0, w, h, depth, (0x20 if top_left else 0) | (8 if alpha else 0))
    with open(path, "wb") as f:
        f.write(hd + id_field + cm + body)


def write_ppm(path, img, maxval=255, ascii=False, comment=True):
    import numpy as np

    h, w = img.shape[:2]
    head = ("P3" if ascii else "P6") + "\n" + ("# made by the tests\n" if comment else "") + f"{w} {h}\n{maxval}\n"
    with open(path, "wb") as f:
        f.write(head.encode())
        if ascii:
            f.write((" ".join(str(int(v)) for v in img.reshape(-1)) + "\n").encode())
        else:
            f.write(img.astype(">u2" if maxval > 255 else np.uint8).tobytes())


def write_qoi(path, img, alpha=None):
    """Reference-style QOI encoder (all op kinds)."""
    import struct

    import numpy as np

    h, w = img.shape[:2]
    ch = 4 if alpha is not None else 3
    px = np.concatenate([img, alpha[:, :, None]], axis=2) if alpha is not None else np.concatenate([img, np.full((h, w, 1), 255)], axis=2)
    px = px.reshape(-1, 4).astype(np.int64)
    out = bytearray(b"qoif" + struct.pack(">IIBB", w, h, ch, 0))
    seen = [(0, 0, 0, 0)] * 64
    prev, run = (0, 0, 0, 255), 0
    for i, p in enumerate(map(tuple, px)):
        if p == prev:
            run += 1
            if run == 62 or i == len(px) - 1:
                out.append(0xC0 | (run - 1))
                run = 0
            continue
        if run:
            out.append(0xC0 | (run - 1))
            run = 0
        k = (p[0] * 3 + p[1] * 5 + p[2] * 7 + p[3] * 11) % 64
        if seen[k] == p:
            out.append(k)
        else:
            seen[k] = p
            if p[3] == prev[3]:
                d = [((p[c] - prev[c] + 128) & 255) - 128 for c in range(3)]
                dr_dg, db_dg = d[0] - d[1], d[2] - d[1]
                if all(-2 <= v <= 1 for v in d):
                    out.append(0x40 | ((d[0] + 2) << 4) | ((d[1] + 2) << 2) | (d[2] + 2))
                elif -32 <= d[1] <= 31 and -8 <= dr_dg <= 7 and -8 <= db_dg <= 7:
                    out += bytes([0x80 | (d[1] + 32), ((dr_dg + 8) << 4) | (db_dg + 8)])
                else:
                    out += bytes([0xFE, p[0], p[1], p[2]])
            else:
                out += bytes([0xFF, p[0], p[1], p[2], p[3]])
        prev = p
    out += b"\0" * 7 + b"\1"
    with open(path, "wb") as f:
        f.write(bytes(out))


def write_farbfeld(path, img16):
    import struct

    import numpy as np

    h, w = img16.shape[:2]
    px = np.concatenate([img16, np.full((h, w, 1), 65535)], axis=2).astype(">u2")
    with open(path, "wb") as f:
        f.write(b"farbfeld" + struct.pack(">II", w, h) + px.tobytes())


def write_exr(path, img, compression=0, half=False, extra_channels=(), data_origin=(0, 0)):
    """Scan-line OpenEXR: img (h, w, 3) float32; compression 0 (none), 2 (ZIPS), 3 (ZIP);
    `extra_channels` adds named FLOAT channels ('A', 'Z', ...) filled with 0.25."""
    import struct
    import zlib

    import numpy as np

    h, w = img.shape[:2]
    chans = sorted([("R", img[:, :, 0]), ("G", img[:, :, 1]), ("B", img[:, :, 2])] + [(n, np.full((h, w), 0.25, dtype=np.float32)) for n in extra_channels])
    pt = 1 if half else 2

    def attr(name, typ, body):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<I", len(body)) + body

    chl = b"".join(n.encode() + b"\0" + struct.pack("<IBxxxII", pt if n in "RGB" else 2, 0, 1, 1) for n, _ in chans) + b"\0"
    x0, y0 = data_origin
    box = struct.pack("<iiii", x0, y0, x0 + w - 1, y0 + h - 1)
    hd = struct.pack("<II", 20000630, 2)
    hd += attr("channels", "chlist", chl) + attr("compression", "compression", bytes([compression]))
    hd += attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0")
    hd += attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0))
    hd += attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    lpc = 16 if compression == 3 else 1
    chunks = []
    for r0 in range(0, h, lpc):
        raw = b""
        for r in range(r0, min(h, r0 + lpc)):
            for n, a in chans:
                raw += a[r].astype("<f2" if (half and n in "RGB") else "<f4").tobytes()
        body = raw
        if compression:
            t = np.frombuffer(raw, dtype=np.uint8)
            t = np.concatenate([t[0::2], t[1::2]])
            d = t.astype(np.int64)
            d[1:] = (d[1:] - t[:-1].astype(np.int64) + 128 + 256) & 255
            z = zlib.compress(d.astype(np.uint8).tobytes(), 6)
            body = z if len(z) < len(raw) else raw
        chunks.append(struct.pack("<iI", y0 + r0, len(body)) + body)
    table_at = len(hd)
    off = table_at + 8 * len(chunks)
    table = b""
    for c in chunks:
        table += struct.pack("<Q", off)
        off += len(c)
    with open(path, "wb") as f:
        f.write(hd + table + b"".join(chunks))


# ----------------------------------------------------------------------------- BASELINE-sized scenes as files
def write_mesh_ply(path, points, faces, normals=None, uvs=None):
    """Binary little-endian PLY of one triangle mesh: float32 x y z [nx ny nz] [u v], faces as `list uchar uint`.
    The arrays are written as they are (float32 bits preserved)."""
    import numpy as np

    points = np.ascontiguousarray(points, dtype=np.float32)
    faces = np.ascontiguousarray(faces, dtype=np.uint32)
    cols = [points]
    h = "ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n" % len(points)
    if normals is not None:
        h += "property float nx\nproperty float ny\nproperty float nz\n"
        cols.append(np.ascontiguousarray(normals, dtype=np.float32))
    if uvs is not None:
        h += "property float u\nproperty float v\n"
        cols.append(np.ascontiguousarray(uvs, dtype=np.float32))
    h += "element face %d\nproperty list uchar uint vertex_indices\nend_header\n" % len(faces)
    rec = np.dtype([("n", "u1"), ("i", "<u4", (3,))])
    fr = np.zeros(len(faces), dtype=rec)
    fr["n"] = 3
    fr["i"] = faces
    with open(path, "wb") as f:
        f.write(h.encode())
        f.write(np.concatenate(cols, axis=1).astype("<f4").tobytes())
        f.write(fr.tobytes())


def write_cfg2_ply(path):
    """BASELINE configs[1]'s mesh ("Stanford-bunny PLY (~70 k tri)": here the bunny-class 69,312-triangle mesh) as the file
    Scene::ply reads: the vertices BEFORE the loader's fit-to-unit-cube transform (scene/ply.rs:99-108), so that loading the
    file reproduces scenes.bunny_class() bit for bit."""
    from yuki_amd import scenes

    raw, tris = scenes.bunny_class_raw()
    write_mesh_ply(path, raw, tris)
    return path


def _g(x):
    return "%.9g" % float(x)  # nine significant digits: a float32 survives the text round trip exactly


def write_scene_as_pbrt(dirname, sd, res=(1920, 1080), name="scene.pbrt"):
    """A generated SceneData (scenes.city: BASELINE configs[2..4]) as a pbrt-v3 file plus one binary PLY per mesh — what
    scene::pbrt::load reads (scene/pbrt/mod.rs:94-857): LookAt / Camera / Film, `LightSource "point"` and "infinite"
    (the loader has no area lights: AreaLightSource is parsed and ignored, a rectangular light cannot be written — its
    quad stays as geometry), one Attribute block with a Material and a `Shape "plymesh"` per mesh under the identity
    CTM, so the loaded vertex, normal, uv and index arrays are the generator's, bit for bit and in the same order.
    Returns (path, per-file statistics)."""
    import numpy as np
    from yuki_amd import abi

    os.makedirs(os.path.join(dirname, "meshes"), exist_ok=True)
    cam = sd.camera
    out = []
    out.append("LookAt %s  %s  %s" % (" ".join(_g(v) for v in cam["position"]), " ".join(_g(v) for v in cam["target"]), " ".join(_g(v) for v in cam["up"])))
    out.append('Camera "perspective" "float fov" %s' % _g(cam["fov_degrees"]))
    out.append('Film "image" "integer xresolution" [%d] "integer yresolution" [%d]' % res)
    out.append("WorldBegin")
    out.append('LightSource "infinite" "rgb L" [%s]' % " ".join(_g(v) for v in sd.background))
    n_lights = 1
    for l in sd.lights:
        if l["kind"] == "point":
            p = np.asarray(l["l2w"], dtype=np.float32)[:3, 3]
            out.append('LightSource "point" "rgb I" [%s] "point from" [%s]' % (" ".join(_g(v) for v in l["I"]), " ".join(_g(v) for v in p)))
            n_lights += 1
    tri_mesh = np.asarray(sd.tri_mesh)
    order = np.argsort(tri_mesh, kind="stable")
    bounds = np.searchsorted(tri_mesh[order], np.arange(len(sd.meshes) + 1))
    n_files = 0
    for m, (has_n, has_uv, _swaps) in enumerate(sd.meshes):
        tri_ids = order[bounds[m] : bounds[m + 1]]
        if len(tri_ids) == 0:
            continue
        tris = sd.indices[tri_ids].astype(np.int64)
        lo, hi = int(tris.min()), int(tris.max()) + 1
        mat = sd.materials[int(sd.tri_material[tri_ids[0]])]
        assert np.all(sd.tri_material[tri_ids] == sd.tri_material[tri_ids[0]]), "one material per mesh"
        if mat["kind"] == abi.MAT_MATTE:
            sigma = float(np.rad2deg(np.rad2deg(mat.get("c", 0.0))))  # the loader applies to_radians twice (pbrt/mod.rs:906-910)
            md = '"matte" "rgb Kd" [%s] "float sigma" %s' % (" ".join(_g(v) for v in mat["a"]), _g(sigma))
        elif mat["kind"] == abi.MAT_GLASS:
            md = '"glass" "rgb Kr" [%s] "rgb Kt" [%s] "float eta" %s' % (" ".join(_g(v) for v in mat["a"]), " ".join(_g(v) for v in mat["b"]), _g(mat["c"]))
        elif mat["kind"] == abi.MAT_METAL:
            md = '"metal" "rgb eta" [%s] "rgb k" [%s] "float roughness" %s "bool remaproughness" "%s"' % (
                " ".join(_g(v) for v in mat["a"]), " ".join(_g(v) for v in mat["b"]), _g(mat["c"]), "true" if mat.get("remap", True) else "false")
        else:
            md = '"glossy" "rgb Rs" [%s] "float roughness" %s' % (" ".join(_g(v) for v in mat["a"]), _g(mat["c"]))
        fn = "meshes/m%05d.ply" % m
        write_mesh_ply(os.path.join(dirname, fn), sd.points[lo:hi], tris - lo, sd.normals[lo:hi] if has_n else None, sd.uvs[lo:hi] if has_uv else None)
        n_files += 1
        out.append("AttributeBegin\n  Material %s\n  Shape \"plymesh\" \"string filename\" \"%s\"\nAttributeEnd" % (md, fn))
    out.append("WorldEnd")
    p = os.path.join(dirname, name)
    with open(p, "w") as f:
        f.write("\n".join(out) + "\n")
    return p, dict(ply_files=n_files, lights=n_lights)
