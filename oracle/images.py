"""TEST INFRASTRUCTURE ONLY — CPU restatement of ImageTexture::new
(yuki/src/textures/image_texture.rs:66-70,114-141) for PNG files.

The reference decodes through the `image` 0.24 crate (a Cargo dependency that is not under
/root/reference); its PNG behaviour is restated from the PNG specification: EXPAND
transformations (palette -> RGB, tRNS -> alpha), 8/16-bit RGB(A) -> c/255 or c/65535 in
f32, no gamma, alpha dropped, gray(+alpha) -> "Unsupported image format".  Decompression is
Python's zlib; filtering/interlace handling is written here independently of
yuki_amd/csrc/yk_image.cpp.  Parity unpinned: the reference ships no image fixtures.
"""
import struct
import zlib

import numpy as np


class ImageError(Exception):
    pass


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def _unfilter(raw, rows, stride, bpp):
    out = np.zeros((rows, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int64)
    pos = 0
    for y in range(rows):
        ft = raw[pos]
        line = np.frombuffer(raw, dtype=np.uint8, count=stride, offset=pos + 1).astype(np.int64)
        pos += stride + 1
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 0xFF
        elif ft in (1, 3, 4):
            cur = line.copy()
            for x in range(stride):
                a = cur[x - bpp] if x >= bpp else 0
                b = prev[x]
                c = prev[x - bpp] if x >= bpp else 0
                pred = a if ft == 1 else ((a + b) >> 1 if ft == 3 else _paeth(a, b, c))
                cur[x] = (cur[x] + pred) & 0xFF
        else:
            raise ImageError("bad filter type")
        out[y] = cur
        prev = cur
    return out, pos


def load_png(path):
    """-> (h, w, 3) float32, row 0 = top row."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ImageError("not a PNG file")
    pos, idat, plte, hdr = 8, b"", None, None
    while pos + 12 <= len(data):
        (n,) = struct.unpack(">I", data[pos : pos + 4])
        typ, body = data[pos + 4 : pos + 8], data[pos + 8 : pos + 8 + n]
        if len(body) != n or pos + 12 + n > len(data):
            raise ImageError("truncated chunk")
        if zlib.crc32(typ + body) != struct.unpack(">I", data[pos + 8 + n : pos + 12 + n])[0]:
            raise ImageError("CRC mismatch")
        pos += 12 + n
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"PLTE":
            plte = np.frombuffer(body, dtype=np.uint8).reshape(-1, 3)
        elif typ == b"IDAT":
            idat += body
        elif typ == b"IEND":
            break
    if hdr is None or not idat:
        raise ImageError("missing IHDR/IDAT")
    w, h, depth, ctype, comp, flt, interlace = hdr
    if ctype in (0, 4):
        raise ImageError("Unsupported image format")
    channels = {2: 3, 6: 4, 3: 1}.get(ctype)
    if channels is None or comp or flt or interlace > 1 or (depth not in ((1, 2, 4, 8) if ctype == 3 else (8, 16))):
        raise ImageError("bad IHDR")
    try:
        raw = zlib.decompress(idat)
    except zlib.error as e:
        raise ImageError(str(e))
    bits_pp = channels * depth
    bpp = max(1, bits_pp // 8)
    img = np.zeros((h, w, 3), dtype=np.float32)
    passes = [(0, 0, 1, 1)] if not interlace else [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
    off = 0
    for x0, y0, dx, dy in passes:
        pw = (w - x0 + dx - 1) // dx if w > x0 else 0
        ph = (h - y0 + dy - 1) // dy if h > y0 else 0
        if pw == 0 or ph == 0:
            continue
        stride = (pw * bits_pp + 7) // 8
        if off + ph * (stride + 1) > len(raw):
            raise ImageError("not enough image data")
        lines, used = _unfilter(raw[off : off + ph * (stride + 1)], ph, stride, bpp)
        off += used
        if ctype == 3:
            bits = np.unpackbits(lines, axis=1)[:, : pw * depth].reshape(ph, pw, depth)
            idx = (bits * (1 << np.arange(depth - 1, -1, -1))).sum(axis=2)
            if idx.max() >= len(plte):
                raise ImageError("palette index out of range")
            px = plte[idx].astype(np.float32) / np.float32(255.0)
        elif depth == 8:
            px = lines.reshape(ph, pw, channels)[:, :, :3].astype(np.float32) / np.float32(255.0)
        else:
            v = lines.reshape(ph, pw, channels, 2).astype(np.uint32)
            px = ((v[..., 0] << 8) | v[..., 1])[:, :, :3].astype(np.float32) / np.float32(65535.0)
        img[y0::dy, x0::dx] = px
    return img


def evaluate(tex, u, v):
    """ImageTexture::evaluate (image_texture.rs:81-111) for one uv, in float32."""
    F = np.float32
    h, w = tex.shape[:2]

    def as_usize(f):
        return 0 if not (f > 0) else int(f)

    sx = F(u) - np.trunc(F(u))
    if sx < 0:
        sx = F(1.0) + sx
    sy = F(v) - np.trunc(F(v))
    if sy < 0:
        sy = F(1.0) + sy
    sy = F(1.0) - sy
    fx = sx * F(w) - F(0.5)
    fy = sy * F(h) - F(0.5)
    return tex[as_usize(fy), as_usize(fx)]
