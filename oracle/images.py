"""TEST INFRASTRUCTURE ONLY — CPU restatement of ImageTexture::new
(yuki/src/textures/image_texture.rs:66-70,114-141) for PNG files.

The reference decodes through the `image` 0.24 crate (a Cargo dependency that is not under
/root/reference); its PNG behaviour is restated from the PNG specification: EXPAND
transformations (palette -> RGB, tRNS -> alpha), 8/16-bit RGB(A) -> c/255 or c/65535 in
f32, no gamma, alpha dropped, gray(+alpha) -> "Unsupported image format".  Decompression is
Python's zlib; filtering/interlace handling is written here independently of
yuki_amd/csrc/yk_image.cpp.  Parity unpinned: the reference ships no image fixtures.

`load_image` picks the decoder from the file extension like image::io::Reader::open and also
reads BMP, TGA, PPM, QOI, farbfeld and scan-line OpenEXR (restated from their specifications,
numpy-style, independently of yuki_amd/csrc/yk_image_formats.cpp).
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


def _u8(a):
    return a.astype(np.float32) / np.float32(255.0)


def load_bmp(data):
    if data[:2] != b"BM":
        raise ImageError("BMP: bad signature")
    try:
        off, hdr = struct.unpack_from("<II", data, 10)
        masks = (0xFF0000, 0xFF00, 0xFF)
        if hdr == 12:
            w, h, planes, bits = struct.unpack_from("<HHHH", data, 18)
            comp, used, pal_at, esz = 0, 0, 26, 3
        elif hdr in (40, 52, 56, 108, 124):
            w, h, planes, bits, comp, _, _, _, used, _ = struct.unpack_from("<iiHHIIiiII", data, 18)
            pal_at, esz = 14 + hdr, 4
            if comp in (3, 6):
                masks = struct.unpack_from("<III", data, 54)
                if hdr == 40:
                    pal_at += 16 if comp == 6 else 12
        else:
            raise ImageError("BMP: unknown header size")
    except struct.error:
        raise ImageError("BMP: unexpected end of file")
    top_down = h < 0
    h = abs(h)
    if w <= 0 or h == 0 or planes != 1:
        raise ImageError("BMP: bad header")
    if comp in (3, 6):
        if bits != 32 or any(((m >> ((m & -m).bit_length() - 1)) != 0xFF) for m in masks):
            raise ImageError("BMP: unsupported bit fields")
    elif comp != 0:
        raise ImageError("BMP: RLE not implemented")
    if bits not in (1, 4, 8, 24, 32):
        raise ImageError("BMP: unsupported bit count")
    stride = (w * bits + 31) // 32 * 4
    if off + stride * h > len(data):
        raise ImageError("BMP: unexpected end of file")
    rows = np.frombuffer(data, dtype=np.uint8, count=stride * h, offset=off).reshape(h, stride)
    if not top_down:
        rows = rows[::-1]
    if bits <= 8:
        n = used or (1 << bits)
        if pal_at + n * esz > len(data) or n > 256:
            raise ImageError("BMP: bad palette")
        pal = np.frombuffer(data, dtype=np.uint8, count=n * esz, offset=pal_at).reshape(n, esz)[:, 2::-1]
        b = np.unpackbits(rows, axis=1)[:, : w * bits].reshape(h, w, bits)
        idx = (b * (1 << np.arange(bits - 1, -1, -1))).sum(axis=2)
        if idx.max() >= n:
            raise ImageError("BMP: palette index out of range")
        return _u8(pal[idx])
    if bits == 24:
        return _u8(rows[:, : w * 3].reshape(h, w, 3)[:, :, ::-1])
    v = rows[:, : w * 4].copy().view("<u4").reshape(h, w)
    return _u8(np.stack([(v >> ((m & -m).bit_length() - 1)) & 0xFF for m in masks], axis=2))


def load_tga(data):
    try:
        idl, cmt, typ, cm_first, cm_len, cm_bits, _, _, w, h, depth, desc = struct.unpack_from("<BBBHHBHHHHBB", data, 0)
    except struct.error:
        raise ImageError("TGA: unexpected end of file")
    if w == 0 or h == 0:
        raise ImageError("TGA: empty image")
    rle, base = typ in (9, 10, 11), typ - 8 if typ in (9, 10, 11) else typ
    if base == 3:
        raise ImageError("Unsupported image format")
    if base not in (1, 2) or cmt not in (0, 1):
        raise ImageError("TGA: bad image type")
    pos = 18 + idl
    cmap = None
    if cmt == 1:
        if cm_bits not in (24, 32):
            raise ImageError("TGA: unsupported colour map")
        nb = cm_len * (cm_bits // 8)
        if pos + nb > len(data):
            raise ImageError("TGA: unexpected end of file")
        cmap = np.frombuffer(data, dtype=np.uint8, count=nb, offset=pos).reshape(cm_len, cm_bits // 8)
        pos += nb
    if (base == 1 and (cmap is None or depth not in (8, 16))) or (base == 2 and depth not in (24, 32)):
        raise ImageError("TGA: unsupported pixel depth")
    pb, total = depth // 8, w * h * (depth // 8)
    if not rle:
        if pos + total > len(data):
            raise ImageError("TGA: unexpected end of file")
        raw = data[pos : pos + total]
    else:
        out = bytearray()
        while len(out) < total:
            if pos >= len(data):
                raise ImageError("TGA: unexpected end of file")
            hd = data[pos]
            pos += 1
            cnt = (hd & 0x7F) + 1
            take = pb if hd & 0x80 else cnt * pb
            if pos + take > len(data) or len(out) + cnt * pb > total:
                raise ImageError("TGA: bad run")
            out += data[pos : pos + pb] * cnt if hd & 0x80 else data[pos : pos + take]
            pos += take
        raw = bytes(out)
    px = np.frombuffer(raw, dtype=np.uint8).reshape(h, w, pb)
    if base == 1:
        idx = px[:, :, 0].astype(np.int64) if pb == 1 else px[:, :, 0].astype(np.int64) | (px[:, :, 1].astype(np.int64) << 8)
        idx = idx - cm_first
        if idx.min() < 0 or idx.max() >= cm_len:
            raise ImageError("TGA: colour-map index out of range")
        px = cmap[idx]
    img = px[:, :, 2::-1]
    if not desc & 0x20:
        img = img[::-1]
    return _u8(img)


def load_pnm(data):
    if data[:1] != b"P" or len(data) < 3:
        raise ImageError("PNM: bad magic")
    kind = data[1:2]
    if kind in (b"1", b"2", b"4", b"5"):
        raise ImageError("Unsupported image format")
    if kind not in (b"3", b"6"):
        raise ImageError("PNM: bad magic")
    pos = 2

    def token():
        nonlocal pos
        while True:
            if pos >= len(data):
                raise ImageError("PNM: unexpected end of file")
            ch = data[pos : pos + 1]
            if ch == b"#":
                while pos < len(data) and data[pos : pos + 1] not in (b"\n", b"\r"):
                    pos += 1
            elif ch in b" \t\n\r\v\f":
                pos += 1
            else:
                break
        start = pos
        while pos < len(data) and data[pos : pos + 1].isdigit():
            pos += 1
        if start == pos:
            raise ImageError("PNM: expected a number")
        return int(data[start:pos])

    w, h, maxval = token(), token(), token()
    if w == 0 or h == 0 or maxval not in (255, 65535):
        raise ImageError("PNM: unsupported header")
    n = w * h * 3
    if kind == b"3":
        v = np.array([token() for _ in range(n)], dtype=np.uint32)
        if v.max() > maxval:
            raise ImageError("PNM: sample exceeds maxval")
    else:
        pos += 1
        bps = 2 if maxval > 255 else 1
        if pos + n * bps > len(data):
            raise ImageError("PNM: unexpected end of file")
        v = np.frombuffer(data, dtype=">u2" if bps == 2 else np.uint8, count=n, offset=pos).astype(np.uint32)
    return (v.astype(np.float32) / np.float32(maxval)).reshape(h, w, 3)


def load_qoi(data):
    if data[:4] != b"qoif" or len(data) < 14:
        raise ImageError("QOI: bad magic")
    w, h, ch, cs = struct.unpack_from(">IIBB", data, 4)
    if ch not in (3, 4) or cs > 1 or w == 0 or h == 0:
        raise ImageError("QOI: bad header")
    out = np.zeros((w * h, 3), dtype=np.uint8)
    seen = [(0, 0, 0, 0)] * 64
    r, g, b, a = 0, 0, 0, 255
    pos, i, n = 14, 0, w * h
    try:
        while i < n:
            t = data[pos]
            pos += 1
            run = 1
            if t == 0xFE:
                r, g, b = data[pos], data[pos + 1], data[pos + 2]
                pos += 3
            elif t == 0xFF:
                r, g, b, a = data[pos], data[pos + 1], data[pos + 2], data[pos + 3]
                pos += 4
            elif t >> 6 == 0:
                r, g, b, a = seen[t]
            elif t >> 6 == 1:
                r, g, b = (r + ((t >> 4) & 3) - 2) & 255, (g + ((t >> 2) & 3) - 2) & 255, (b + (t & 3) - 2) & 255
            elif t >> 6 == 2:
                t2 = data[pos]
                pos += 1
                dg = (t & 63) - 32
                r, g, b = (r + dg - 8 + (t2 >> 4)) & 255, (g + dg) & 255, (b + dg - 8 + (t2 & 15)) & 255
            else:
                run = (t & 63) + 1
            seen[(r * 3 + g * 5 + b * 7 + a * 11) % 64] = (r, g, b, a)
            run = min(run, n - i)
            out[i : i + run] = (r, g, b)
            i += run
    except IndexError:
        raise ImageError("QOI: unexpected end of file")
    return _u8(out.reshape(h, w, 3))


def load_farbfeld(data):
    if data[:8] != b"farbfeld" or len(data) < 16:
        raise ImageError("farbfeld: bad magic")
    w, h = struct.unpack_from(">II", data, 8)
    if w == 0 or h == 0 or 16 + w * h * 8 > len(data):
        raise ImageError("farbfeld: bad size")
    v = np.frombuffer(data, dtype=">u2", count=w * h * 4, offset=16).reshape(h, w, 4)[:, :, :3]
    return v.astype(np.float32) / np.float32(65535.0)


def load_exr(data):
    try:
        magic, version = struct.unpack_from("<II", data, 0)
    except struct.error:
        raise ImageError("EXR: unexpected end of file")
    if magic != 20000630 or version & 0xFF != 2 or version & 0x1A00:
        raise ImageError("EXR: unsupported file")
    pos, attrs = 8, {}

    def cstr():
        nonlocal pos
        end = data.index(b"\0", pos)
        s0 = data[pos:end]
        pos = end + 1
        return s0

    try:
        while True:
            name = cstr()
            if not name:
                break
            cstr()
            (size,) = struct.unpack_from("<I", data, pos)
            attrs[name] = data[pos + 4 : pos + 4 + size]
            pos += 4 + size
        chl, chans, q = attrs[b"channels"], [], 0
        while chl[q : q + 1] != b"\0":
            end = chl.index(b"\0", q)
            pt, _, xs, ys = struct.unpack_from("<IIII", chl, end + 1)
            if xs != 1 or ys != 1 or pt > 2:
                raise ImageError("EXR: unsupported channel")
            chans.append((chl[q:end], pt))
            q = end + 17
        comp = attrs[b"compression"][0]
        x0, y0, x1, y1 = struct.unpack("<iiii", attrs[b"dataWindow"][:16])
    except (KeyError, ValueError, struct.error, IndexError):
        raise ImageError("EXR: bad header")
    if comp not in (0, 2, 3) or attrs.get(b"lineOrder", b"\0")[0] > 1:
        raise ImageError("EXR: unsupported compression / line order")
    w, h = x1 - x0 + 1, y1 - y0 + 1
    if w <= 0 or h <= 0:
        raise ImageError("EXR: empty data window")
    names = [c[0] for c in chans]
    if names != sorted(names) or any(k not in names for k in (b"R", b"G", b"B")):
        raise ImageError("EXR: no R, G, B channels")
    sizes = [w * (2 if pt == 1 else 4) for _, pt in chans]
    line_bytes = sum(sizes)
    lpc = 16 if comp == 3 else 1
    n_chunks = (h + lpc - 1) // lpc
    img = np.zeros((h, w, 3), dtype=np.float32)
    try:
        offs = struct.unpack_from("<%dQ" % n_chunks, data, pos)
        for o in offs:
            y, size = struct.unpack_from("<iI", data, o)
            body = data[o + 8 : o + 8 + size]
            rows = min(lpc, y1 - y + 1)
            if len(body) != size or y < y0 or (y - y0) % lpc:
                raise ImageError("EXR: bad chunk")
            if comp and size != rows * line_bytes:
                t = np.frombuffer(zlib.decompress(body), dtype=np.uint8)
                if len(t) != rows * line_bytes:
                    raise ImageError("EXR: bad chunk size")
                t = (np.cumsum(t.astype(np.int64) - 128) + 128).astype(np.uint8)  # predictor: t[i] = t[i-1] + t[i] - 128, t[0] kept
                half = (len(t) + 1) // 2
                u = np.empty(len(t), dtype=np.uint8)
                u[0::2] = t[:half]
                u[1::2] = t[half:]
                body = u.tobytes()
            if len(body) != rows * line_bytes:
                raise ImageError("EXR: bad chunk size")
            for r in range(rows):
                q = r * line_bytes
                for (nm, pt), sz in zip(chans, sizes):
                    if nm in (b"R", b"G", b"B"):
                        dt = {0: "<u4", 1: "<f2", 2: "<f4"}[pt]
                        img[y - y0 + r, :, (b"R", b"G", b"B").index(nm)] = np.frombuffer(body, dtype=dt, count=w, offset=q).astype(np.float32)
                    q += sz
    except (struct.error, zlib.error) as e:
        raise ImageError("EXR: " + str(e))
    return img


def load_image(path):
    """ImageTexture::new: decoder by file extension -> (h, w, 3) float32."""
    ext = path.rsplit(".", 1)[-1].lower() if "." in path.rsplit("/", 1)[-1] else ""
    if ext == "png":
        return load_png(path)
    table = {"bmp": load_bmp, "tga": load_tga, "ppm": load_pnm, "pnm": load_pnm, "pbm": load_pnm, "pgm": load_pnm, "qoi": load_qoi, "ff": load_farbfeld, "exr": load_exr}
    if ext not in table:
        raise ImageError("image format not implemented or not determined")
    with open(path, "rb") as f:
        return table[ext](f.read())


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
