// yk_image_formats.cpp — the other container formats ImageTexture::new accepts
// (textures/image_texture.rs:66-70,114-141): the reference opens a texture with
// `image::io::Reader::open(path)`, which picks the decoder from the FILE EXTENSION, and keeps
// Rgb8 / Rgba8 (c / 255), Rgb16 / Rgba16 (c / 65535) and Rgb32F / Rgba32F (as is); everything
// else (Luma*, LumaA*) is its "Unsupported image format".  The `image` 0.24 crate is not
// under /root/reference, so each decoder is restated from the format's specification:
//   .bmp          Windows BMP: BITMAPCOREHEADER / BITMAPINFOHEADER..V5, BI_RGB 1/4/8-bit
//                 palette, 24-bit, 32-bit (4th byte dropped); bottom-up or top-down
//   .tga          Truevision TGA: true-colour 24/32-bit and colour-mapped with 24/32-bit
//                 entries, raw or RLE (types 1, 2, 9, 10); grey (3, 11) -> Luma -> unsupported;
//                 rows flipped unless the top-left origin bit is set
//   .ppm/.pnm     Netpbm P3 / P6, maxval 255 (Rgb8) or 65535 (Rgb16, big endian);
//   .pbm/.pgm     P1/P2/P4/P5 are Luma -> unsupported
//   .qoi          QOI 1.0, 3 or 4 channels
//   .ff           farbfeld: RGBA 16-bit big endian
//   .exr          OpenEXR 2 single-part scan-line files with R, G, B channels (HALF or FLOAT,
//                 no sub-sampling), NO_COMPRESSION / ZIPS / ZIP — what yk_write_exr and the
//                 reference's own writer (film output) produce
// JPEG, GIF, TIFF, WebP, ICO, DDS, Radiance HDR and the remaining EXR compressions are named
// in the error message and not implemented.  No colour-space conversion anywhere (the
// reference does none), alpha dropped.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/yuki_hip.h"
#include "yk_image_internal.h"

namespace yk_img {

namespace {

struct Fail {
    std::string msg;
    yk_status st;
};
[[noreturn]] void bad(const std::string& m) { throw Fail{m, YK_ERR_INVALID_ARGUMENT}; }
[[noreturn]] void unsupported(const std::string& m) { throw Fail{m, YK_ERR_UNSUPPORTED}; }

struct Cursor {
    const uint8_t* p;
    size_t n, pos = 0;
    const char* what;
    void need(size_t k) const {
        if (k > n - pos) bad(std::string(what) + ": unexpected end of file");
    }
    uint8_t u8() {
        need(1);
        return p[pos++];
    }
    uint16_t le16() {
        need(2);
        uint16_t v = (uint16_t)(p[pos] | (p[pos + 1] << 8));
        pos += 2;
        return v;
    }
    uint32_t le32() {
        need(4);
        uint32_t v = (uint32_t)p[pos] | ((uint32_t)p[pos + 1] << 8) | ((uint32_t)p[pos + 2] << 16) | ((uint32_t)p[pos + 3] << 24);
        pos += 4;
        return v;
    }
    uint32_t be32() {
        need(4);
        uint32_t v = ((uint32_t)p[pos] << 24) | ((uint32_t)p[pos + 1] << 16) | ((uint32_t)p[pos + 2] << 8) | (uint32_t)p[pos + 3];
        pos += 4;
        return v;
    }
    void skip(size_t k) {
        need(k);
        pos += k;
    }
};

void check_size(const char* what, uint64_t w, uint64_t h) {
    if (w == 0 || h == 0) bad(std::string(what) + ": empty image");
    if (w * h > (1ull << 30)) unsupported(std::string(what) + ": image too large");
}

// ------------------------------------------------------------------ BMP
void decode_bmp(const std::vector<uint8_t>& f, uint32_t& W, uint32_t& H, std::vector<float>& rgb) {
    Cursor c{f.data(), f.size(), 0, "BMP"};
    if (f.size() < 2 || f[0] != 'B' || f[1] != 'M') bad("BMP: bad signature");
    c.skip(10);
    const uint32_t data_off = c.le32();
    const uint32_t hdr = c.le32();
    int64_t w, h;
    uint32_t bits, compression = 0, colors_used = 0;
    uint32_t mask[3] = {0x00ff0000u, 0x0000ff00u, 0x000000ffu};  // r g b of a 32-bit pixel read little-endian
    if (hdr == 12) {
        w = c.le16();
        h = c.le16();
        if (c.le16() != 1) bad("BMP: bad plane count");
        bits = c.le16();
    } else if (hdr == 40 || hdr == 52 || hdr == 56 || hdr == 108 || hdr == 124) {
        w = (int32_t)c.le32();
        h = (int32_t)c.le32();
        if (c.le16() != 1) bad("BMP: bad plane count");
        bits = c.le16();
        compression = c.le32();
        c.skip(12);
        colors_used = c.le32();
        c.skip(4);
        if (compression == 3 || compression == 6) {  // BI_BITFIELDS / BI_ALPHABITFIELDS: masks follow (inside the header from V2 on)
            for (int k = 0; k < 3; ++k) mask[k] = c.le32();
            if (hdr == 40) c.skip(compression == 6 ? 4 : 0);
            else c.skip(hdr - 52);
        } else {
            c.skip(hdr - 40);
        }
    } else {
        bad("BMP: unknown header size");
    }
    const bool top_down = h < 0;
    if (top_down) h = -h;
    if (w <= 0) bad("BMP: bad width");
    check_size("BMP", (uint64_t)w, (uint64_t)h);
    int shift[3] = {16, 8, 0};
    if (compression == 3 || compression == 6) {
        if (bits != 32) unsupported("BMP: bit fields are implemented for 32-bit pixels only");
        for (int k = 0; k < 3; ++k) {
            int sh = 0;
            while (sh < 32 && !((mask[k] >> sh) & 1u)) ++sh;
            if (sh > 24 || (mask[k] >> sh) != 0xffu) unsupported("BMP: only 8-bit-wide channel masks are implemented");
            shift[k] = sh;
        }
    } else if (compression != 0) {
        unsupported("BMP: RLE-compressed files are not implemented");
    }
    if (bits != 1 && bits != 4 && bits != 8 && bits != 24 && bits != 32) unsupported("BMP: only 1/4/8-bit palette, 24-bit and 32-bit pixels are implemented");
    std::vector<uint8_t> pal;  // r g b per entry
    if (bits <= 8) {
        uint32_t n = colors_used ? colors_used : (1u << bits);
        if (n > 256) bad("BMP: palette too large");
        const size_t esz = hdr == 12 ? 3 : 4;
        pal.resize((size_t)n * 3);
        for (uint32_t i = 0; i < n; ++i) {
            c.need(esz);
            pal[i * 3 + 2] = c.p[c.pos];
            pal[i * 3 + 1] = c.p[c.pos + 1];
            pal[i * 3 + 0] = c.p[c.pos + 2];
            c.pos += esz;
        }
    }
    if (data_off < c.pos || data_off > f.size()) bad("BMP: bad pixel data offset");
    const size_t stride = (((size_t)w * bits + 31) / 32) * 4;
    if (stride * (size_t)h > f.size() - data_off) bad("BMP: unexpected end of file");
    W = (uint32_t)w;
    H = (uint32_t)h;
    rgb.assign((size_t)W * H * 3, 0.0f);
    for (uint32_t y = 0; y < H; ++y) {
        const uint8_t* row = f.data() + data_off + stride * (top_down ? y : H - 1 - y);
        float* o = &rgb[(size_t)y * W * 3];
        for (uint32_t x = 0; x < W; ++x, o += 3) {
            if (bits <= 8) {
                const size_t bit = (size_t)x * bits;
                const uint32_t idx = (row[bit >> 3] >> (8 - bits - (bit & 7))) & ((1u << bits) - 1u);
                if ((size_t)idx * 3 + 2 >= pal.size()) bad("BMP: palette index out of range");
                for (int k = 0; k < 3; ++k) o[k] = (float)pal[idx * 3 + k] / 255.0f;
            } else if (bits == 24) {
                const uint8_t* q = row + (size_t)x * 3;
                o[0] = (float)q[2] / 255.0f;
                o[1] = (float)q[1] / 255.0f;
                o[2] = (float)q[0] / 255.0f;
            } else {
                const uint8_t* q = row + (size_t)x * 4;
                const uint32_t v = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
                for (int k = 0; k < 3; ++k) o[k] = (float)((v >> shift[k]) & 0xffu) / 255.0f;
            }
        }
    }
}

// ------------------------------------------------------------------ TGA
void decode_tga(const std::vector<uint8_t>& f, uint32_t& W, uint32_t& H, std::vector<float>& rgb) {
    Cursor c{f.data(), f.size(), 0, "TGA"};
    const uint8_t id_len = c.u8(), cmap_type = c.u8(), type = c.u8();
    const uint16_t cmap_first = c.le16(), cmap_len = c.le16();
    const uint8_t cmap_bits = c.u8();
    c.skip(4);
    const uint16_t w = c.le16(), h = c.le16();
    const uint8_t depth = c.u8(), desc = c.u8();
    check_size("TGA", w, h);
    const bool rle = type == 9 || type == 10 || type == 11;
    const uint8_t base = rle ? type - 8 : type;
    if (base == 3) unsupported("Unsupported image format");  // grey -> Luma8 / LumaA8
    if (base != 1 && base != 2) bad("TGA: bad image type");
    c.skip(id_len);
    std::vector<uint8_t> cmap;
    size_t cmap_bytes = 0;
    if (cmap_type == 1) {
        if (cmap_bits != 24 && cmap_bits != 32) unsupported("TGA: only 24- and 32-bit colour-map entries are implemented");
        cmap_bytes = cmap_bits / 8;
        c.need((size_t)cmap_len * cmap_bytes);
        cmap.assign(c.p + c.pos, c.p + c.pos + (size_t)cmap_len * cmap_bytes);
        c.pos += (size_t)cmap_len * cmap_bytes;
    } else if (cmap_type != 0) {
        bad("TGA: bad colour-map type");
    }
    size_t px_bytes;
    if (base == 1) {
        if (cmap_type != 1) bad("TGA: colour-mapped image without a colour map");
        if (depth != 8 && depth != 16) unsupported("TGA: only 8- and 16-bit colour-map indices are implemented");
        px_bytes = depth / 8;
    } else {
        if (depth != 24 && depth != 32) unsupported("TGA: only 24- and 32-bit true-colour pixels are implemented");
        px_bytes = depth / 8;
    }
    const size_t n_px = (size_t)w * h;
    std::vector<uint8_t> raw(n_px * px_bytes);
    if (!rle) {
        c.need(raw.size());
        std::memcpy(raw.data(), c.p + c.pos, raw.size());
    } else {
        size_t o = 0;
        while (o < raw.size()) {
            const uint8_t hd = c.u8();
            const size_t count = (size_t)(hd & 0x7f) + 1;
            if (count * px_bytes > raw.size() - o) bad("TGA: run crosses the end of the image");
            if (hd & 0x80) {
                c.need(px_bytes);
                for (size_t k = 0; k < count; ++k, o += px_bytes) std::memcpy(&raw[o], c.p + c.pos, px_bytes);
                c.pos += px_bytes;
            } else {
                c.need(count * px_bytes);
                std::memcpy(&raw[o], c.p + c.pos, count * px_bytes);
                c.pos += count * px_bytes;
                o += count * px_bytes;
            }
        }
    }
    W = w;
    H = h;
    rgb.assign(n_px * 3, 0.0f);
    const bool top_left = (desc & 0x20) != 0;
    for (uint32_t y = 0; y < H; ++y) {
        const uint8_t* row = &raw[(size_t)(top_left ? y : H - 1 - y) * W * px_bytes];
        float* o = &rgb[(size_t)y * W * 3];
        for (uint32_t x = 0; x < W; ++x, o += 3) {
            const uint8_t* q = row + (size_t)x * px_bytes;
            if (base == 1) {
                uint32_t idx = px_bytes == 1 ? q[0] : (uint32_t)(q[0] | (q[1] << 8));
                if (idx < cmap_first || idx - cmap_first >= cmap_len) bad("TGA: colour-map index out of range");
                q = &cmap[(size_t)(idx - cmap_first) * cmap_bytes];
            }
            o[0] = (float)q[2] / 255.0f;  // stored B G R [A]
            o[1] = (float)q[1] / 255.0f;
            o[2] = (float)q[0] / 255.0f;
        }
    }
}

// ------------------------------------------------------------------ Netpbm
void decode_pnm(const std::vector<uint8_t>& f, uint32_t& W, uint32_t& H, std::vector<float>& rgb) {
    if (f.size() < 3 || f[0] != 'P') bad("PNM: bad magic");
    const char kind = (char)f[1];
    if (kind == '1' || kind == '2' || kind == '4' || kind == '5') unsupported("Unsupported image format");  // bitmap / greymap -> Luma
    if (kind == '7') unsupported("PNM: PAM (P7) files are not implemented");
    if (kind != '3' && kind != '6') bad("PNM: bad magic");
    size_t pos = 2;
    auto token = [&]() -> uint64_t {
        for (;;) {  // white space and comments
            if (pos >= f.size()) bad("PNM: unexpected end of file");
            if (f[pos] == '#') {
                while (pos < f.size() && f[pos] != '\n' && f[pos] != '\r') ++pos;
            } else if (f[pos] == ' ' || f[pos] == '\t' || f[pos] == '\n' || f[pos] == '\r' || f[pos] == '\v' || f[pos] == '\f') {
                ++pos;
            } else {
                break;
            }
        }
        uint64_t v = 0;
        size_t digits = 0;
        while (pos < f.size() && f[pos] >= '0' && f[pos] <= '9') {
            v = v * 10 + (uint64_t)(f[pos] - '0');
            if (v > 0xFFFFFFFFull) bad("PNM: number too large");
            ++pos;
            ++digits;
        }
        if (!digits) bad("PNM: expected a number");
        return v;
    };
    const uint64_t w = token(), h = token(), maxval = token();
    check_size("PNM", w, h);
    if (maxval != 255 && maxval != 65535) unsupported("PNM: only maxval 255 and 65535 are implemented");
    const float scale = (float)maxval;
    W = (uint32_t)w;
    H = (uint32_t)h;
    const size_t n = (size_t)W * H * 3;
    rgb.assign(n, 0.0f);
    if (kind == '3') {
        for (size_t i = 0; i < n; ++i) {
            const uint64_t v = token();
            if (v > maxval) bad("PNM: sample exceeds maxval");
            rgb[i] = (float)v / scale;
        }
        return;
    }
    if (pos >= f.size()) bad("PNM: unexpected end of file");
    ++pos;  // the single white-space byte after maxval
    const size_t bps = maxval > 255 ? 2 : 1;
    if (n * bps > f.size() - pos) bad("PNM: unexpected end of file");
    const uint8_t* q = f.data() + pos;
    for (size_t i = 0; i < n; ++i) rgb[i] = bps == 1 ? (float)q[i] / scale : (float)(((uint32_t)q[2 * i] << 8) | q[2 * i + 1]) / scale;
}

// ------------------------------------------------------------------ QOI
void decode_qoi(const std::vector<uint8_t>& f, uint32_t& W, uint32_t& H, std::vector<float>& rgb) {
    Cursor c{f.data(), f.size(), 0, "QOI"};
    if (f.size() < 14 || std::memcmp(f.data(), "qoif", 4) != 0) bad("QOI: bad magic");
    c.skip(4);
    const uint32_t w = c.be32(), h = c.be32();
    const uint8_t channels = c.u8(), colorspace = c.u8();
    if ((channels != 3 && channels != 4) || colorspace > 1) bad("QOI: bad header");
    check_size("QOI", w, h);
    W = w;
    H = h;
    const size_t n = (size_t)w * h;
    rgb.assign(n * 3, 0.0f);
    uint8_t px[4] = {0, 0, 0, 255};
    uint8_t index[64][4];
    std::memset(index, 0, sizeof(index));
    size_t run = 0;
    for (size_t i = 0; i < n; ++i) {
        if (run > 0) {
            --run;
        } else {
            const uint8_t b = c.u8();
            if (b == 0xFE) {
                px[0] = c.u8();
                px[1] = c.u8();
                px[2] = c.u8();
            } else if (b == 0xFF) {
                px[0] = c.u8();
                px[1] = c.u8();
                px[2] = c.u8();
                px[3] = c.u8();
            } else if ((b & 0xC0) == 0x00) {
                std::memcpy(px, index[b & 63], 4);
            } else if ((b & 0xC0) == 0x40) {
                px[0] = (uint8_t)(px[0] + ((b >> 4) & 3) - 2);
                px[1] = (uint8_t)(px[1] + ((b >> 2) & 3) - 2);
                px[2] = (uint8_t)(px[2] + (b & 3) - 2);
            } else if ((b & 0xC0) == 0x80) {
                const uint8_t b2 = c.u8();
                const int dg = (b & 63) - 32;
                px[0] = (uint8_t)(px[0] + dg - 8 + ((b2 >> 4) & 15));
                px[1] = (uint8_t)(px[1] + dg);
                px[2] = (uint8_t)(px[2] + dg - 8 + (b2 & 15));
            } else {
                run = b & 63;  // run of (b & 63) + 1 pixels, this one included
            }
            std::memcpy(index[(px[0] * 3 + px[1] * 5 + px[2] * 7 + px[3] * 11) & 63], px, 4);
        }
        for (int k = 0; k < 3; ++k) rgb[i * 3 + k] = (float)px[k] / 255.0f;
    }
}

// ------------------------------------------------------------------ farbfeld
void decode_farbfeld(const std::vector<uint8_t>& f, uint32_t& W, uint32_t& H, std::vector<float>& rgb) {
    Cursor c{f.data(), f.size(), 0, "farbfeld"};
    if (f.size() < 16 || std::memcmp(f.data(), "farbfeld", 8) != 0) bad("farbfeld: bad magic");
    c.skip(8);
    const uint32_t w = c.be32(), h = c.be32();
    check_size("farbfeld", w, h);
    const size_t n = (size_t)w * h;
    if (n * 8 > f.size() - 16) bad("farbfeld: unexpected end of file");
    W = w;
    H = h;
    rgb.assign(n * 3, 0.0f);
    const uint8_t* q = f.data() + 16;
    for (size_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) rgb[i * 3 + k] = (float)(((uint32_t)q[i * 8 + 2 * k] << 8) | q[i * 8 + 2 * k + 1]) / 65535.0f;
}

// ------------------------------------------------------------------ OpenEXR
float half_to_float(uint16_t hbits) {
    const uint32_t s = (uint32_t)(hbits >> 15) << 31, e = (hbits >> 10) & 31u, m = hbits & 1023u;
    uint32_t out;
    if (e == 0) {
        if (m == 0) {
            out = s;
        } else {  // subnormal: normalise
            int sh = 0;
            uint32_t mm = m;
            while (!(mm & 1024u)) {
                mm <<= 1;
                ++sh;
            }
            out = s | ((uint32_t)(127 - 15 - sh + 1) << 23) | ((mm & 1023u) << 13);
        }
    } else if (e == 31) {
        out = s | 0x7f800000u | (m << 13);
    } else {
        out = s | ((e + 127 - 15) << 23) | (m << 13);
    }
    float v;
    std::memcpy(&v, &out, 4);
    return v;
}

void decode_exr(const std::vector<uint8_t>& f, uint32_t& W, uint32_t& H, std::vector<float>& rgb) {
    Cursor c{f.data(), f.size(), 0, "EXR"};
    if (c.le32() != 20000630u) bad("EXR: bad magic");
    const uint32_t version = c.le32();
    if ((version & 0xff) != 2) unsupported("EXR: unknown file version");
    if (version & 0x1a00) unsupported("EXR: tiled, deep and multi-part files are not implemented");  // bits 9 (tiles), 11 (deep), 12 (multipart)
    struct Channel {
        std::string name;
        uint32_t type;
    };
    std::vector<Channel> channels;
    int32_t dw[4] = {0, 0, -1, -1};
    int compression = -1, line_order = 0;
    bool have_dw = false;
    for (;;) {
        std::string name;
        for (uint8_t b; (b = c.u8()) != 0;) name.push_back((char)b);
        if (name.empty()) break;
        std::string type;
        for (uint8_t b; (b = c.u8()) != 0;) type.push_back((char)b);
        const uint32_t size = c.le32();
        c.need(size);
        Cursor a{c.p + c.pos, size, 0, "EXR"};
        if (name == "channels") {
            for (;;) {
                std::string cn;
                for (uint8_t b; (b = a.u8()) != 0;) cn.push_back((char)b);
                if (cn.empty()) break;
                const uint32_t pt = a.le32();
                a.skip(4);
                const uint32_t xs = a.le32(), ys = a.le32();
                if (xs != 1 || ys != 1) unsupported("EXR: sub-sampled channels are not implemented");
                if (pt > 2) bad("EXR: bad pixel type");
                channels.push_back({cn, pt});
            }
        } else if (name == "compression") {
            compression = a.u8();
        } else if (name == "dataWindow") {
            for (int k = 0; k < 4; ++k) dw[k] = (int32_t)a.le32();
            have_dw = true;
        } else if (name == "lineOrder") {
            line_order = a.u8();
        }
        c.pos += size;
    }
    if (channels.empty() || !have_dw || compression < 0) bad("EXR: missing channels / dataWindow / compression");
    if (compression != 0 && compression != 2 && compression != 3) unsupported("EXR: only NO_COMPRESSION, ZIPS and ZIP are implemented (not RLE, PIZ, PXR24, B44, DWA)");
    if (line_order > 1) unsupported("EXR: random-y files are not implemented");
    if (dw[2] < dw[0] || dw[3] < dw[1]) bad("EXR: empty data window");
    const uint64_t w = (uint64_t)((int64_t)dw[2] - dw[0] + 1), h = (uint64_t)((int64_t)dw[3] - dw[1] + 1);
    check_size("EXR", w, h);
    int want[3] = {-1, -1, -1};
    std::vector<size_t> ch_off(channels.size());  // byte offset of each channel inside one scan line
    size_t line_bytes = 0;
    for (size_t k = 0; k < channels.size(); ++k) {  // the list is stored sorted by name, and so is the pixel data
        if (k && !(channels[k - 1].name < channels[k].name)) bad("EXR: channel list not sorted");
        ch_off[k] = line_bytes;
        line_bytes += (size_t)w * (channels[k].type == 1 ? 2 : 4);
        if (channels[k].name == "R") want[0] = (int)k;
        if (channels[k].name == "G") want[1] = (int)k;
        if (channels[k].name == "B") want[2] = (int)k;
    }
    if (want[0] < 0 || want[1] < 0 || want[2] < 0) unsupported("EXR: no R, G, B channels in the first layer");
    const uint32_t lines_per_chunk = compression == 3 ? 16 : 1;
    const size_t n_chunks = (size_t)((h + lines_per_chunk - 1) / lines_per_chunk);
    c.need(n_chunks * 8);
    std::vector<uint64_t> offsets(n_chunks);
    for (size_t k = 0; k < n_chunks; ++k) {
        const uint64_t lo = c.le32(), hi = c.le32();
        offsets[k] = lo | (hi << 32);
    }
    W = (uint32_t)w;
    H = (uint32_t)h;
    rgb.assign((size_t)W * H * 3, 0.0f);
    std::vector<uint8_t> buf, tmp;
    for (size_t k = 0; k < n_chunks; ++k) {
        if (offsets[k] > f.size() || f.size() - offsets[k] < 8) bad("EXR: bad chunk offset");
        Cursor d{f.data(), f.size(), (size_t)offsets[k], "EXR"};
        const int64_t y0 = (int32_t)d.le32();
        const uint32_t size = d.le32();
        d.need(size);
        if (y0 < dw[1] || y0 > dw[3] || (y0 - dw[1]) % lines_per_chunk) bad("EXR: bad chunk coordinate");
        const size_t rows = (size_t)std::min<int64_t>(lines_per_chunk, (int64_t)dw[3] - y0 + 1);
        const size_t expect = rows * line_bytes;
        const uint8_t* data;
        if (compression == 0 || size == expect) {  // a chunk that does not shrink is stored raw
            if (size != expect) bad("EXR: bad chunk size");
            data = d.p + d.pos;
        } else {
            if (size < 6) bad("EXR: bad chunk size");
            if (!yk_inflate_zlib(d.p + d.pos, size, tmp) || tmp.size() != expect) bad("EXR: corrupt zlib chunk");
            // undo the predictor, then the byte interleave (OpenEXR ImfZip)
            for (size_t i = 1; i < tmp.size(); ++i) tmp[i] = (uint8_t)(tmp[i - 1] + tmp[i] - 128);
            buf.resize(expect);
            const size_t half = (expect + 1) / 2;
            for (size_t i = 0; i < expect; ++i) buf[i] = (i & 1) ? tmp[half + i / 2] : tmp[i / 2];
            data = buf.data();
        }
        for (size_t r = 0; r < rows; ++r) {
            const uint8_t* line = data + r * line_bytes;
            float* o = &rgb[((size_t)(y0 - dw[1]) + r) * W * 3];
            for (int ch = 0; ch < 3; ++ch) {
                const Channel& cd = channels[(size_t)want[ch]];
                const uint8_t* q = line + ch_off[(size_t)want[ch]];
                for (uint32_t x = 0; x < W; ++x) {
                    float v;
                    if (cd.type == 1) {
                        v = half_to_float((uint16_t)(q[2 * x] | (q[2 * x + 1] << 8)));
                    } else if (cd.type == 2) {
                        std::memcpy(&v, q + 4 * (size_t)x, 4);
                    } else {
                        uint32_t u;
                        std::memcpy(&u, q + 4 * (size_t)x, 4);
                        v = (float)u;
                    }
                    o[(size_t)x * 3 + ch] = v;
                }
            }
        }
    }
}

std::string lower_extension(const std::string& path) {
    const size_t slash = path.find_last_of("/\\");
    const size_t dot = path.find_last_of('.');
    if (dot == std::string::npos || (slash != std::string::npos && dot < slash) || dot + 1 >= path.size()) return "";
    std::string e = path.substr(dot + 1);
    for (char& ch : e)
        if (ch >= 'A' && ch <= 'Z') ch = (char)(ch - 'A' + 'a');
    return e;
}

}  // namespace

// Decoder chosen by extension, like image::io::Reader::open.  Returns YK_OK, or a status with `err` set.
yk_status decode_by_extension(const std::string& path, const std::vector<uint8_t>& bytes, uint32_t& w, uint32_t& h, std::vector<float>& rgb, std::string& err,
                              bool& is_png) {
    const std::string e = lower_extension(path);
    is_png = e == "png";
    if (is_png) return YK_OK;  // the caller owns the PNG decoder
    try {
        if (e == "bmp") decode_bmp(bytes, w, h, rgb);
        else if (e == "tga") decode_tga(bytes, w, h, rgb);
        else if (e == "ppm" || e == "pnm" || e == "pbm" || e == "pgm") decode_pnm(bytes, w, h, rgb);
        else if (e == "qoi") decode_qoi(bytes, w, h, rgb);
        else if (e == "ff") decode_farbfeld(bytes, w, h, rgb);
        else if (e == "exr") decode_exr(bytes, w, h, rgb);
        else if (e == "jpg" || e == "jpeg" || e == "gif" || e == "tif" || e == "tiff" || e == "webp" || e == "ico" || e == "dds" || e == "hdr" || e == "pam" || e == "avif")
            unsupported("image format '." + e + "' is not implemented (PNG, BMP, TGA, PPM, QOI, farbfeld and EXR are)");
        else
            unsupported("The image format could not be determined");
    } catch (const Fail& x) {
        err = x.msg;
        return x.st;
    } catch (const std::exception& x) {
        err = std::string("image: ") + x.what();
        return YK_ERR_INVALID_ARGUMENT;
    }
    return YK_OK;
}

}  // namespace yk_img
