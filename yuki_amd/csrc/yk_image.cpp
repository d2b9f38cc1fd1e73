// yk_image.cpp — ImageTexture::new (textures/image_texture.rs:66-70,114-141): decode an
// image file into row-major RGB f32.  The reference decodes through the `image` 0.24
// crate (not under /root/reference); what it does for PNG is restated here from the PNG
// specification (ISO/IEC 15948) and RFC 1950/1951:
//   * 8-bit RGB / RGBA            -> ImageRgb8 / ImageRgba8   -> c / 255
//   * 16-bit RGB / RGBA           -> ImageRgb16 / ImageRgba16 -> c / 65535
//   * palette (1,2,4,8 bit)       -> expanded to Rgb8 (Rgba8 with tRNS) by the decoder
//   * gray, gray+alpha            -> Luma*/LumaA* -> the reference's "Unsupported image format"
//   * no gamma / sRGB conversion, alpha dropped, Adam7 interlace supported, CRC and
//     Adler-32 verified (a corrupt file is a decode error in the reference too).
// The decoder is chosen from the file extension, like image::io::Reader::open; the other
// containers (BMP, TGA, PPM, QOI, farbfeld, EXR) live in yk_image_formats.cpp.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/yuki_hip.h"
#include "yk_host.h"
#include "yk_image_internal.h"

namespace {

thread_local std::string g_image_error;

// ---------------------------------------------------------------- inflate (RFC 1951)
struct BitReader {
    const uint8_t* p;
    size_t n, pos = 0;
    uint32_t acc = 0;
    int cnt = 0;
    bool bad = false;
    uint32_t bits(int k) {
        while (cnt < k) {
            if (pos >= n) {
                bad = true;
                return 0;
            }
            acc |= (uint32_t)p[pos++] << cnt;
            cnt += 8;
        }
        uint32_t v = k ? (acc & ((1u << k) - 1u)) : 0u;
        acc >>= k;
        cnt -= k;
        return v;
    }
    void align() {
        acc = 0;
        cnt = 0;
    }
};

struct Huffman {
    uint16_t count[16], symbol[288];
    bool build(const uint8_t* len, int n) {
        std::memset(count, 0, sizeof(count));
        for (int i = 0; i < n; ++i) count[len[i]]++;
        int left = 1;
        for (int l = 1; l < 16; ++l) {
            left <<= 1;
            left -= count[l];
            if (left < 0) return false;  // over-subscribed
        }
        uint16_t offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + count[l];
        for (int i = 0; i < n; ++i)
            if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
        return true;
    }
    int decode(BitReader& br) const {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; ++l) {
            code |= (int)br.bits(1);
            if (br.bad) return -1;
            int c = count[l];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
};

bool inflate(const uint8_t* src, size_t n, std::vector<uint8_t>& out) {
    static const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint16_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    BitReader br{src, n};
    for (;;) {
        uint32_t last = br.bits(1), type = br.bits(2);
        if (br.bad) return false;
        if (type == 0) {
            br.align();
            if (br.pos + 4 > n) return false;
            uint32_t len = src[br.pos] | (src[br.pos + 1] << 8), nlen = src[br.pos + 2] | (src[br.pos + 3] << 8);
            br.pos += 4;
            if ((len ^ 0xffffu) != nlen || br.pos + len > n) return false;
            out.insert(out.end(), src + br.pos, src + br.pos + len);
            br.pos += len;
        } else if (type == 1 || type == 2) {
            Huffman lit, dist;
            uint8_t lens[320];
            if (type == 1) {
                for (int i = 0; i < 144; ++i) lens[i] = 8;
                for (int i = 144; i < 256; ++i) lens[i] = 9;
                for (int i = 256; i < 280; ++i) lens[i] = 7;
                for (int i = 280; i < 288; ++i) lens[i] = 8;
                lit.build(lens, 288);
                for (int i = 0; i < 30; ++i) lens[i] = 5;
                dist.build(lens, 30);
            } else {
                static const uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
                if (br.bad || nlen > 286 || ndist > 30) return false;
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; ++i) cl[ORDER[i]] = (uint8_t)br.bits(3);
                Huffman ch;
                if (!ch.build(cl, 19)) return false;
                int i = 0;
                while (i < nlen + ndist) {
                    int sym = ch.decode(br);
                    if (sym < 0) return false;
                    if (sym < 16) {
                        lens[i++] = (uint8_t)sym;
                    } else {
                        int rep, val = 0;
                        if (sym == 16) {
                            if (i == 0) return false;
                            val = lens[i - 1];
                            rep = 3 + (int)br.bits(2);
                        } else if (sym == 17) {
                            rep = 3 + (int)br.bits(3);
                        } else {
                            rep = 11 + (int)br.bits(7);
                        }
                        if (i + rep > nlen + ndist) return false;
                        while (rep--) lens[i++] = (uint8_t)val;
                    }
                }
                if (br.bad || lens[256] == 0) return false;
                if (!lit.build(lens, nlen)) return false;
                dist.build(lens + nlen, ndist);  // incomplete distance codes are legal
            }
            for (;;) {
                int sym = lit.decode(br);
                if (sym < 0) return false;
                if (sym < 256) {
                    out.push_back((uint8_t)sym);
                } else if (sym == 256) {
                    break;
                } else {
                    sym -= 257;
                    if (sym >= 29) return false;
                    size_t len = LBASE[sym] + br.bits(LEXT[sym]);
                    int ds = dist.decode(br);
                    if (ds < 0 || ds >= 30) return false;
                    size_t d = DBASE[ds] + br.bits(DEXT[ds]);
                    if (br.bad || d > out.size()) return false;
                    size_t from = out.size() - d;
                    for (size_t k = 0; k < len; ++k) out.push_back(out[from + k]);
                }
            }
        } else {
            return false;
        }
        if (last) return !br.bad;
    }
}

uint32_t crc32(const uint8_t* p, size_t n) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    uint32_t c = 0xffffffffu;
    for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xffu] ^ (c >> 8);
    return c ^ 0xffffffffu;
}

uint32_t adler32(const std::vector<uint8_t>& v) {
    uint32_t a = 1, b = 0;
    for (uint8_t x : v) {
        a = (a + x) % 65521u;
        b = (b + a) % 65521u;
    }
    return (b << 16) | a;
}

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

yk_status ifail(yk_status st, const std::string& m) {
    g_image_error = m;
    return st;
}

// PNG spec 9.2: reverse the per-scanline filters of one (sub)image in place
bool unfilter(uint8_t* data, size_t rows, size_t stride, size_t bpp) {
    const uint8_t* prev = nullptr;
    for (size_t y = 0; y < rows; ++y) {
        uint8_t* line = data + y * (stride + 1);
        const uint8_t ft = line[0];
        uint8_t* cur = line + 1;
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= bpp ? cur[x - bpp] : 0, b = prev ? prev[x] : 0, c = (prev && x >= bpp) ? prev[x - bpp] : 0;
            int pred;
            switch (ft) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: {
                    int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    break;
                }
                default: return false;
            }
            cur[x] = (uint8_t)(cur[x] + pred);
        }
        prev = cur;
    }
    return true;
}

yk_status decode_png(const std::vector<uint8_t>& f, uint32_t& W, uint32_t& H, std::vector<float>& rgb) {
    static const uint8_t SIG[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (f.size() < 8 || std::memcmp(f.data(), SIG, 8) != 0) return ifail(YK_ERR_UNSUPPORTED, "PNG: bad signature");
    size_t pos = 8;
    uint32_t depth = 0, ctype = 0, interlace = 0;
    bool have_hdr = false, have_trns = false, ended = false;
    std::vector<uint8_t> idat, plte;
    while (pos + 12 <= f.size() && !ended) {
        uint32_t len = be32(&f[pos]);
        if (pos + 12 + (size_t)len > f.size()) return ifail(YK_ERR_INVALID_ARGUMENT, "PNG: truncated chunk");
        const uint8_t* type = &f[pos + 4];
        const uint8_t* body = &f[pos + 8];
        if (crc32(type, 4 + (size_t)len) != be32(body + len)) return ifail(YK_ERR_INVALID_ARGUMENT, "PNG: CRC mismatch");
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) return ifail(YK_ERR_INVALID_ARGUMENT, "PNG: bad IHDR");
            W = be32(body);
            H = be32(body + 4);
            depth = body[8];
            ctype = body[9];
            interlace = body[12];
            if (body[10] != 0 || body[11] != 0 || interlace > 1 || W == 0 || H == 0) return ifail(YK_ERR_INVALID_ARGUMENT, "PNG: bad IHDR");
            have_hdr = true;
        } else if (!std::memcmp(type, "PLTE", 4)) {
            plte.assign(body, body + len);
        } else if (!std::memcmp(type, "tRNS", 4)) {
            have_trns = true;
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            ended = true;
        }
        pos += 12 + (size_t)len;
    }
    (void)have_trns;  // only adds an alpha channel, which the texture loader drops
    if (!have_hdr || idat.empty()) return ifail(YK_ERR_INVALID_ARGUMENT, "PNG: missing IHDR/IDAT");
    if (ctype == 0 || ctype == 4) return ifail(YK_ERR_UNSUPPORTED, "Unsupported image format");  // Luma / LumaA
    uint32_t channels;
    if (ctype == 2) channels = 3;
    else if (ctype == 6) channels = 4;
    else if (ctype == 3) channels = 1;
    else return ifail(YK_ERR_INVALID_ARGUMENT, "PNG: bad colour type");
    const bool depth_ok = ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8) : (depth == 8 || depth == 16);
    if (!depth_ok) return ifail(YK_ERR_INVALID_ARGUMENT, "PNG: bad bit depth");
    if (ctype == 3 && (plte.empty() || plte.size() % 3)) return ifail(YK_ERR_INVALID_ARGUMENT, "PNG: missing palette");
    if ((uint64_t)W * H > (1ull << 30)) return ifail(YK_ERR_UNSUPPORTED, "PNG: image too large");
    // zlib wrapper (RFC 1950)
    if (idat.size() < 6 || (idat[0] & 0x0f) != 8 || ((idat[0] << 8) | idat[1]) % 31 != 0 || (idat[1] & 0x20)) return ifail(YK_ERR_INVALID_ARGUMENT, "PNG: bad zlib header");
    std::vector<uint8_t> raw;
    if (!inflate(idat.data() + 2, idat.size() - 6, raw)) return ifail(YK_ERR_INVALID_ARGUMENT, "PNG: corrupt deflate stream");
    if (adler32(raw) != be32(&idat[idat.size() - 4])) return ifail(YK_ERR_INVALID_ARGUMENT, "PNG: Adler-32 mismatch");
    const size_t bits_pp = (size_t)channels * depth, bpp = bits_pp >= 8 ? bits_pp / 8 : 1;
    rgb.assign((size_t)W * H * 3, 0.0f);
    // one pass for non-interlaced, seven for Adam7 (PNG spec 8.2)
    static const uint32_t X0[7] = {0, 4, 0, 2, 0, 1, 0}, Y0[7] = {0, 0, 4, 0, 2, 0, 1}, DX[7] = {8, 8, 4, 4, 2, 2, 1}, DY[7] = {8, 8, 8, 4, 4, 2, 2};
    size_t off = 0;
    const int passes = interlace ? 7 : 1;
    for (int ps = 0; ps < passes; ++ps) {
        const uint32_t x0 = interlace ? X0[ps] : 0, y0 = interlace ? Y0[ps] : 0, dx = interlace ? DX[ps] : 1, dy = interlace ? DY[ps] : 1;
        const size_t pw = W > x0 ? (W - x0 + dx - 1) / dx : 0, ph = H > y0 ? (H - y0 + dy - 1) / dy : 0;
        if (pw == 0 || ph == 0) continue;
        const size_t stride = (pw * bits_pp + 7) / 8;
        if (off + ph * (stride + 1) > raw.size()) return ifail(YK_ERR_INVALID_ARGUMENT, "PNG: not enough image data");
        if (!unfilter(raw.data() + off, ph, stride, bpp)) return ifail(YK_ERR_INVALID_ARGUMENT, "PNG: bad filter type");
        for (size_t y = 0; y < ph; ++y) {
            const uint8_t* line = raw.data() + off + y * (stride + 1) + 1;
            for (size_t x = 0; x < pw; ++x) {
                float* o = &rgb[(((size_t)y0 + y * dy) * W + (x0 + x * dx)) * 3];
                if (ctype == 3) {
                    const size_t bit = x * depth;
                    const uint32_t idx = (line[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
                    if ((size_t)idx * 3 + 2 >= plte.size()) return ifail(YK_ERR_INVALID_ARGUMENT, "PNG: palette index out of range");
                    for (int c = 0; c < 3; ++c) o[c] = (float)plte[idx * 3 + c] / 255.0f;
                } else if (depth == 8) {
                    for (int c = 0; c < 3; ++c) o[c] = (float)line[x * channels + c] / 255.0f;
                } else {
                    for (int c = 0; c < 3; ++c) {
                        const uint8_t* q = &line[(x * channels + c) * 2];
                        o[c] = (float)(((uint32_t)q[0] << 8) | q[1]) / 65535.0f;
                    }
                }
            }
        }
        off += ph * (stride + 1);
    }
    return YK_OK;
}

}  // namespace

bool yk_img::yk_inflate_zlib(const uint8_t* src, size_t n, std::vector<uint8_t>& out) {
    out.clear();
    if (n < 6 || (src[0] & 0x0f) != 8 || ((src[0] << 8) | src[1]) % 31 != 0 || (src[1] & 0x20)) return false;
    if (!inflate(src + 2, n - 6, out)) return false;
    return adler32(out) == be32(src + n - 4);
}

// used by the pbrt loader (yk_loaders.cpp)
yk_status yk_image_decode_file(const std::string& path, uint32_t& w, uint32_t& h, std::vector<float>& rgb, std::string& err) {
    std::vector<uint8_t> bytes;
    if (!yk::read_file(path, bytes)) {
        err = "Could not open '" + path + "'";
        return YK_ERR_INVALID_ARGUMENT;
    }
    yk_status st;
    bool is_png = false;
    st = yk_img::decode_by_extension(path, bytes, w, h, rgb, err, is_png);
    if (!is_png) {
        if (st != YK_OK) err += " (" + path + ")";
        return st;
    }
    try {
        st = decode_png(bytes, w, h, rgb);
    } catch (const std::exception& e) {
        st = ifail(YK_ERR_INVALID_ARGUMENT, std::string("PNG: ") + e.what());
    }
    if (st != YK_OK) err = g_image_error + " (" + path + ")";
    return st;
}

extern "C" {

const char* yk_loader_last_error(void);
void yk_loader_set_error(const char* msg);

yk_status yk_image_texture_load(const char* path, yk_texture_desc* out) {
    if (!path || !out) return YK_ERR_INVALID_ARGUMENT;
    out->width = out->height = 0;
    out->rgb = nullptr;
    uint32_t w = 0, h = 0;
    std::vector<float> rgb;
    std::string err;
    yk_status st = yk_image_decode_file(path, w, h, rgb, err);
    if (st != YK_OK) {
        yk_loader_set_error(err.c_str());
        return st;
    }
    float* p = new float[rgb.size()];
    std::memcpy(p, rgb.data(), rgb.size() * sizeof(float));
    out->width = w;
    out->height = h;
    out->rgb = p;
    return YK_OK;
}

// ---- film output ---------------------------------------------------------------------
// OpenEXR 2 single-part scan-line file, three FLOAT channels, NO_COMPRESSION (one scan line
// per chunk), INCREASING_Y.  Layout per the OpenEXR file-layout specification.
yk_status yk_write_exr(const char* path, uint32_t width, uint32_t height, const float* rgb) {
    if (!path || !rgb || width == 0 || height == 0 || width > (1u << 26) || height > (1u << 26)) return YK_ERR_INVALID_ARGUMENT;
    std::vector<uint8_t> hd;
    auto put = [&](const void* p, size_t n) { hd.insert(hd.end(), (const uint8_t*)p, (const uint8_t*)p + n); };
    auto str = [&](const char* z) { put(z, std::strlen(z) + 1); };
    auto i32 = [&](int32_t v) { put(&v, 4); };
    auto f32 = [&](float v) { put(&v, 4); };
    auto attr = [&](const char* name, const char* type, int32_t size) {
        str(name);
        str(type);
        i32(size);
    };
    const uint32_t magic = 20000630u, version = 2u;
    put(&magic, 4);
    put(&version, 4);
    attr("channels", "chlist", 3 * (2 + 16) + 1);
    for (const char* c : {"B", "G", "R"}) {
        str(c);
        i32(2);  // FLOAT
        const uint8_t lin[4] = {0, 0, 0, 0};
        put(lin, 4);
        i32(1);
        i32(1);
    }
    hd.push_back(0);
    attr("compression", "compression", 1);
    hd.push_back(0);
    attr("dataWindow", "box2i", 16);
    i32(0); i32(0); i32((int32_t)width - 1); i32((int32_t)height - 1);
    attr("displayWindow", "box2i", 16);
    i32(0); i32(0); i32((int32_t)width - 1); i32((int32_t)height - 1);
    attr("lineOrder", "lineOrder", 1);
    hd.push_back(0);
    attr("pixelAspectRatio", "float", 4);
    f32(1.0f);
    attr("screenWindowCenter", "v2f", 8);
    f32(0.0f); f32(0.0f);
    attr("screenWindowWidth", "float", 4);
    f32(1.0f);
    hd.push_back(0);
    std::FILE* f = std::fopen(path, "wb");
    if (!f) return ifail(YK_ERR_INVALID_ARGUMENT, std::string("Error writing EXR to '") + path + "'");
    const uint64_t line_bytes = 8ull + 12ull * width;
    bool ok = std::fwrite(hd.data(), 1, hd.size(), f) == hd.size();
    uint64_t off = hd.size() + 8ull * height;
    for (uint32_t y = 0; y < height && ok; ++y, off += line_bytes) ok = std::fwrite(&off, 8, 1, f) == 1;
    std::vector<float> line(3 * (size_t)width);
    for (uint32_t y = 0; y < height && ok; ++y) {
        const float* src = rgb + 3 * (size_t)y * width;
        for (uint32_t x = 0; x < width; ++x) {
            line[x] = src[3 * x + 2];              // B
            line[width + x] = src[3 * x + 1];      // G
            line[2 * (size_t)width + x] = src[3 * x];  // R
        }
        const int32_t yy = (int32_t)y, sz = (int32_t)(12u * width);
        ok = std::fwrite(&yy, 4, 1, f) == 1 && std::fwrite(&sz, 4, 1, f) == 1 && std::fwrite(line.data(), 4, line.size(), f) == line.size();
    }
    ok = (std::fclose(f) == 0) && ok;
    return ok ? YK_OK : ifail(YK_ERR_INVALID_ARGUMENT, std::string("Error writing EXR to '") + path + "'");
}

// Portable Float Map: "PF", little endian (negative scale), rows stored bottom to top
yk_status yk_write_pfm(const char* path, uint32_t width, uint32_t height, const float* rgb) {
    if (!path || !rgb || width == 0 || height == 0) return YK_ERR_INVALID_ARGUMENT;
    std::FILE* f = std::fopen(path, "wb");
    if (!f) return ifail(YK_ERR_INVALID_ARGUMENT, std::string("Error writing PFM to '") + path + "'");
    bool ok = std::fprintf(f, "PF\n%u %u\n-1.0\n", width, height) > 0;
    for (uint32_t y = height; y-- > 0 && ok;) ok = std::fwrite(rgb + 3 * (size_t)y * width, 4, 3 * (size_t)width, f) == 3 * (size_t)width;
    ok = (std::fclose(f) == 0) && ok;
    return ok ? YK_OK : YK_ERR_INVALID_ARGUMENT;
}

void yk_image_texture_free(yk_texture_desc* tex) {
    if (!tex) return;
    delete[] tex->rgb;
    tex->rgb = nullptr;
    tex->width = tex->height = 0;
}

}  // extern "C"
