// shared by yk_image.cpp (PNG, inflate, writers) and yk_image_formats.cpp (the other containers)
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/yuki_hip.h"

namespace yk_img {

// RFC 1950 stream (2-byte header, deflate data, Adler-32) -> bytes; false on any corruption
bool yk_inflate_zlib(const uint8_t* src, size_t n, std::vector<uint8_t>& out);

// Decoder picked from the file extension like image::io::Reader::open.  `is_png` = the caller's
// PNG decoder must handle it (nothing decoded yet).
yk_status decode_by_extension(const std::string& path, const std::vector<uint8_t>& bytes, uint32_t& w, uint32_t& h, std::vector<float>& rgb, std::string& err,
                              bool& is_png);

}  // namespace yk_img
