// yk_loaders.cpp — scene input for the hot path: the
// reference's PLY loader and a subset of its pbrt-v3 loader, producing the flattened
// world-space arrays yk_scene_create consumes.
//
//   Scene::ply + ply::load          yuki/src/scene/mod.rs:99-152, scene/ply.rs:19-130,217-284
//   pbrt::load (subset)             yuki/src/scene/pbrt/mod.rs:94-936, lexer.rs, param_set.rs, cie.rs
//
// Behaviour follows the reference, including its quirks: only
// float32 vertex properties are read, faces must be int/uint lists, fan
// triangulation; pbrt: camera fov is FoV::Y switched to X when res.y >= res.x,
// LookAt up is normalised, AreaLightSource / Integrator / Sampler are parsed and
// ignored, matte sigma goes through to_radians twice, TransformEnd pops the
// graphics-state stack, unknown directives (Transform, ConcatTransform, Identity,
// ...) abort the load.  Where the reference panics we return an error.
// Texture "spectrum" "imagemap" loads the file through yk_image_texture_load (yk_image.cpp) and a
// matte Kd may name it (scene/pbrt/mod.rs:719-735, textures/image_texture.rs:66-141).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <atomic>
#include <map>
#include <thread>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/yuki_hip.h"
#include "yk_bsdf.h"
#include "yk_host.h"
#include "yk_libm.h"
#include "yk_math.h"

using namespace yk;

struct yk_loaded_scene {
    std::vector<float> points, normals, uvs;
    std::vector<uint32_t> indices, tri_mesh;
    std::vector<uint32_t> shape_order, shape_order_flat;  // file order of shapes: triangle id | 0x80000000+sphere id
    std::vector<int32_t> tri_material, tri_area_light;
    std::vector<yk_mesh_desc> meshes;
    std::vector<yk_sphere_desc> spheres;
    std::vector<yk_material_desc> materials;
    std::vector<yk_light_desc> lights;
    std::vector<std::vector<float>> texture_data;
    std::vector<yk_texture_desc> textures;
    float background[3] = {0, 0, 0};
    yk_camera_params camera;
    uint16_t tile_dim = 16;
    uint32_t split_method = YK_SPLIT_SAH, max_shapes_in_node = 1;
    bool any_normals = false, any_uvs = false;
};

yk_status yk_image_decode_file(const std::string& path, uint32_t& w, uint32_t& h, std::vector<float>& rgb, std::string& err);  // yk_image.cpp

static thread_local std::string g_loader_error;
static yk_status lfail(yk_status st, const std::string& msg) {
    g_loader_error = msg;
    return st;
}

// ------------------------------------------------------------------ helpers
static yk_material_desc make_mat(uint32_t kind, const float a[3], const float b[3], float c, bool remap) {
    yk_material_desc m;
    std::memset(&m, 0, sizeof(m));
    m.kind = kind;
    for (int k = 0; k < 3; ++k) {
        m.a[k] = a ? a[k] : 0.0f;
        m.b[k] = b ? b[k] : 0.0f;
    }
    m.c = c;
    m.flags = remap ? 1u : 0u;
    return m;
}

// Transform::swaps_handedness, transform.rs:85-91
static bool swaps_handedness(const float* m) {
    float det = m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) + m[2] * (m[4] * m[9] - m[5] * m[8]);
    return det < 0.0f;
}

// Mesh::new (shapes/mesh.rs:20-43): appends a mesh, pre-transforming points and normals
static void add_mesh(yk_loaded_scene& s, const Xf& t, const std::vector<uint32_t>& idx, const std::vector<float>& pts, const std::vector<float>& nrm,
                     const std::vector<float>& uv, int material) {
    const uint32_t base = (uint32_t)(s.points.size() / 3);
    const size_t nv = pts.size() / 3;
    for (size_t i = 0; i < nv; ++i) {
        V3 p = xf_point(t.m, V3{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]});
        s.points.push_back(p.x);
        s.points.push_back(p.y);
        s.points.push_back(p.z);
    }
    const bool hn = !nrm.empty(), hu = !uv.empty();
    s.normals.resize(s.points.size(), 0.0f);
    s.uvs.resize(s.points.size() / 3 * 2, 0.0f);
    if (hn) {
        s.any_normals = true;
        for (size_t i = 0; i < nv; ++i) {
            V3 n = xf_normal(t.mi, V3{nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]});
            s.normals[3 * (base + i)] = n.x;
            s.normals[3 * (base + i) + 1] = n.y;
            s.normals[3 * (base + i) + 2] = n.z;
        }
    }
    if (hu) {
        s.any_uvs = true;
        for (size_t i = 0; i < nv; ++i) {
            s.uvs[2 * (base + i)] = uv[2 * i];
            s.uvs[2 * (base + i) + 1] = uv[2 * i + 1];
        }
    }
    yk_mesh_desc md;
    md.has_normals = hn;
    md.has_uvs = hu;
    md.swaps_handedness = swaps_handedness(t.m);
    md.pad = 0;
    const uint32_t mesh_id = (uint32_t)s.meshes.size();
    s.meshes.push_back(md);
    for (size_t k = 0; k + 2 < idx.size(); k += 3) {
        s.indices.push_back(base + idx[k]);
        s.indices.push_back(base + idx[k + 1]);
        s.indices.push_back(base + idx[k + 2]);
        s.shape_order.push_back((uint32_t)s.tri_mesh.size());
        s.tri_mesh.push_back(mesh_id);
        s.tri_material.push_back(material);
        s.tri_area_light.push_back(-1);
    }
}

// ------------------------------------------------------------------ PLY
namespace {

enum PlyType { T_I8, T_U8, T_I16, T_U16, T_I32, T_U32, T_F32, T_F64, T_BAD };
PlyType ply_type(const std::string& s) {
    if (s == "char" || s == "int8") return T_I8;
    if (s == "uchar" || s == "uint8") return T_U8;
    if (s == "short" || s == "int16") return T_I16;
    if (s == "ushort" || s == "uint16") return T_U16;
    if (s == "int" || s == "int32") return T_I32;
    if (s == "uint" || s == "uint32") return T_U32;
    if (s == "float" || s == "float32") return T_F32;
    if (s == "double" || s == "float64") return T_F64;
    return T_BAD;
}
size_t ply_size(PlyType t) {
    switch (t) {
        case T_I8: case T_U8: return 1;
        case T_I16: case T_U16: return 2;
        case T_I32: case T_U32: case T_F32: return 4;
        case T_F64: return 8;
        default: return 0;
    }
}
struct PlyProp {
    std::string name;
    bool is_list = false;
    PlyType type = T_BAD, count_type = T_BAD;
};
struct PlyElement {
    std::string name;
    size_t count = 0;
    std::vector<PlyProp> props;
};

struct PlyReader {
    const std::vector<unsigned char>& buf;
    size_t pos;
    int format;  // 0 ascii, 1 little, 2 big
    bool ok = true;
    PlyReader(const std::vector<unsigned char>& b, size_t p, int f) : buf(b), pos(p), format(f) {}
    // one scalar as double (and raw float bits when the type is f32)
    bool scalar(PlyType t, double& v, float& f32v) {
        if (format == 0) {
            while (pos < buf.size() && (buf[pos] == ' ' || buf[pos] == '\t' || buf[pos] == '\n' || buf[pos] == '\r')) ++pos;
            size_t st = pos;
            while (pos < buf.size() && !(buf[pos] == ' ' || buf[pos] == '\t' || buf[pos] == '\n' || buf[pos] == '\r')) ++pos;
            if (st == pos) return ok = false;
            std::string tok(buf.begin() + st, buf.begin() + pos);
            char* end = nullptr;
            if (t == T_F32) {
                f32v = std::strtof(tok.c_str(), &end);  // Rust str::parse::<f32> is correctly rounded, as is strtof
                v = f32v;
            } else if (t == T_F64) {
                v = std::strtod(tok.c_str(), &end);
                f32v = (float)v;
            } else {
                v = (double)std::strtoll(tok.c_str(), &end, 10);
                f32v = (float)v;
            }
            return ok = (end && *end == 0);
        }
        size_t n = ply_size(t);
        if (pos + n > buf.size()) return ok = false;
        unsigned char b[8];
        for (size_t i = 0; i < n; ++i) b[i] = format == 1 ? buf[pos + i] : buf[pos + n - 1 - i];
        pos += n;
        switch (t) {
            case T_I8: v = (int8_t)b[0]; break;
            case T_U8: v = b[0]; break;
            case T_I16: { int16_t x; std::memcpy(&x, b, 2); v = x; break; }
            case T_U16: { uint16_t x; std::memcpy(&x, b, 2); v = x; break; }
            case T_I32: { int32_t x; std::memcpy(&x, b, 4); v = x; break; }
            case T_U32: { uint32_t x; std::memcpy(&x, b, 4); v = x; break; }
            case T_F32: { float x; std::memcpy(&x, b, 4); f32v = x; v = x; break; }
            case T_F64: { double x; std::memcpy(&x, b, 8); v = x; f32v = (float)x; break; }
            default: return ok = false;
        }
        return true;
    }
};

}  // namespace

// The payload of one PLY file as ply::load reads it (scene/ply.rs:19-130), before any transform.  Pure: touches nothing but its
// arguments (errors come back through the thread-local loader error of the CALLING thread), so several files are read at once.
struct PlyMesh {
    std::vector<float> pts, nrm, uv;
    std::vector<uint32_t> indices;
};
static yk_status read_ply_mesh(const std::string& path, PlyMesh& out) {
    std::vector<unsigned char> buf;
    if (!read_file(path, buf)) return lfail(YK_ERR_INVALID_ARGUMENT, "Could not open '" + path + "'");
    // ---- header
    size_t pos = 0;
    auto next_line = [&](std::string& line) -> bool {
        if (pos >= buf.size()) return false;
        size_t e = pos;
        while (e < buf.size() && buf[e] != '\n') ++e;
        line.assign(buf.begin() + pos, buf.begin() + e);
        if (!line.empty() && line.back() == '\r') line.pop_back();
        pos = e < buf.size() ? e + 1 : e;
        return true;
    };
    std::string line;
    if (!next_line(line) || line != "ply") return lfail(YK_ERR_INVALID_ARGUMENT, "PLY: missing magic");
    int format = -1;
    std::vector<PlyElement> elements;
    bool ended = false;
    while (next_line(line)) {
        std::istringstream ls(line);
        std::string kw;
        ls >> kw;
        if (kw == "format") {
            std::string fm;
            ls >> fm;
            format = fm == "ascii" ? 0 : (fm == "binary_little_endian" ? 1 : (fm == "binary_big_endian" ? 2 : -1));
        } else if (kw == "element") {
            PlyElement e;
            ls >> e.name >> e.count;
            elements.push_back(e);
        } else if (kw == "property") {
            if (elements.empty()) return lfail(YK_ERR_INVALID_ARGUMENT, "PLY: property before element");
            PlyProp p;
            std::string t;
            ls >> t;
            if (t == "list") {
                std::string ct, it;
                ls >> ct >> it >> p.name;
                p.is_list = true;
                p.count_type = ply_type(ct);
                p.type = ply_type(it);
            } else {
                p.type = ply_type(t);
                ls >> p.name;
            }
            if (p.type == T_BAD || (p.is_list && p.count_type == T_BAD)) return lfail(YK_ERR_INVALID_ARGUMENT, "PLY: unknown property type");
            elements.back().props.push_back(p);
        } else if (kw == "end_header") {
            ended = true;
            break;
        }  // comment / obj_info: ignored
    }
    if (!ended || format < 0) return lfail(YK_ERR_INVALID_ARGUMENT, "PLY: bad header");
    // ---- is_valid (ply.rs:146-215)
    const PlyElement *ve = nullptr, *fe = nullptr;
    for (const PlyElement& e : elements) {
        if (e.name == "vertex") ve = &e;
        if (e.name == "face") fe = &e;
    }
    auto has = [](const PlyElement* e, const char* n) {
        for (const PlyProp& p : e->props)
            if (p.name == n) return true;
        return false;
    };
    if (!ve || !fe || !has(ve, "x") || !has(ve, "y") || !has(ve, "z") || !(has(fe, "vertex_index") || has(fe, "vertex_indices")))
        return lfail(YK_ERR_UNSUPPORTED, "PLY: Unsupported content");
    // ---- payload
    PlyReader rd(buf, pos, format);
    std::vector<float>&pts = out.pts, &nrm = out.nrm, &uv = out.uv;
    std::vector<uint32_t>& indices = out.indices;
    bool saw_normal = false, saw_uv = false;
    for (const PlyElement& e : elements) {
        const bool is_v = &e == ve, is_f = &e == fe;
        for (size_t i = 0; i < e.count; ++i) {
            float P[3] = {0, 0, 0}, N[3] = {0, 0, 0}, UV[2] = {0, 0};
            bool hn = false, hu = false;
            std::vector<long long> face;
            bool face_set = false;
            for (const PlyProp& p : e.props) {
                double v;
                float fv;
                if (!p.is_list) {
                    if (!rd.scalar(p.type, v, fv)) return lfail(YK_ERR_INVALID_ARGUMENT, "PLY: truncated payload");
                    if (is_v && p.type == T_F32) {  // only Property::Float is consumed (ply.rs:237-256)
                        if (p.name == "x") P[0] = fv;
                        else if (p.name == "y") P[1] = fv;
                        else if (p.name == "z") P[2] = fv;
                        else if (p.name == "nx") { hn = true; N[0] = fv; N[1] = N[2] = 0.0f; }
                        else if (p.name == "ny") { if (!hn) return lfail(YK_ERR_INVALID_ARGUMENT, "PLY: 'ny' before 'nx'"); N[1] = fv; }
                        else if (p.name == "nz") { if (!hn) return lfail(YK_ERR_INVALID_ARGUMENT, "PLY: 'nz' before 'nx'"); N[2] = fv; }
                        else if (p.name == "u") { hu = true; UV[0] = fv; UV[1] = 0.0f; }
                        else if (p.name == "v") { if (!hu) return lfail(YK_ERR_INVALID_ARGUMENT, "PLY: 'v' before 'u'"); UV[1] = fv; }
                    }
                } else {
                    if (!rd.scalar(p.count_type, v, fv) || v < 0) return lfail(YK_ERR_INVALID_ARGUMENT, "PLY: truncated payload");
                    size_t cnt = (size_t)v;
                    std::vector<long long> items(cnt);
                    for (size_t k = 0; k < cnt; ++k) {
                        if (!rd.scalar(p.type, v, fv)) return lfail(YK_ERR_INVALID_ARGUMENT, "PLY: truncated payload");
                        items[k] = (long long)v;
                    }
                    // only ListInt / ListUInt are consumed (ply.rs:272-281)
                    if (is_f && (p.name == "vertex_index" || p.name == "vertex_indices") && (p.type == T_I32 || p.type == T_U32)) {
                        face = items;
                        face_set = true;
                    }
                }
            }
            if (is_v) {
                pts.insert(pts.end(), P, P + 3);
                if (hn) { nrm.insert(nrm.end(), N, N + 3); saw_normal = true; }
                if (hu) { uv.insert(uv.end(), UV, UV + 2); saw_uv = true; }
            }
            if (is_f) {
                if (!face_set || face.empty()) return lfail(YK_ERR_INVALID_ARGUMENT, "PLY: face without int/uint vertex indices");
                for (long long q : face)
                    if (q < 0) return lfail(YK_ERR_INVALID_ARGUMENT, "Negative PLY index");
                for (size_t k = 1; k + 1 < face.size(); ++k) {  // fan, ply.rs:81-93
                    indices.push_back((uint32_t)face[0]);
                    indices.push_back((uint32_t)face[k]);
                    indices.push_back((uint32_t)face[k + 1]);
                }
            }
        }
    }
    const size_t nv = pts.size() / 3;
    if (nv == 0 || indices.empty()) return lfail(YK_ERR_INVALID_ARGUMENT, "PLY: empty mesh");
    for (uint32_t q : indices)
        if (q >= nv) return lfail(YK_ERR_INVALID_ARGUMENT, "PLY: vertex index out of range");
    if (saw_normal && nrm.size() != pts.size()) return lfail(YK_ERR_INVALID_ARGUMENT, "PLY: normals on a subset of the vertices");
    if (saw_uv && uv.size() / 2 != nv) return lfail(YK_ERR_INVALID_ARGUMENT, "PLY: uvs on a subset of the vertices");
    return YK_OK;
}

// Mesh::new for a PLY payload.  No transform given (Scene::ply): scale / translate into the unit cube (ply.rs:99-108).
static void add_ply_mesh(yk_loaded_scene& s, const PlyMesh& m, const Xf* transform, int material) {
    const std::vector<float>& pts = m.pts;
    const size_t nv = pts.size() / 3;
    Xf t;
    if (transform) {
        t = *transform;
    } else {  // ply.rs:99-108
        float lo[3] = {3.40282347e+38f, 3.40282347e+38f, 3.40282347e+38f}, hi[3] = {-3.40282347e+38f, -3.40282347e+38f, -3.40282347e+38f};
        for (size_t i = 0; i < nv; ++i)
            for (int k = 0; k < 3; ++k) {
                lo[k] = rmin(lo[k], pts[3 * i + k]);
                hi[k] = rmax(hi[k], pts[3 * i + k]);
            }
        float dg[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        float center[3] = {lo[0] + dg[0] / 2.0f, lo[1] + dg[1] / 2.0f, lo[2] + dg[2] / 2.0f};
        float mesh_scale = 1.0f / rmax(dg[0], rmax(dg[1], dg[2]));
        t = xf_mul(xf_scale(mesh_scale, mesh_scale, mesh_scale), xf_translation(-center[0], -center[1], -center[2]));
    }
    add_mesh(s, t, m.indices, pts, m.nrm, m.uv, material);
}

static yk_status load_ply_mesh(const std::string& path, const Xf* transform, yk_loaded_scene& s, int material) {
    PlyMesh m;
    yk_status st = read_ply_mesh(path, m);
    if (st != YK_OK) return st;
    add_ply_mesh(s, m, transform, material);
    return YK_OK;
}

// ------------------------------------------------------------------ pbrt-v3 subset
namespace {

struct Tok {
    enum Kind { Number, String, LBracket, RBracket, Ident, End, Error } kind = End;
    double num = 0;
    std::string text;
};


// str::parse::<f64> grammar (core::num::dec2flt): [+-]? (inf | infinity | nan | digits[.digits][e[+-]digits])
bool rust_f64_grammar(const std::string& s) {
    size_t i = 0, n = s.size();
    if (i < n && (s[i] == '+' || s[i] == '-')) ++i;
    std::string rest = s.substr(i);
    for (auto& c : rest) c = (char)std::tolower((unsigned char)c);
    if (rest == "inf" || rest == "infinity" || rest == "nan") return true;
    size_t d0 = i;
    while (i < n && s[i] >= '0' && s[i] <= '9') ++i;
    size_t nd = i - d0;
    if (i < n && s[i] == '.') {
        ++i;
        size_t f0 = i;
        while (i < n && s[i] >= '0' && s[i] <= '9') ++i;
        nd += i - f0;
    }
    if (nd == 0) return false;
    if (i < n && (s[i] == 'e' || s[i] == 'E')) {
        ++i;
        if (i < n && (s[i] == '+' || s[i] == '-')) ++i;
        size_t e0 = i;
        while (i < n && s[i] >= '0' && s[i] <= '9') ++i;
        if (i == e0) return false;
    }
    return i == n;
}

// the identifiers the reference's lexer knows (pbrt/lexer.rs:304-345); anything else is a lexer error
bool is_directive(const std::string& s) {
    static const char* k[] = {"Accelerator", "ActiveTransform", "All", "AreaLightSource", "AttributeBegin", "AttributeEnd", "Camera", "ConcatTransform",
        "CoordinateSystem", "CoordSysTransform", "EndTime", "Film", "Identity", "Include", "Integrator", "LightSource", "LookAt", "MakeNamedMedium",
        "MakeNamedMaterial", "Material", "MediumInterface", "NamedMaterial", "ObjectBegin", "ObjectEnd", "ObjectInstance", "PixelFilter",
        "ReverseOrientation", "Rotate", "Sampler", "Scale", "Shape", "StartTime", "Texture", "TransformBegin", "TransformEnd", "TransformTimes",
        "Transform", "Translate", "WorldBegin", "WorldEnd"};
    for (const char* d : k)
        if (s == d) return true;
    return false;
}

// pbrt/lexer.rs:68-245
struct Lexer {
    std::string in;
    size_t pos = 0;
    std::string error;
    Tok next() {
        Tok t;
        for (;;) {
            while (pos < in.size() && (in[pos] == ' ' || in[pos] == '\t' || in[pos] == '\n' || in[pos] == '\r')) ++pos;
            if (pos >= in.size()) return t;  // End
            if (in[pos] == '#') {
                while (pos < in.size() && in[pos] != '\n' && in[pos] != '\r') ++pos;
                continue;
            }
            break;
        }
        char c = in[pos];
        if (c == '"') {
            size_t st = ++pos;
            for (;;) {
                if (pos >= in.size()) { t.kind = Tok::Error; error = "UnexpectedEndOfInput"; return t; }
                char d = in[pos++];
                if (d == '"') { t.kind = Tok::String; t.text = in.substr(st, pos - 1 - st); return t; }
                if (d == '\\') { if (pos >= in.size()) { t.kind = Tok::Error; error = "UnexpectedEndOfInput"; return t; } ++pos; }
                else if (d == '\n') { t.kind = Tok::Error; error = "UnterminatedString"; return t; }
            }
        }
        if (c == '[') { ++pos; t.kind = Tok::LBracket; return t; }
        if (c == ']') { ++pos; t.kind = Tok::RBracket; return t; }
        // identifier / number: runs to whitespace or ']' (left for the next call).  As in the
        // reference, '#', '"' and '[' inside an identifier abandon it and start their own token.
        size_t st = pos;
        for (;;) {
            if (pos >= in.size()) return t;  // the reference reports EndOfInput for an identifier that runs into EOF
            char d = in[pos];
            if (d == ' ' || d == '\t' || d == '\n' || d == '\r' || d == ']') break;
            if (d == '#' || d == '"' || d == '[') return next();
            ++pos;
        }
        t.text = in.substr(st, pos - st);
        if (pos < in.size() && in[pos] != ']') ++pos;  // the terminating whitespace is consumed
        char f0 = t.text[0];
        if (f0 == '-' || f0 == '.' || (f0 >= '0' && f0 <= '9')) {
            if (!rust_f64_grammar(t.text)) { t.kind = Tok::Error; error = "InvalidNumber"; return t; }
            t.num = std::strtod(t.text.c_str(), nullptr);
            t.kind = Tok::Number;
        } else {
            t.kind = is_directive(t.text) ? Tok::Ident : Tok::Error;
            if (t.kind == Tok::Error) error = "UnknownIdentifier '" + t.text + "'";
        }
        return t;
    }
};

// pbrt/param_set.rs: typed (name, values) lists searched linearly.  find_* of ONE value
// return the first item of that name that holds exactly one value; find_*s return the
// first item of that name.
template <class T>
struct Items {
    std::vector<std::pair<std::string, std::vector<T>>> v;
    void add(const std::string& n, const std::vector<T>& x) { v.emplace_back(n, x); }
    const std::vector<T>* one(const char* n, size_t width) const {
        for (auto& it : v)
            if (it.first == n && it.second.size() == width) return &it.second;
        return nullptr;
    }
    const std::vector<T>* many(const char* n) const {
        for (auto& it : v)
            if (it.first == n) return &it.second;
        return nullptr;
    }
};
struct ParamSet {
    Items<bool> bools;
    Items<float> floats, uvs, spectra, points, normals;  // vector-valued items are stored flattened
    Items<int> ints;
    Items<std::string> strings;
    float f32(const char* n, float d) const { auto i = floats.one(n, 1); return i ? (*i)[0] : d; }
    int i32(const char* n, int d) const { auto i = ints.one(n, 1); return i ? (*i)[0] : d; }
    bool boolean(const char* n, bool d) const { auto i = bools.one(n, 1); return i ? (*i)[0] : d; }
    std::string str(const char* n, const char* d) const { auto i = strings.one(n, 1); return i ? (*i)[0] : std::string(d); }
    void vec3(const Items<float>& m, const char* n, const float d[3], float out[3]) const {
        auto i = m.one(n, 3);
        const float* s = i ? i->data() : d;
        out[0] = s[0]; out[1] = s[1]; out[2] = s[2];
    }
};

// pbrt/cie.rs (Wyman, Sloan, Shirley fits) — f32::exp = glibc's expf, restated in yk_libm.h
float expf_once(float x) { return yk::det_expf(x); }
float x_fit_1931(float l) {
    float t1 = (l - 442.0f) * (l < 442.0f ? 0.0624f : 0.0374f);
    float t2 = (l - 599.8f) * (l < 599.8f ? 0.0264f : 0.0323f);
    float t3 = (l - 501.1f) * (l < 501.1f ? 0.0490f : 0.0382f);
    return 0.362f * expf_once(-0.5f * t1 * t1) + 1.056f * expf_once(-0.5f * t2 * t2) - 0.065f * expf_once(-0.5f * t3 * t3);
}
float y_fit_1931(float l) {
    float t1 = (l - 568.8f) * (l < 568.8f ? 0.0213f : 0.0247f);
    float t2 = (l - 530.9f) * (l < 530.9f ? 0.0613f : 0.0322f);
    return 0.821f * expf_once(-0.5f * t1 * t1) + 0.286f * expf_once(-0.5f * t2 * t2);
}
float z_fit_1931(float l) {
    float t1 = (l - 437.0f) * (l < 437.0f ? 0.0845f : 0.0278f);
    float t2 = (l - 459.0f) * (l < 459.0f ? 0.0385f : 0.0725f);
    return 1.217f * expf_once(-0.5f * t1 * t1) + 0.681f * expf_once(-0.5f * t2 * t2);
}
// pbrt/mod.rs:979-1016 — note: the reference sorts unsorted input into temporaries
// and then DISCARDS the recursive result, integrating the unsorted data (sic)
void sampled_spectrum_into_rgb(const std::vector<float>& lambda, const std::vector<float>& samples, float rgb[3]) {
    float X = 0.0f, Y = 0.0f, Z = 0.0f;
    for (size_t i = 0; i < lambda.size(); ++i) {
        X += x_fit_1931(lambda[i]) * samples[i];
        Y += y_fit_1931(lambda[i]) * samples[i];
        Z += z_fit_1931(lambda[i]) * samples[i];
    }
    float sum_scale = (lambda.back() - lambda.front()) / (float)lambda.size();
    X *= sum_scale;
    Y *= sum_scale;
    Z *= sum_scale;
    rgb[0] = 3.240479f * X - 1.537150f * Y - 0.498535f * Z;
    rgb[1] = -0.969256f * X + 1.875991f * Y + 0.041556f * Z;
    rgb[2] = 0.055648f * X - 0.204043f * Y + 1.057311f * Z;
}

// measured copper n/k used as the `metal` defaults (pbrt/mod.rs:1027-1105; data)
const float COPPER_WAVELENGTHS[56] = {298.7570554f, 302.4004341f, 306.1337728f, 309.960445f, 313.8839949f, 317.9081487f, 322.036826f, 326.2741526f,
    330.6244747f, 335.092373f, 339.6826795f, 344.4004944f, 349.2512056f, 354.2405086f, 359.374429f, 364.6593471f, 370.1020239f, 375.7096303f,
    381.4897785f, 387.4505563f, 393.6005651f, 399.9489613f, 406.5055016f, 413.2805933f, 420.2853492f, 427.5316483f, 435.0322035f, 442.8006357f,
    450.8515564f, 459.2006593f, 467.8648226f, 476.8622231f, 486.2124627f, 495.936712f, 506.0578694f, 516.6007417f, 527.5922468f, 539.0616435f,
    551.0407911f, 563.5644455f, 576.6705953f, 590.4008476f, 604.8008683f, 619.92089f, 635.8162974f, 652.5483053f, 670.1847459f, 688.8009889f,
    708.4810171f, 729.3186941f, 751.4192606f, 774.9011125f, 799.8979226f, 826.5611867f, 855.0632966f, 885.6012714f};
const float COPPER_N[56] = {1.400313f, 1.38f, 1.358438f, 1.34f, 1.329063f, 1.325f, 1.3325f, 1.34f, 1.334375f, 1.325f, 1.317812f, 1.31f, 1.300313f, 1.29f,
    1.281563f, 1.27f, 1.249062f, 1.225f, 1.2f, 1.18f, 1.174375f, 1.175f, 1.1775f, 1.18f, 1.178125f, 1.175f, 1.172812f, 1.17f, 1.165312f, 1.16f, 1.155312f,
    1.15f, 1.142812f, 1.135f, 1.131562f, 1.12f, 1.092437f, 1.04f, 0.950375f, 0.826f, 0.645875f, 0.468f, 0.35125f, 0.272f, 0.230813f, 0.214f, 0.20925f,
    0.213f, 0.21625f, 0.223f, 0.2365f, 0.25f, 0.254188f, 0.26f, 0.28f, 0.3f};
const float COPPER_K[56] = {1.662125f, 1.687f, 1.703313f, 1.72f, 1.744563f, 1.77f, 1.791625f, 1.81f, 1.822125f, 1.834f, 1.85175f, 1.872f, 1.89425f, 1.916f,
    1.931688f, 1.95f, 1.972438f, 2.015f, 2.121562f, 2.21f, 2.177188f, 2.13f, 2.160063f, 2.21f, 2.249938f, 2.289f, 2.326f, 2.362f, 2.397625f, 2.433f,
    2.469187f, 2.504f, 2.535875f, 2.564f, 2.589625f, 2.605f, 2.595562f, 2.583f, 2.5765f, 2.599f, 2.678062f, 2.809f, 3.01075f, 3.24f, 3.458187f, 3.67f,
    3.863125f, 4.05f, 4.239563f, 4.43f, 4.619563f, 4.817f, 5.034125f, 5.26f, 5.485625f, 5.717f};

// transforms::rotation, math/transforms.rs:98-127 (sin/cos through yk_libm.h: glibc's sinf / cosf)
Xf xf_rotation(float theta, V3 axis) {
    V3 a = normalize(axis);
    float c = det_cosf(theta), s = det_sinf(theta);
    Xf t = xf_identity();
    float m[16] = {a.x * a.x + (1.0f - a.x * a.x) * c, a.x * a.y * (1.0f - c) - a.z * s, a.x * a.z * (1.0f - c) + a.y * s, 0.0f,
                   a.x * a.y * (1.0f - c) + a.z * s, a.y * a.y + (1.0f - a.y * a.y) * c, a.y * a.z * (1.0f - c) - a.x * s, 0.0f,
                   a.x * a.z * (1.0f - c) - a.y * s, a.y * a.z * (1.0f - c) + a.x * s, a.z * a.z + (1.0f - a.z * a.z) * c, 0.0f,
                   0.0f, 0.0f, 0.0f, 1.0f};
    std::memcpy(t.m, m, 64);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) t.mi[4 * i + j] = m[4 * j + i];
    return t;
}

// get_material, pbrt/mod.rs:860-936
yk_status get_material(const std::string& type, const ParamSet& p, const std::map<std::string, int>& textures, yk_material_desc& out) {
    const float ones[3] = {1, 1, 1}, half[3] = {0.5f, 0.5f, 0.5f};
    float a[3], b[3];
    if (type == "glass") {
        p.vec3(p.spectra, "Kr", ones, a);
        p.vec3(p.spectra, "Kt", ones, b);
        out = make_mat(YK_MAT_GLASS, a, b, p.f32("eta", 1.5f), false);
    } else if (type == "glossy") {
        p.vec3(p.spectra, "Rs", half, a);
        out = make_mat(YK_MAT_GLOSSY, a, nullptr, p.f32("roughness", 0.5f), false);
    } else if (type == "matte") {
        std::string kd_tex = p.str("Kd", "");
        int tex = -1;
        if (!kd_tex.empty()) {
            auto it = textures.find(kd_tex);
            if (it == textures.end()) return lfail(YK_ERR_INVALID_ARGUMENT, "Texture '" + kd_tex + "' not found");
            tex = it->second;
        } else {
            p.vec3(p.spectra, "Kd", half, a);
        }
        const float rpd = YK_PI / 180.0f;  // f32::to_radians, applied twice (quirk 12)
        float sigma = p.f32("sigma", 0.0f) * rpd;
        out = make_mat(YK_MAT_MATTE, tex < 0 ? a : nullptr, nullptr, sigma * rpd, false);
        if (tex >= 0) {
            out.flags |= YK_MAT_FLAG_TEXTURED_A;
            out.a_texture = (uint32_t)tex;
        }
    } else if (type == "metal") {
        std::vector<float> l(COPPER_WAVELENGTHS, COPPER_WAVELENGTHS + 56), n(COPPER_N, COPPER_N + 56), k(COPPER_K, COPPER_K + 56);
        float eta_d[3], k_d[3];
        sampled_spectrum_into_rgb(l, n, eta_d);
        sampled_spectrum_into_rgb(l, k, k_d);
        p.vec3(p.spectra, "eta", eta_d, a);
        p.vec3(p.spectra, "k", k_d, b);
        out = make_mat(YK_MAT_METAL, a, b, p.f32("roughness", 0.01f), p.boolean("remaproughness", true));
    } else {
        const float v = 1.0f * 0.5f;
        const float g[3] = {v, v, v};
        out = make_mat(YK_MAT_MATTE, g, nullptr, 0.0f, false);
    }
    return YK_OK;
}

}  // namespace

// `n as i32` from f64: saturating, NaN -> 0
static int sat_i32(double v) {
    if (v != v) return 0;
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (-2147483647 - 1);
    return (int)v;
}

// scene/pbrt/mod.rs: ParseShape — shapes in file order; `plymesh` files are read after the parse, in parallel (:786-800)
struct ParseShape {
    int kind = 0;  // 0 sphere, 1 trianglemesh, 2 plymesh
    yk_sphere_desc sphere;
    Xf transform;
    int material = 0;
    std::vector<uint32_t> idx;
    std::vector<float> P, N, UV;
    std::string ply_path;
    PlyMesh ply;
    yk_status status = YK_OK;
    std::string error;
};

struct PbrtState {
    std::vector<ParseShape> shapes;
    Xf current = xf_identity();
    std::vector<Xf> xf_stack;
    std::vector<int> gs_stack;      // GraphicsState = the current material
    std::vector<bool> atb_stack;    // active-transform bits: START set?
    std::map<std::string, int> named;
    std::map<std::string, int> textures;   // imagemap name -> index into yk_loaded_scene::textures
    int cur_material = 0;
    bool start_active = true;
    Tok fetched;                    // the reference's `fetched_token` lives across file scopes
    bool have_fetched = false;
};

// A file that ends in the middle of a directive silently drops that directive: the
// reference's get_next_token! does `break 'top_parse` on EndOfInput (pbrt/mod.rs:140-160).
struct EndOfFile {};
struct Failure {
    yk_status st;
};

static yk_status load_pbrt_file(const std::string& path, yk_loaded_scene& s, PbrtState& S, int depth) {
    if (depth > 32) return lfail(YK_ERR_INVALID_ARGUMENT, "Include nesting too deep");
    std::vector<unsigned char> bytes;
    if (!read_file(path, bytes)) return lfail(YK_ERR_INVALID_ARGUMENT, "Could not open '" + path + "'");
    Lexer lx;
    lx.in.assign(bytes.begin(), bytes.end());
    std::string parent = ".";
    {
        size_t slash = path.find_last_of('/');
        if (slash != std::string::npos) parent = path.substr(0, slash);
    }
    auto fail = [&](yk_status st, const std::string& msg) -> Failure { return Failure{lfail(st, msg + " (" + path + ")")}; };
    auto next = [&]() -> Tok {
        if (S.have_fetched) {
            S.have_fetched = false;
            return S.fetched;
        }
        Tok t = lx.next();
        if (t.kind == Tok::End) throw EndOfFile();
        if (t.kind == Tok::Error) throw fail(YK_ERR_INVALID_ARGUMENT, "pbrt lexer: " + lx.error);
        return t;
    };
    auto unfetch = [&](const Tok& t) {
        S.fetched = t;
        S.have_fetched = true;
    };
    auto unexpected = [&](const Tok& t) -> Failure { return fail(YK_ERR_INVALID_ARGUMENT, "pbrt: UnexpectedToken '" + t.text + "'"); };
    auto num = [&]() -> float {
        Tok t = next();
        if (t.kind != Tok::Number) throw unexpected(t);
        return (float)t.num;  // Token::Number(f64) `as f32`
    };
    auto str = [&]() -> std::string {
        Tok t = next();
        if (t.kind != Tok::String) throw unexpected(t);
        return t.text;
    };
    // get_num_params! / get_{two,three}_component_vector_params!
    auto numbers = [&](bool allow_single, size_t width) -> std::vector<double> {
        std::vector<double> out;
        Tok a = next();
        if (a.kind == Tok::Number && allow_single) {
            out.push_back(a.num);
            return out;
        }
        if (a.kind != Tok::LBracket) throw unexpected(a);
        for (;;) {
            Tok b = next();
            if (b.kind == Tok::RBracket) return out;
            if (b.kind != Tok::Number) throw unexpected(b);
            out.push_back(b.num);
            for (size_t k = 1; k < width; ++k) {  // a vector must be complete before ']'
                Tok c = next();
                if (c.kind != Tok::Number) throw unexpected(c);
                out.push_back(c.num);
            }
        }
    };
    auto strings = [&]() -> std::vector<std::string> {
        std::vector<std::string> out;
        Tok a = next();
        if (a.kind == Tok::String) {
            out.push_back(a.text);
            return out;
        }
        if (a.kind != Tok::LBracket) throw unexpected(a);
        for (;;) {
            Tok b = next();
            if (b.kind == Tok::RBracket) return out;
            if (b.kind != Tok::String) throw unexpected(b);
            out.push_back(b.text);
        }
    };
    auto to_f32 = [](const std::vector<double>& v) {
        std::vector<float> o;
        for (double x : v) o.push_back((float)x);
        return o;
    };
    // get_param_set!, pbrt/mod.rs:381-470
    auto param_set = [&]() -> ParamSet {
        ParamSet ps;
        for (;;) {
            Tok t = next();
            if (t.kind != Tok::String) {
                unfetch(t);
                return ps;
            }
            std::istringstream ds(t.text);
            std::string type_name, param_name, extra;
            if (!(ds >> type_name >> param_name) || (ds >> extra)) throw unexpected(t);
            if (type_name == "bool") {
                std::vector<bool> bv;
                for (auto& x : strings()) {
                    if (x != "true" && x != "false") throw fail(YK_ERR_INVALID_ARGUMENT, "pbrt: UnexpectedToken '" + x + "'");
                    bv.push_back(x == "true");
                }
                ps.bools.add(param_name, bv);
            } else if (type_name == "float") {
                if (param_name == "uv") ps.uvs.add(param_name, to_f32(numbers(false, 2)));
                else ps.floats.add(param_name, to_f32(numbers(true, 1)));
            } else if (type_name == "integer") {
                std::vector<int> iv;
                for (double v : numbers(true, 1)) iv.push_back(sat_i32(v));
                ps.ints.add(param_name, iv);
            } else if (type_name == "string" || type_name == "texture") {
                ps.strings.add(param_name, strings());
            } else if (type_name == "color" || type_name == "rgb") {
                ps.spectra.add(param_name, to_f32(numbers(false, 3)));
            } else if (type_name == "point") {
                ps.points.add(param_name, to_f32(numbers(false, 3)));
            } else if (type_name == "normal") {
                ps.normals.add(param_name, to_f32(numbers(false, 3)));
            } else if (type_name == "spectrum") {
                std::vector<float> values;
                Tok a = next();
                if (a.kind == Tok::String) {
                    std::vector<unsigned char> spd;
                    if (!read_file(parent + "/" + a.text, spd)) throw fail(YK_ERR_INVALID_ARGUMENT, "Could not open spd '" + a.text + "'");
                    std::istringstream sf(std::string(spd.begin(), spd.end()));
                    std::string l;
                    while (std::getline(sf, l)) {
                        std::istringstream ls(l.substr(0, l.find('#')));
                        std::string w;
                        while (ls >> w) {
                            char* end = nullptr;
                            values.push_back(std::strtof(w.c_str(), &end));
                            if (!end || *end) throw fail(YK_ERR_INVALID_ARGUMENT, "bad number in spd '" + a.text + "'");
                        }
                    }
                } else {
                    unfetch(a);
                    values = to_f32(numbers(true, 1));
                }
                if (values.empty() || values.size() % 2) throw fail(YK_ERR_INVALID_ARGUMENT, "pbrt: spectrum needs (lambda, value) pairs");
                std::vector<float> lam, smp;
                for (size_t k = 0; k + 1 < values.size(); k += 2) {
                    lam.push_back(values[k]);
                    smp.push_back(values[k + 1]);
                }
                float rgb[3];
                sampled_spectrum_into_rgb(lam, smp, rgb);
                ps.spectra.add(param_name, std::vector<float>(rgb, rgb + 3));
            } else if (type_name == "blackbody") {
                numbers(true, 1);  // "not supported, falling back to default"
            } else {
                throw fail(YK_ERR_INVALID_ARGUMENT, "pbrt: UnknownParamType '" + type_name + " " + param_name + "'");
            }
        }
    };

    try {
        for (;;) {
            Tok t = next();
            if (t.kind != Tok::Ident) throw fail(YK_ERR_UNSUPPORTED, "pbrt: UnimplementedToken '" + t.text + "'");
            const std::string d = t.text;
            if (d == "ActiveTransform") {
                Tok a = next();
                if (a.kind == Tok::Ident && (a.text == "All" || a.text == "StartTime")) S.start_active = true;
                else if (a.kind == Tok::Ident && a.text == "EndTime") S.start_active = false;
                else throw unexpected(a);
            } else if (d == "AreaLightSource" || d == "Integrator" || d == "Sampler") {  // ignore_type_definition!
                str();
                param_set();
            } else if (d == "AttributeBegin") {
                S.gs_stack.push_back(S.cur_material);
                S.xf_stack.push_back(S.current);
                S.atb_stack.push_back(S.start_active);
            } else if (d == "AttributeEnd") {
                if (!S.gs_stack.empty()) {
                    if (S.xf_stack.empty() || S.atb_stack.empty()) throw fail(YK_ERR_INVALID_ARGUMENT, "pbrt: unbalanced attribute stack");
                    S.cur_material = S.gs_stack.back();
                    S.gs_stack.pop_back();
                    S.current = S.xf_stack.back();
                    S.xf_stack.pop_back();
                    S.start_active = S.atb_stack.back();
                    S.atb_stack.pop_back();
                }
            } else if (d == "Camera") {
                if (str() != "perspective") throw fail(YK_ERR_UNSUPPORTED, "Only perspective camera is supported");
                ParamSet ps = param_set();
                s.camera.fov_degrees = ps.f32("fov", 45.0f);
            } else if (d == "Film") {
                str();
                ParamSet ps = param_set();
                s.camera.res_x = (uint16_t)(uint32_t)ps.i32("xresolution", 640);
                s.camera.res_y = (uint16_t)(uint32_t)ps.i32("yresolution", 480);
            } else if (d == "Include") {
                std::string inc = str();
                yk_status st = load_pbrt_file(parent + "/" + inc, s, S, depth + 1);
                if (st != YK_OK) return st;
            } else if (d == "LightSource") {
                std::string type_name = str();
                ParamSet ps = param_set();
                const float ones[3] = {1, 1, 1}, zero[3] = {0, 0, 0}, zaxis[3] = {0, 0, 1};
                if (type_name == "infinite") {
                    ps.vec3(ps.spectra, "L", ones, s.background);
                } else if (type_name == "distant") {
                    float L[3], from[3], to[3];
                    ps.vec3(ps.spectra, "L", ones, L);
                    if (!(L[0] == 0.0f && L[1] == 0.0f && L[2] == 0.0f)) {
                        ps.vec3(ps.points, "from", zero, from);
                        ps.vec3(ps.points, "to", zaxis, to);
                        V3 w = normalize(V3{from[0] - to[0], from[1] - to[1], from[2] - to[2]});
                        yk_light_desc l;
                        std::memset(&l, 0, sizeof(l));
                        l.kind = YK_LIGHT_DISTANT;
                        l.p[0] = w.x; l.p[1] = w.y; l.p[2] = w.z;
                        for (int k = 0; k < 3; ++k) l.i[k] = L[k];
                        s.lights.push_back(l);
                    }
                } else if (type_name == "point") {
                    float I[3], from[3];
                    ps.vec3(ps.spectra, "I", ones, I);
                    if (!(I[0] == 0.0f && I[1] == 0.0f && I[2] == 0.0f)) {
                        ps.vec3(ps.points, "from", zero, from);
                        Xf tr = xf_translation(from[0], from[1], from[2]);
                        yk_light_desc l;
                        yk_make_point_light(tr.m, I, &l);
                        s.lights.push_back(l);
                    }
                }  // other light types: "not implemented", skipped
            } else if (d == "LookAt") {
                if (S.start_active) {  // otherwise the nine numbers are left for the parser to trip over
                    for (int k = 0; k < 3; ++k) s.camera.position[k] = num();
                    for (int k = 0; k < 3; ++k) s.camera.target[k] = num();
                    float u0 = num(), u1 = num(), u2 = num();
                    V3 up = normalize(V3{u0, u1, u2});
                    s.camera.up[0] = up.x; s.camera.up[1] = up.y; s.camera.up[2] = up.z;
                }
            } else if (d == "NamedMaterial") {
                std::string name = str();
                auto it = S.named.find(name);
                S.cur_material = it != S.named.end() ? it->second : 0;  // default_material
            } else if (d == "Material" || d == "MakeNamedMaterial") {
                std::string name;
                if (d == "MakeNamedMaterial") {
                    name = str();
                    if (str() != "string type") throw fail(YK_ERR_INVALID_ARGUMENT, "pbrt: UnknownParamType in MakeNamedMaterial");
                }
                std::string type = str();
                ParamSet ps = param_set();
                yk_material_desc m;
                yk_status st = get_material(type, ps, S.textures, m);
                if (st != YK_OK) return st;
                s.materials.push_back(m);
                if (d == "Material") S.cur_material = (int)s.materials.size() - 1;
                else S.named[name] = (int)s.materials.size() - 1;  // HashMap::insert replaces
            } else if (d == "Rotate") {
                float angle = num(), ax = num(), ay = num(), az = num();
                S.current = xf_mul(S.current, xf_rotation(angle * (YK_PI / 180.0f), V3{ax, ay, az}));
            } else if (d == "Scale") {
                float x = num(), y = num(), z = num();
                S.current = xf_mul(S.current, xf_scale(x, y, z));
            } else if (d == "Translate") {
                float x = num(), y = num(), z = num();
                S.current = xf_mul(S.current, xf_translation(x, y, z));
            } else if (d == "Shape") {
                std::string shape_type = str();
                ParamSet ps = param_set();
                if (shape_type == "sphere") {
                    yk_sphere_desc sp;
                    std::memcpy(sp.object_to_world, S.current.m, 64);
                    std::memcpy(sp.world_to_object, S.current.mi, 64);
                    sp.radius = ps.f32("radius", 1.0f);
                    sp.material = S.cur_material;
                    ParseShape ph;
                    ph.kind = 0;
                    ph.sphere = sp;
                    S.shapes.push_back(std::move(ph));
                } else if (shape_type == "trianglemesh") {
                    std::vector<uint32_t> idx;
                    if (auto ii = ps.ints.many("indices"))
                        for (int v : *ii) idx.push_back((uint32_t)v);
                    if (idx.size() < 3 || idx.size() % 3) continue;  // "Invalid 'trianglemesh'": skipped
                    auto fetch = [](const Items<float>& m, const char* n) {
                        auto i = m.many(n);
                        return i ? *i : std::vector<float>();
                    };
                    std::vector<float> P = fetch(ps.points, "P"), N = fetch(ps.normals, "N"), UV = fetch(ps.uvs, "uv");
                    const size_t nv = P.size() / 3;
                    for (uint32_t q : idx)
                        if (q >= nv) throw fail(YK_ERR_INVALID_ARGUMENT, "pbrt: trianglemesh index out of range");
                    if (!N.empty() && N.size() != P.size()) throw fail(YK_ERR_INVALID_ARGUMENT, "pbrt: N count differs from P");
                    if (!UV.empty() && UV.size() / 2 != nv) throw fail(YK_ERR_INVALID_ARGUMENT, "pbrt: uv count differs from P");
                    ParseShape ph;
                    ph.kind = 1;
                    ph.transform = S.current;
                    ph.material = S.cur_material;
                    ph.idx = std::move(idx);
                    ph.P = std::move(P);
                    ph.N = std::move(N);
                    ph.UV = std::move(UV);
                    S.shapes.push_back(std::move(ph));
                } else if (shape_type == "plymesh") {
                    std::string fn = ps.str("filename", "");
                    if (fn.empty()) throw fail(YK_ERR_INVALID_ARGUMENT, "Empty PLY filename");
                    // the path is resolved while parsing (canonicalize(): a missing file ends the load here, pbrt/mod.rs:689-699);
                    // the file is READ after the parse, with the others
                    ParseShape ph;
                    ph.kind = 2;
                    ph.transform = S.current;
                    ph.material = S.cur_material;
                    ph.ply_path = parent + "/" + fn;
                    if (!std::ifstream(ph.ply_path, std::ios::binary).good()) return lfail(YK_ERR_INVALID_ARGUMENT, "Could not open '" + ph.ply_path + "'");
                    S.shapes.push_back(std::move(ph));
                }  // other shapes: "Unsupported shape type", skipped
            } else if (d == "Texture") {
                std::string name = str(), ttype = str(), cls = str();
                ParamSet ps = param_set();
                if (ttype == "spectrum" && cls == "imagemap") {  // pbrt/mod.rs:719-735
                    std::string fn = ps.str("filename", "");
                    if (fn.empty()) throw fail(YK_ERR_INVALID_ARGUMENT, "missing file for texture '" + name + "'");
                    uint32_t w = 0, h = 0;
                    std::vector<float> rgb;
                    std::string err;
                    yk_status st = yk_image_decode_file(parent + "/" + fn, w, h, rgb, err);
                    if (st != YK_OK) return lfail(st, err);
                    s.texture_data.push_back(std::move(rgb));
                    yk_texture_desc td;
                    td.width = w;
                    td.height = h;
                    td.rgb = nullptr;  // fixed up in yk_loaded_scene_get (the vectors may still move)
                    s.textures.push_back(td);
                    S.textures[name] = (int)s.textures.size() - 1;  // HashMap::insert replaces
                }
            } else if (d == "TransformBegin") {
                S.xf_stack.push_back(S.current);
            } else if (d == "TransformEnd") {
                if (!S.gs_stack.empty()) {  // sic: pops the graphics-state stack (pbrt/mod.rs:749-755)
                    S.cur_material = S.gs_stack.back();
                    S.gs_stack.pop_back();
                }
            } else if (d == "WorldBegin") {
                S.current = xf_identity();
            } else if (d == "WorldEnd") {
            } else {
                throw fail(YK_ERR_UNSUPPORTED, "pbrt: UnimplementedToken '" + d + "'");
            }
        }
    } catch (const EndOfFile&) {
        return YK_OK;
    } catch (const Failure& e) {
        return e.st;
    }
}

static void default_camera(yk_loaded_scene& s) {
    std::memset(&s.camera, 0, sizeof(s.camera));
    s.camera.up[1] = 1.0f;
    s.camera.fov_axis = 0;
    s.camera.res_x = 640;
    s.camera.res_y = 480;
}

extern "C" {

const char* yk_loader_last_error(void) { return g_loader_error.c_str(); }
void yk_loader_set_error(const char* msg) { g_loader_error = msg ? msg : ""; }

yk_status yk_load_ply(const char* path, uint32_t split_method, uint32_t max_shapes_in_node, yk_loaded_scene** out) {
    if (!path || !out) return lfail(YK_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    yk_loaded_scene* s = new yk_loaded_scene();
    default_camera(*s);
    s->split_method = split_method;
    s->max_shapes_in_node = max_shapes_in_node;
    const float white[3] = {1, 1, 1};
    s->materials.push_back(make_mat(YK_MAT_MATTE, white, nullptr, 0.0f, false));  // scene/mod.rs:104-107
    yk_status st;
    try {
        st = load_ply_mesh(path, nullptr, *s, 0);
    } catch (const std::exception& e) {  // e.g. bad_alloc on an absurd element count
        st = lfail(YK_ERR_INVALID_ARGUMENT, std::string("PLY: ") + e.what());
    }
    if (st != YK_OK) {
        delete s;
        return st;
    }
    Xf tr = xf_translation(5.0f, 5.0f, 0.0f);  // scene/mod.rs:118-121
    const float I[3] = {1.0f * 600.0f, 1.0f * 600.0f, 1.0f * 600.0f};
    yk_light_desc l;
    yk_make_point_light(tr.m, I, &l);
    s->lights.push_back(l);
    const float pos[3] = {2, 2, 2};
    for (int k = 0; k < 3; ++k) s->camera.position[k] = pos[k];
    s->camera.fov_axis = 0;
    s->camera.fov_degrees = 40.0f;
    *out = s;
    return YK_OK;
}

yk_status yk_load_pbrt(const char* path, uint32_t split_method, uint32_t max_shapes_in_node, yk_loaded_scene** out) {
    if (!path || !out) return lfail(YK_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    yk_loaded_scene* s = new yk_loaded_scene();
    default_camera(*s);
    s->camera.fov_degrees = 0.0f;
    s->split_method = split_method;
    s->max_shapes_in_node = max_shapes_in_node;
    yk_material_desc def;
    PbrtState S;
    get_material("matte", ParamSet(), S.textures, def);
    s->materials.push_back(def);  // default_material
    yk_status st;
    try {
        st = load_pbrt_file(path, *s, S, 0);
    } catch (const std::exception& e) {
        st = lfail(YK_ERR_INVALID_ARGUMENT, std::string("pbrt: ") + e.what());
    }
    if (st == YK_OK) {
        // "load plys" (pbrt/mod.rs:786-800: parse_shapes.par_iter_mut().try_for_each(ply::load)): the files are read by a few
        // threads, each into its own ParseShape; the meshes then join the scene in file order ("collect meshes", :807-822)
        std::vector<size_t> ply;
        for (size_t i = 0; i < S.shapes.size(); ++i)
            if (S.shapes[i].kind == 2) ply.push_back(i);
        unsigned n_threads = (unsigned)std::min<size_t>(ply.size(), std::min(16u, std::max(1u, std::thread::hardware_concurrency())));
        if (const char* e = std::getenv("YK_LOADER_THREADS")) n_threads = (unsigned)std::min<size_t>(ply.size(), (size_t)std::max(1, std::atoi(e)));  // measurements
        std::atomic<size_t> next_ply{0};
        auto work = [&]() {
            for (;;) {
                const size_t k = next_ply.fetch_add(1);
                if (k >= ply.size()) return;
                ParseShape& ph = S.shapes[ply[k]];
                try {
                    ph.status = read_ply_mesh(ph.ply_path, ph.ply);
                    if (ph.status != YK_OK) ph.error = g_loader_error;  // this thread's own copy
                } catch (const std::exception& e) {  // e.g. bad_alloc on an absurd element count
                    ph.status = YK_ERR_INVALID_ARGUMENT;
                    ph.error = std::string("PLY: ") + e.what();
                }
            }
        };
        if (n_threads <= 1) {
            work();
        } else {
            std::vector<std::thread> pool;
            try {
                for (unsigned t = 0; t < n_threads; ++t) pool.emplace_back(work);
            } catch (const std::exception&) {  // no more threads: the ones that started (and this one) finish the list
            }
            work();
            for (std::thread& t : pool) t.join();
        }
        for (ParseShape& ph : S.shapes) {
            if (ph.kind == 2 && ph.status != YK_OK) {  // the first failing file in file order
                st = lfail(ph.status, ph.error);
                break;
            }
        }
    }
    if (st == YK_OK) {
        try {
            size_t nv = 0, nt = 0;  // one allocation per array instead of a growth series (10 M triangles: cfg5)
            for (const ParseShape& ph : S.shapes) {
                nv += (ph.kind == 1 ? ph.P.size() : ph.ply.pts.size()) / 3;
                nt += (ph.kind == 1 ? ph.idx.size() : ph.ply.indices.size()) / 3;
            }
            s->points.reserve(3 * nv);
            s->normals.reserve(3 * nv);
            s->uvs.reserve(2 * nv);
            s->indices.reserve(3 * nt);
            s->tri_mesh.reserve(nt);
            s->tri_material.reserve(nt);
            s->tri_area_light.reserve(nt);
            s->shape_order.reserve(nt + S.shapes.size());
            for (ParseShape& ph : S.shapes) {
                if (ph.kind == 0) {
                    s->shape_order.push_back(0x80000000u | (uint32_t)s->spheres.size());
                    s->spheres.push_back(ph.sphere);
                } else if (ph.kind == 1) {
                    add_mesh(*s, ph.transform, ph.idx, ph.P, ph.N, ph.UV, ph.material);
                } else {
                    add_ply_mesh(*s, ph.ply, &ph.transform, ph.material);
                    PlyMesh().pts.swap(ph.ply.pts);  // the payload has been copied into the scene
                }
            }
        } catch (const std::exception& e) {
            st = lfail(YK_ERR_INVALID_ARGUMENT, std::string("pbrt: ") + e.what());
        }
    }
    if (st != YK_OK) {
        delete s;
        return st;
    }
    // pbrt/mod.rs:827-835
    s->camera.fov_axis = s->camera.res_y < s->camera.res_x ? 1u : 0u;
    if (s->indices.empty() && s->spheres.empty()) {
        delete s;
        return lfail(YK_ERR_INVALID_ARGUMENT, "pbrt: scene has no shapes");
    }
    // shapes keep their file order (spheres and meshes interleave, pbrt/mod.rs:807-822)
    const uint32_t nt = (uint32_t)(s->indices.size() / 3);
    for (uint32_t e : s->shape_order) s->shape_order_flat.push_back((e & 0x80000000u) ? nt + (e & 0x7fffffffu) : e);
    *out = s;
    return YK_OK;
}

yk_status yk_loaded_scene_get(const yk_loaded_scene* s, yk_scene_desc* d, yk_camera_params* camera, uint16_t* tile_dim) {
    if (!s || !d) return YK_ERR_INVALID_ARGUMENT;
    std::memset(d, 0, sizeof(*d));
    d->n_vertices = (uint32_t)(s->points.size() / 3);
    d->points = s->points.data();
    d->normals = s->any_normals ? s->normals.data() : nullptr;
    d->uvs = s->any_uvs ? s->uvs.data() : nullptr;
    d->n_triangles = (uint32_t)(s->indices.size() / 3);
    d->indices = s->indices.data();
    d->tri_mesh = s->tri_mesh.data();
    d->tri_material = s->tri_material.data();
    d->tri_area_light = s->tri_area_light.data();
    d->n_meshes = (uint32_t)s->meshes.size();
    d->meshes = s->meshes.data();
    d->n_spheres = (uint32_t)s->spheres.size();
    d->spheres = s->spheres.data();
    d->n_materials = (uint32_t)s->materials.size();
    d->materials = s->materials.data();
    d->n_lights = (uint32_t)s->lights.size();
    d->lights = s->lights.data();
    for (int k = 0; k < 3; ++k) d->background[k] = s->background[k];
    d->shape_order = s->shape_order_flat.empty() ? nullptr : s->shape_order_flat.data();
    yk_loaded_scene* ms = const_cast<yk_loaded_scene*>(s);
    for (size_t t = 0; t < ms->textures.size(); ++t) ms->textures[t].rgb = ms->texture_data[t].data();
    d->n_textures = (uint32_t)s->textures.size();
    d->textures = s->textures.empty() ? nullptr : s->textures.data();
    d->split_method = s->split_method;
    d->max_shapes_in_node = s->max_shapes_in_node;
    if (camera) *camera = s->camera;
    if (tile_dim) *tile_dim = s->tile_dim;
    return YK_OK;
}

void yk_loaded_scene_destroy(yk_loaded_scene* s) { delete s; }

}  // extern "C"
