#include <cstdio>
#include <cstring>
#include <string>
#include "../../include/yuki_hip.h"
extern "C" yk_status yk_make_point_light(const float*, const float*, yk_light_desc* out) { std::memset(out, 0, sizeof(*out)); return YK_OK; }
int main(int argc, char** argv) {
    int ok = 0, bad = 0;
    for (int i = 1; i < argc; ++i) {
        std::string p = argv[i];
        yk_status st;
        if (p.size() > 4 && p.substr(p.size() - 4) == ".ply") { yk_loaded_scene* s = nullptr; st = yk_load_ply(p.c_str(), 0, 1, &s); if (s) { yk_scene_desc d; yk_loaded_scene_get(s, &d, nullptr, nullptr); yk_loaded_scene_destroy(s);} }
        else if (p.size() > 5 && p.substr(p.size() - 5) != ".pbrt") { /* every image container */ yk_texture_desc t; st = yk_image_texture_load(p.c_str(), &t); if (st == YK_OK) yk_image_texture_free(&t); }
        else { yk_loaded_scene* s = nullptr; st = yk_load_pbrt(p.c_str(), 0, 1, &s); if (s) { yk_scene_desc d; yk_loaded_scene_get(s, &d, nullptr, nullptr); yk_loaded_scene_destroy(s);} }
        (st == YK_OK ? ok : bad)++;
    }
    std::printf("ok %d rejected %d\n", ok, bad);
    return 0;
}
