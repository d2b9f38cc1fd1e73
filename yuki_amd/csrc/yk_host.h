// yk_host.h — host-side pieces of the hot path: Camera::new, film tiles, light
// construction and the BVH builder.  Pure C++ (no HIP calls) so that they work —
// and are tested — on a machine without a GPU.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/yuki_hip.h"

namespace yk {

// row-major 4x4 transform pair (math/transform.rs:12-19)
struct Xf {
    float m[16];
    float mi[16];
};
Xf xf_identity();
Xf xf_from(const float* m, const float* mi);
Xf xf_from_matrix(const float* m, bool* ok);  // Transform::new: inverse by Gauss-Jordan
Xf xf_mul(const Xf& a, const Xf& b);          // transform.rs:211-223
Xf xf_inverse(const Xf& a);
Xf xf_translation(float x, float y, float z);  // transforms.rs:4-23
Xf xf_scale(float x, float y, float z);        // transforms.rs:26-45
Xf xf_look_at(const float pos[3], const float target[3], const float up[3], bool* ok);  // transforms.rs:138-153
bool mat4_inverse(const float* m, float* out);  // matrix.rs:107-215

yk_status camera_init(const yk_camera_params* p, yk_camera* out);
std::vector<yk_tile> film_tiles(uint16_t res_x, uint16_t res_y, uint16_t tile_dim);

// Whole regular file into memory; false for directories, unreadable files and files > 2 GiB.
bool read_file(const std::string& path, std::vector<unsigned char>& out);

// ---- BVH ---------------------------------------------------------------------
struct ShapeBounds {
    float bmin[3], bmax[3];
};

struct HostBvh {
    std::vector<yk_bvh_node> nodes;     // reference layout, depth-first (bvh.rs:396-420)
    std::vector<uint32_t> shape_order;  // leaf order -> source shape index (bvh.rs:96)
    uint32_t max_leaf_shapes = 0;
    uint32_t depth = 0;
    bool split_failed = false;  // reference: assert_ne!(mid, start)
};
// BoundingVolumeHierarchy::new, bvh.rs:39-115,305-523
void build_bvh(const std::vector<ShapeBounds>& bounds, uint32_t max_shapes_in_node, uint32_t split_method, HostBvh& out);

}  // namespace yk
