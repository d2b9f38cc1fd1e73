// yk_host.cpp — host-side restatements that feed the device path.
//
//   Matrix4x4::inverted / Transform / transforms::*   yuki/src/math/{matrix,transform,transforms}.rs
//   Camera::new                                       yuki/src/camera.rs:52-102
//   generate_tiles / outward_spiral                   yuki/src/film.rs:299-376
//   RectangularLight::new / SpotLight::new / PointLight::new   yuki/src/lights/*.rs
//   BoundingVolumeHierarchy::new                      yuki/src/bvh.rs:39-115,305-523
//
// f32 arithmetic in the reference's operation order; built with -ffp-contract=off.
#include "yk_host.h"

#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include <cmath>
#include <cstring>
#include <utility>

#include "yk_bsdf.h"
#include "yk_libm.h"
#include "yk_math.h"

namespace yk {

bool read_file(const std::string& path, std::vector<unsigned char>& out) {
    out.clear();
    struct stat sb;
    if (stat(path.c_str(), &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 0 || (unsigned long long)sb.st_size > (2ull << 30)) return false;
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    out.resize((size_t)sb.st_size);
    const size_t got = out.empty() ? 0 : std::fread(out.data(), 1, out.size(), f);
    std::fclose(f);
    out.resize(got);
    return true;
}


// ------------------------------------------------------------------ matrices
bool mat4_inverse(const float* src, float* out) {
    float a[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) a[i][j] = src[4 * i + j];
    int col_of[4] = {0, 0, 0, 0}, row_of[4] = {0, 0, 0, 0}, used[4] = {0, 0, 0, 0};
    for (int step = 0; step < 4; ++step) {
        int pr = 0, pc = 0;
        float best = 0.0f;
        for (int r = 0; r < 4; ++r) {
            if (used[r] == 1) continue;
            for (int c = 0; c < 4; ++c) {
                if (used[c] == 0 && fabsf(a[r][c]) > best) {
                    best = fabsf(a[r][c]);
                    pr = r;
                    pc = c;
                }
            }
        }
        used[pc] += 1;
        if (pr != pc)
            for (int k = 0; k < 4; ++k) std::swap(a[pr][k], a[pc][k]);
        row_of[step] = pr;
        col_of[step] = pc;
        if (a[pc][pc] == 0.0f) return false;  // reference: assert "singular matrix"
        float pivinv = 1.0f / a[pc][pc];
        a[pc][pc] = 1.0f;
        for (int k = 0; k < 4; ++k) a[pc][k] *= pivinv;
        for (int r = 0; r < 4; ++r) {
            if (r == pc) continue;
            float factor = a[r][pc];
            a[r][pc] = 0.0f;
            for (int k = 0; k < 4; ++k) a[r][k] -= factor * a[pc][k];
        }
    }
    for (int step = 3; step >= 0; --step) {
        if (row_of[step] != col_of[step])
            for (int r = 0; r < 4; ++r) std::swap(a[r][row_of[step]], a[r][col_of[step]]);
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) out[4 * i + j] = a[i][j];
    return true;
}

static void mat4_mul(const float* a, const float* b, float* out) {
    float r[16];
    for (int row = 0; row < 4; ++row)
        for (int col = 0; col < 4; ++col)
            r[4 * row + col] = a[4 * row + 0] * b[0 + col] + a[4 * row + 1] * b[4 + col] + a[4 * row + 2] * b[8 + col] +
                               a[4 * row + 3] * b[12 + col];
    std::memcpy(out, r, sizeof(r));
}

Xf xf_identity() {
    Xf t;
    for (int i = 0; i < 16; ++i) t.m[i] = t.mi[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    return t;
}
Xf xf_from(const float* m, const float* mi) {
    Xf t;
    std::memcpy(t.m, m, 64);
    std::memcpy(t.mi, mi, 64);
    return t;
}
Xf xf_from_matrix(const float* m, bool* ok) {
    Xf t;
    std::memcpy(t.m, m, 64);
    bool good = mat4_inverse(m, t.mi);
    if (ok) *ok = good;
    return t;
}
Xf xf_mul(const Xf& a, const Xf& b) {
    Xf t;
    mat4_mul(a.m, b.m, t.m);
    mat4_mul(b.mi, a.mi, t.mi);
    return t;
}
Xf xf_inverse(const Xf& a) { return xf_from(a.mi, a.m); }
Xf xf_translation(float x, float y, float z) {
    Xf t = xf_identity();
    t.m[3] = x;
    t.m[7] = y;
    t.m[11] = z;
    t.mi[3] = -x;
    t.mi[7] = -y;
    t.mi[11] = -z;
    return t;
}
Xf xf_scale(float x, float y, float z) {
    Xf t = xf_identity();
    t.m[0] = x;
    t.m[5] = y;
    t.m[10] = z;
    t.mi[0] = 1.0f / x;
    t.mi[5] = 1.0f / y;
    t.mi[10] = 1.0f / z;
    return t;
}
Xf xf_look_at(const float pos[3], const float target[3], const float up[3], bool* ok) {
    V3 p = V3{pos[0], pos[1], pos[2]}, tg = V3{target[0], target[1], target[2]}, u = V3{up[0], up[1], up[2]};
    V3 dir = normalize(tg - p);
    V3 right = normalize(cross(normalize(u), dir));
    V3 new_up = cross(dir, right);
    float c2w[16] = {right.x, new_up.x, dir.x, p.x, right.y, new_up.y, dir.y, p.y, right.z, new_up.z, dir.z, p.z, 0.0f, 0.0f, 0.0f, 1.0f};
    Xf t;
    std::memcpy(t.mi, c2w, 64);
    bool good = mat4_inverse(c2w, t.m);
    if (ok) *ok = good;
    return t;
}

// ------------------------------------------------------------------ camera
yk_status camera_init(const yk_camera_params* p, yk_camera* out) {
    if (!p || !out || p->res_x == 0 || p->res_y == 0) return YK_ERR_INVALID_ARGUMENT;
    bool ok = true;
    Xf world_to_camera = xf_look_at(p->position, p->target, p->up, &ok);
    if (!ok) return YK_ERR_INVALID_ARGUMENT;
    Xf camera_to_world = xf_inverse(world_to_camera);
    const float near_z = 1e-2f, far_z = 1000.0f;
    float rad = p->fov_degrees * (YK_PI / 180.0f);  // f32::to_radians
    float inv_tan = 1.0f / det_tanf(rad / 2.0f);
    float persp[16] = {1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, far_z / (far_z - near_z), -(far_z * near_z) / (far_z - near_z),
                       0.0f, 0.0f, 1.0f, 0.0f};
    Xf persp_t = xf_from_matrix(persp, &ok);
    if (!ok) return YK_ERR_INVALID_ARGUMENT;
    Xf camera_to_screen = xf_mul(xf_scale(inv_tan, inv_tan, 1.0f), persp_t);
    float film_x = (float)p->res_x, film_y = (float)p->res_y;
    float smin_x, smin_y, smax_x, smax_y;
    if (p->fov_axis == 0) {
        float ar = film_x / film_y;
        smin_x = -1.0f;
        smin_y = -1.0f / ar;
        smax_x = 1.0f;
        smax_y = 1.0f / ar;
    } else {
        float ar = film_y / film_x;
        smin_x = -1.0f / ar;
        smin_y = -1.0f;
        smax_x = 1.0f / ar;
        smax_y = 1.0f;
    }
    Xf screen_to_raster = xf_mul(xf_scale(film_x, film_y, 1.0f), xf_mul(xf_scale(1.0f / (smax_x - smin_x), 1.0f / (smin_y - smax_y), 1.0f),
                                                                      xf_translation(-smin_x, -smax_y, 0.0f)));
    Xf raster_to_screen = xf_inverse(screen_to_raster);
    Xf raster_to_camera = xf_mul(xf_inverse(camera_to_screen), raster_to_screen);
    std::memcpy(out->camera_to_world, camera_to_world.m, 64);
    std::memcpy(out->camera_to_world_inv, camera_to_world.mi, 64);
    std::memcpy(out->raster_to_camera, raster_to_camera.m, 64);
    std::memcpy(out->raster_to_camera_inv, raster_to_camera.mi, 64);
    return YK_OK;
}

// ------------------------------------------------------------------ film tiles
std::vector<yk_tile> film_tiles(uint16_t res_x, uint16_t res_y, uint16_t tile_dim) {
    std::vector<yk_tile> order;
    if (res_x == 0 || res_y == 0 || tile_dim == 0) return order;
    const int cols = (int)std::ceil((float)res_x / (float)tile_dim);
    const int rows = (int)std::ceil((float)res_y / (float)tile_dim);
    std::vector<uint8_t> taken((size_t)cols * rows, 0);
    const int cx = (cols / 2) - (1 - cols % 2);
    const int cy = (rows / 2) - (1 - rows % 2);
    const int side = cols > rows ? cols : rows;
    int x = 0, y = 0, dx = 0, dy = -1;
    order.reserve((size_t)cols * rows);
    for (int k = 0; k < side * side; ++k) {
        const int tx = cx + x, ty = cy + y;
        if (tx >= 0 && tx < cols && ty >= 0 && ty < rows && !taken[(size_t)ty * cols + tx]) {
            taken[(size_t)ty * cols + tx] = 1;
            const uint32_t px = (uint32_t)tx * tile_dim, py = (uint32_t)ty * tile_dim;
            yk_tile t;
            t.x0 = (uint16_t)px;
            t.y0 = (uint16_t)py;
            t.x1 = (uint16_t)(px + tile_dim < res_x ? px + tile_dim : res_x);
            t.y1 = (uint16_t)(py + tile_dim < res_y ? py + tile_dim : res_y);
            order.push_back(t);
        }
        if (x == y || (x < 0 && x == -y) || (x > 0 && x == 1 - y)) {
            const int t = dx;
            dx = -dy;
            dy = t;
        }
        x += dx;
        y += dy;
    }
    return order;
}

// ------------------------------------------------------------------ BVH build
namespace {

struct Prim {
    uint32_t shape;
    float bmin[3], bmax[3];
    float c[3];  // "centroid" = p_min + diagonal/0.5 (sic, bvh.rs:56)
};

struct Box {
    float lo[3], hi[3];
};
inline Box box_empty() {
    const float big = 3.40282347e+38f;
    return Box{{big, big, big}, {-big, -big, -big}};
}
inline void box_add_box(Box& b, const float* lo, const float* hi) {
    for (int k = 0; k < 3; ++k) {
        b.lo[k] = rmin(b.lo[k], lo[k]);
        b.hi[k] = rmax(b.hi[k], hi[k]);
    }
}
inline void box_add_point(Box& b, const float* p) {
    for (int k = 0; k < 3; ++k) {
        b.lo[k] = rmin(b.lo[k], p[k]);
        b.hi[k] = rmax(b.hi[k], p[k]);
    }
}
// bounds.rs:134-138
inline float box_area(const Box& b) {
    float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return 2.0f * (dx * dy + dz * dy + dx * dz);
}
// bounds.rs:147-156
inline int box_max_extent(const Box& b) {
    float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    if (dx > dy && dx > dz) return 0;
    if (dy > dz) return 1;
    return 2;
}

const size_t kNoSplit = (size_t)-1;
const int kBuckets = 12;

// SAH bucket of a primitive: impl_bounds.rs offset() then `(12*o).max(0) as usize`
inline int sah_bucket(const Box& cb, const Prim& p, int axis) {
    float o = p.c[axis] - cb.lo[axis];
    if (cb.hi[axis] != cb.lo[axis]) o /= cb.hi[axis] - cb.lo[axis];
    float bf = (float)kBuckets * o;
    float m = rmax(bf, 0.0f);
    if (m != m) return 0;
    if (m >= (float)kBuckets) return kBuckets - 1;
    int b = (int)m;
    return b < kBuckets - 1 ? b : kBuckets - 1;
}

// two-ended swap partition (the algorithm of itertools::partition)
template <class Pred> size_t swap_partition(Prim* a, size_t lo, size_t hi, Pred pred) {
    size_t count = 0, front = lo, back = hi;
    while (front < back) {
        size_t f = front++;
        if (!pred(a[f])) {
            bool swapped = false;
            while (front < back) {
                size_t b = --back;
                if (pred(a[b])) {
                    std::swap(a[f], a[b]);
                    swapped = true;
                    break;
                }
            }
            if (!swapped) return count;
        }
        ++count;
    }
    return count;
}

// "select_nth spec" (DESIGN.md): 3-way quickselect, middle pivot, on c[axis]
void select_nth(Prim* a, size_t lo, size_t hi, size_t k, int axis) {
    while (hi - lo > 1) {
        const float pivot = a[lo + (hi - lo) / 2].c[axis];
        size_t i = lo, lt = lo, gt = hi;
        while (i < gt) {
            const float v = a[i].c[axis];
            if (v < pivot) {
                std::swap(a[lt], a[i]);
                ++lt;
                ++i;
            } else if (v > pivot) {
                --gt;
                std::swap(a[i], a[gt]);
            } else {
                ++i;
            }
        }
        if (k < lt)
            hi = lt;
        else if (k >= gt)
            lo = gt;
        else
            return;
    }
}

// fn(chunk_index, lo, hi) on `threads` contiguous chunks of [start, end), chunk 0 on the caller
template <class Fn> void for_chunks(size_t start, size_t end, unsigned threads, Fn fn) {
    const size_t n = end - start, per = (n + threads - 1) / threads;
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < threads; ++t) {
        const size_t lo = start + std::min(n, per * t), hi = start + std::min(n, per * (t + 1));
        pool.emplace_back([=] { fn(t, lo, hi); });
    }
    fn(0u, start, start + std::min(n, per));
    for (std::thread& th : pool) th.join();
}
const size_t kParallelRange = (size_t)1 << 18;  // ranges at least this long are reduced by several threads
const unsigned kMaxThreads = 32;

// A subtree cut off the top of the tree and built by a worker thread into its own arena.
struct alignas(256) SubtreeTask {  // own cache lines: every leaf a worker emits moves its arena's vector ends
    size_t start, end;
    uint32_t depth;
    HostBvh arena;
};
const uint8_t kTaskMarker = 2;  // yk_bvh_node::is_leaf of a placeholder in the top arena (a = task index)

struct Builder {
    Prim* prims;
    uint32_t max_shapes;
    uint32_t method;
    HostBvh* out;
    // top phase of the parallel build: ranges of at most `grain` shapes become tasks
    std::vector<SubtreeTask>* tasks = nullptr;
    size_t grain = 0;
    unsigned par_threads = 1;  // > 1 in the top phase: the three reductions over a long range (bounds, centroid
                               // bounds, SAH buckets) are split over threads; min / max / counts merge exactly

    Box range_bounds(size_t start, size_t end, bool centroids) const {
        auto one = [&](size_t lo, size_t hi) {
            Box b = box_empty();
            if (centroids)
                for (size_t i = lo; i < hi; ++i) box_add_point(b, prims[i].c);
            else
                for (size_t i = lo; i < hi; ++i) box_add_box(b, prims[i].bmin, prims[i].bmax);
            return b;
        };
        if (par_threads <= 1 || end - start < kParallelRange) return one(start, end);
        Box part[kMaxThreads];
        for_chunks(start, end, par_threads, [&](unsigned t, size_t lo, size_t hi) { part[t] = one(lo, hi); });
        Box b = part[0];
        for (unsigned t = 1; t < par_threads; ++t) box_add_box(b, part[t].lo, part[t].hi);
        return b;
    }

    size_t split_equal_counts(size_t start, size_t end, int axis) {
        size_t mid = (start + end) / 2;
        select_nth(prims, start, end, mid, axis);
        return mid;
    }
    size_t split_middle(const Box& cb, size_t start, size_t end, int axis) {
        const float mid_value = (cb.lo[axis] + cb.hi[axis]) / 2.0f;
        return swap_partition(prims, start, end, [&](const Prim& p) { return p.c[axis] < mid_value; }) + start;
    }
    size_t split_sah(const Box& bounds, const Box& cb, size_t start, size_t end, int axis) {
        const size_t n = end - start;
        if (n <= 2) return start;
        size_t counts[kBuckets];
        Box boxes[kBuckets];
        for (int b = 0; b < kBuckets; ++b) {
            counts[b] = 0;
            boxes[b] = box_empty();
        }
        auto fill = [&](size_t lo, size_t hi, size_t* cnt, Box* bx) {
            for (size_t i = lo; i < hi; ++i) {
                int b = sah_bucket(cb, prims[i], axis);
                cnt[b] += 1;
                box_add_box(bx[b], prims[i].bmin, prims[i].bmax);
            }
        };
        if (par_threads <= 1 || n < kParallelRange) {
            fill(start, end, counts, boxes);
        } else {
            std::vector<size_t> pc((size_t)par_threads * kBuckets, 0);
            std::vector<Box> pb((size_t)par_threads * kBuckets, box_empty());
            for_chunks(start, end, par_threads, [&](unsigned t, size_t lo, size_t hi) { fill(lo, hi, &pc[(size_t)t * kBuckets], &pb[(size_t)t * kBuckets]); });
            for (unsigned t = 0; t < par_threads; ++t)
                for (int b = 0; b < kBuckets; ++b) {
                    counts[b] += pc[(size_t)t * kBuckets + b];
                    box_add_box(boxes[b], pb[(size_t)t * kBuckets + b].lo, pb[(size_t)t * kBuckets + b].hi);
                }
        }
        float best_cost = 0.0f;
        int best = 0;
        const float denom = rmax(box_area(bounds), 1e-10f);
        for (int i = 0; i < kBuckets - 1; ++i) {
            Box b0 = box_empty(), b1 = box_empty();
            size_t c0 = 0, c1 = 0;
            for (int j = 0; j <= i; ++j) {
                box_add_box(b0, boxes[j].lo, boxes[j].hi);
                c0 += counts[j];
            }
            for (int j = i + 1; j < kBuckets; ++j) {
                box_add_box(b1, boxes[j].lo, boxes[j].hi);
                c1 += counts[j];
            }
            float cost = 1.0f + ((float)c0 * box_area(b0) + (float)c1 * box_area(b1)) / denom;
            if (i == 0 || cost < best_cost) {  // min_by keeps the first minimum
                best_cost = cost;
                best = i;
            }
        }
        if (best_cost < (float)n)
            return swap_partition(prims, start, end, [&](const Prim& p) { return sah_bucket(cb, p, axis) <= best; }) + start;
        return kNoSplit;
    }

    uint32_t emit_leaf(const Box& b, size_t start, size_t end) {
        yk_bvh_node n;
        for (int k = 0; k < 3; ++k) {
            n.bmin[k] = b.lo[k];
            n.bmax[k] = b.hi[k];
        }
        n.a = (uint32_t)out->shape_order.size();
        n.count = (uint16_t)(end - start);
        n.axis = 0;
        n.is_leaf = 1;
        for (size_t i = start; i < end; ++i) out->shape_order.push_back(prims[i].shape);
        if (end - start > out->max_leaf_shapes) out->max_leaf_shapes = (uint32_t)(end - start);
        out->nodes.push_back(n);
        return (uint32_t)out->nodes.size() - 1;
    }

    // returns the node index (== depth-first position, as flatten_tree assigns it)
    uint32_t build(size_t start, size_t end, uint32_t depth) {
        if (tasks && depth > 1 && end - start <= grain) {  // cut here: a worker builds this subtree
            SubtreeTask t;
            t.start = start;
            t.end = end;
            t.depth = depth;
            tasks->push_back(std::move(t));
            yk_bvh_node ph;
            std::memset(&ph, 0, sizeof(ph));
            ph.a = (uint32_t)tasks->size() - 1;
            ph.is_leaf = kTaskMarker;
            out->nodes.push_back(ph);
            return (uint32_t)out->nodes.size() - 1;
        }
        if (depth > out->depth) out->depth = depth;
        const Box bounds = range_bounds(start, end, false);
        const size_t n = end - start;
        if (n <= max_shapes) return emit_leaf(bounds, start, end);
        const Box cb = range_bounds(start, end, true);
        const int axis = box_max_extent(cb);
        if (cb.hi[axis] == cb.lo[axis]) return emit_leaf(bounds, start, end);
        size_t mid;
        if (method == YK_SPLIT_SAH) {
            mid = split_sah(bounds, cb, start, end, axis);
            if (!(mid != start && mid != end)) mid = split_equal_counts(start, end, axis);
        } else if (method == YK_SPLIT_MIDDLE) {
            mid = split_middle(cb, start, end, axis);
            if (!(mid != start && mid != end)) mid = split_equal_counts(start, end, axis);
        } else {
            mid = split_equal_counts(start, end, axis);
        }
        if (mid == start) {
            out->split_failed = true;
            return emit_leaf(bounds, start, end);
        }
        if (mid == kNoSplit) return emit_leaf(bounds, start, end);
        const uint32_t self = (uint32_t)out->nodes.size();
        out->nodes.push_back(yk_bvh_node());
        const uint32_t c0 = build(start, mid, depth + 1);
        const uint32_t c1 = build(mid, end, depth + 1);
        yk_bvh_node n0 = out->nodes[c0], n1 = out->nodes[c1];
        yk_bvh_node me;
        for (int k = 0; k < 3; ++k) {  // BVHBuildNode::interior: child0.bounds.union_b(child1.bounds)
            me.bmin[k] = rmin(n0.bmin[k], n1.bmin[k]);
            me.bmax[k] = rmax(n0.bmax[k], n1.bmax[k]);
        }
        me.a = c1;
        me.count = 0;
        me.axis = (uint8_t)axis;
        me.is_leaf = 0;
        out->nodes[self] = me;
        return self;
    }
};

}  // namespace

// The build is the reference's recursion (bvh.rs:305-420), node for node.  For large inputs the
// recursion is cut where a range holds at most n / (8 * threads) shapes: the top of the tree is
// built on the calling thread exactly as before (same partitions in the same order on the same
// data), the cut-off ranges — disjoint slices of the primitive array, which nothing else touches
// — are built by worker threads into arenas of their own, and the arenas are spliced back in
// depth-first order with their node and shape indices rebased.  The result is the sequential
// build's, bit for bit (tests/test_bvh.py compares it with the oracle's sequential builder).
void build_bvh(const std::vector<ShapeBounds>& bounds, uint32_t max_shapes_in_node, uint32_t split_method, HostBvh& out) {
    out.nodes.clear();
    out.shape_order.clear();
    out.max_leaf_shapes = 0;
    out.depth = 0;
    out.split_failed = false;
    if (bounds.empty()) return;
    std::vector<Prim> prims(bounds.size());
    for (size_t i = 0; i < bounds.size(); ++i) {
        Prim& p = prims[i];
        p.shape = (uint32_t)i;
        for (int k = 0; k < 3; ++k) {
            p.bmin[k] = bounds[i].bmin[k];
            p.bmax[k] = bounds[i].bmax[k];
            p.c[k] = p.bmin[k] + ((p.bmax[k] - p.bmin[k]) / 0.5f);
        }
    }
    Builder b;
    b.prims = prims.data();
    b.max_shapes = max_shapes_in_node;
    b.method = split_method;
    unsigned n_threads = std::thread::hardware_concurrency();
    if (const char* e = std::getenv("YK_BVH_THREADS")) n_threads = (unsigned)std::max(1, std::atoi(e));
    n_threads = std::min(n_threads, kMaxThreads);
    if (n_threads <= 1 || bounds.size() < (1u << 16)) {  // small inputs: the plain recursion
        b.out = &out;
        out.nodes.reserve(bounds.size() * 2);
        out.shape_order.reserve(bounds.size());
        b.build(0, bounds.size(), 1);
        return;
    }
    // ---- top of the tree, with placeholders where subtrees are cut off
    HostBvh top;
    std::vector<SubtreeTask> tasks;
    b.out = &top;
    b.tasks = &tasks;
    b.grain = std::max<size_t>(bounds.size() / (8u * n_threads), std::max<size_t>(4096, max_shapes_in_node));
    b.par_threads = n_threads;
    const bool tm = std::getenv("YK_BVH_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    b.build(0, bounds.size(), 1);
    const double t1 = now();
    // ---- the subtrees, largest first
    std::vector<size_t> order(tasks.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return tasks[x].end - tasks[x].start > tasks[y].end - tasks[y].start; });
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= order.size()) return;
            SubtreeTask& t = tasks[order[k]];
            Builder w;
            w.prims = prims.data();
            w.max_shapes = max_shapes_in_node;
            w.method = split_method;
            w.out = &t.arena;
            t.arena.nodes.reserve((t.end - t.start) * 2);
            t.arena.shape_order.reserve(t.end - t.start);
            w.build(t.start, t.end, t.depth);
        }
    };
    std::vector<std::thread> pool;
    for (unsigned i = 1; i < n_threads && i < tasks.size(); ++i) pool.emplace_back(work);
    work();
    for (std::thread& th : pool) th.join();
    const double t2 = now();
    // ---- splice: the top arena is in depth-first order already, so one pass in index order lays
    // out the final array; interior nodes of the top get their second child and bounds afterwards
    out.depth = top.depth;
    out.max_leaf_shapes = top.max_leaf_shapes;
    out.split_failed = top.split_failed;
    out.nodes.reserve(bounds.size() * 2);
    out.shape_order.reserve(bounds.size());
    std::vector<uint32_t> final_index(top.nodes.size());
    for (size_t i = 0; i < top.nodes.size(); ++i) {
        const yk_bvh_node& tn = top.nodes[i];
        final_index[i] = (uint32_t)out.nodes.size();
        if (tn.is_leaf == kTaskMarker) {
            const HostBvh& a = tasks[tn.a].arena;
            const uint32_t node_base = (uint32_t)out.nodes.size(), shape_base = (uint32_t)out.shape_order.size();
            for (yk_bvh_node n : a.nodes) {
                n.a += n.is_leaf ? shape_base : node_base;
                out.nodes.push_back(n);
            }
            out.shape_order.insert(out.shape_order.end(), a.shape_order.begin(), a.shape_order.end());
            out.depth = std::max(out.depth, a.depth);
            out.max_leaf_shapes = std::max(out.max_leaf_shapes, a.max_leaf_shapes);
            out.split_failed = out.split_failed || a.split_failed;
        } else if (tn.is_leaf) {
            yk_bvh_node n = tn;
            n.a = (uint32_t)out.shape_order.size();
            out.shape_order.insert(out.shape_order.end(), top.shape_order.begin() + tn.a, top.shape_order.begin() + tn.a + tn.count);
            out.nodes.push_back(n);
        } else {
            out.nodes.push_back(tn);
        }
    }
    for (size_t i = top.nodes.size(); i-- > 0;) {
        const yk_bvh_node& tn = top.nodes[i];
        if (tn.is_leaf) continue;
        yk_bvh_node& me = out.nodes[final_index[i]];
        const uint32_t c0 = final_index[i + 1], c1 = final_index[tn.a];
        const yk_bvh_node &n0 = out.nodes[c0], &n1 = out.nodes[c1];
        for (int k = 0; k < 3; ++k) {  // BVHBuildNode::interior: child0.bounds.union_b(child1.bounds)
            me.bmin[k] = rmin(n0.bmin[k], n1.bmin[k]);
            me.bmax[k] = rmax(n0.bmax[k], n1.bmax[k]);
        }
        me.a = c1;
    }
    if (tm) std::fprintf(stderr, "[bvh] top %.3f s (%zu tasks, grain %zu), subtrees %.3f s on %u threads, splice %.3f s\n", t1 - t0, tasks.size(), b.grain, t2 - t1, n_threads, now() - t2);
}

}  // namespace yk
