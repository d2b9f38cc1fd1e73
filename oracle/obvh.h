// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of yuki/src/bvh.rs (build: :39-115, :305-523; flatten:
// :396-420; intersect :160-232; any_intersect :235-302).
//
// Two library calls in the reference's builder are not reproducible from its
// sources: `itertools::partition` (restated below from the crate's published
// two-ended swap algorithm) and `slice::select_nth_unstable_by` (Rust std's
// pdqselect, version dependent).  They only influence the ORDER of primitives
// and which of several equal-key primitives lands on which side of a median
// split; closest-hit results do not depend on it except for exact-t ties.
// The selection used here is the deterministic 3-way quickselect specified in
// DESIGN.md ("select_nth spec"); the HIP-side host builder implements the
// same spec so both trees are identical node for node.  Parity unpinned.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#include "oshapes.h"

namespace orc {

enum SplitMethod { SPLIT_SAH = 0, SPLIT_MIDDLE = 1, SPLIT_EQUAL_COUNTS = 2 };

struct BVHPrimitiveInfo {
    uint32_t shape_index;
    Bounds3f bounds;
    Point3f centroid;
};

// bvh.rs:536-556 — 32-byte node
struct BVHNode {
    Bounds3f bounds;
    uint32_t a;      // interior: second_child_index ; leaf: first_shape_index
    uint16_t count;  // leaf: shape_count ; interior: 0
    uint8_t axis;    // interior: split axis
    uint8_t is_leaf;
};
static_assert(sizeof(BVHNode) == 32, "node is 32 bytes like the reference's");

struct IntersectionResult {
    bool has_hit;
    Hit hit;
    size_t intersection_test_count;
    size_t intersection_count;
    size_t shape_test_count;  // not in the reference: leaf primitive tests, for the roofline's N_tri
};

struct BuildNode {
    Bounds3f bounds;
    int child0, child1;  // indices into the build arena, -1 for leaf
    int split_axis;
    size_t first_shape_index, shape_count;
};

// deterministic 3-way quickselect on centroid[axis] (see header comment)
inline void select_nth(std::vector<BVHPrimitiveInfo>& a, size_t lo, size_t hi, size_t k, int axis) {
#ifdef ORC_STD_NTH_ELEMENT
    // sensitivity flavour (liboracle_nth.so, tools/libm_sensitivity.py): another legitimate selection
    // algorithm (libstdc++'s introselect) in place of the select_nth spec — like Rust's pdqselect it
    // leaves equal keys in an order of its own.  Measures what that freedom does to the image.
    std::nth_element(a.begin() + lo, a.begin() + k, a.begin() + hi,
                     [axis](const BVHPrimitiveInfo& x, const BVHPrimitiveInfo& y) { return x.centroid[axis] < y.centroid[axis]; });
    return;
#endif
    while (hi - lo > 1) {
        float pivot = a[lo + (hi - lo) / 2].centroid[axis];
        size_t i = lo, lt = lo, gt = hi;
        while (i < gt) {
            float v = a[i].centroid[axis];
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

// itertools::partition: returns number of elements satisfying pred, which end up first
template <class Pred> inline size_t itertools_partition(std::vector<BVHPrimitiveInfo>& a, size_t lo, size_t hi, Pred pred) {
    size_t split_index = 0;
    size_t front = lo, back = hi;  // [front, back) unvisited
    while (front < back) {
        size_t f = front++;
        if (!pred(a[f])) {
            bool found = false;
            while (front < back) {
                size_t b = --back;
                if (pred(a[b])) {
                    std::swap(a[f], a[b]);
                    found = true;
                    break;
                }
            }
            if (!found) return split_index;
        }
        split_index += 1;
    }
    return split_index;
}

struct BVH {
    int split_method;
    size_t max_shapes_in_node;
    std::vector<BVHNode> nodes;
    std::vector<Shape> shapes;  // leaf order
    const Geometry* geom;

    // ---------------------------------------------------------------- build
    std::vector<BuildNode> arena;

    static const size_t NO_SPLIT = (size_t)-1;

    // bvh.rs:422-436
    static size_t split_equal_counts(std::vector<BVHPrimitiveInfo>& si, size_t start, size_t end, int axis) {
        size_t mid = (start + end) / 2;
        select_nth(si, start, end, mid, axis);
        return mid;
    }
    // bvh.rs:438-450
    static size_t split_middle(std::vector<BVHPrimitiveInfo>& si, const Bounds3f& cb, size_t start, size_t end, int axis) {
        float mid_value = (cb.p_min[axis] + cb.p_max[axis]) / 2.0f;
        return itertools_partition(si, start, end, [&](const BVHPrimitiveInfo& s) { return s.centroid[axis] < mid_value; }) +
               start;
    }
    static inline size_t bucket_of(const Bounds3f& cb, const BVHPrimitiveInfo& s, int axis) {
        const size_t N_BUCKETS = 12;
        float bf = (float)N_BUCKETS * cb.offset(s.centroid)[axis];
        float m = rmax(bf, 0.0f);
        // Rust `as usize`: saturating, NaN -> 0
        size_t b;
        if (m != m)
            b = 0;
        else if (m >= 1.8446744e19f)
            b = (size_t)-1;
        else
            b = (size_t)m;
        return b < N_BUCKETS - 1 ? b : N_BUCKETS - 1;
    }
    // bvh.rs:452-523
    static size_t split_sah(std::vector<BVHPrimitiveInfo>& si, const Bounds3f& bounds, const Bounds3f& cb, size_t start,
                            size_t end, int axis) {
        size_t shape_count = end - start;
        if (shape_count <= 2) return start;
        const size_t N_BUCKETS = 12;
        size_t counts[N_BUCKETS];
        Bounds3f bbs[N_BUCKETS];
        for (size_t i = 0; i < N_BUCKETS; ++i) counts[i] = 0;
        for (size_t i = start; i < end; ++i) {
            size_t b = bucket_of(cb, si[i], axis);
            counts[b] += 1;
            bbs[b] = bbs[b].union_b(si[i].bounds);
        }
        float costs[N_BUCKETS - 1];
        for (size_t i = 0; i < N_BUCKETS - 1; ++i) {
            Bounds3f b0, b1;
            size_t c0 = 0, c1 = 0;
            for (size_t j = 0; j <= i; ++j) {
                b0 = b0.union_b(bbs[j]);
                c0 += counts[j];
            }
            for (size_t j = i + 1; j < N_BUCKETS; ++j) {
                b1 = b1.union_b(bbs[j]);
                c1 += counts[j];
            }
            costs[i] = 1.0f + ((float)c0 * b0.surface_area() + (float)c1 * b1.surface_area()) /
                                  rmax(bounds.surface_area(), 1e-10f);
        }
        // Iterator::min_by returns the first of equal minima
        size_t min_bucket = 0;
        float min_cost = costs[0];
        for (size_t i = 1; i < N_BUCKETS - 1; ++i)
            if (costs[i] < min_cost) {
                min_cost = costs[i];
                min_bucket = i;
            }
        float leaf_cost = (float)shape_count;
        if (min_cost < leaf_cost) {
            return itertools_partition(si, start, end,
                                       [&](const BVHPrimitiveInfo& s) { return bucket_of(cb, s, axis) <= min_bucket; }) +
                   start;
        }
        return NO_SPLIT;
    }

    int make_leaf(const std::vector<Shape>& src, const std::vector<BVHPrimitiveInfo>& si, size_t start, size_t end,
                  const Bounds3f& bounds, std::vector<Shape>& ordered) {
        BuildNode n;
        n.bounds = bounds;
        n.child0 = n.child1 = -1;
        n.split_axis = 0;
        n.first_shape_index = ordered.size();
        n.shape_count = end - start;
        for (size_t i = start; i < end; ++i) ordered.push_back(src[si[i].shape_index]);
        arena.push_back(n);
        return (int)arena.size() - 1;
    }

    // bvh.rs:305-390.  Returns arena index; *nodes_in_tree accumulates.
    int recursive_build(const std::vector<Shape>& src, std::vector<BVHPrimitiveInfo>& si, size_t start, size_t end,
                        std::vector<Shape>& ordered, size_t& nodes_in_tree, bool& failed) {
        Bounds3f bounds;
        for (size_t i = start; i < end; ++i) bounds = bounds.union_b(si[i].bounds);
        size_t shape_count = end - start;
        if (shape_count <= max_shapes_in_node) {
            nodes_in_tree += 1;
            return make_leaf(src, si, start, end, bounds, ordered);
        }
        Bounds3f cb;
        for (size_t i = start; i < end; ++i) cb = cb.union_p(si[i].centroid);
        int axis = cb.maximum_extent();
        if (cb.p_max[axis] == cb.p_min[axis]) {
            nodes_in_tree += 1;
            return make_leaf(src, si, start, end, bounds, ordered);
        }
        size_t mid;
        if (split_method == SPLIT_SAH) {
            mid = split_sah(si, bounds, cb, start, end, axis);
            if (!(mid != start && mid != end)) mid = split_equal_counts(si, start, end, axis);
        } else if (split_method == SPLIT_MIDDLE) {
            mid = split_middle(si, cb, start, end, axis);
            if (!(mid != start && mid != end)) mid = split_equal_counts(si, start, end, axis);
        } else {
            mid = split_equal_counts(si, start, end, axis);
        }
        if (mid == start) {  // assert_ne!(mid, start) in the reference
            failed = true;
            nodes_in_tree += 1;
            return make_leaf(src, si, start, end, bounds, ordered);
        }
        if (mid == NO_SPLIT) {
            nodes_in_tree += 1;
            return make_leaf(src, si, start, end, bounds, ordered);
        }
        int c0 = recursive_build(src, si, start, mid, ordered, nodes_in_tree, failed);
        int c1 = recursive_build(src, si, mid, end, ordered, nodes_in_tree, failed);
        BuildNode n;
        n.bounds = arena[c0].bounds.union_b(arena[c1].bounds);
        n.child0 = c0;
        n.child1 = c1;
        n.split_axis = axis;
        n.first_shape_index = 0;
        n.shape_count = 0;
        arena.push_back(n);
        nodes_in_tree += 1;
        return (int)arena.size() - 1;
    }

    // bvh.rs:396-420
    size_t flatten_tree(int root, size_t next_index) {
        const BuildNode bn = arena[root];
        if (bn.child0 >= 0) {
            size_t self_index = next_index;
            size_t second_child_index = flatten_tree(bn.child0, self_index + 1);
            next_index = flatten_tree(bn.child1, second_child_index);
            BVHNode& n = nodes[self_index];
            n.bounds = bn.bounds;
            n.a = (uint32_t)second_child_index;
            n.count = 0;
            n.axis = (uint8_t)bn.split_axis;
            n.is_leaf = 0;
        } else {
            BVHNode& n = nodes[next_index];
            n.bounds = bn.bounds;
            n.a = (uint32_t)bn.first_shape_index;
            n.count = (uint16_t)bn.shape_count;
            n.axis = 0;
            n.is_leaf = 1;
            next_index += 1;
        }
        return next_index;
    }

    // bvh.rs:39-115.  Returns false when the reference would have panicked
    // (empty scene or a failed split).
    bool build(const Geometry* g, const std::vector<Shape>& src, size_t max_shapes, int method) {
        geom = g;
        split_method = method;
        max_shapes_in_node = max_shapes;
        nodes.clear();
        shapes.clear();
        arena.clear();
        if (src.empty()) return false;
        std::vector<BVHPrimitiveInfo> si(src.size());
        for (size_t i = 0; i < src.size(); ++i) {
            Bounds3f b = shape_world_bound(*g, src[i]);
            si[i].shape_index = (uint32_t)i;
            si[i].bounds = b;
            si[i].centroid = b.p_min + (b.diagonal() / 0.5f);  // sic (quirk 1)
        }
        arena.reserve(src.size() * 2);
        std::vector<Shape> ordered;
        ordered.reserve(src.size());
        size_t nodes_in_tree = 0;
        bool failed = false;
        int root = recursive_build(src, si, 0, src.size(), ordered, nodes_in_tree, failed);
        shapes.swap(ordered);
        nodes.resize(nodes_in_tree);
        flatten_tree(root, 0);
        arena.clear();
        arena.shrink_to_fit();
        return !failed;
    }

    Bounds3f bounds() const { return nodes[0].bounds; }

    // ------------------------------------------------------------ traversal
    // bvh.rs:160-232
    IntersectionResult intersect(Rayf ray) const {
        IntersectionResult res;
        res.has_hit = false;
        res.intersection_test_count = 0;
        res.intersection_count = 0;
        res.shape_test_count = 0;
        Vec3f inv_dir(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
        bool dir_is_neg[3] = {inv_dir.x < 0.0f, inv_dir.y < 0.0f, inv_dir.z < 0.0f};
        size_t current = 0, to_visit_index = 0;
        size_t stack[64];
        for (;;) {
            const BVHNode& node = nodes[current];
            res.intersection_test_count += 1;
            if (node.bounds.intersect(ray, inv_dir)) {
                res.intersection_count += 1;
                if (!node.is_leaf) {
                    if (dir_is_neg[node.axis]) {
                        stack[to_visit_index++] = current + 1;
                        current = node.a;
                    } else {
                        stack[to_visit_index++] = node.a;
                        current += 1;
                    }
                } else {
                    for (uint32_t i = node.a; i < node.a + (uint32_t)node.count; ++i) {
                        Hit h;
                        res.shape_test_count += 1;
                        if (shape_intersect(*geom, shapes[i], ray, h)) {
                            res.hit = h;
                            res.has_hit = true;
                            ray.t_max = h.t;
                        }
                    }
                    if (to_visit_index == 0) break;
                    current = stack[--to_visit_index];
                }
            } else {
                if (to_visit_index == 0) break;
                current = stack[--to_visit_index];
            }
        }
        return res;
    }

    // bvh.rs:235-302.  area_light = index of the sampled light or -1.
    bool any_intersect(const Rayf& ray, int area_light, size_t* node_tests = nullptr, size_t* shape_tests = nullptr) const {
        Vec3f inv_dir(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
        size_t current = 0, to_visit_index = 0;
        size_t stack[64];
        for (;;) {
            const BVHNode& node = nodes[current];
            if (node_tests) *node_tests += 1;
            if (node.bounds.intersect(ray, inv_dir)) {
                if (!node.is_leaf) {
                    if (inv_dir[node.axis] < 0.0f) {
                        stack[to_visit_index++] = current + 1;
                        current = node.a;
                    } else {
                        stack[to_visit_index++] = node.a;
                        current += 1;
                    }
                } else {
                    for (uint32_t i = node.a; i < node.a + (uint32_t)node.count; ++i) {
                        Hit h;
                        if (shape_tests) *shape_tests += 1;
                        if (shape_intersect(*geom, shapes[i], ray, h)) {
                            if (area_light >= 0 && h.si.area_light >= 0) {
                                if (h.si.area_light != area_light) return true;
                            } else {
                                return true;
                            }
                        }
                    }
                    if (to_visit_index == 0) break;
                    current = stack[--to_visit_index];
                }
            } else {
                if (to_visit_index == 0) break;
                current = stack[--to_visit_index];
            }
        }
        return false;
    }
};

}  // namespace orc
