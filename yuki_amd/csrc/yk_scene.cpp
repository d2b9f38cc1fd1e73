// yk_scene.cpp — scene description -> the reference's BVH (host) -> device records -> one copy per device.
// (yk_scene_create, bvh.rs:39-115 via yk_host.cpp; the record layouts are in yk_device.h and DESIGN.md §3.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "yk_internal.h"

template <class T> static yk_status upload(yk_context* ctx, DevBuf& buf, const T* src, size_t count) {
    size_t bytes = std::max<size_t>(count * sizeof(T), 16);
    HIP_TRY(ctx, buf.ensure(bytes));
    if (count) HIP_TRY(ctx, hipMemcpy(buf.p, src, count * sizeof(T), hipMemcpyHostToDevice));
    return YK_OK;
}

Material make_material(const yk_material_desc& m) {
    Material r;
    std::memset(&r, 0, sizeof(r));
    for (int k = 0; k < 3; ++k) {
        r.a[k] = m.a[k];
        r.b[k] = m.b[k];
    }
    const bool remap = (m.flags & YK_MAT_FLAG_REMAP) != 0;
    const bool textured = m.kind == YK_MAT_MATTE && (m.flags & YK_MAT_FLAG_TEXTURED_A) != 0;
    r.tex = textured ? m.a_texture + 1u : 0u;
    switch (m.kind) {
        case YK_MAT_MATTE: {  // matte.rs:27-39 (a textured Kd is tested for black per hit)
            if (!textured && m.a[0] == 0.0f && m.a[1] == 0.0f && m.a[2] == 0.0f) {
                r.kind = MK_BLACK;
            } else if (m.c == 0.0f) {
                r.kind = MK_LAMBERT;
            } else {  // oren_nayar.rs:20-27
                r.kind = MK_OREN_NAYAR;
                float sigma2 = m.c * m.c;
                r.c = 1.0f - (sigma2 / (2.0f * (sigma2 + 0.33f)));
                r.d = 0.45f * sigma2 / (sigma2 + 0.09f);
            }
            break;
        }
        case YK_MAT_GLASS:
            r.kind = MK_GLASS;
            r.c = m.c;
            break;
        case YK_MAT_METAL: {  // metal.rs:39-50, trowbridge_reitz.rs:15-20
            r.kind = MK_METAL;
            float roughness = remap ? roughness_to_alpha(m.c) : m.c;
            r.c = rmax(roughness, 0.001f);
            break;
        }
        default: {  // glossy.rs:37-49
            r.kind = MK_GLOSSY;
            float roughness = remap ? roughness_to_alpha(m.c) : m.c;
            r.c = rmax(roughness * roughness, 0.001f);
            break;
        }
    }
    return r;
}

DevLight make_light(const yk_light_desc& l) {
    DevLight d;
    std::memset(&d, 0, sizeof(d));
    d.kind = l.kind;
    for (int k = 0; k < 3; ++k) {
        d.p[k] = l.p[k];
        d.i[k] = l.i[k];
    }
    d.cos_total_width = l.cos_total_width;
    d.cos_falloff_start = l.cos_falloff_start;
    std::memcpy(d.w2l, l.world_to_light, 64);
    std::memcpy(d.s2w, l.sample_to_world, 64);
    V3 n = xf_normal(l.sample_to_world_inv, V3{0.0f, -1.0f, 0.0f});  // rectangular_light.rs:48
    d.n[0] = n.x;
    d.n[1] = n.y;
    d.n[2] = n.z;
    d.area = l.area;
    return d;
}

// Everything yk_scene_create derives from a scene description on the host (yk_internal.h).
struct SceneImage {
    std::shared_ptr<const HostBvh> bvh;
    const yk_scene_desc* d = nullptr;  // BORROWED: the caller's arrays (indices, points, normals, uvs, tri_material) are uploaded straight
                                       // from the description, so an image is only valid inside the call that built it
    uint32_t n_triangles = 0, n_spheres = 0, n_lights = 0, n_delta_lights = 0;
    yk_scene_info info;  // host part: node counts, bounds, build time
    bool has_device_records = false, wide = false, wide_auto = false;
    uint32_t root_ref = 0;
    std::vector<DevNode> dn, top, top_any;
    std::vector<DevNode4> dn4;
    std::vector<float4> tris, texels, prim_attr;
    std::vector<uint4> prim_shade, tex_info;
    std::vector<uint32_t> mesh_flags, tri_mesh;
    std::vector<int32_t> tri_al;
    std::vector<Material> mats;
    std::vector<DevSphere> spheres;
    std::vector<DevLight> lights;
};

// Host half of yk_scene_create: validation, BoundingVolumeHierarchy::new (bvh.rs:39-115) and — when `ctx` is given (its
// "top_nodes" / "wide_bvh" options apply) — the device records laid out from the tree.
yk_status yk_build_scene_image(yk_context* ctx, const yk_scene_desc* d, std::shared_ptr<SceneImage>& out) try {
    if (!d) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null scene description");
    out.reset();
    if ((uint64_t)d->n_triangles + d->n_spheres == 0) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "empty scene");
    if (d->n_triangles && (!d->points || !d->indices)) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "missing geometry arrays");
    if (d->max_shapes_in_node == 0 || d->max_shapes_in_node > 65535u || d->split_method > 2) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "bad BVH settings");
    for (uint32_t i = 0; i < d->n_triangles; ++i) {
        for (int k = 0; k < 3; ++k)
            if (d->indices[3 * i + k] >= d->n_vertices) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "vertex index out of range");
        if (d->tri_mesh && d->tri_mesh[i] >= d->n_meshes) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "mesh index out of range");
        if (d->tri_material && (d->tri_material[i] < 0 || (uint32_t)d->tri_material[i] >= d->n_materials))
            return fail(ctx, YK_ERR_INVALID_ARGUMENT, "material index out of range");
        if (d->tri_area_light && d->tri_area_light[i] >= (int32_t)d->n_lights) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "light index out of range");
    }
    if ((d->n_spheres && !d->spheres) || (d->n_materials && !d->materials) || (d->n_lights && !d->lights) || (d->n_meshes && !d->meshes))
        return fail(ctx, YK_ERR_INVALID_ARGUMENT, "a count is non-zero but its array is NULL");
    if (d->tri_area_light)  // Triangle.area_light is Option<Arc<RectangularLight>> (triangle.rs:22): -1 or a rectangular light
        for (uint32_t i = 0; i < d->n_triangles; ++i) {
            const int32_t al = d->tri_area_light[i];
            if (al < -1 || (al >= 0 && d->lights[al].kind != YK_LIGHT_RECT)) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "tri_area_light must be -1 or index a rectangular light");
        }
    for (uint32_t k = 0; k < d->n_spheres; ++k)
        if (d->spheres[k].material < 0 || (uint32_t)d->spheres[k].material >= d->n_materials) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "sphere material out of range");
    for (uint32_t m = 0; m < d->n_materials; ++m)
        if ((d->materials[m].flags & YK_MAT_FLAG_TEXTURED_A) && d->materials[m].kind == YK_MAT_MATTE && d->materials[m].a_texture >= d->n_textures)
            return fail(ctx, YK_ERR_INVALID_ARGUMENT, "material texture index out of range");
    if (d->n_materials >= (1u << 26)) return fail(ctx, YK_ERR_UNSUPPORTED, "more than 2^26 materials");
    for (uint32_t t = 0; t < d->n_textures; ++t)
        if (!d->textures || !d->textures[t].rgb || d->textures[t].width == 0 || d->textures[t].height == 0 || d->textures[t].width >= (1u << 24) ||
            d->textures[t].height >= (1u << 24))
            return fail(ctx, YK_ERR_INVALID_ARGUMENT, "bad texture");
    if (d->n_triangles && (!d->tri_material || d->n_materials == 0 || d->n_meshes == 0))
        return fail(ctx, YK_ERR_INVALID_ARGUMENT, "triangles need materials and meshes");
    for (uint32_t m = 0; m < d->n_meshes; ++m) {
        if (d->meshes[m].has_normals && !d->normals) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "mesh has_normals without a normals array");
        if (d->meshes[m].has_uvs && !d->uvs) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "mesh has_uvs without a uvs array");
    }

    std::shared_ptr<SceneImage> img = std::make_shared<SceneImage>();
    SceneImage* s = img.get();
    std::shared_ptr<HostBvh> bvh = std::make_shared<HostBvh>();
    s->bvh = bvh;
    s->d = d;
    s->n_triangles = d->n_triangles;
    s->n_spheres = d->n_spheres;
    s->n_lights = d->n_lights;
    for (uint32_t l = 0; l < d->n_lights; ++l) s->n_delta_lights += d->lights[l].kind != YK_LIGHT_RECT ? 1u : 0u;
    std::memset(&s->info, 0, sizeof(s->info));

    // world bounds of every shape: Triangle::world_bound (triangle.rs:229-235),
    // Sphere::world_bound (sphere.rs:121-123)
    std::vector<ShapeBounds> sb((size_t)d->n_triangles + d->n_spheres);
    for (uint32_t i = 0; i < d->n_triangles; ++i) {
        const float* p0 = d->points + 3 * (size_t)d->indices[3 * i];
        const float* p1 = d->points + 3 * (size_t)d->indices[3 * i + 1];
        const float* p2 = d->points + 3 * (size_t)d->indices[3 * i + 2];
        for (int k = 0; k < 3; ++k) {
            sb[i].bmin[k] = rmin(rmin(p0[k], p1[k]), p2[k]);
            sb[i].bmax[k] = rmax(rmax(p0[k], p1[k]), p2[k]);
        }
    }
    for (uint32_t i = 0; i < d->n_spheres; ++i) {
        const yk_sphere_desc& sp = d->spheres[i];
        const float r = sp.radius;
        const float lo[3] = {-r, -r, -r}, hi[3] = {r, r, r};
        const float big = 3.40282347e+38f;
        ShapeBounds b = {{big, big, big}, {-big, -big, -big}};
        const int corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}};  // transform.rs:194-206
        for (int c = 0; c < 8; ++c) {
            V3 q = xf_point(sp.object_to_world, V3{corner[c][0] ? hi[0] : lo[0], corner[c][1] ? hi[1] : lo[1], corner[c][2] ? hi[2] : lo[2]});
            const float qq[3] = {q.x, q.y, q.z};
            for (int k = 0; k < 3; ++k) {
                b.bmin[k] = rmin(b.bmin[k], qq[k]);
                b.bmax[k] = rmax(b.bmax[k], qq[k]);
            }
        }
        sb[(size_t)d->n_triangles + i] = b;
    }
    if (d->shape_order) {  // the caller's Scene.shapes order (a permutation of all shapes)
        std::vector<uint8_t> seen(sb.size(), 0);
        std::vector<ShapeBounds> ordered(sb.size());
        for (size_t i = 0; i < sb.size(); ++i) {
            const uint32_t src = d->shape_order[i];
            if (src >= sb.size() || seen[src]) {
                return fail(ctx, YK_ERR_INVALID_ARGUMENT, "shape_order is not a permutation of the shapes");
            }
            seen[src] = 1;
            ordered[i] = sb[src];
        }
        sb.swap(ordered);
    }
    double t0 = now_seconds();
    build_bvh(sb, d->max_shapes_in_node, d->split_method, *bvh);
    s->info.build_seconds = now_seconds() - t0;
    if (d->shape_order)  // leaf order -> position in Scene.shapes -> source shape
        for (uint32_t& o : bvh->shape_order) o = d->shape_order[o];
    if (bvh->split_failed || bvh->nodes.empty()) {
        return fail(ctx, YK_ERR_BVH_BUILD, "BVH split failed (reference: assert_ne!(mid, start))");
    }
    s->info.n_nodes = bvh->nodes.size();
    s->info.n_shapes = bvh->shape_order.size();
    s->info.max_leaf_shapes = bvh->max_leaf_shapes;
    s->info.tree_depth = bvh->depth;
    for (int k = 0; k < 3; ++k) {
        s->info.bounds_min[k] = bvh->nodes[0].bmin[k];
        s->info.bounds_max[k] = bvh->nodes[0].bmax[k];
    }
    uint64_t n_interior = 0;
    for (const yk_bvh_node& n : bvh->nodes) n_interior += n.is_leaf ? 0 : 1;
    s->info.n_interior = n_interior;

    if (ctx) {  // device records (a host-only scene — ctx == NULL — stops at the tree)
        const std::vector<yk_bvh_node>& nodes = bvh->nodes;
        // interior index of each reference node = number of interior nodes before it
        std::vector<uint32_t> interior_index(nodes.size());
        uint32_t cnt = 0;
        for (size_t i = 0; i < nodes.size(); ++i) {
            interior_index[i] = cnt;
            if (!nodes[i].is_leaf) ++cnt;
        }
        if (nodes.size() > YK_REF_INDEX_MAX || bvh->shape_order.size() > YK_REF_INDEX_MAX) {
            return fail(ctx, YK_ERR_UNSUPPORTED, "more than 2^28 BVH nodes or shapes");
        }
        auto ref_of = [&](uint32_t idx) -> uint32_t { return nodes[idx].is_leaf ? (YK_LEAF_BIT | nodes[idx].a) : interior_index[idx]; };
        std::vector<DevNode>& dn = s->dn;
        dn.assign(std::max<size_t>(n_interior, 1), DevNode());
        for (size_t i = 0; i < nodes.size(); ++i) {
            if (nodes[i].is_leaf) continue;
            const yk_bvh_node& c0 = nodes[i + 1];
            const yk_bvh_node& c1 = nodes[nodes[i].a];
            DevNode& o = dn[interior_index[i]];
            o.q0 = make_float4(c0.bmin[0], c0.bmin[1], c0.bmin[2], c0.bmax[0]);
            o.q1 = make_float4(c0.bmax[1], c0.bmax[2], c1.bmin[0], c1.bmin[1]);
            o.q2 = make_float4(c1.bmin[2], c1.bmax[0], c1.bmax[1], c1.bmax[2]);
            o.q3 = make_uint4(ref_of((uint32_t)i + 1), ref_of(nodes[i].a) | ((uint32_t)nodes[i].axis << YK_AXIS_SHIFT), 0u, 0u);
        }
        // top of the tree, breadth first, for the LDS-resident copies (YK_TOP_BIT refs).  Two sets: the closest-hit kernels
        // keep 8-byte stack entries (ref, entry distance) in LDS and have room for trace_top_nodes() nodes beside them; the
        // any-hit kernel's entries are a bare ref (4 bytes), which leaves room for trace_top_nodes_any() — more than twice as many.
        auto build_top = [&](size_t cap, std::vector<DevNode>& top) {
            top.clear();
            if (nodes[0].is_leaf || cap == 0) return;
            std::vector<uint32_t> order;  // reference node indices, breadth first
            std::vector<uint32_t> top_id(nodes.size(), 0xffffffffu);
            order.push_back(0);
            top_id[0] = 0;
            for (size_t q = 0; q < order.size() && order.size() < cap; ++q) {
                const uint32_t P = order[q];
                for (uint32_t c : {P + 1, nodes[P].a}) {
                    if (!nodes[c].is_leaf && order.size() < cap) {
                        top_id[c] = (uint32_t)order.size();
                        order.push_back(c);
                    }
                }
            }
            for (uint32_t P : order) {
                DevNode t = dn[interior_index[P]];
                const uint32_t c0 = P + 1, c1 = nodes[P].a;
                if (top_id[c0] != 0xffffffffu) t.q3.x = YK_TOP_BIT | top_id[c0];
                if (top_id[c1] != 0xffffffffu) t.q3.y = YK_TOP_BIT | top_id[c1] | ((uint32_t)nodes[P].axis << YK_AXIS_SHIFT);
                top.push_back(t);
            }
        };
        build_top((size_t)std::min<int64_t>(ctx->top_nodes, trace_top_nodes()), s->top);
        build_top((size_t)std::min<int64_t>(ctx->top_nodes, trace_top_nodes_any()), s->top_any);
        // 4-wide collapse (DevNode4): one node per reference interior node reached at even depth
        // below the root.  Built only while the traversal stack of the collapsed tree is
        // guaranteed to fit (the reference asserts on its own stack depth, bvh.rs:172-174).
        std::vector<DevNode4>& dn4 = s->dn4;
        const bool wide = s->wide = ctx->wide_bvh != 0 && !nodes[0].is_leaf && bvh->depth <= 64;
        if (wide) {
            dn4.reserve(n_interior / 2 + 1);
            struct Todo {
                uint32_t binary;  // reference node index of P
                uint32_t slot;    // DevNode4 index to fill
            };
            std::vector<Todo> stack;
            dn4.emplace_back();
            stack.push_back(Todo{0u, 0u});
            while (!stack.empty()) {
                const Todo td = stack.back();
                stack.pop_back();
                const uint32_t P = td.binary, A = P + 1, B = nodes[P].a;
                uint32_t child[4] = {YK_REF_NONE, YK_REF_NONE, YK_REF_NONE, YK_REF_NONE};  // reference node index per slot
                if (nodes[A].is_leaf) {
                    child[0] = A;
                } else {
                    child[0] = A + 1;
                    child[1] = nodes[A].a;
                }
                if (nodes[B].is_leaf) {
                    child[2] = B;
                } else {
                    child[2] = B + 1;
                    child[3] = nodes[B].a;
                }
                float box[4][6] = {};
                uint32_t ref[4];
                for (int k = 0; k < 4; ++k) {
                    ref[k] = YK_REF_NONE;
                    if (child[k] == YK_REF_NONE) continue;
                    const yk_bvh_node& c = nodes[child[k]];
                    for (int a = 0; a < 3; ++a) {
                        box[k][a] = c.bmin[a];
                        box[k][3 + a] = c.bmax[a];
                    }
                    if (c.is_leaf) {
                        ref[k] = YK_LEAF_BIT | c.a;
                    } else {
                        ref[k] = (uint32_t)dn4.size();
                        dn4.emplace_back();
                    }
                }
                // children are expanded so that the first visited subtree (for a positive ray) follows in memory
                for (int k = 3; k >= 0; --k)
                    if (ref[k] != YK_REF_NONE && !(ref[k] & YK_LEAF_BIT)) stack.push_back(Todo{child[k], ref[k]});
                DevNode4& o = dn4[td.slot];
                o.q0 = make_float4(box[0][0], box[0][1], box[0][2], box[0][3]);
                o.q1 = make_float4(box[0][4], box[0][5], box[1][0], box[1][1]);
                o.q2 = make_float4(box[1][2], box[1][3], box[1][4], box[1][5]);
                o.q3 = make_float4(box[2][0], box[2][1], box[2][2], box[2][3]);
                o.q4 = make_float4(box[2][4], box[2][5], box[3][0], box[3][1]);
                o.q5 = make_float4(box[3][2], box[3][3], box[3][4], box[3][5]);
                o.q6 = make_uint4(ref[0], ref[1], ref[2], ref[3]);
                const uint32_t axA = nodes[A].is_leaf ? 0u : nodes[A].axis, axB = nodes[B].is_leaf ? 0u : nodes[B].axis;
                o.q7 = make_uint4((uint32_t)nodes[P].axis | (axA << 2) | (axB << 4), 0u, 0u, 0u);
            }
        }
        const size_t np = bvh->shape_order.size();
        std::vector<float4>& tris = s->tris;
        tris.assign(3 * np, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        std::vector<uint4>& prim_shade = s->prim_shade;
        prim_shade.assign(np, make_uint4(0u, 0u, 0u, 0u));
        std::vector<uint32_t> mat_kind(std::max<uint32_t>(d->n_materials, 1), 0u);  // device BSDF kind (MK_*) per material
        for (uint32_t m = 0; m < d->n_materials; ++m) mat_kind[m] = make_material(d->materials[m]).kind & 7u;
        std::vector<uint8_t> last(np, 0);
        for (const yk_bvh_node& n : nodes)
            if (n.is_leaf) last[(size_t)n.a + n.count - 1] = 1;
        for (size_t p = 0; p < np; ++p) {
            uint32_t src = bvh->shape_order[p];
            if (src >= d->n_triangles) {  // sphere: only the source index and the flags are read
                uint32_t none = 0xffffffffu, fl = (last[p] ? YK_PRIM_LAST : 0u) | YK_PRIM_SPHERE | (mat_kind[d->spheres[src - d->n_triangles].material] << YK_PRIM_KIND_SHIFT);
                float w0, w1, w2;
                std::memcpy(&w0, &none, 4);
                std::memcpy(&w1, &src, 4);
                std::memcpy(&w2, &fl, 4);
                tris[3 * p + 0] = make_float4(0.0f, 0.0f, 0.0f, w0);
                tris[3 * p + 1] = make_float4(0.0f, 0.0f, 0.0f, w1);
                tris[3 * p + 2] = make_float4(0.0f, 0.0f, 0.0f, w2);
                prim_shade[p] = make_uint4(0u, 0u, 0u, ((uint32_t)d->spheres[src - d->n_triangles].material << 6) | (mat_kind[d->spheres[src - d->n_triangles].material] << 3));
                continue;
            }
            const float* p0 = d->points + 3 * (size_t)d->indices[3 * src];
            const float* p1 = d->points + 3 * (size_t)d->indices[3 * src + 1];
            const float* p2 = d->points + 3 * (size_t)d->indices[3 * src + 2];
            int al = d->tri_area_light ? d->tri_area_light[src] : -1;
            uint32_t alb = (uint32_t)al, lastb = (last[p] ? YK_PRIM_LAST : 0u) | (mat_kind[d->tri_material[src]] << YK_PRIM_KIND_SHIFT);
            float w0, w1, w2;
            std::memcpy(&w0, &alb, 4);
            std::memcpy(&w1, &src, 4);
            std::memcpy(&w2, &lastb, 4);
            tris[3 * p + 0] = make_float4(p0[0], p0[1], p0[2], w0);
            tris[3 * p + 1] = make_float4(p1[0], p1[1], p1[2], w1);
            tris[3 * p + 2] = make_float4(p2[0], p2[1], p2[2], w2);
            const yk_mesh_desc& md = d->meshes[d->tri_mesh ? d->tri_mesh[src] : 0];
            const uint32_t mfl = (md.has_normals ? YK_MESH_NORMALS : 0u) | (md.has_uvs ? YK_MESH_UVS : 0u) | (md.swaps_handedness ? YK_MESH_SWAPS : 0u);
            prim_shade[p] = make_uint4(d->indices[3 * src], d->indices[3 * src + 1], d->indices[3 * src + 2],
                                       ((uint32_t)d->tri_material[src] << 6) | (mat_kind[d->tri_material[src]] << 3) | mfl);
        }
        if (d->normals || d->uvs) {  // leaf-order copy of the per-vertex normals / uvs (yk_device.h: DevScene::prim_attr)
            s->prim_attr.assign(4 * np, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
            for (size_t p = 0; p < np; ++p) {
                const uint32_t src = bvh->shape_order[p];
                if (src >= d->n_triangles) continue;
                const yk_mesh_desc& md = d->meshes[d->tri_mesh ? d->tri_mesh[src] : 0];
                float nrm[3][3] = {}, uv[3][2] = {};
                for (int k = 0; k < 3; ++k) {
                    const size_t vi = d->indices[3 * (size_t)src + k];
                    if (md.has_normals)
                        for (int c = 0; c < 3; ++c) nrm[k][c] = d->normals[3 * vi + c];
                    if (md.has_uvs)
                        for (int c = 0; c < 2; ++c) uv[k][c] = d->uvs[2 * vi + c];
                }
                s->prim_attr[4 * p + 0] = make_float4(nrm[0][0], nrm[0][1], nrm[0][2], uv[0][0]);
                s->prim_attr[4 * p + 1] = make_float4(nrm[1][0], nrm[1][1], nrm[1][2], uv[0][1]);
                s->prim_attr[4 * p + 2] = make_float4(nrm[2][0], nrm[2][1], nrm[2][2], uv[1][0]);
                s->prim_attr[4 * p + 3] = make_float4(uv[1][1], uv[2][0], uv[2][1], 0.0f);
            }
        }
        std::vector<uint32_t>& mesh_flags = s->mesh_flags;
        mesh_flags.assign(std::max<uint32_t>(d->n_meshes, 1), 0);
        for (uint32_t m = 0; m < d->n_meshes; ++m)
            mesh_flags[m] = (d->meshes[m].has_normals ? YK_MESH_NORMALS : 0u) | (d->meshes[m].has_uvs ? YK_MESH_UVS : 0u) |
                            (d->meshes[m].swaps_handedness ? YK_MESH_SWAPS : 0u);
        std::vector<Material>& mats = s->mats;
        mats.resize(std::max<uint32_t>(d->n_materials, 1));
        for (uint32_t m = 0; m < d->n_materials; ++m) mats[m] = make_material(d->materials[m]);
        std::vector<DevSphere>& spheres = s->spheres;
        spheres.resize(std::max<uint32_t>(d->n_spheres, 1));
        for (uint32_t k = 0; k < d->n_spheres; ++k) {
            DevSphere& o = spheres[k];
            std::memcpy(o.o2w, d->spheres[k].object_to_world, 64);
            std::memcpy(o.w2o, d->spheres[k].world_to_object, 64);
            o.radius = d->spheres[k].radius;
            o.material = d->spheres[k].material;
            const float* m = o.o2w;  // Transform::swaps_handedness, transform.rs:85-91
            float det = m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) + m[2] * (m[4] * m[9] - m[5] * m[8]);
            o.swaps_handedness = det < 0.0f ? 1u : 0u;
            o.pad = 0;
        }
        std::vector<DevLight>& lights = s->lights;
        lights.resize(std::max<uint32_t>(d->n_lights, 1));
        for (uint32_t l = 0; l < d->n_lights; ++l) lights[l] = make_light(d->lights[l]);
        std::vector<uint32_t>& tri_mesh = s->tri_mesh;
        tri_mesh.assign(d->n_triangles, 0);
        if (d->tri_mesh) std::memcpy(tri_mesh.data(), d->tri_mesh, sizeof(uint32_t) * d->n_triangles);
        std::vector<int32_t>& tri_al = s->tri_al;
        tri_al.assign(d->n_triangles, -1);
        if (d->tri_area_light) std::memcpy(tri_al.data(), d->tri_area_light, sizeof(int32_t) * d->n_triangles);

        for (uint32_t t = 0; t < d->n_textures; ++t) {
            const yk_texture_desc& td = d->textures[t];
            s->tex_info.push_back(make_uint4((unsigned)s->texels.size(), td.width, td.height, 0u));
            const size_t n = (size_t)td.width * td.height;
            if (s->texels.size() + n > 0xffffffffull) return fail(ctx, YK_ERR_UNSUPPORTED, "more than 2^32 texels");
            for (size_t k = 0; k < n; ++k) s->texels.push_back(make_float4(td.rgb[3 * k], td.rgb[3 * k + 1], td.rgb[3 * k + 2], 0.0f));
        }
        s->root_ref = ref_of(0);
        s->wide_auto = wide && ctx->wide_bvh == 2;
        s->has_device_records = true;
    }
    out = img;
    return YK_OK;
} YK_CATCH(ctx)

// Device half: one copy of the image in the HBM of ctx's device.
yk_status yk_upload_scene_image(yk_context* ctx, const std::shared_ptr<SceneImage>& img, yk_scene** out) try {
    if (!out) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null out");
    *out = nullptr;
    if (!img || !img->bvh) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null scene image");
    yk_scene* s = new yk_scene();
    struct SceneGuard {  // frees the half-built scene on every early return and on an exception
        yk_scene* s;
        ~SceneGuard() {
            if (s) yk_scene_destroy(s);
        }
    } guard{s};
    s->device = ctx ? ctx->device : -1;
    s->bvh = img->bvh;
    s->n_triangles = img->n_triangles;
    s->n_spheres = img->n_spheres;
    s->n_lights = img->n_lights;
    s->n_delta_lights = img->n_delta_lights;
    s->info = img->info;
    if (ctx) {
        if (!img->has_device_records) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "scene image was built without device records");
        const yk_scene_desc* d = img->d;
        (void)hipSetDevice(ctx->device);
        double u0 = now_seconds();
        yk_status st;
#define UP(buf, ptr, n) \
    if ((st = upload(ctx, s->buf, ptr, n)) != YK_OK) return st;
        UP(nodes, img->dn.data(), img->dn.size());
        UP(nodes4, img->dn4.data(), img->dn4.size());
        UP(top_nodes, img->top.data(), img->top.size());
        UP(top_nodes_any, img->top_any.data(), img->top_any.size());
        UP(tris, img->tris.data(), img->tris.size());
        UP(prim_shade, img->prim_shade.data(), img->prim_shade.size());
        UP(prim_attr, img->prim_attr.data(), img->prim_attr.size());
        UP(indices, d->indices, 3 * (size_t)d->n_triangles);
        UP(points, d->points, 3 * (size_t)d->n_vertices);
        UP(normals, d->normals, d->normals ? 3 * (size_t)d->n_vertices : 0);
        UP(uvs, d->uvs, d->uvs ? 2 * (size_t)d->n_vertices : 0);
        UP(tri_mesh, img->tri_mesh.data(), img->tri_mesh.size());
        UP(tri_material, d->tri_material, (size_t)d->n_triangles);
        UP(tri_area_light, img->tri_al.data(), img->tri_al.size());
        UP(mesh_flags, img->mesh_flags.data(), img->mesh_flags.size());
        UP(materials, img->mats.data(), img->mats.size());
        UP(lights, img->lights.data(), img->lights.size());
        UP(spheres, img->spheres.data(), img->spheres.size());
        UP(texels, img->texels.data(), img->texels.size());
        UP(tex_info, img->tex_info.data(), img->tex_info.size());
#undef UP
        const std::vector<yk_bvh_node>& nodes = img->bvh->nodes;
        DevScene& ds = s->dev;
        ds.nodes = s->nodes.as<DevNode>();
        ds.nodes4 = img->wide ? s->nodes4.as<DevNode4>() : nullptr;
        s->wide_auto = img->wide_auto;
        ds.top_nodes = s->top_nodes.as<DevNode>();
        ds.n_top = (uint32_t)img->top.size();
        ds.top_nodes_any = s->top_nodes_any.as<DevNode>();
        ds.n_top_any = (uint32_t)img->top_any.size();
        ds.tris = s->tris.as<float4>();
        ds.prim_shade = s->prim_shade.as<uint4>();
        ds.prim_attr = img->prim_attr.empty() ? nullptr : s->prim_attr.as<float4>();
        ds.spheres = d->n_spheres ? s->spheres.as<DevSphere>() : nullptr;
        ds.n_triangles = d->n_triangles;
        ds.root_ref = img->root_ref;
        for (int k = 0; k < 3; ++k) {
            ds.root_bmin[k] = nodes[0].bmin[k];
            ds.root_bmax[k] = nodes[0].bmax[k];
            ds.background[k] = d->background[k];
        }
        ds.indices = s->indices.as<uint32_t>();
        ds.points = s->points.as<float>();
        ds.normals = s->normals.as<float>();
        ds.uvs = s->uvs.as<float>();
        ds.tri_mesh = s->tri_mesh.as<uint32_t>();
        ds.tri_material = s->tri_material.as<int32_t>();
        ds.tri_area_light = s->tri_area_light.as<int32_t>();
        ds.mesh_flags = s->mesh_flags.as<uint32_t>();
        ds.materials = s->materials.as<Material>();
        ds.lights = s->lights.as<DevLight>();
        ds.n_lights = d->n_lights;
        ds.texels = d->n_textures ? s->texels.as<float4>() : nullptr;
        ds.tex_info = d->n_textures ? s->tex_info.as<uint4>() : nullptr;
        s->on_device = true;
        s->info.upload_seconds = now_seconds() - u0;
        DevBuf* all[] = {&s->nodes, &s->nodes4, &s->top_nodes, &s->top_nodes_any, &s->tris, &s->prim_shade, &s->prim_attr, &s->indices, &s->points, &s->normals, &s->uvs, &s->tri_mesh, &s->tri_material, &s->tri_area_light,
                         &s->mesh_flags, &s->materials, &s->lights, &s->spheres, &s->texels, &s->tex_info};
        for (DevBuf* b : all) s->info.device_bytes += b->bytes;
    }
    guard.s = nullptr;
    *out = s;
    return YK_OK;
} YK_CATCH(ctx)

extern "C" {

yk_status yk_scene_create(yk_context* ctx, const yk_scene_desc* d, yk_scene** out) {
    std::unique_lock<std::recursive_mutex> yk_lock_;
    if (ctx) yk_lock_ = std::unique_lock<std::recursive_mutex>(ctx->mu);
    if (!d || !out) return fail(ctx, YK_ERR_INVALID_ARGUMENT, "null scene description");
    *out = nullptr;
    std::shared_ptr<SceneImage> img;
    yk_status st = yk_build_scene_image(ctx, d, img);
    if (st != YK_OK) return st;
    return yk_upload_scene_image(ctx, img, out);
}


void yk_scene_destroy(yk_scene* s) {
    if (!s) return;
    if (s->device >= 0) (void)hipSetDevice(s->device);
    DevBuf* all[] = {&s->nodes, &s->nodes4, &s->top_nodes, &s->top_nodes_any, &s->tris, &s->prim_shade, &s->prim_attr, &s->indices, &s->points, &s->normals, &s->uvs, &s->tri_mesh, &s->tri_material, &s->tri_area_light,
                     &s->mesh_flags, &s->materials, &s->lights, &s->spheres, &s->texels, &s->tex_info};
    for (DevBuf* b : all) b->release();
    delete s;
}

yk_status yk_scene_get_info(const yk_scene* s, yk_scene_info* out) {
    if (!s || !out) return YK_ERR_INVALID_ARGUMENT;
    *out = s->info;
    return YK_OK;
}

yk_status yk_scene_export_bvh(const yk_scene* s, yk_bvh_node* nodes, uint32_t* shape_order) {
    if (!s) return YK_ERR_INVALID_ARGUMENT;
    if (nodes) std::memcpy(nodes, s->bvh->nodes.data(), s->bvh->nodes.size() * sizeof(yk_bvh_node));
    if (shape_order) std::memcpy(shape_order, s->bvh->shape_order.data(), s->bvh->shape_order.size() * sizeof(uint32_t));
    return YK_OK;
}

}  // extern "C"
