// yk_kernels.hip — gfx950 kernels of the wavefront Path integrator.
//
// One bounce of every in-flight path = four launches over dense, compacted arrays:
//
//   trace_closest   BoundingVolumeHierarchy::intersect      bvh.rs:160-232
//   shade           Path::li_internal body                   path.rs:89-169
//                   (surface reconstruction, NEE light sampling, emission,
//                    BSDF sampling, Russian roulette, wave-ballot compaction of
//                    survivors into the other state buffer)
//   trace_any       BoundingVolumeHierarchy::any_intersect   bvh.rs:235-302
//   accumulate      `incoming_radiance += beta * radiance`   path.rs:102-129
//
// plus raygen (Integrator::render's sample loop + Camera::ray) and resolve
// (the per-pixel mean, integrators/mod.rs:172-182).  All floating-point work
// follows the reference's operation order; built with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "yk_device.h"
#include "yk_geom.h"
#include "yk_kernels.h"
#include "yk_rng.h"

namespace yk {

#define YK_WAVE 64

__device__ __forceinline__ V3 f4_xyz(float4 v) { return V3{v.x, v.y, v.z}; }
__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & (YK_WAVE - 1); }

// wave-level append: every lane of the wave calls this in converged control flow;
// lanes with `want` get consecutive slots from one atomic per wave.
__device__ __forceinline__ unsigned wave_append(bool want, unsigned* counter) {
    unsigned long long mask = __ballot(want);
    unsigned total = (unsigned)__popcll(mask);
    unsigned lane = lane_id();
    unsigned prefix = (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
    unsigned base = 0;
    int leader = total ? (int)__ffsll((long long)mask) - 1 : 0;
    if (total && (int)lane == leader) base = atomicAdd(counter, total);
    base = __shfl(base, leader);
    return base + prefix;
}

// ------------------------------------------------------------------ pixel table
// chunk-local pixel index -> pixel coordinates, tile-major / row-major in tile
// (the order Integrator::render visits them, integrators/mod.rs:145)
__global__ void k_pixel_table(const yk_tile* tiles, const uint32_t* tile_offset, uint32_t n_tiles, uint32_t n_pixels,
                              uint32_t* pixel_xy) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    uint32_t lo = 0, hi = n_tiles;  // last tile with offset <= i
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (tile_offset[mid] <= i)
            lo = mid;
        else
            hi = mid;
    }
    yk_tile t = tiles[lo];
    uint32_t w = (uint32_t)t.x1 - t.x0;
    uint32_t r = i - tile_offset[lo];
    uint32_t x = t.x0 + r % w, y = t.y0 + r / w;
    pixel_xy[i] = x | (y << 16);
}

// ------------------------------------------------------------------ raygen
// sampler.start_pixel_sample(p, sample_index, 0); p_film = p + get_2d();
// ray = camera.ray(p_film)     integrators/mod.rs:163-169, camera.rs:105-114
__device__ __forceinline__ void camera_ray(const DevCamera& cam, float fx, float fy, V3& o, V3& d) {
    V3 p_camera = xf_point(cam.r2c, V3{fx, fy, 0.0f});
    V3 dir = normalize(p_camera);
    o = xf_point(cam.c2w, V3{0.0f, 0.0f, 0.0f});
    d = xf_vector(cam.c2w, dir);
}

__global__ void k_raygen(DevCamera cam, RenderParams prm, const uint32_t* pixel_xy, uint64_t work0, uint32_t n, PathBuffers out,
                         float4* sample_buf, unsigned* ctrl) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) ctrl[0] = n;
    if (i >= n) return;
    uint64_t w = work0 + i;
    uint32_t spp = prm.sampler.spp;
    uint32_t pix = (uint32_t)(w / spp), s = (uint32_t)(w % spp);
    uint32_t xy = pixel_xy[pix];
    uint32_t px = xy & 0xffffu, py = xy >> 16;
    SamplerState st = sampler_start(prm.sampler, px, py, s, 0);
    float ux, uy;
    sampler_get_2d(prm.sampler, st, ux, uy);
    V3 o, d;
    camera_ray(cam, (float)px + ux, (float)py + uy, o, d);
    out.rayO[i] = make_float4(o.x, o.y, o.z, __uint_as_float(0u));
    out.rayD[i] = make_float4(d.x, d.y, d.z, __uint_as_float((uint32_t)w));
    out.thru[i] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(st.dimension));
    out.rngs[i] = make_uint4((unsigned)st.rng.state, (unsigned)(st.rng.state >> 32), (unsigned)st.rng.inc, (unsigned)(st.rng.inc >> 32));
    sample_buf[w] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// Integrator::li entry: caller-supplied rays (yk_li)
__global__ void k_raygen_user(RenderParams prm, const float* ray_o, const float* ray_d, const uint16_t* pixel, const uint32_t* sample_index,
                              uint32_t dimension, uint32_t n, PathBuffers out, float4* sample_buf, uint32_t* pixel_xy, unsigned* ctrl) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) ctrl[0] = n;
    if (i >= n) return;
    uint32_t px = pixel[2 * i], py = pixel[2 * i + 1];
    pixel_xy[i] = px | (py << 16);
    SamplerState st = sampler_start(prm.sampler, px, py, sample_index[i], dimension);
    if (prm.sampler.kind == 1) st.dimension = dimension;  // caller already consumed `dimension` draws
    out.rayO[i] = make_float4(ray_o[3 * i], ray_o[3 * i + 1], ray_o[3 * i + 2], __uint_as_float(0u));
    out.rayD[i] = make_float4(ray_d[3 * i], ray_d[3 * i + 1], ray_d[3 * i + 2], __uint_as_float(i));
    out.thru[i] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(st.dimension));
    out.rngs[i] = make_uint4((unsigned)st.rng.state, (unsigned)(st.rng.state >> 32), (unsigned)st.rng.inc, (unsigned)(st.rng.inc >> 32));
    sample_buf[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// ------------------------------------------------------------------ traversal
// Traversal stack: entries [0, LDS_DEPTH) live in LDS laid out [depth][thread]
// (conflict-free: the bank depends on the lane only), deeper entries overflow to
// a per-thread slice of HBM scratch.  Capacity 64 like the reference (bvh.rs:172).
#define YK_STACK_CAP 64

template <int BLOCK, int LDS_DEPTH> struct TravStack {
    uint2* lds;      // [LDS_DEPTH][BLOCK]
    uint2* spill;    // [YK_STACK_CAP - LDS_DEPTH][spill_stride]
    unsigned spill_stride, gtid;
    __device__ __forceinline__ void push(int sp, unsigned ref, float tmin) {
        uint2 e = make_uint2(ref, __float_as_uint(tmin));
        if (sp < LDS_DEPTH)
            lds[sp * BLOCK + threadIdx.x] = e;
        else
            spill[(size_t)(sp - LDS_DEPTH) * spill_stride + gtid] = e;
    }
    __device__ __forceinline__ uint2 at(int sp) const {
        if (sp < LDS_DEPTH) return lds[sp * BLOCK + threadIdx.x];
        return spill[(size_t)(sp - LDS_DEPTH) * spill_stride + gtid];
    }
};

struct NodeBoxes {
    V3 lo0, hi0, lo1, hi1;
    unsigned ref0, ref1, axis;
};
__device__ __forceinline__ NodeBoxes load_node(const DevNode* nodes, unsigned idx) {
    const float4* q = reinterpret_cast<const float4*>(nodes + idx);
    float4 a = q[0], b = q[1], c = q[2];
    uint4 d = reinterpret_cast<const uint4*>(q)[3];
    NodeBoxes n;
    n.lo0 = V3{a.x, a.y, a.z};
    n.hi0 = V3{a.w, b.x, b.y};
    n.lo1 = V3{b.z, b.w, c.x};
    n.hi1 = V3{c.y, c.z, c.w};
    n.ref0 = d.x;
    n.ref1 = d.y;
    n.axis = d.z;
    return n;
}

// Closest hit with the reference's visiting order (near child first by the sign
// of the direction along the split axis, far child deferred, leaves in shape
// order, a later hit with t == t_max replaces the earlier one).  Box tests of a
// deferred child are evaluated when its parent is visited and completed at pop
// time by `tmin <= t_max`, which is exactly the reference's test at pop time
// because t_max only shrinks (DESIGN.md §traversal equivalence).
template <int BLOCK, int LDS_DEPTH, bool STATS>
__device__ __forceinline__ void traverse_closest(const DevScene& sc, V3 o, V3 d, float t_max_in, TravStack<BLOCK, LDS_DEPTH>& stk, int& out_tri,
                                                 TriHit& out_hit, unsigned& node_tests, unsigned& node_hits, unsigned& shape_tests,
                                                 unsigned* err) {
    V3 inv = V3{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    bool neg[3] = {inv.x < 0.0f, inv.y < 0.0f, inv.z < 0.0f};
    RayTri rt = ray_tri_setup(d);
    float t_max = t_max_in;
    out_tri = -1;
    int sp = 0;
    float tmin;
    if (STATS) node_tests += 1;
    if (!slab(V3{sc.root_bmin[0], sc.root_bmin[1], sc.root_bmin[2]}, V3{sc.root_bmax[0], sc.root_bmax[1], sc.root_bmax[2]}, o, inv, t_max, tmin)) return;
    if (STATS) node_hits += 1;
    unsigned cur = sc.root_ref;
    for (;;) {
        if (!(cur & YK_LEAF_BIT)) {
            NodeBoxes nb = load_node(sc.nodes, cur);
            float t0, t1;
            bool h0 = slab(nb.lo0, nb.hi0, o, inv, t_max, t0);
            bool h1 = slab(nb.lo1, nb.hi1, o, inv, t_max, t1);
            bool swap = neg[nb.axis];
            unsigned near_ref = swap ? nb.ref1 : nb.ref0, far_ref = swap ? nb.ref0 : nb.ref1;
            bool near_hit = swap ? h1 : h0, far_hit = swap ? h0 : h1;
            float far_t = swap ? t0 : t1;
            if (STATS) {
                node_tests += 1;  // the near child is tested right away; the far one is counted when popped
                if (near_hit) node_hits += 1;
            }
            if (STATS || far_hit) {
                // with STATS the far child is pushed even when its box is missed so
                // that the test is counted at pop time like the reference does
                if (sp >= YK_STACK_CAP) {
                    atomicOr(err, 1u);
                    return;
                }
                stk.push(sp, far_ref, far_hit ? far_t : __builtin_nanf(""));
                ++sp;
            }
            if (near_hit) {
                cur = near_ref;
                continue;
            }
        } else {
            unsigned prim = cur & ~YK_LEAF_BIT;
            for (;;) {
                float4 v0 = sc.tris[3 * prim], v1 = sc.tris[3 * prim + 1], v2 = sc.tris[3 * prim + 2];
                TriHit h;
                if (STATS) shape_tests += 1;
                if (tri_intersect(o, rt, t_max, f4_xyz(v0), f4_xyz(v1), f4_xyz(v2), h)) {
                    out_hit = h;
                    out_tri = (int)__float_as_uint(v1.w);
                    t_max = h.t;
                }
                if (__float_as_uint(v2.w) & 1u) break;
                ++prim;
            }
        }
        // pop
        bool found = false;
        while (sp > 0) {
            --sp;
            uint2 e = stk.at(sp);
            float et = __uint_as_float(e.y);
            if (STATS) node_tests += 1;
            if (et <= t_max) {
                if (STATS) node_hits += 1;
                cur = e.x;
                found = true;
                break;
            }
        }
        if (!found) return;
    }
}

// any_intersect: boolean, order independent; t_max is fixed so a deferred
// child's box test is final.
template <int BLOCK, int LDS_DEPTH>
__device__ __forceinline__ bool traverse_any(const DevScene& sc, V3 o, V3 d, float t_max, int area_light, TravStack<BLOCK, LDS_DEPTH>& stk,
                                             unsigned* err) {
    V3 inv = V3{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    bool neg[3] = {inv.x < 0.0f, inv.y < 0.0f, inv.z < 0.0f};
    RayTri rt = ray_tri_setup(d);
    int sp = 0;
    float tmin;
    if (!slab(V3{sc.root_bmin[0], sc.root_bmin[1], sc.root_bmin[2]}, V3{sc.root_bmax[0], sc.root_bmax[1], sc.root_bmax[2]}, o, inv, t_max, tmin)) return false;
    unsigned cur = sc.root_ref;
    for (;;) {
        if (!(cur & YK_LEAF_BIT)) {
            NodeBoxes nb = load_node(sc.nodes, cur);
            float t0, t1;
            bool h0 = slab(nb.lo0, nb.hi0, o, inv, t_max, t0);
            bool h1 = slab(nb.lo1, nb.hi1, o, inv, t_max, t1);
            bool swap = neg[nb.axis];
            unsigned near_ref = swap ? nb.ref1 : nb.ref0, far_ref = swap ? nb.ref0 : nb.ref1;
            bool near_hit = swap ? h1 : h0, far_hit = swap ? h0 : h1;
            if (near_hit) {
                if (far_hit) {
                    if (sp >= YK_STACK_CAP) {
                        atomicOr(err, 1u);
                        return true;
                    }
                    stk.push(sp, far_ref, 0.0f);
                    ++sp;
                }
                cur = near_ref;
                continue;
            }
            if (far_hit) {
                cur = far_ref;
                continue;
            }
        } else {
            unsigned prim = cur & ~YK_LEAF_BIT;
            for (;;) {
                float4 v0 = sc.tris[3 * prim], v1 = sc.tris[3 * prim + 1], v2 = sc.tris[3 * prim + 2];
                TriHit h;
                if (tri_intersect(o, rt, t_max, f4_xyz(v0), f4_xyz(v1), f4_xyz(v2), h)) {
                    // bvh.rs:269-280: a hit on the sampled area light's own surface does not occlude
                    int prim_light = (int)__float_as_uint(v0.w);
                    if (!(area_light >= 0 && prim_light >= 0 && prim_light == area_light)) return true;
                }
                if (__float_as_uint(v2.w) & 1u) break;
                ++prim;
            }
        }
        if (sp == 0) return false;
        --sp;
        cur = stk.at(sp).x;
    }
}

// Persistent waves: each wave pulls 64 consecutive rays from a global head until
// the queue (whose length only the device knows) is drained.
template <int BLOCK, int LDS_DEPTH, bool STATS>
__global__ __launch_bounds__(BLOCK) void k_trace_closest(DevScene sc, const float4* rayO, const float4* rayD, const float* t_max_opt,
                                                         const unsigned* count_ptr, unsigned* head, int* hit_tri, float4* hit_out,
                                                         uint4* stats_out, uint2* spill, unsigned spill_stride, unsigned* ctrl,
                                                         unsigned long long* ray_counter) {
    __shared__ uint2 lds_stack[LDS_DEPTH * BLOCK];
    TravStack<BLOCK, LDS_DEPTH> stk;
    stk.lds = lds_stack;
    stk.spill = spill;
    stk.spill_stride = spill_stride;
    stk.gtid = blockIdx.x * BLOCK + threadIdx.x;
    const unsigned n = *count_ptr;
    if (ray_counter && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(ray_counter, (unsigned long long)n);
    for (;;) {
        unsigned base = 0;
        if (lane_id() == 0) base = atomicAdd(head, YK_WAVE);
        base = __shfl(base, 0);
        if (base >= n) break;
        unsigned i = base + lane_id();
        if (i < n) {
            float4 ro = rayO[i], rd = rayD[i];
            float tm = t_max_opt ? t_max_opt[i] : __builtin_inff();
            int tri;
            TriHit h = TriHit{0.0f, 0.0f, 0.0f, 0.0f};
            unsigned nt = 0, nh = 0, st = 0;
            traverse_closest<BLOCK, LDS_DEPTH, STATS>(sc, f4_xyz(ro), f4_xyz(rd), tm, stk, tri, h, nt, nh, st, ctrl + YK_CTRL_ERR);
            hit_tri[i] = tri;
            if (hit_out) hit_out[i] = make_float4(h.t, h.b0, h.b1, h.b2);
            if (STATS) stats_out[i] = make_uint4(nt, nh, st, 0u);
        }
    }
}

template <int BLOCK, int LDS_DEPTH>
__global__ __launch_bounds__(BLOCK) void k_trace_any(DevScene sc, const float4* shO, const float4* shD, const unsigned* queue,
                                                     const unsigned* count_ptr, unsigned* head, unsigned char* vis, uint2* spill,
                                                     unsigned spill_stride, unsigned* ctrl, unsigned long long* shadow_counter) {
    __shared__ uint2 lds_stack[LDS_DEPTH * BLOCK];
    TravStack<BLOCK, LDS_DEPTH> stk;
    stk.lds = lds_stack;
    stk.spill = spill;
    stk.spill_stride = spill_stride;
    stk.gtid = blockIdx.x * BLOCK + threadIdx.x;
    const unsigned n = *count_ptr;
    if (shadow_counter && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(shadow_counter, (unsigned long long)n);
    for (;;) {
        unsigned base = 0;
        if (lane_id() == 0) base = atomicAdd(head, YK_WAVE);
        base = __shfl(base, 0);
        if (base >= n) break;
        unsigned k = base + lane_id();
        if (k < n) {
            unsigned slot = queue ? queue[k] : k;
            float4 so = shO[slot], sd = shD[slot];
            bool occluded = traverse_any<BLOCK, LDS_DEPTH>(sc, f4_xyz(so), f4_xyz(sd), so.w, (int)__float_as_uint(sd.w), stk, ctrl + YK_CTRL_ERR);
            if (queue) {
                if (occluded) vis[slot] = 2;
            } else {
                vis[slot] = occluded ? 1 : 0;
            }
        }
    }
}

// ------------------------------------------------------------------ shade
// Path::li_internal for one vertex of every active path (path.rs:89-169).
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_shade(DevScene sc, RenderParams prm, const uint32_t* pixel_xy, const uint32_t* sample_index_tab,
                                                 PathBuffers cur, PathBuffers nxt,
                                                 const int* hit_tri, float4* pend, float4* shO, float4* shD, float4* shC, unsigned char* vis,
                                                 unsigned* shq, unsigned* ctrl, unsigned cur_slot) {
    const unsigned n = ctrl[cur_slot];
    const unsigned nl = sc.n_lights;
    unsigned* next_count = ctrl + (cur_slot ^ 1u);
    unsigned* shq_count = ctrl + YK_CTRL_SHQ;
    // all lanes stay in the loop together so the ballots below see whole waves
    const unsigned n_round = (n + BLOCK - 1) / BLOCK * BLOCK;
    for (unsigned i = blockIdx.x * BLOCK + threadIdx.x; i < n_round; i += gridDim.x * BLOCK) {
        const bool valid = i < n;
        bool alive = false;
        float4 nO = make_float4(0, 0, 0, 0), nD = nO, nT = nO;
        uint4 nR = make_uint4(0, 0, 0, 0);
        // per-light scratch lives in registers only for the current light
        V3 o = V3{0, 0, 0}, d = V3{0, 0, 1};
        RGB beta = RGB{0, 0, 0};
        unsigned flags = 0, sid = 0, bounces = 0;
        bool specular_bounce = false, hit = false;
        SamplerState st;
        st.rng.state = 0;
        st.rng.inc = 1;
        st.px = st.py = st.sample_index = st.dimension = 0;
        Surface sf;
        sf.p = sf.n = sf.ns = sf.dpdus = V3{0, 0, 1};
        sf.material = 0;
        sf.area_light = -1;
        Material mat;
        mat.kind = MK_BLACK;
        Frame fr;
        fr.s = fr.t = fr.n = fr.ng = V3{0, 0, 1};
        V3 wo = V3{0, 0, 1};
        if (valid) {
            float4 a = cur.rayO[i], b = cur.rayD[i], c = cur.thru[i];
            uint4 r = cur.rngs[i];
            o = f4_xyz(a);
            d = f4_xyz(b);
            flags = __float_as_uint(a.w);
            sid = __float_as_uint(b.w);
            beta = RGB{c.x, c.y, c.z};
            bounces = flags & 0xffu;
            specular_bounce = (flags >> 8) & 1u;
            st.rng.state = (u64)r.x | ((u64)r.y << 32);
            st.rng.inc = (u64)r.z | ((u64)r.w << 32);
            st.dimension = __float_as_uint(c.w);
            // film renders: sample_id = pixel*spp + sample ; yk_li: one table entry per ray
            uint32_t xy = pixel_xy[sample_index_tab ? sid : sid / prm.sampler.spp];
            st.px = xy & 0xffffu;
            st.py = xy >> 16;
            st.sample_index = sample_index_tab ? sample_index_tab[sid] : sid % prm.sampler.spp;
            int tri = hit_tri[i];
            hit = tri >= 0;
            if (hit) {
                // recompute the accepted intersection: same ray, same triangle, same
                // arithmetic -> same (t, b0, b1, b2) as inside the traversal
                uint32_t i0 = sc.indices[3 * tri], i1 = sc.indices[3 * tri + 1], i2 = sc.indices[3 * tri + 2];
                RayTri rt = ray_tri_setup(d);
                TriHit th;
                tri_intersect(o, rt, __builtin_inff(), ld3(sc.points, i0), ld3(sc.points, i1), ld3(sc.points, i2), th);
                sf = make_surface(sc, (uint32_t)tri, th);
                mat = sc.materials[sf.material];
                fr = make_frame(sf.n, sf.ns, sf.dpdus);
                wo = -d;
            }
        }
        // ---- next-event estimation over ALL lights (path.rs:102-119); two sampler
        // dimensions are consumed per light whether or not it contributes.
        for (unsigned l = 0; l < nl; ++l) {
            bool want = false;
            unsigned slot = i * nl + l;
            RGB contrib = RGB{0, 0, 0};
            V3 so = V3{0, 0, 0}, sd = V3{0, 0, 1};
            int al = -1;
            if (valid && hit) {
                float ux, uy;
                sampler_get_2d(prm.sampler, st, ux, uy);
                LightSample ls = sample_light(sc.lights[l], (int)l, sf.p, ux, uy);
                if (!is_black(ls.li)) {
                    RGB f = bsdf_f(mat, fr, wo, ls.l);
                    if (ls.has_vis && !is_black(f)) {
                        contrib = f * ls.li * rclamp(dot_nv(sf.ns, ls.l), 0.0f, 1.0f) / ls.pdf;
                        // VisibilityTester::ray = p0.spawn_ray_to(p1), interaction.rs:44-59
                        V3 offset = sf.n * 0.001f;
                        so = dot(ls.p1 - sf.p, sf.n) > 0.0f ? sf.p + offset : sf.p - offset;
                        sd = ls.p1 - so;
                        al = ls.area_light;
                        want = true;
                    }
                }
            }
            if (valid) vis[slot] = want ? 1 : 0;
            unsigned q = wave_append(want, shq_count);
            if (want) {
                shO[slot] = make_float4(so.x, so.y, so.z, 0.9999f);
                shD[slot] = make_float4(sd.x, sd.y, sd.z, __uint_as_float((unsigned)al));
                shC[slot] = make_float4(contrib.r, contrib.g, contrib.b, 0.0f);
                shq[q] = slot;
            }
        }
        if (valid) {
            // kind bits: 1 = miss, 2 = emission term present, 4 = indirect clamp applies
            unsigned kind = 0;
            RGB term = RGB{0, 0, 0};
            if (!hit) {
                // path.rs:155-160: incoming_radiance += beta * scene.background; break
                term = beta * RGB{sc.background[0], sc.background[1], sc.background[2]};
                kind = 1;
            } else {
                if (bounces == 0 || specular_bounce) {  // path.rs:121-123
                    RGB le = RGB{0, 0, 0};
                    if (sf.area_light >= 0) {
                        const DevLight& L = sc.lights[sf.area_light];
                        le = dot_nv(sf.n, wo) > 0.0f ? RGB{L.i[0], L.i[1], L.i[2]} : RGB{0, 0, 0};  // rectangular_light.rs:75-81
                    }
                    term = beta * le;
                    kind |= 2;
                }
                if (bounces > 0 && prm.has_clamp) kind |= 4;
                // path.rs:131-145
                float ux, uy;
                sampler_get_2d(prm.sampler, st, ux, uy);
                BsdfSample bs = bsdf_sample_f(mat, fr, wo, ux, uy);
                if (!(is_black(bs.f) || bs.pdf == 0.0f)) {
                    specular_bounce = (bs.type & BX_SPECULAR) != 0;
                    beta = beta * (bs.f * fabsf(dot_nv(bs.wi, sf.ns)) / bs.pdf);
                    V3 no = spawn_origin(sf.p, sf.n, bs.wi);
                    alive = true;
                    // Russian roulette, path.rs:162-169
                    if (bounces > 3) {
                        float q = rmax(1.0f - beta.g, 0.05f);
                        if (sampler_get_1d(prm.sampler, st) < q)
                            alive = false;
                        else
                            beta = beta * (RGB{1.0f, 1.0f, 1.0f} / (1.0f - q));
                    }
                    bounces += 1;
                    if (!(bounces < prm.max_depth)) alive = false;  // while bounces < max_depth
                    nO = make_float4(no.x, no.y, no.z, __uint_as_float((bounces & 0xffu) | (specular_bounce ? 0x100u : 0u)));
                    nD = make_float4(bs.wi.x, bs.wi.y, bs.wi.z, __uint_as_float(sid));
                    nT = make_float4(beta.r, beta.g, beta.b, __uint_as_float(st.dimension));
                    nR = make_uint4((unsigned)st.rng.state, (unsigned)(st.rng.state >> 32), (unsigned)st.rng.inc, (unsigned)(st.rng.inc >> 32));
                }
            }
            pend[i] = make_float4(term.r, term.g, term.b, __uint_as_float(kind));
        }
        // ---- stream compaction of the survivors into the other buffer
        unsigned j = wave_append(alive, next_count);
        if (alive) {
            nxt.rayO[j] = nO;
            nxt.rayD[j] = nD;
            nxt.thru[j] = nT;
            nxt.rngs[j] = nR;
        }
    }
}

// ------------------------------------------------------------------ accumulate
// radiance = fold over lights (in light order) of the unoccluded contributions,
// + beta*Le, clamp, then incoming_radiance += beta * radiance   (path.rs:102-129)
__global__ void k_accumulate(RenderParams prm, PathBuffers cur, const float4* pend, const float4* shC, const unsigned char* vis, unsigned nl,
                             float4* sample_buf, const unsigned* ctrl, unsigned cur_slot) {
    const unsigned n = ctrl[cur_slot];
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 p = pend[i];
        unsigned kind = __float_as_uint(p.w);
        unsigned sid = __float_as_uint(cur.rayD[i].w);
        float4 c = cur.thru[i];
        RGB beta = RGB{c.x, c.y, c.z};
        float4 acc = sample_buf[sid];
        RGB L = RGB{acc.x, acc.y, acc.z};
        if (kind & 1u) {
            L = L + RGB{p.x, p.y, p.z};
        } else {
            RGB radiance = RGB{0.0f, 0.0f, 0.0f};
            for (unsigned l = 0; l < nl; ++l) {
                unsigned slot = i * nl + l;
                if (vis[slot] == 1) {
                    float4 ct = shC[slot];
                    radiance = radiance + RGB{ct.x, ct.y, ct.z};
                }
            }
            if (kind & 2u) radiance = radiance + RGB{p.x, p.y, p.z};
            if (kind & 4u) radiance = rgb_min(radiance, RGB{1.0f, 1.0f, 1.0f} * prm.clamp);
            L = L + beta * radiance;
        }
        sample_buf[sid] = make_float4(L.r, L.g, L.b, 0.0f);
    }
}

// ------------------------------------------------------------------ resolve
// color = sum of the pixel's samples in sample order; color /= spp
// (integrators/mod.rs:172-175); output tile-major like tile_pixels.
__global__ void k_resolve(const float4* sample_buf, uint32_t n_pixels, uint32_t spp, float* out_rgb) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pixels) return;
    RGB color = RGB{0.0f, 0.0f, 0.0f};
    const float4* s = sample_buf + (size_t)p * spp;
    for (uint32_t k = 0; k < spp; ++k) {
        float4 v = s[k];
        color = color + RGB{v.x, v.y, v.z};
    }
    color = color / (float)spp;
    out_rgb[3 * (size_t)p + 0] = color.r;
    out_rgb[3 * (size_t)p + 1] = color.g;
    out_rgb[3 * (size_t)p + 2] = color.b;
}

// Film::update_tile on the device (film.rs:236-278): tile-major -> row-major film
__global__ void k_film_scatter(const uint32_t* pixel_xy, uint32_t n_pixels, const float* tile_rgb, uint32_t res_x, float* film_rgb) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pixels) return;
    uint32_t xy = pixel_xy[p];
    size_t dst = (size_t)(xy >> 16) * res_x + (xy & 0xffffu);
    film_rgb[3 * dst + 0] = tile_rgb[3 * (size_t)p + 0];
    film_rgb[3 * dst + 1] = tile_rgb[3 * (size_t)p + 1];
    film_rgb[3 * dst + 2] = tile_rgb[3 * (size_t)p + 2];
}

// ------------------------------------------------------------------ debug integrators
// BVHIntersections / GeometryNormals / ShadingNormals::li (bvh_heatmap.rs:25-40,
// geometry_normals.rs:24-33, shading_normals.rs)
__global__ void k_debug_shade(DevScene sc, uint32_t integrator, PathBuffers cur, const int* hit_tri, const uint4* stats, uint32_t n,
                              float4* sample_buf) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned sid = __float_as_uint(cur.rayD[i].w);
    int tri = hit_tri[i];
    RGB c = RGB{0.0f, 0.0f, 0.0f};
    if (integrator == YK_INTEGRATOR_BVH_INTERSECTIONS) {
        uint4 s = stats[i];
        c = RGB{(float)s.x, (float)s.y, tri >= 0 ? (float)s.y : 0.0f};
    } else if (tri >= 0) {
        V3 o = f4_xyz(cur.rayO[i]), d = f4_xyz(cur.rayD[i]);
        uint32_t i0 = sc.indices[3 * tri], i1 = sc.indices[3 * tri + 1], i2 = sc.indices[3 * tri + 2];
        RayTri rt = ray_tri_setup(d);
        TriHit th;
        tri_intersect(o, rt, __builtin_inff(), ld3(sc.points, i0), ld3(sc.points, i1), ld3(sc.points, i2), th);
        Surface sf = make_surface(sc, (uint32_t)tri, th);
        V3 nn = integrator == YK_INTEGRATOR_GEOMETRY_NORMALS ? sf.n : sf.ns;
        c = RGB{nn.x, nn.y, nn.z} / 2.0f + 0.5f;
    }
    sample_buf[sid] = make_float4(c.r, c.g, c.b, 0.0f);
}

// ------------------------------------------------------------------ unit-test kernels
__global__ void k_device_math(int fn, size_t n, const float* a, const float* b, float* out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b ? b[i] : 0.0f, r;
    switch (fn) {
        case 0: r = det_sinf(x); break;
        case 1: r = det_cosf(x); break;
        case 2: r = det_tanf(x); break;
        case 3: r = det_logf(x); break;
        case 4: r = det_acosf(x); break;
        case 5: r = det_atan2f(x, y); break;
        case 6: r = sqrtf(x); break;
        case 7: r = x / y; break;
        case 8: r = (float)sqrt((double)x); break;
        case 9: r = rmin(x, y); break;
        case 10: r = rmax(x, y); break;
        default: r = 0.0f;
    }
    out[i] = r;
}

__global__ void k_sampler_sequence(SamplerCfg cfg, uint32_t px, uint32_t py, uint32_t sample_index, const uint8_t* dims, size_t n_draws, float* out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    SamplerState st = sampler_start(cfg, px, py, sample_index, 0);
    for (size_t k = 0; k < n_draws; ++k) {
        if (dims[k] == 1) {
            out[2 * k] = sampler_get_1d(cfg, st);
            out[2 * k + 1] = 0.0f;
        } else {
            float ux, uy;
            sampler_get_2d(cfg, st, ux, uy);
            out[2 * k] = ux;
            out[2 * k + 1] = uy;
        }
    }
}

__global__ void k_bsdf_test(Material m, size_t n, const float* ng, const float* ns, const float* dpdu, const float* wo, const float* wi_or_u,
                            int sample, float* out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    Frame fr = make_frame(ld3(ng, (uint32_t)i), ld3(ns, (uint32_t)i), ld3(dpdu, (uint32_t)i));
    V3 w = ld3(wo, (uint32_t)i);
    if (!sample) {
        RGB f = bsdf_f(m, fr, w, ld3(wi_or_u, (uint32_t)i));
        out[3 * i] = f.r;
        out[3 * i + 1] = f.g;
        out[3 * i + 2] = f.b;
    } else {
        BsdfSample s = bsdf_sample_f(m, fr, w, wi_or_u[2 * i], wi_or_u[2 * i + 1]);
        out[8 * i + 0] = s.wi.x;
        out[8 * i + 1] = s.wi.y;
        out[8 * i + 2] = s.wi.z;
        out[8 * i + 3] = s.f.r;
        out[8 * i + 4] = s.f.g;
        out[8 * i + 5] = s.f.b;
        out[8 * i + 6] = s.pdf;
        out[8 * i + 7] = (float)s.type;
    }
}

__global__ void k_pack_rays(size_t n, const float* o, const float* d, float4* rayO, float4* rayD) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    rayO[i] = make_float4(o[3 * i], o[3 * i + 1], o[3 * i + 2], 0.0f);
    rayD[i] = make_float4(d[3 * i], d[3 * i + 1], d[3 * i + 2], 0.0f);
}
__global__ void k_pack_shadow_rays(size_t n, const float* o, const float* d, const float* t_max, const int* area_light, float4* shO, float4* shD) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    shO[i] = make_float4(o[3 * i], o[3 * i + 1], o[3 * i + 2], t_max[i]);
    shD[i] = make_float4(d[3 * i], d[3 * i + 1], d[3 * i + 2], __uint_as_float((unsigned)(area_light ? area_light[i] : -1)));
}
__global__ void k_unpack_rays(size_t n, const float4* rayO, const float4* rayD, float* o, float* d) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 a = rayO[i], b = rayD[i];
    o[3 * i] = a.x; o[3 * i + 1] = a.y; o[3 * i + 2] = a.z;
    d[3 * i] = b.x; d[3 * i + 1] = b.y; d[3 * i + 2] = b.z;
}

// ------------------------------------------------------------------ launchers
static inline unsigned blocks_for(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

#define TRACE_BLOCK 256
#define TRACE_LDS 16

unsigned trace_block_size() { return TRACE_BLOCK; }
unsigned trace_spill_depth() { return YK_STACK_CAP - TRACE_LDS; }

void launch_pixel_table(hipStream_t s, const yk_tile* tiles, const uint32_t* tile_offset, uint32_t n_tiles, uint32_t n_pixels, uint32_t* pixel_xy) {
    if (!n_pixels) return;
    hipLaunchKernelGGL(k_pixel_table, dim3(blocks_for(n_pixels, 256)), dim3(256), 0, s, tiles, tile_offset, n_tiles, n_pixels, pixel_xy);
}
void launch_raygen(hipStream_t s, const DevCamera& cam, const RenderParams& prm, const uint32_t* pixel_xy, uint64_t work0, uint32_t n,
                   PathBuffers out, float4* sample_buf, unsigned* ctrl) {
    hipLaunchKernelGGL(k_raygen, dim3(blocks_for(n, 256)), dim3(256), 0, s, cam, prm, pixel_xy, work0, n, out, sample_buf, ctrl);
}
void launch_raygen_user(hipStream_t s, const RenderParams& prm, const float* o, const float* d, const uint16_t* pixel, const uint32_t* sample_index,
                        uint32_t dimension, uint32_t n, PathBuffers out, float4* sample_buf, uint32_t* pixel_xy, unsigned* ctrl) {
    hipLaunchKernelGGL(k_raygen_user, dim3(blocks_for(n, 256)), dim3(256), 0, s, prm, o, d, pixel, sample_index, dimension, n, out, sample_buf,
                       pixel_xy, ctrl);
}
void launch_trace_closest(hipStream_t s, unsigned grid, const DevScene& sc, const float4* rayO, const float4* rayD, const float* t_max_opt,
                          const unsigned* count_ptr, unsigned* head, int* hit_tri, float4* hit_out, uint4* stats_out, uint2* spill,
                          unsigned spill_stride, unsigned* ctrl, unsigned long long* ray_counter) {
    if (stats_out)
        hipLaunchKernelGGL((k_trace_closest<TRACE_BLOCK, TRACE_LDS, true>), dim3(grid), dim3(TRACE_BLOCK), 0, s, sc, rayO, rayD, t_max_opt, count_ptr,
                           head, hit_tri, hit_out, stats_out, spill, spill_stride, ctrl, ray_counter);
    else
        hipLaunchKernelGGL((k_trace_closest<TRACE_BLOCK, TRACE_LDS, false>), dim3(grid), dim3(TRACE_BLOCK), 0, s, sc, rayO, rayD, t_max_opt, count_ptr,
                           head, hit_tri, hit_out, stats_out, spill, spill_stride, ctrl, ray_counter);
}
void launch_trace_any(hipStream_t s, unsigned grid, const DevScene& sc, const float4* shO, const float4* shD, const unsigned* queue,
                      const unsigned* count_ptr, unsigned* head, unsigned char* vis, uint2* spill, unsigned spill_stride, unsigned* ctrl,
                      unsigned long long* shadow_counter) {
    hipLaunchKernelGGL((k_trace_any<TRACE_BLOCK, TRACE_LDS>), dim3(grid), dim3(TRACE_BLOCK), 0, s, sc, shO, shD, queue, count_ptr, head, vis, spill,
                       spill_stride, ctrl, shadow_counter);
}
void launch_shade(hipStream_t s, unsigned grid, const DevScene& sc, const RenderParams& prm, const uint32_t* pixel_xy, const uint32_t* sample_index_tab,
                  PathBuffers cur, PathBuffers nxt,
                  const int* hit_tri, float4* pend, float4* shO, float4* shD, float4* shC, unsigned char* vis, unsigned* shq, unsigned* ctrl,
                  unsigned cur_slot) {
    hipLaunchKernelGGL((k_shade<256>), dim3(grid), dim3(256), 0, s, sc, prm, pixel_xy, sample_index_tab, cur, nxt, hit_tri, pend, shO, shD, shC, vis, shq, ctrl, cur_slot);
}
void launch_accumulate(hipStream_t s, unsigned grid, const RenderParams& prm, PathBuffers cur, const float4* pend, const float4* shC,
                       const unsigned char* vis, unsigned nl, float4* sample_buf, const unsigned* ctrl, unsigned cur_slot) {
    hipLaunchKernelGGL(k_accumulate, dim3(grid), dim3(256), 0, s, prm, cur, pend, shC, vis, nl, sample_buf, ctrl, cur_slot);
}
void launch_resolve(hipStream_t s, const float4* sample_buf, uint32_t n_pixels, uint32_t spp, float* out_rgb) {
    if (!n_pixels) return;
    hipLaunchKernelGGL(k_resolve, dim3(blocks_for(n_pixels, 256)), dim3(256), 0, s, sample_buf, n_pixels, spp, out_rgb);
}
void launch_film_scatter(hipStream_t s, const uint32_t* pixel_xy, uint32_t n_pixels, const float* tile_rgb, uint32_t res_x, float* film_rgb) {
    if (!n_pixels) return;
    hipLaunchKernelGGL(k_film_scatter, dim3(blocks_for(n_pixels, 256)), dim3(256), 0, s, pixel_xy, n_pixels, tile_rgb, res_x, film_rgb);
}
void launch_debug_shade(hipStream_t s, const DevScene& sc, uint32_t integrator, PathBuffers cur, const int* hit_tri, const uint4* stats, uint32_t n,
                        float4* sample_buf) {
    hipLaunchKernelGGL(k_debug_shade, dim3(blocks_for(n, 256)), dim3(256), 0, s, sc, integrator, cur, hit_tri, stats, n, sample_buf);
}
void launch_device_math(hipStream_t s, int fn, size_t n, const float* a, const float* b, float* out) {
    hipLaunchKernelGGL(k_device_math, dim3(blocks_for(n, 256)), dim3(256), 0, s, fn, n, a, b, out);
}
void launch_sampler_sequence(hipStream_t s, const SamplerCfg& cfg, uint32_t px, uint32_t py, uint32_t sample_index, const uint8_t* dims, size_t n_draws,
                             float* out) {
    hipLaunchKernelGGL(k_sampler_sequence, dim3(1), dim3(64), 0, s, cfg, px, py, sample_index, dims, n_draws, out);
}
void launch_bsdf_test(hipStream_t s, const Material& m, size_t n, const float* ng, const float* ns, const float* dpdu, const float* wo,
                      const float* wi_or_u, int sample, float* out) {
    hipLaunchKernelGGL(k_bsdf_test, dim3(blocks_for(n, 256)), dim3(256), 0, s, m, n, ng, ns, dpdu, wo, wi_or_u, sample, out);
}
void launch_pack_rays(hipStream_t s, size_t n, const float* o, const float* d, float4* rayO, float4* rayD) {
    hipLaunchKernelGGL(k_pack_rays, dim3(blocks_for(n, 256)), dim3(256), 0, s, n, o, d, rayO, rayD);
}
void launch_pack_shadow_rays(hipStream_t s, size_t n, const float* o, const float* d, const float* t_max, const int* area_light, float4* shO,
                             float4* shD) {
    hipLaunchKernelGGL(k_pack_shadow_rays, dim3(blocks_for(n, 256)), dim3(256), 0, s, n, o, d, t_max, area_light, shO, shD);
}
void launch_unpack_rays(hipStream_t s, size_t n, const float4* rayO, const float4* rayD, float* o, float* d) {
    hipLaunchKernelGGL(k_unpack_rays, dim3(blocks_for(n, 256)), dim3(256), 0, s, n, rayO, rayD, o, d);
}

}  // namespace yk
