// yk_packet.hip — wave-packet BVH traversal for COHERENT rays (camera rays: the 64 lanes of
// a wave are consecutive samples, i.e. one pixel at 64 spp; and the shadow rays they spawn).
//
// The generic kernels (yk_trace.hip) let every lane walk the tree on its own, which costs four
// 16-byte gathers per lane and node — the CU's vector-memory path is their bound.
// Here the wave walks the tree ONCE: the current node / triangle is wave-uniform and arrives
// through the scalar cache (s_load), each lane only tests its own ray against it, and a
// 64-bit lane mask records which rays are inside the subtree.
//
// Per-ray equivalence with BoundingVolumeHierarchy::intersect (bvh.rs:160-232):
//   * a lane takes part in a node only if its own box test of that node passed, so the
//     leaves a ray is tested against are exactly those of its own traversal;
//   * the reference orders the two children by the sign of the ray direction along the split
//     axis; the lanes of a packet are first split into groups of equal direction signs and
//     every group is traversed separately, so the wave-uniform order is each ray's own order
//     (camera rays through one pixel almost always form a single group);
//   * the far child is pushed untested by the reference and tested when popped with the
//     then-current t_max: the packet stack keeps (parent, which child, lanes that visited the
//     parent and hit the child's box at push time) and re-tests the child's box per lane at
//     pop time — lanes are remembered by a relaxed bound at push time (t_max can rise by ulps on tie hits, yk_geom.h);
//   * leaf primitives are tested in order and a later hit with t == t_max replaces the
//     earlier one (triangle.rs:126-127), per lane as in the scalar code.
// any_intersect (bvh.rs:235-302) is order-independent: one group, lanes retire when occluded.
#include <hip/hip_runtime.h>

#include "yk_device.h"
#include "yk_geom.h"
#include "yk_kernels.h"
#include "yk_wave.h"

namespace yk {

#define YK_PKT_STACK 64  // one entry per tree level at most; packets are used only when depth <= 64
#define YK_PKT_BLOCK 256

struct PktNode {
    V3 lo0, hi0, lo1, hi1;
    unsigned ref0, ref1, axis;
};
// Scene data never changes during a launch: reading it through the constant address space
// tells the compiler so, and a wave-uniform address then becomes a scalar load (s_load_*,
// served by the scalar cache) instead of 64 identical vector requests.
typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) f4v cf4;
typedef const __attribute__((address_space(4))) u2v cu2;
__device__ __forceinline__ cf4* as_const(const float4* p) { return (cf4*)(unsigned long long)p; }
__device__ __forceinline__ float4 ldc(cf4* p, int i) {
    const f4v v = p[i];
    return make_float4(v.x, v.y, v.z, v.w);
}

// `idx` is wave-uniform
__device__ __forceinline__ PktNode pkt_load_node(const DevNode* nodes, unsigned idx) {
    cf4* q = as_const(reinterpret_cast<const float4*>(nodes + idx));
    const float4 a = ldc(q, 0), b = ldc(q, 1), c = ldc(q, 2);
    const u2v dv = ((cu2*)q)[6];
    const uint2 d = make_uint2(dv.x, dv.y);
    PktNode n;
    n.lo0 = V3{a.x, a.y, a.z};
    n.hi0 = V3{a.w, b.x, b.y};
    n.lo1 = V3{b.z, b.w, c.x};
    n.hi1 = V3{c.y, c.z, c.w};
    n.ref0 = d.x;
    n.ref1 = d.y & ~YK_AXIS_MASK;
    n.axis = (d.y >> YK_AXIS_SHIFT) & 3u;
    return n;
}

__device__ __forceinline__ unsigned uni(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ unsigned long long uni64(unsigned long long v) { return (unsigned long long)uni((unsigned)v) | ((unsigned long long)uni((unsigned)(v >> 32)) << 32); }
__device__ __forceinline__ bool in_mask(unsigned long long m) { return (m >> lane_id()) & 1ull; }

// wave-uniform stack in LDS: entry = (ref, which | reserved, mask)
struct PktStack {
    uint4* base;  // this wave's YK_PKT_STACK entries
    __device__ __forceinline__ void push(int sp, unsigned ref, unsigned which, unsigned long long mask) {
        if (lane_id() == 0) base[sp] = make_uint4(ref, which, (unsigned)mask, (unsigned)(mask >> 32));
    }
    __device__ __forceinline__ void at(int sp, unsigned& ref, unsigned& which, unsigned long long& mask) const {
        const uint4 e = base[sp];
        ref = uni(e.x);
        which = uni(e.y);
        mask = (unsigned long long)uni(e.z) | ((unsigned long long)uni(e.w) << 32);
    }
};

// rays claimed per atomic: up to YK_PKT_CHUNK packets, fewer when the queue is short so that
// every resident wave makes about eight claims (two claims per wave left the last waves of a
// 16 M-ray launch with 50 % more work than the rest)
#ifndef YK_PKT_CHUNK
#define YK_PKT_CHUNK 16  // packets per claim at most (sweep 4 / 8 / 16 / 32 / 64)
#endif
__device__ __forceinline__ unsigned pkt_claim_size(unsigned n) {
    const unsigned waves = gridDim.x * (blockDim.x / YK_WAVE);
    unsigned packets = n / (waves * YK_WAVE * 8u);
    packets = packets < 1u ? 1u : (packets > (unsigned)YK_PKT_CHUNK ? (unsigned)YK_PKT_CHUNK : packets);
    return packets * YK_WAVE;
}

struct PktRay {
    V3 o, inv, d;
    RayTri rt;
    float t_max;
};

// one group of lanes with equal direction signs `sg`; updates best / r.t_max of its lanes
template <bool SPHERES>
__device__ __forceinline__ void pkt_closest_group(const DevScene& sc, PktStack& stk, PktRay& r, int& best, unsigned long long group, unsigned sg) {
    unsigned cur = sc.root_ref;
    unsigned long long cmask = group;
    int sp = 0;
    for (;;) {
        bool need_pop = false;
        if (cur & YK_LEAF_BIT) {
            unsigned prim = cur & ~YK_LEAF_BIT;
            const bool mine = in_mask(cmask);
            for (;;) {
                cf4* tq = as_const(sc.tris + 3 * prim);
                const float4 v0 = ldc(tq, 0), v1 = ldc(tq, 1), v2 = ldc(tq, 2);
                const unsigned pflags = __float_as_uint(v2.w);
                if (mine) {
                    TriHit h = TriHit{0.0f, 0.0f, 0.0f, 0.0f};
                    bool got;
                    if (SPHERES && (pflags & YK_PRIM_SPHERE)) {
                        V3 ro, rd;
                        got = sphere_hit_t(sc.spheres[__float_as_uint(v1.w) - sc.n_triangles], r.o, r.d, r.t_max, h.t, ro, rd);
                    } else {
                        got = tri_intersect(r.o, r.rt, r.t_max, f4_xyz(v0), f4_xyz(v1), f4_xyz(v2), h);
                    }
                    if (got) {
                        best = YK_HIT_WORD(prim, pflags);  // leaf-order slot | BSDF kind, like the render-loop flavour of k_trace_closest_pt
                        r.t_max = h.t;
                    }
                }
                if (pflags & YK_PRIM_LAST) break;
                ++prim;
            }
            need_pop = true;
        } else {
            const PktNode nb = pkt_load_node(sc.nodes, cur);
            const bool mine = in_mask(cmask);
            float t0, t1;
            // exact bound for a child entered now; a deferred (far) child is re-tested exactly when popped, so the
            // lanes remembered for it are screened with the relaxed bound (yk_geom.h: t_max can rise by ulps)
            const float t_def = deferred_t_max(r.t_max);
            const bool r0 = mine && slab(nb.lo0, nb.hi0, r.o, r.inv, t_def, t0);
            const bool r1 = mine && slab(nb.lo1, nb.hi1, r.o, r.inv, t_def, t1);
            const bool h0 = r0 && t0 <= r.t_max, h1 = r1 && t1 <= r.t_max;
            const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1);
            const bool swap = (sg >> nb.axis) & 1u;
            const unsigned near_ref = swap ? nb.ref1 : nb.ref0, far_ref = swap ? nb.ref0 : nb.ref1;
            const unsigned long long m_near = swap ? m1 : m0, m_far = swap ? m0 : m1;
            const unsigned long long m_far_def = __ballot(swap ? r0 : r1);
            if (m_near) {
                if (m_far_def) {
                    stk.push(sp, cur, swap ? 0u : 1u, m_far_def);  // which = index of the far child in the parent
                    ++sp;
                }
                cur = near_ref;
                cmask = m_near;
            } else if (m_far) {
                cur = far_ref;  // no leaf in between: the test is final
                cmask = m_far;
            } else {
                need_pop = true;
            }
        }
        if (need_pop) {
            for (;;) {
                if (sp == 0) return;
                --sp;
                unsigned parent, which;
                unsigned long long mask;
                stk.at(sp, parent, which, mask);
                const PktNode nb = pkt_load_node(sc.nodes, parent);
                float t;
                const bool h = in_mask(mask) && (which ? slab(nb.lo1, nb.hi1, r.o, r.inv, r.t_max, t) : slab(nb.lo0, nb.hi0, r.o, r.inv, r.t_max, t));
                const unsigned long long m = __ballot(h);
                if (m) {
                    cur = which ? nb.ref1 : nb.ref0;
                    cmask = m;
                    break;
                }
            }
        }
    }
}

// Camera rays of one batch, 64 consecutive rays per packet.  hit_tri[i] = primitive slot (leaf order) or -1.
template <bool SPHERES>
__global__ __launch_bounds__(YK_PKT_BLOCK, 8) void k_trace_closest_packet(DevScene sc, const float4* __restrict__ rayO, const float4* __restrict__ rayD,
                                                                          const unsigned* count_ptr, unsigned* head, int* __restrict__ hit_tri,
                                                                          unsigned long long* ray_counter, const float4* __restrict__ lean_origin, CancelRef cancel) {
    __shared__ uint4 lds_stack[(YK_PKT_BLOCK / YK_WAVE) * YK_PKT_STACK];
    PktStack stk;
    stk.base = lds_stack + (threadIdx.x / YK_WAVE) * YK_PKT_STACK;
    const unsigned n = cancel_raised(cancel) ? 0u : *count_ptr;
    if (ray_counter && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(ray_counter, (unsigned long long)n);
    const V3 root_lo = V3{sc.root_bmin[0], sc.root_bmin[1], sc.root_bmin[2]}, root_hi = V3{sc.root_bmax[0], sc.root_bmax[1], sc.root_bmax[2]};
    // one atomic per YK_PKT_CHUNK packets: a per-packet atomic on the single head word would
    // cap the launch at ~100 M atomics/s (2 M packets = 20 ms)
    const unsigned per_claim = pkt_claim_size(n);
    for (;;) {
        unsigned claim = 0;
        if (lane_id() == 0) {
            claim = atomicAdd(head, per_claim);
            if (blockIdx.x == 0 && threadIdx.x == 0 && cancel_relay(cancel, head)) claim = 0xffffffffu;  // interrupted (yk_device.h): the head is poisoned, nobody claims again
        }
        claim = uni(claim);
        if (claim >= n) break;
        const unsigned claim_end = claim + per_claim < n ? claim + per_claim : n;
    for (unsigned base = claim; base < claim_end; base += YK_WAVE) {
        const unsigned idx = base + lane_id();
        const bool valid = idx < n;
        PktRay r;
        unsigned negmask = 0;
        bool alive = false;
        int best = -1;
        {
            // rayO == null: the lean camera bounce, all rays start at the camera's origin (yk_device.h, YK_CTRL_CAM_O)
            const float4 ro = !valid ? make_float4(0, 0, 0, 0) : (rayO ? rayO[idx] : *lean_origin), rd = valid ? rayD[idx] : make_float4(0, 0, 1, 0);
            r.o = f4_xyz(ro);
            r.d = f4_xyz(rd);
            r.inv = V3{1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
            negmask = (r.inv.x < 0.0f ? 1u : 0u) | (r.inv.y < 0.0f ? 2u : 0u) | (r.inv.z < 0.0f ? 4u : 0u);
            r.rt = ray_tri_setup(r.d);
            r.t_max = __builtin_inff();
            float tmin;
            alive = valid && slab(root_lo, root_hi, r.o, r.inv, r.t_max, tmin);
        }
        unsigned long long remaining = __ballot(alive);
        while (remaining) {
            const int first = __ffsll((long long)remaining) - 1;
            const unsigned sg = (unsigned)__builtin_amdgcn_readlane((int)negmask, first);
            const unsigned long long group = __ballot(alive && negmask == sg);
            remaining &= ~group;
            pkt_closest_group<SPHERES>(sc, stk, r, best, group, sg);
        }
        if (valid) hit_tri[idx] = best;
    }
    }
}

// Shadow rays (dense queue); vis[slot] = 2 when occluded (slot_of given) — as k_trace_any_pt.
template <bool SPHERES>
__global__ __launch_bounds__(YK_PKT_BLOCK, 8) void k_trace_any_packet(DevScene sc, const float4* __restrict__ shO, const float4* __restrict__ shD,
                                                                      const unsigned* __restrict__ slot_of, const unsigned* count_ptr, unsigned* head,
                                                                      unsigned char* __restrict__ vis, unsigned long long* shadow_counter, CancelRef cancel) {
    __shared__ uint4 lds_stack[(YK_PKT_BLOCK / YK_WAVE) * YK_PKT_STACK];
    PktStack stk;
    stk.base = lds_stack + (threadIdx.x / YK_WAVE) * YK_PKT_STACK;
    const unsigned n = cancel_raised(cancel) ? 0u : *count_ptr;
    if (shadow_counter && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(shadow_counter, (unsigned long long)n);
    const V3 root_lo = V3{sc.root_bmin[0], sc.root_bmin[1], sc.root_bmin[2]}, root_hi = V3{sc.root_bmax[0], sc.root_bmax[1], sc.root_bmax[2]};
    const unsigned per_claim = pkt_claim_size(n);
    for (;;) {
        unsigned claim = 0;
        if (lane_id() == 0) {
            claim = atomicAdd(head, per_claim);
            if (blockIdx.x == 0 && threadIdx.x == 0 && cancel_relay(cancel, head)) claim = 0xffffffffu;  // interrupted (yk_device.h): the head is poisoned, nobody claims again
        }
        claim = uni(claim);
        if (claim >= n) break;
        const unsigned claim_end = claim + per_claim < n ? claim + per_claim : n;
    for (unsigned base = claim; base < claim_end; base += YK_WAVE) {
        const unsigned k = base + lane_id();
        const bool valid = k < n;
        PktRay r;
        int area_light = -1;
        unsigned slot = 0, negmask = 0;
        bool alive = false, occluded = false;
        {
            const float4 ro = valid ? shO[k] : make_float4(0, 0, 0, 0), rd = valid ? shD[k] : make_float4(0, 0, 1, 0);
            r.o = f4_xyz(ro);
            r.d = f4_xyz(rd);
            r.inv = V3{1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
            negmask = (r.inv.x < 0.0f ? 1u : 0u) | (r.inv.y < 0.0f ? 2u : 0u) | (r.inv.z < 0.0f ? 4u : 0u);
            r.rt = ray_tri_setup(r.d);
            r.t_max = ro.w;
            area_light = (int)__float_as_uint(rd.w);
            slot = valid ? (slot_of ? slot_of[k] : k) : 0u;
            float tmin;
            alive = valid && slab(root_lo, root_hi, r.o, r.inv, r.t_max, tmin);
        }
        unsigned long long cmask = __ballot(alive);
        if (cmask) {
            // visiting order only affects how soon occluders are found: follow the first lane's signs
            const unsigned sg = (unsigned)__builtin_amdgcn_readlane((int)negmask, __ffsll((long long)cmask) - 1);
            unsigned cur = sc.root_ref;
            int sp = 0;
            for (;;) {
                bool need_pop = false;
                if (cur & YK_LEAF_BIT) {
                    unsigned prim = cur & ~YK_LEAF_BIT;
                    for (;;) {
                        cf4* tq = as_const(sc.tris + 3 * prim);
                const float4 v0 = ldc(tq, 0), v1 = ldc(tq, 1), v2 = ldc(tq, 2);
                        const unsigned pflags = __float_as_uint(v2.w);
                        if (in_mask(cmask) && !occluded) {
                            TriHit h;
                            bool got;
                            if (SPHERES && (pflags & YK_PRIM_SPHERE)) {
                                V3 ro, rd;
                                got = sphere_hit_t(sc.spheres[__float_as_uint(v1.w) - sc.n_triangles], r.o, r.d, r.t_max, h.t, ro, rd);
                            } else {
                                got = tri_intersect(r.o, r.rt, r.t_max, f4_xyz(v0), f4_xyz(v1), f4_xyz(v2), h);
                            }
                            if (got) {
                                // bvh.rs:269-280: the sampled area light's own surface does not occlude
                                const int prim_light = (int)__float_as_uint(v0.w);
                                if (!(area_light >= 0 && prim_light >= 0 && prim_light == area_light)) occluded = true;
                            }
                        }
                        if (pflags & YK_PRIM_LAST) break;
                        ++prim;
                    }
                    need_pop = true;
                } else {
                    const PktNode nb = pkt_load_node(sc.nodes, cur);
                    const bool mine = in_mask(cmask) && !occluded;
                    float t0, t1;
                    const bool h0 = mine && slab(nb.lo0, nb.hi0, r.o, r.inv, r.t_max, t0);
                    const bool h1 = mine && slab(nb.lo1, nb.hi1, r.o, r.inv, r.t_max, t1);
                    const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1);
                    const bool swap = (sg >> nb.axis) & 1u;
                    const unsigned near_ref = swap ? nb.ref1 : nb.ref0, far_ref = swap ? nb.ref0 : nb.ref1;
                    const unsigned long long m_near = swap ? m1 : m0, m_far = swap ? m0 : m1;
                    if (m_near) {
                        if (m_far) {
                            stk.push(sp, far_ref, 0u, m_far);  // t_max is fixed: the box test stays valid
                            ++sp;
                        }
                        cur = near_ref;
                        cmask = m_near;
                    } else if (m_far) {
                        cur = far_ref;
                        cmask = m_far;
                    } else {
                        need_pop = true;
                    }
                }
                if (need_pop) {
                    const unsigned long long occ = __ballot(occluded);
                    bool done = false;
                    for (;;) {
                        if (sp == 0) {
                            done = true;
                            break;
                        }
                        --sp;
                        unsigned ref, which;
                        unsigned long long mask;
                        stk.at(sp, ref, which, mask);
                        mask &= ~occ;
                        if (mask) {
                            cur = ref;
                            cmask = mask;
                            break;
                        }
                    }
                    if (done) break;
                }
            }
        }
        if (valid) {
            if (occluded)
                vis[slot] = slot_of ? 2 : 1;
            else if (!slot_of)
                vis[slot] = 0;
        }
    }
    }
}

unsigned packet_blocks_per_cu() { return 8u; }

void launch_trace_closest_packet(hipStream_t s, unsigned grid, const DevScene& sc, const float4* rayO, const float4* rayD, const unsigned* count_ptr,
                                 unsigned* head, int* hit_tri, unsigned long long* ray_counter, const float4* lean_origin, CancelRef cancel) {
    if (sc.spheres)
        hipLaunchKernelGGL((k_trace_closest_packet<true>), dim3(grid), dim3(YK_PKT_BLOCK), 0, s, sc, rayO, rayD, count_ptr, head, hit_tri, ray_counter, lean_origin, cancel);
    else
        hipLaunchKernelGGL((k_trace_closest_packet<false>), dim3(grid), dim3(YK_PKT_BLOCK), 0, s, sc, rayO, rayD, count_ptr, head, hit_tri, ray_counter, lean_origin, cancel);
}
void launch_trace_any_packet(hipStream_t s, unsigned grid, const DevScene& sc, const float4* shO, const float4* shD, const unsigned* slot_of,
                             const unsigned* count_ptr, unsigned* head, unsigned char* vis, unsigned long long* shadow_counter, CancelRef cancel) {
    if (sc.spheres)
        hipLaunchKernelGGL((k_trace_any_packet<true>), dim3(grid), dim3(YK_PKT_BLOCK), 0, s, sc, shO, shD, slot_of, count_ptr, head, vis, shadow_counter, cancel);
    else
        hipLaunchKernelGGL((k_trace_any_packet<false>), dim3(grid), dim3(YK_PKT_BLOCK), 0, s, sc, shO, shD, slot_of, count_ptr, head, vis, shadow_counter, cancel);
}

}  // namespace yk
