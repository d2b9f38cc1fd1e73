// yk_trace.hip — BVH traversal kernels for gfx950.
//
//   k_trace_closest_pt / k_trace_any_pt   production kernels: persistent waves,
//       per-lane ray replacement with software prefetch, wave-uniform choice
//       between an interior-node step and a leaf step (LDS-staged stacks)
//   k_trace_closest<STATS> / k_trace_any  plain one-ray-per-lane loops; the STATS
//       flavour reproduces IntersectionResult's counters (bvh.rs:167-179) for the
//       BVHIntersections integrator and the parity tests
//
// Both families implement BoundingVolumeHierarchy::intersect (bvh.rs:160-232) and
// ::any_intersect (bvh.rs:235-302) with the reference's visiting order.
#include <hip/hip_runtime.h>

#include "yk_device.h"
#include "yk_geom.h"
#include "yk_kernels.h"
#ifdef YK_EXPERIMENT_XCD
#include "../../tools/micro/xcd_claim_experiment.h"
#endif
#include "yk_wave.h"

// build-time tuning knobs of the persistent traversal kernels
#ifndef TRACE_BLOCK
#define TRACE_BLOCK 256
#endif
#ifndef TRACE_LDS
#define TRACE_LDS 8  // stack entries per lane kept in LDS
#endif
#ifndef TRACE_TOP
#define TRACE_TOP 95  // interior nodes of the first tree levels kept in LDS (<= YK_TOP_MAX)
#endif
#ifndef TRACE_ANY_TOP
#define TRACE_ANY_TOP 224  // any-hit kernel: its stack entries are 4 bytes (a ref, no entry distance), which leaves LDS for this many top nodes
#endif
#ifndef TRACE_ANY_LDS
#define TRACE_ANY_LDS TRACE_LDS  // any-hit kernel: stack entries per lane kept in LDS (>= TRACE_LDS: both kernels share the spill buffer's depth)
#endif
#ifndef TRACE_MIN_WAVES
#define TRACE_MIN_WAVES 7  // waves per SIMD the register allocator must leave room for
#endif

namespace yk {

// Sensitivity experiments are compiled in only by tools/build_variant.sh (-DYK_TRACE_EXPERIMENTS pulls in
// tools/micro/trace_experiments.h); the product build sees an empty hook.
#ifdef YK_TRACE_EXPERIMENTS
#include "../../tools/micro/trace_experiments.h"
#else
#define YK_EXPERIMENT_NODE(p, nb)
#endif

// ------------------------------------------------------------------ traversal
// Traversal stack: entries [0, LDS_DEPTH) live in LDS laid out [depth][thread]
// (conflict-free: the bank depends on the lane only), deeper entries overflow to
// a per-thread slice of HBM scratch.  Capacity 64 like the reference (bvh.rs:172).
#define YK_REF_STACK_CAP 64  // the reference's to_visit_stack (bvh.rs:172-174): binary traversal
#define YK_STACK_CAP 96      // storage: 64 binary entries can become 96 in the 4-wide traversal

// LDS words are addressed through an address_space(3) pointer so the compiler
// emits ds_read_b64 / ds_write_b64 (a generic pointer in a struct degrades to
// flat_load/flat_store, which also ties the access to vmcnt).
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
typedef __attribute__((address_space(1))) unsigned long long glb_u64;

template <int BLOCK, int LDS_DEPTH> struct TravStack {
    lds_u64* lds;    // [LDS_DEPTH][BLOCK], entry = ref | tmin_bits << 32
    glb_u64* spill;  // [YK_STACK_CAP - LDS_DEPTH][spill_stride]
    unsigned spill_stride, gtid;
    __device__ __forceinline__ void push(int sp, unsigned ref, float tmin) {
        unsigned long long e = (unsigned long long)ref | ((unsigned long long)__float_as_uint(tmin) << 32);
        if (sp < LDS_DEPTH)
            lds[sp * BLOCK + threadIdx.x] = e;
        else
            spill[(size_t)(sp - LDS_DEPTH) * spill_stride + gtid] = e;
    }
    __device__ __forceinline__ uint2 at(int sp) const {
        unsigned long long e;
        if (sp < LDS_DEPTH)
            e = lds[sp * BLOCK + threadIdx.x];
        else
            e = spill[(size_t)(sp - LDS_DEPTH) * spill_stride + gtid];
        return make_uint2((unsigned)e, (unsigned)(e >> 32));
    }
};

// any-hit traversal never re-tests a deferred box, so its stack holds bare refs: half the LDS
typedef __attribute__((address_space(3))) unsigned lds_u32;
typedef __attribute__((address_space(1))) unsigned glb_u32;
template <int BLOCK, int LDS_DEPTH> struct TravStack32 {
    lds_u32* lds;    // [LDS_DEPTH][BLOCK]
    glb_u32* spill;  // [YK_STACK_CAP - LDS_DEPTH][spill_stride] (the 8-byte-entry spill buffer, used as words)
    unsigned spill_stride, gtid;
    __device__ __forceinline__ void push(int sp, unsigned ref) {
        if (sp < LDS_DEPTH)
            lds[sp * BLOCK + threadIdx.x] = ref;
        else
            spill[(size_t)(sp - LDS_DEPTH) * spill_stride + gtid] = ref;
    }
    __device__ __forceinline__ unsigned at(int sp) const {
        return sp < LDS_DEPTH ? lds[sp * BLOCK + threadIdx.x] : spill[(size_t)(sp - LDS_DEPTH) * spill_stride + gtid];
    }
};

struct NodeBoxes {
    V3 lo0, hi0, lo1, hi1;
    unsigned ref0, ref1, axis;
};
__device__ __forceinline__ NodeBoxes load_node(const DevNode* nodes, unsigned idx) {
    const float4* q = reinterpret_cast<const float4*>(nodes + idx);
    float4 a = q[0], b = q[1], c = q[2];
    uint2 d = reinterpret_cast<const uint2*>(q)[6];  // only 56 of the node's 64 bytes are fetched
    NodeBoxes n;
    n.lo0 = V3{a.x, a.y, a.z};
    n.hi0 = V3{a.w, b.x, b.y};
    n.lo1 = V3{b.z, b.w, c.x};
    n.hi1 = V3{c.y, c.z, c.w};
    n.ref0 = d.x;
    n.ref1 = d.y & ~YK_AXIS_MASK;
    n.axis = (d.y >> YK_AXIS_SHIFT) & 3u;
    return n;
}

// the first tree levels live in LDS (YK_TOP_BIT refs): a block copies them once
typedef __attribute__((address_space(3))) float4 lds_f4;
__device__ __forceinline__ NodeBoxes load_node_lds(const float4* top, unsigned idx) {
    const float4* q = top + 4 * idx;
    float4 a = q[0], b = q[1], c = q[2];
    uint2 d = reinterpret_cast<const uint2*>(q)[6];  // only 56 of the node's 64 bytes are fetched
    NodeBoxes n;
    n.lo0 = V3{a.x, a.y, a.z};
    n.hi0 = V3{a.w, b.x, b.y};
    n.lo1 = V3{b.z, b.w, c.x};
    n.hi1 = V3{c.y, c.z, c.w};
    n.ref0 = d.x;
    n.ref1 = d.y & ~YK_AXIS_MASK;
    n.axis = (d.y >> YK_AXIS_SHIFT) & 3u;
    return n;
}
template <int BLOCK> __device__ __forceinline__ void fill_top(float4* lds_top, const DevNode* top_nodes, unsigned n_top) {
    const float4* src = reinterpret_cast<const float4*>(top_nodes);
    for (unsigned i = threadIdx.x; i < n_top * 4u; i += BLOCK) lds_top[i] = src[i];
    __syncthreads();
}

// ---- 4-wide node step ------------------------------------------------------------
// Tests the four grandchild boxes of a collapsed node and returns them in the reference's
// visiting order for this ray (slot k of the result is visited before slot k+1); a missed
// or absent slot has ref == YK_REF_NONE.
//
// Equivalence with the binary traversal (traversal equivalence): the reference
// would first test the intermediate child box A (or B) and only then its children.  For
// the slab test of bounds.rs:176-193 a box that contains another yields, axis by axis, an
// interval that contains the other's ((p - o) * inv is monotone in p, min/max keep order and
// drop the same NaNs), so hit(grandchild) implies hit(child) with any t_max: skipping the
// intermediate test visits exactly the same leaves.  Deferred slots keep their entry
// distance and are re-checked against the current t_max when popped, as in the 2-wide step.
struct Step4 {
    unsigned ref[4];
    float t[4];
};
__device__ __forceinline__ Step4 node4_step(const DevNode4* nodes, unsigned idx, const V3& o, const V3& inv, float t_max, unsigned negmask) {
    const float4* q = reinterpret_cast<const float4*>(nodes + idx);
    const float4 a0 = q[0], a1 = q[1], a2 = q[2], b0 = q[3], b1 = q[4], b2 = q[5];
    const uint4 refs = reinterpret_cast<const uint4*>(q)[6];
    const unsigned axes = reinterpret_cast<const uint4*>(q)[7].x;
    Step4 s;
    bool h0 = slab(V3{a0.x, a0.y, a0.z}, V3{a0.w, a1.x, a1.y}, o, inv, t_max, s.t[0]);
    bool h1 = slab(V3{a1.z, a1.w, a2.x}, V3{a2.y, a2.z, a2.w}, o, inv, t_max, s.t[1]);
    bool h2 = slab(V3{b0.x, b0.y, b0.z}, V3{b0.w, b1.x, b1.y}, o, inv, t_max, s.t[2]);
    bool h3 = slab(V3{b1.z, b1.w, b2.x}, V3{b2.y, b2.z, b2.w}, o, inv, t_max, s.t[3]);
    s.ref[0] = h0 ? refs.x : YK_REF_NONE;  // an absent slot already holds YK_REF_NONE
    s.ref[1] = h1 ? refs.y : YK_REF_NONE;
    s.ref[2] = h2 ? refs.z : YK_REF_NONE;
    s.ref[3] = h3 ? refs.w : YK_REF_NONE;
    const bool sp = (negmask >> (axes & 3u)) & 1u, sa = (negmask >> ((axes >> 2) & 3u)) & 1u, sb = (negmask >> ((axes >> 4) & 3u)) & 1u;
#define YK_CSWAP(c, i, j)                         \
    {                                             \
        const unsigned ri = s.ref[i], rj = s.ref[j]; \
        const float ti = s.t[i], tj = s.t[j];     \
        s.ref[i] = (c) ? rj : ri;                 \
        s.ref[j] = (c) ? ri : rj;                 \
        s.t[i] = (c) ? tj : ti;                   \
        s.t[j] = (c) ? ti : tj;                   \
    }
    YK_CSWAP(sa, 0, 1)
    YK_CSWAP(sb, 2, 3)
    YK_CSWAP(sp, 0, 2)
    YK_CSWAP(sp, 1, 3)
#undef YK_CSWAP
    return s;
}

// Closest hit with the reference's visiting order (near child first by the sign
// of the direction along the split axis, far child deferred, leaves in shape
// order, a later hit with t == t_max replaces the earlier one).  The box of a
// deferred child is evaluated when its parent is visited — against a slightly
// relaxed bound, because a tie hit can raise t_max by a few ulps (deferred_t_max,
// yk_geom.h) — and completed at pop time by the exact `tmin <= t_max`, which is the
// reference's test at pop time (DESIGN.md §traversal equivalence).
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
            const float t_def = deferred_t_max(t_max);
            bool h0 = slab(nb.lo0, nb.hi0, o, inv, t_def, t0);
            bool h1 = slab(nb.lo1, nb.hi1, o, inv, t_def, t1);
            bool swap = neg[nb.axis];
            unsigned near_ref = swap ? nb.ref1 : nb.ref0, far_ref = swap ? nb.ref0 : nb.ref1;
            // the near child is entered now: exact bound; the far child is deferred: relaxed now, exact at pop
            bool near_hit = (swap ? h1 : h0) && (swap ? t1 : t0) <= t_max, far_hit = swap ? h0 : h1;
            float far_t = swap ? t0 : t1;
            if (STATS) {
                node_tests += 1;  // the near child is tested right away; the far one is counted when popped
                if (near_hit) node_hits += 1;
            }
            if (STATS || far_hit) {
                // with STATS the far child is pushed even when its box is missed so
                // that the test is counted at pop time like the reference does
                if (sp >= YK_REF_STACK_CAP) {
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
                const unsigned pflags = __float_as_uint(v2.w);
                TriHit h = TriHit{0.0f, 0.0f, 0.0f, 0.0f};
                if (STATS) shape_tests += 1;
                bool got;
                if (pflags & YK_PRIM_SPHERE) {
                    V3 ro, rd;
                    got = sphere_hit_t(sc.spheres[__float_as_uint(v1.w) - sc.n_triangles], o, d, t_max, h.t, ro, rd);
                } else {
                    got = tri_intersect(o, rt, t_max, f4_xyz(v0), f4_xyz(v1), f4_xyz(v2), h);
                }
                if (got) {
                    out_hit = h;
                    out_tri = (int)__float_as_uint(v1.w);
                    t_max = h.t;
                }
                if (pflags & YK_PRIM_LAST) break;
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


// ------------------------------------------------------------------ persistent-thread traversal
// A wave owns 64 ray slots.  Work arrives in CHUNK-sized index ranges claimed with
// one atomic on the queue head; inside a chunk indices are handed to lanes by a
// wave-local cursor.  Every lane keeps the NEXT ray it will trace already loaded
// in registers (issued as soon as PF_MIN lanes lack one, consumed iterations
// later), so replacing a finished ray costs no memory round trip: with plain
// "fetch when idle" the wave stalled 2-4 us per refill and lost more than the
// idle lanes had cost (profiles/r01_b_sweep.txt).  Each iteration the wave then
// runs ONE of two bodies — an interior-node step for all lanes on interior nodes,
// or, once LEAF_MIN lanes are parked on leaves (or none is on a node), the leaf
// intersection for the parked lanes — so neither body executes at a handful of
// lanes (lane utilisation of the plain loop: 26 %, profiles/r01_a_pmc_summary.json).
// Per-ray arithmetic and visiting order are exactly those of traverse_closest /
// traverse_any above.
struct LaneRay {
    V3 o, inv, d;
    RayTri rt;
    float t_max;
    unsigned negmask;
};

__device__ __forceinline__ void lane_ray_setup(LaneRay& r, V3 o, V3 d, float t_max) {
    r.o = o;
    r.d = d;
    r.inv = V3{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    r.negmask = (r.inv.x < 0.0f ? 1u : 0u) | (r.inv.y < 0.0f ? 2u : 0u) | (r.inv.z < 0.0f ? 4u : 0u);
    r.rt = ray_tri_setup(d);
    r.t_max = t_max;
}

// pops until an entry whose stored entry distance still satisfies tmin <= t_max
template <int BLOCK, int LDS_DEPTH> __device__ __forceinline__ bool pop_closest(TravStack<BLOCK, LDS_DEPTH>& stk, int& sp, float t_max, unsigned& cur) {
    while (sp > 0) {
        --sp;
        uint2 e = stk.at(sp);
        if (__uint_as_float(e.y) <= t_max) {
            cur = e.x;
            return true;
        }
    }
    return false;
}

// Wave-local work distribution: claims CHUNK indices at a time from *head.
struct ChunkCursor {
    unsigned cur, end;  // wave-uniform
    bool exhausted, took_share;
#ifdef YK_EXPERIMENT_XCD
    unsigned xcd_step;  // ranges this wave has found drained
#endif
    __device__ __forceinline__ void init() {
        cur = end = 0;
        exhausted = took_share = false;
#ifdef YK_EXPERIMENT_XCD
        xcd_step = 0;
#endif
    }
    // hands `want` lanes consecutive indices; returns the index of this lane or
    // 0xffffffff.  All lanes of the wave call it (converged).
    // Long queues: CHUNK indices per claim, one atomic on *head each.  A queue that one chunk
    // per wave covers (n <= waves * CHUNK: late bounces, small per-GPU shares, interactive
    // passes) is cut into equal shares instead, wave w takes share w and nothing else: the
    // two atomics per wave of the dynamic scheme (a claim and a failed claim, ~14 K per launch on
    // one address at ~100 M/s) were most of what such a launch cost beyond its longest ray.
    // `cancel`: the launch's relay wave (first wave of block 0) also looks at the host's interruption word when it claims
    // (yk_device.h); finding it set it poisons the head, and every wave's next claim comes back beyond the queue's end.
    template <int CHUNK> __device__ __forceinline__ unsigned take(bool want, unsigned n, unsigned* head, const CancelRef& cancel) {
        unsigned long long mask = __ballot(want);
        if (mask == 0ull || exhausted) return 0xffffffffu;
        if (cur >= end) {
            const unsigned waves = gridDim.x * (blockDim.x / YK_WAVE);
            unsigned base;
            unsigned chunk;
            if (n <= waves * (unsigned)CHUNK) {
                if (took_share) {
                    exhausted = true;
                    return 0xffffffffu;
                }
                took_share = true;
                chunk = (n + waves - 1u) / waves;
                chunk = chunk < 8u ? 8u : chunk;  // fewer, fuller waves for very short queues
                base = (blockIdx.x * (blockDim.x / YK_WAVE) + threadIdx.x / YK_WAVE) * chunk;
            } else {
                chunk = (unsigned)CHUNK;
                base = 0;
#ifdef YK_EXPERIMENT_XCD  // timing builds only (tools/micro/xcd_claim_experiment.h): XCD-affine claims
                YK_XCD_CLAIM
#endif
                if (lane_id() == 0) {
                    base = atomicAdd(head, chunk);
                    if (blockIdx.x == 0 && threadIdx.x == 0 && cancel_relay(cancel, head)) base = 0xffffffffu;
                }
                base = __shfl(base, 0);
            }
            if (base >= n) {
                exhausted = true;
                return 0xffffffffu;
            }
            cur = base;
            end = base + chunk < n ? base + chunk : n;
#ifdef YK_EXPERIMENT_XCD
        claimed:;
#endif
        }
        unsigned rank = (unsigned)__popcll(mask & ((1ull << lane_id()) - 1ull));
        unsigned idx = cur + rank;
        unsigned total = (unsigned)__popcll(mask);
        cur = cur + total < end ? cur + total : end;
        return (want && idx < end) ? idx : 0xffffffffu;
    }
};

template <int BLOCK, int LDS_DEPTH, int PF_MIN, int START_MIN, int LEAF_MIN, int CHUNK, bool SPHERES, bool API, bool WIDE>
__global__ __launch_bounds__(BLOCK, TRACE_MIN_WAVES) void k_trace_closest_pt(DevScene sc, const float4* __restrict__ rayO, const float4* __restrict__ rayD,
                                                            const float* __restrict__ t_max_opt, const unsigned* count_ptr, unsigned* head,
                                                            int* __restrict__ hit_tri, float4* __restrict__ hit_out, uint2* spill,
                                                            unsigned spill_stride, unsigned* ctrl, unsigned long long* ray_counter, const unsigned* cancel_host) {
    __shared__ unsigned long long lds_stack[LDS_DEPTH * BLOCK];
    __shared__ float4 lds_top[WIDE ? 1 : TRACE_TOP * 4];
    if (!WIDE) fill_top<BLOCK>(lds_top, sc.top_nodes, sc.n_top);
    TravStack<BLOCK, LDS_DEPTH> stk;
    stk.lds = (lds_u64*)lds_stack;
    stk.spill = (glb_u64*)spill;
    stk.spill_stride = spill_stride;
    stk.gtid = blockIdx.x * BLOCK + threadIdx.x;
    const CancelRef cancel = CancelRef{cancel_host, cancel_host ? ctrl + YK_CTRL_CANCELLED : nullptr};  // render loop: ctrl is the context's error block
    const unsigned n = cancel_raised(cancel) ? 0u : *count_ptr;
    if (ray_counter && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(ray_counter, (unsigned long long)n);
    const unsigned root = WIDE ? 0u : (sc.n_top ? YK_TOP_BIT : sc.root_ref);
    const V3 root_lo = V3{sc.root_bmin[0], sc.root_bmin[1], sc.root_bmin[2]}, root_hi = V3{sc.root_bmax[0], sc.root_bmax[1], sc.root_bmax[2]};

    ChunkCursor work;
    work.init();
    bool active = false, pf_valid = false;
    float4 pf_o = make_float4(0, 0, 0, 0), pf_d = make_float4(0, 0, 1, 0);
    float pf_t = 0.0f;
    unsigned pf_idx = 0;
    LaneRay r;
    r.o = r.inv = r.d = V3{0, 0, 0};
    r.rt = RayTri{0, 1, 2, 0, 0, 0};
    r.t_max = 0.0f;
    r.negmask = 0;
    unsigned ray_i = 0, cur = 0;
    int sp = 0, best = -1;
    TriHit best_hit = TriHit{0, 0, 0, 0};

    for (;;) {
        // ---- start prefetched rays once START_MIN lanes are idle (the setup code —
        // six IEEE divisions and the root test — runs divergently, so it is batched)
        const bool startable = !active && pf_valid;
        const bool go = (unsigned)__popcll(__ballot(startable)) >= (unsigned)START_MIN || !__any(active);
        if (go && startable) {
            pf_valid = false;
            lane_ray_setup(r, f4_xyz(pf_o), f4_xyz(pf_d), API ? pf_t : __builtin_inff());
            ray_i = pf_idx;
            float tmin;
            if (slab(root_lo, root_hi, r.o, r.inv, r.t_max, tmin)) {
                active = true;
                cur = root;
                sp = 0;
                best = -1;
            } else {
                hit_tri[ray_i] = -1;
            }
        }
        // ---- top up the prefetch registers (loads are consumed in a later iteration)
        {
            unsigned n_need = (unsigned)__popcll(__ballot(!pf_valid));
            if (!work.exhausted && (n_need >= (unsigned)PF_MIN || !__any(active))) {
                unsigned idx = work.take<CHUNK>(!pf_valid, n, head, cancel);
                if (idx != 0xffffffffu) {
                    pf_o = rayO[idx];
                    pf_d = rayD[idx];
                    pf_t = (API && t_max_opt) ? t_max_opt[idx] : __builtin_inff();
                    pf_idx = idx;
                    pf_valid = true;
                }
            }
        }
        if (!__any(active)) {
            if (work.exhausted && !__any(pf_valid)) break;
            continue;
        }
        // ---- one step, chosen for the whole wave
        const bool on_leaf = active && (cur & YK_LEAF_BIT);
        const bool on_node = active && !(cur & YK_LEAF_BIT);
        const unsigned n_leaf = (unsigned)__popcll(__ballot(on_leaf));
        if (__any(on_node) && n_leaf < (unsigned)LEAF_MIN) {
            if (WIDE) {
                if (on_node) {
                    Step4 st = node4_step(sc.nodes4, cur, r.o, r.inv, deferred_t_max(r.t_max), r.negmask);
                    // The first slot (in visiting order) that passes the exact bound is entered right away; slots
                    // before it fail now as they would in the reference (t_max cannot change before they are
                    // tested); the later ones are deferred: relaxed here, exact when popped (yk_geom.h).
                    bool entered = false;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool exact = st.ref[k] != YK_REF_NONE && st.t[k] <= r.t_max;
                        if (!entered && !exact) st.ref[k] = YK_REF_NONE;
                        entered = entered || exact;
                    }
                    // push the later-visited hits (last first)
                    unsigned next = YK_REF_NONE;
                    float next_t = 0.0f;
#pragma unroll
                    for (int k = 3; k >= 0; --k) {
                        if (st.ref[k] != YK_REF_NONE) {
                            if (next != YK_REF_NONE) {
                                if (sp >= YK_STACK_CAP) {
                                    atomicOr(ctrl + YK_CTRL_ERR, 1u);
                                    sp = 0;
                                } else {
                                    stk.push(sp, next, next_t);
                                    ++sp;
                                }
                            }
                            next = st.ref[k];
                            next_t = st.t[k];
                        }
                    }
                    if (next != YK_REF_NONE) {
                        cur = next;  // no leaf was visited since the test: its result is final
                    } else if (!pop_closest(stk, sp, r.t_max, cur)) {
                        hit_tri[ray_i] = best;
                        if (API && hit_out) hit_out[ray_i] = make_float4(best_hit.t, best_hit.b0, best_hit.b1, best_hit.b2);
                        active = false;
                    }
                }
            } else if (on_node) {
                NodeBoxes nb = (cur & YK_TOP_BIT) ? load_node_lds(lds_top, cur & ~YK_TOP_BIT) : load_node(sc.nodes, cur);
                YK_EXPERIMENT_NODE(sc.nodes + cur, nb);
                float t0, t1;
                const float t_def = deferred_t_max(r.t_max);  // yk_geom.h: a deferred box is screened with a relaxed bound
                bool h0 = slab(nb.lo0, nb.hi0, r.o, r.inv, t_def, t0);
                bool h1 = slab(nb.lo1, nb.hi1, r.o, r.inv, t_def, t1);
                bool swap = (r.negmask >> nb.axis) & 1u;
                unsigned near_ref = swap ? nb.ref1 : nb.ref0, far_ref = swap ? nb.ref0 : nb.ref1;
                bool near_hit = (swap ? h1 : h0) && (swap ? t1 : t0) <= r.t_max, far_hit = swap ? h0 : h1;
                float far_t = swap ? t0 : t1;
                if (near_hit) {
                    if (far_hit) {
                        if (sp >= YK_REF_STACK_CAP) {
                            atomicOr(ctrl + YK_CTRL_ERR, 1u);
                            sp = 0;  // abandon this ray; the host reports YK_ERR_STACK_OVERFLOW
                        } else {
                            stk.push(sp, far_ref, far_t);
                            ++sp;
                        }
                    }
                    cur = near_ref;
                } else if (far_hit && far_t <= r.t_max) {
                    cur = far_ref;  // entered now (no leaf in between): the exact bound decides
                } else if (!pop_closest(stk, sp, r.t_max, cur)) {
                    hit_tri[ray_i] = best;
                    if (API && hit_out) hit_out[ray_i] = make_float4(best_hit.t, best_hit.b0, best_hit.b1, best_hit.b2);
                    active = false;
                }
            }
        } else if (on_leaf) {
            unsigned prim = cur & ~YK_LEAF_BIT;
            for (;;) {
                float4 v0 = sc.tris[3 * prim], v1 = sc.tris[3 * prim + 1], v2 = sc.tris[3 * prim + 2];
                const unsigned pflags = __float_as_uint(v2.w);
                TriHit h = TriHit{0.0f, 0.0f, 0.0f, 0.0f};
                bool got;
                if (SPHERES && (pflags & YK_PRIM_SPHERE)) {
                    V3 ro, rd;
                    got = sphere_hit_t(sc.spheres[__float_as_uint(v1.w) - sc.n_triangles], r.o, r.d, r.t_max, h.t, ro, rd);
                } else {
                    got = tri_intersect(r.o, r.rt, r.t_max, f4_xyz(v0), f4_xyz(v1), f4_xyz(v2), h);
                }
                if (got) {
                    if (API) best_hit = h;
                    // API callers get the source shape; the render loop gets the leaf-order slot, from
                    // which k_shade reaches everything it needs in one hop (hit_surface_prim)
                    best = API ? (int)__float_as_uint(v1.w) : YK_HIT_WORD(prim, pflags);
                    r.t_max = h.t;
                }
                if (pflags & YK_PRIM_LAST) break;
                ++prim;
            }
            if (!pop_closest(stk, sp, r.t_max, cur)) {
                hit_tri[ray_i] = best;
                if (API && hit_out) hit_out[ray_i] = make_float4(best_hit.t, best_hit.b0, best_hit.b1, best_hit.b2);
                active = false;
            }
        }
    }
}

// Shadow rays: shO/shD are dense (compacted by `shade`); slot_of[k] is where the
// verdict goes (vis[slot] = 2 when occluded); slot_of == NULL (API mode): vis[k] = 0/1.
template <int BLOCK, int LDS_DEPTH, int PF_MIN, int START_MIN, int LEAF_MIN, int CHUNK, bool SPHERES, bool WIDE>
__global__ __launch_bounds__(BLOCK, TRACE_MIN_WAVES) void k_trace_any_pt(DevScene sc, const float4* __restrict__ shO, const float4* __restrict__ shD,
                                                        const unsigned* __restrict__ slot_of, const unsigned* count_ptr, unsigned* head,
                                                        unsigned char* __restrict__ vis, uint2* spill, unsigned spill_stride, unsigned* ctrl,
                                                        unsigned long long* shadow_counter, const unsigned* cancel_host) {
    __shared__ unsigned lds_stack[LDS_DEPTH * BLOCK];
    __shared__ float4 lds_top[WIDE ? 1 : TRACE_ANY_TOP * 4];
    if (!WIDE) fill_top<BLOCK>(lds_top, sc.top_nodes_any, sc.n_top_any);
    TravStack32<BLOCK, LDS_DEPTH> stk;
    stk.lds = (lds_u32*)lds_stack;
    stk.spill = (glb_u32*)spill;
    stk.spill_stride = spill_stride;
    stk.gtid = blockIdx.x * BLOCK + threadIdx.x;
    const CancelRef cancel = CancelRef{cancel_host, cancel_host ? ctrl + YK_CTRL_CANCELLED : nullptr};
    const unsigned n = cancel_raised(cancel) ? 0u : *count_ptr;
    if (shadow_counter && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(shadow_counter, (unsigned long long)n);
    const unsigned root = WIDE ? 0u : (sc.n_top_any ? YK_TOP_BIT : sc.root_ref);
    const V3 root_lo = V3{sc.root_bmin[0], sc.root_bmin[1], sc.root_bmin[2]}, root_hi = V3{sc.root_bmax[0], sc.root_bmax[1], sc.root_bmax[2]};

    ChunkCursor work;
    work.init();
    bool active = false, pf_valid = false;
    float4 pf_o = make_float4(0, 0, 0, 0), pf_d = make_float4(0, 0, 1, 0);
    unsigned pf_slot = 0;
    LaneRay r;
    r.o = r.inv = r.d = V3{0, 0, 0};
    r.rt = RayTri{0, 1, 2, 0, 0, 0};
    r.t_max = 0.0f;
    r.negmask = 0;
    unsigned slot = 0, cur = 0;
    int sp = 0, area_light = -1;

    for (;;) {
        const bool startable = !active && pf_valid;
        const bool go = (unsigned)__popcll(__ballot(startable)) >= (unsigned)START_MIN || !__any(active);
        if (go && startable) {
            pf_valid = false;
            lane_ray_setup(r, f4_xyz(pf_o), f4_xyz(pf_d), pf_o.w);
            area_light = (int)__float_as_uint(pf_d.w);
            slot = pf_slot;
            float tmin;
            if (slab(root_lo, root_hi, r.o, r.inv, r.t_max, tmin)) {
                active = true;
                cur = root;
                sp = 0;
            } else if (!slot_of) {
                vis[slot] = 0;
            }
        }
        {
            unsigned n_need = (unsigned)__popcll(__ballot(!pf_valid));
            if (!work.exhausted && (n_need >= (unsigned)PF_MIN || !__any(active))) {
                unsigned k = work.take<CHUNK>(!pf_valid, n, head, cancel);
                if (k != 0xffffffffu) {
                    pf_o = shO[k];
                    pf_d = shD[k];
                    pf_slot = slot_of ? slot_of[k] : k;
                    pf_valid = true;
                }
            }
        }
        if (!__any(active)) {
            if (work.exhausted && !__any(pf_valid)) break;
            continue;
        }
        const bool on_leaf = active && (cur & YK_LEAF_BIT);
        const bool on_node = active && !(cur & YK_LEAF_BIT);
        const unsigned n_leaf = (unsigned)__popcll(__ballot(on_leaf));
        if (__any(on_node) && n_leaf < (unsigned)LEAF_MIN) {
            if (WIDE) {
                if (on_node) {
                    // any-hit: the verdict does not depend on the visiting order; near-first finds occluders sooner
                    Step4 st = node4_step(sc.nodes4, cur, r.o, r.inv, r.t_max, r.negmask);
                    unsigned next = YK_REF_NONE;
#pragma unroll
                    for (int k = 3; k >= 0; --k) {
                        if (st.ref[k] != YK_REF_NONE) {
                            if (next != YK_REF_NONE) {
                                if (sp >= YK_STACK_CAP) {
                                    atomicOr(ctrl + YK_CTRL_ERR, 1u);
                                    sp = 0;
                                } else {
                                    stk.push(sp, next);
                                    ++sp;
                                }
                            }
                            next = st.ref[k];
                        }
                    }
                    if (next != YK_REF_NONE) {
                        cur = next;
                    } else if (sp > 0) {
                        --sp;
                        cur = stk.at(sp);
                    } else {
                        if (!slot_of) vis[slot] = 0;
                        active = false;  // unoccluded
                    }
                }
            } else if (on_node) {
                NodeBoxes nb = (cur & YK_TOP_BIT) ? load_node_lds(lds_top, cur & ~YK_TOP_BIT) : load_node(sc.nodes, cur);
                float t0, t1;
                bool h0 = slab(nb.lo0, nb.hi0, r.o, r.inv, r.t_max, t0);
                bool h1 = slab(nb.lo1, nb.hi1, r.o, r.inv, r.t_max, t1);
                bool swap = (r.negmask >> nb.axis) & 1u;
                unsigned near_ref = swap ? nb.ref1 : nb.ref0, far_ref = swap ? nb.ref0 : nb.ref1;
                bool near_hit = swap ? h1 : h0, far_hit = swap ? h0 : h1;
                if (near_hit) {
                    if (far_hit) {
                        if (sp >= YK_REF_STACK_CAP) {
                            atomicOr(ctrl + YK_CTRL_ERR, 1u);
                            sp = 0;
                        } else {
                            stk.push(sp, far_ref);
                            ++sp;
                        }
                    }
                    cur = near_ref;
                } else if (far_hit) {
                    cur = far_ref;
                } else if (sp > 0) {
                    --sp;
                    cur = stk.at(sp);
                } else {
                    if (!slot_of) vis[slot] = 0;
                    active = false;  // unoccluded
                }
            }
        } else if (on_leaf) {
            unsigned prim = cur & ~YK_LEAF_BIT;
            bool occluded = false;
            for (;;) {
                float4 v0 = sc.tris[3 * prim], v1 = sc.tris[3 * prim + 1], v2 = sc.tris[3 * prim + 2];
                const unsigned pflags = __float_as_uint(v2.w);
                TriHit h;
                bool got;
                if (SPHERES && (pflags & YK_PRIM_SPHERE)) {
                    V3 ro, rd;
                    got = sphere_hit_t(sc.spheres[__float_as_uint(v1.w) - sc.n_triangles], r.o, r.d, r.t_max, h.t, ro, rd);
                } else {
                    got = tri_intersect(r.o, r.rt, r.t_max, f4_xyz(v0), f4_xyz(v1), f4_xyz(v2), h);
                }
                if (got) {
                    // bvh.rs:269-280: a hit on the sampled area light's own surface does not occlude
                    // (spheres carry no area light: v0.w = -1)
                    int prim_light = (int)__float_as_uint(v0.w);
                    if (!(area_light >= 0 && prim_light >= 0 && prim_light == area_light)) {
                        occluded = true;
                        break;
                    }
                }
                if (pflags & YK_PRIM_LAST) break;
                ++prim;
            }
            if (occluded) {
                vis[slot] = slot_of ? 2 : 1;
                active = false;
            } else if (sp > 0) {
                --sp;
                cur = stk.at(sp);
            } else {
                if (!slot_of) vis[slot] = 0;
                active = false;
            }
        }
    }
}

// Persistent waves: each wave pulls 64 consecutive rays from a global head until
// the queue (whose length only the device knows) is drained.
template <int BLOCK, int LDS_DEPTH, bool STATS>
__global__ __launch_bounds__(BLOCK) void k_trace_closest(DevScene sc, const float4* rayO, const float4* rayD, const float* t_max_opt,
                                                         const unsigned* count_ptr, unsigned* head, int* hit_tri, float4* hit_out,
                                                         uint4* stats_out, uint2* spill, unsigned spill_stride, unsigned* ctrl,
                                                         unsigned long long* ray_counter) {
    __shared__ unsigned long long lds_stack[LDS_DEPTH * BLOCK];
    TravStack<BLOCK, LDS_DEPTH> stk;
    stk.lds = (lds_u64*)lds_stack;
    stk.spill = (glb_u64*)spill;
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


// ------------------------------------------------------------------ Whitted
// BoundingVolumeHierarchy::any_intersect (bvh.rs:235-302) for one lane: true = occluded.  A hit
// on the sampled area light's own surface does not occlude (bvh.rs:269-280).
template <int BLOCK, int LDS_DEPTH>
__device__ __forceinline__ bool traverse_any(const DevScene& sc, V3 o, V3 d, float t_max, int area_light, TravStack<BLOCK, LDS_DEPTH>& stk, unsigned* err) {
    V3 inv = V3{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    bool neg[3] = {inv.x < 0.0f, inv.y < 0.0f, inv.z < 0.0f};
    RayTri rt = ray_tri_setup(d);
    float tmin;
    if (!slab(V3{sc.root_bmin[0], sc.root_bmin[1], sc.root_bmin[2]}, V3{sc.root_bmax[0], sc.root_bmax[1], sc.root_bmax[2]}, o, inv, t_max, tmin)) return false;
    int sp = 0;
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
                    if (sp >= YK_REF_STACK_CAP) {
                        atomicOr(err, 1u);
                        return false;
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
                const unsigned pflags = __float_as_uint(v2.w);
                TriHit h = TriHit{0.0f, 0.0f, 0.0f, 0.0f};
                bool got;
                if (pflags & YK_PRIM_SPHERE) {
                    V3 ro, rd;
                    got = sphere_hit_t(sc.spheres[__float_as_uint(v1.w) - sc.n_triangles], o, d, t_max, h.t, ro, rd);
                } else {
                    got = tri_intersect(o, rt, t_max, f4_xyz(v0), f4_xyz(v1), f4_xyz(v2), h);
                }
                if (got) {
                    int prim_light = (int)__float_as_uint(v0.w);
                    if (!(area_light >= 0 && prim_light >= 0 && prim_light == area_light)) return true;
                }
                if (pflags & YK_PRIM_LAST) break;
                ++prim;
            }
        }
        if (sp == 0) return false;
        --sp;
        cur = stk.at(sp).x;
    }
}

// Whitted::li_internal (whitted.rs:74-181), one lane per camera sample.  The recursion —
// direct lighting, then the specular reflection subtree, then the specular transmission
// subtree, each child weighted as (f * li) * |wi . ns| with no pdf (whitted.rs:66) — is run
// depth first with an explicit frame stack, because the ONE sampler of the pixel sample is
// drawn from in exactly that order (2 dimensions per light at every hit).  Child rays come from
// sample_f with u = (0, 0) (whitted.rs:49-51): the tree itself does not depend on the sampler,
// and only glass has specular lobes, so only glass hits push a frame.
#define YK_WHITTED_MAX_DEPTH 16
struct WhittedFrame {
    RGB sum;          // sum_li of the suspended call
    RGB pend_f;       // weight of the child being evaluated
    float pend_cos;
    bool t_valid;     // the transmission child, evaluated after the reflection subtree
    V3 t_o, t_d;
    RGB t_f;
    float t_cos;
};

template <int BLOCK, int LDS_DEPTH>
__global__ __launch_bounds__(BLOCK) void k_whitted(DevScene sc, RenderParams prm, const uint32_t* pixel_xy, const uint32_t* sample_index_tab, PathBuffers cur,
                                                   uint32_t n, float4* sample_buf, uint2* spill, unsigned spill_stride, unsigned* ctrl,
                                                   unsigned long long* counters) {
    __shared__ unsigned long long lds_stack[LDS_DEPTH * BLOCK];
    TravStack<BLOCK, LDS_DEPTH> stk;
    stk.lds = (lds_u64*)lds_stack;
    stk.spill = (glb_u64*)spill;
    stk.spill_stride = spill_stride;
    stk.gtid = blockIdx.x * BLOCK + threadIdx.x;
    unsigned* err = ctrl + YK_CTRL_ERR;
    unsigned long long n_rays = 0, n_shadow = 0;
    if (cancel_raised(prm.cancel)) n = 0;  // interrupted before this launch started (yk_device.h, CancelRef)
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const float4 a = cur.rayO[i], b = cur.rayD[i], c = cur.thru[i];
        const uint4 r = cur.rngs[i];
        V3 o = f4_xyz(a), d = f4_xyz(b);
        const unsigned sid = __float_as_uint(b.w);
        SamplerState st;
        st.rng.state = (u64)r.x | ((u64)r.y << 32);
        st.rng.inc = (u64)r.z | ((u64)r.w << 32);
        st.dimension = __float_as_uint(c.w);
        uint32_t pix, ks;
        split_sample_id(sid, prm.spe, pix, ks);
        const uint32_t xy = pixel_xy[pix];
        st.px = xy & 0xffffu;
        st.py = xy >> 16;
        st.sample_index = (sample_index_tab ? sample_index_tab[pix] : prm.sample_base) + ks;
        WhittedFrame frames[YK_WHITTED_MAX_DEPTH];
        int sp = 0;  // frames on the stack = depth of the call being evaluated
        bool is_specular = false;
        RGB ret = RGB{0.0f, 0.0f, 0.0f};
        for (;;) {
            // ---- call: li_internal(ray (o, d), depth = sp, is_specular)
            n_rays += 1;
            int shape;
            TriHit th = TriHit{0.0f, 0.0f, 0.0f, 0.0f};
            unsigned nt = 0, nh = 0, ns = 0;
            traverse_closest<BLOCK, LDS_DEPTH, false>(sc, o, d, __builtin_inff(), stk, shape, th, nt, nh, ns, err);
            bool called = false;
            if (shape < 0) {
                ret = RGB{sc.background[0], sc.background[1], sc.background[2]};  // whitted.rs:171
            } else {
                Surface sf = hit_surface(sc, (uint32_t)shape, o, d);
                Material mat = sc.materials[sf.material];
                if (mat.tex) {  // matte.rs:29-30
                    RGB kd = texture_eval(sc, mat.tex - 1u, sf.u, sf.v);
                    mat.a[0] = kd.r;
                    mat.a[1] = kd.g;
                    mat.a[2] = kd.b;
                    if (is_black(kd)) mat.kind = MK_BLACK;
                }
                const Frame fr = make_frame(sf.n, sf.ns, sf.dpdus);
                RGB sum = RGB{0.0f, 0.0f, 0.0f};
                for (unsigned l = 0; l < sc.n_lights; ++l) {  // whitted.rs:113-133, the fold of path.rs:102-119
                    float ux, uy;
                    sampler_get_2d(prm.sampler, st, ux, uy);
                    LightSample ls = sample_light(sc.lights[l], (int)l, sf.p, ux, uy);
                    if (is_black(ls.li)) continue;
                    RGB f = bsdf_f(mat, fr, sf.wo, ls.l);
                    if (!ls.has_vis || is_black(f)) continue;
                    V3 offset = sf.n * 0.001f;  // VisibilityTester::ray = p0.spawn_ray_to(p1), interaction.rs:44-59
                    V3 so = dot(ls.p1 - sf.p, sf.n) > 0.0f ? sf.p + offset : sf.p - offset;
                    n_shadow += 1;
                    if (!traverse_any<BLOCK, LDS_DEPTH>(sc, so, ls.p1 - so, 0.9999f, ls.area_light, stk, err))
                        sum = sum + f * ls.li * rclamp(dot_nv(sf.ns, ls.l), 0.0f, 1.0f) / ls.pdf;
                }
                if (sp == 0 || is_specular) {  // whitted.rs:135-137, rectangular_light.rs:75-81
                    if (sf.area_light >= 0) {
                        const DevLight& L = sc.lights[sf.area_light];
                        sum = sum + (dot_nv(sf.n, -d) > 0.0f ? RGB{L.i[0], L.i[1], L.i[2]} : RGB{0.0f, 0.0f, 0.0f});
                    } else {
                        sum = sum + RGB{0.0f, 0.0f, 0.0f};
                    }
                }
                ret = sum;
                if ((unsigned)sp + 1u < prm.max_depth && mat.kind == MK_GLASS) {  // whitted.rs:139-167
                    BsdfSample rs = bsdf_sample_specular(mat, fr, sf.wo, BX_REFLECTION);
                    BsdfSample ts = bsdf_sample_specular(mat, fr, sf.wo, BX_TRANSMISSION);
                    if (rs.type != BX_NONE || ts.type != BX_NONE) {
                        WhittedFrame& fm = frames[sp];
                        fm.sum = sum;
                        fm.t_valid = false;
                        const BsdfSample& first = rs.type != BX_NONE ? rs : ts;
                        if (rs.type != BX_NONE && ts.type != BX_NONE) {
                            fm.t_valid = true;
                            fm.t_o = spawn_origin(sf.p, sf.n, ts.wi);
                            fm.t_d = ts.wi;
                            fm.t_f = ts.f;
                            fm.t_cos = fabsf(dot_nv(ts.wi, sf.ns));
                        }
                        fm.pend_f = first.f;
                        fm.pend_cos = fabsf(dot_nv(first.wi, sf.ns));
                        o = spawn_origin(sf.p, sf.n, first.wi);
                        d = first.wi;
                        is_specular = true;  // sample_type.contains(SPECULAR), whitted.rs:64
                        ++sp;
                        called = true;
                    }
                }
            }
            if (called) continue;
            // ---- return: fold `ret` into the suspended callers
            bool again = false;
            while (sp > 0) {
                WhittedFrame& fm = frames[sp - 1];
                fm.sum = fm.sum + fm.pend_f * ret * fm.pend_cos;  // ret.li = f * ret.li * |wi . ns| ; sum_li += li
                if (fm.t_valid) {
                    fm.t_valid = false;
                    fm.pend_f = fm.t_f;
                    fm.pend_cos = fm.t_cos;
                    o = fm.t_o;
                    d = fm.t_d;
                    is_specular = true;
                    again = true;
                    break;
                }
                ret = fm.sum;
                --sp;
            }
            if (!again) break;
        }
        sample_buf[sid] = make_float4(ret.r, ret.g, ret.b, 0.0f);
    }
    // ray counts: one atomic per wave
    for (int off = 32; off > 0; off >>= 1) {
        n_rays += __shfl_down(n_rays, off);
        n_shadow += __shfl_down(n_shadow, off);
    }
    if (lane_id() == 0 && counters) {
        if (n_rays) atomicAdd(counters, n_rays);
        if (n_shadow) atomicAdd(counters + 1, n_shadow);
    }
}

// ------------------------------------------------------------------ launchers
#ifndef TRACE_PF_MIN
#define TRACE_PF_MIN 16
#endif
#ifndef TRACE_START_MIN
#define TRACE_START_MIN 8
#endif
#ifndef TRACE_LEAF_MIN
#define TRACE_LEAF_MIN 16
#endif
#ifndef TRACE_CHUNK
#define TRACE_CHUNK 128
#endif

unsigned trace_block_size() { return TRACE_BLOCK; }
unsigned trace_spill_depth() { return YK_STACK_CAP - TRACE_LDS; }
unsigned trace_top_nodes() { return TRACE_TOP; }
unsigned trace_top_nodes_any() { return TRACE_ANY_TOP; }
static_assert(TRACE_ANY_LDS * TRACE_BLOCK * 4 + TRACE_ANY_TOP * 64 <= TRACE_LDS * TRACE_BLOCK * 8 + TRACE_TOP * 64 + 1024, "the any-hit kernel must fit the block count of the closest-hit kernel");
static_assert(TRACE_ANY_LDS >= TRACE_LDS, "the spill buffer is sized for TRACE_LDS entries in LDS");
unsigned trace_blocks_per_cu() {
    unsigned by_lds = (160u * 1024u) / (TRACE_LDS * TRACE_BLOCK * 8u + TRACE_TOP * 64u);
    unsigned by_waves = (unsigned)TRACE_MIN_WAVES * 256u / TRACE_BLOCK;
    return by_lds < by_waves ? by_lds : by_waves;
}

void launch_trace_closest(hipStream_t s, unsigned grid, const DevScene& sc, const float4* rayO, const float4* rayD, const float* t_max_opt,
                          const unsigned* count_ptr, unsigned* head, int* hit_tri, float4* hit_out, uint4* stats_out, uint2* spill,
                          unsigned spill_stride, unsigned* ctrl, unsigned long long* ray_counter, const unsigned* cancel_host) {
    if (stats_out)
        hipLaunchKernelGGL((k_trace_closest<TRACE_BLOCK, TRACE_LDS, true>), dim3(grid), dim3(TRACE_BLOCK), 0, s, sc, rayO, rayD, t_max_opt, count_ptr,
                           head, hit_tri, hit_out, stats_out, spill, spill_stride, ctrl, ray_counter);
    else {
        const bool api = t_max_opt != nullptr || hit_out != nullptr;
#define YK_LAUNCH_CLOSEST(SPH, API, WIDE)                                                                                                              \
    hipLaunchKernelGGL((k_trace_closest_pt<TRACE_BLOCK, TRACE_LDS, TRACE_PF_MIN, TRACE_START_MIN, TRACE_LEAF_MIN, TRACE_CHUNK, SPH, API, WIDE>), dim3(grid), \
                       dim3(TRACE_BLOCK), 0, s, sc, rayO, rayD, t_max_opt, count_ptr, head, hit_tri, hit_out, spill, spill_stride, ctrl, ray_counter, cancel_host)
#define YK_LAUNCH_CLOSEST_W(SPH, API) \
    if (sc.nodes4) YK_LAUNCH_CLOSEST(SPH, API, true); else YK_LAUNCH_CLOSEST(SPH, API, false)
        if (sc.spheres) {
            if (api) { YK_LAUNCH_CLOSEST_W(true, true); } else { YK_LAUNCH_CLOSEST_W(true, false); }
        } else {
            if (api) { YK_LAUNCH_CLOSEST_W(false, true); } else { YK_LAUNCH_CLOSEST_W(false, false); }
        }
#undef YK_LAUNCH_CLOSEST_W
#undef YK_LAUNCH_CLOSEST
    }
}
void launch_whitted(hipStream_t s, unsigned grid, const DevScene& sc, const RenderParams& prm, const uint32_t* pixel_xy, const uint32_t* sample_index_tab,
                    PathBuffers cur, uint32_t n, float4* sample_buf, uint2* spill, unsigned spill_stride, unsigned* ctrl, unsigned long long* counters) {
    hipLaunchKernelGGL((k_whitted<TRACE_BLOCK, TRACE_LDS>), dim3(grid), dim3(TRACE_BLOCK), 0, s, sc, prm, pixel_xy, sample_index_tab, cur, n, sample_buf, spill,
                       spill_stride, ctrl, counters);
}
unsigned whitted_max_depth() { return YK_WHITTED_MAX_DEPTH; }
void launch_trace_any(hipStream_t s, unsigned grid, const DevScene& sc, const float4* shO, const float4* shD, const unsigned* slot_of,
                      const unsigned* count_ptr, unsigned* head, unsigned char* vis, uint2* spill, unsigned spill_stride, unsigned* ctrl,
                      unsigned long long* shadow_counter, const unsigned* cancel_host) {
#define YK_LAUNCH_ANY(SPH, WIDE)                                                                                                                       \
    hipLaunchKernelGGL((k_trace_any_pt<TRACE_BLOCK, TRACE_ANY_LDS, TRACE_PF_MIN, TRACE_START_MIN, TRACE_LEAF_MIN, TRACE_CHUNK, SPH, WIDE>), dim3(grid),     \
                       dim3(TRACE_BLOCK), 0, s, sc, shO, shD, slot_of, count_ptr, head, vis, spill, spill_stride, ctrl, shadow_counter, cancel_host)
    if (sc.spheres) {
        if (sc.nodes4) YK_LAUNCH_ANY(true, true); else YK_LAUNCH_ANY(true, false);
    } else {
        if (sc.nodes4) YK_LAUNCH_ANY(false, true); else YK_LAUNCH_ANY(false, false);
    }
#undef YK_LAUNCH_ANY
}

}  // namespace yk
