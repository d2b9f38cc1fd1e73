// yk_shade.h — one vertex of Path::li_internal (integrators/path.rs:89-169) as three inline steps.
//
// `k_shade` (yk_kernels.hip) runs them once per bounce over every in-flight path — vertex_setup, vertex_light for every
// light, vertex_finish — and `k_accumulate` ends the vertex with vertex_accumulate once the any-hit verdicts are in.  The
// order is the reference's: two sampler dimensions per light whether or not it contributes, then the BSDF sample, then
// Russian roulette; every expression keeps its operation order.  (A per-lane kernel that took the last few paths of a
// batch to their end with the same four steps was measured and dropped: profiles/r02_tail_kernel_sweep.txt)
#pragma once
#include "yk_device.h"
#include "yk_geom.h"
#include "yk_rng.h"

namespace yk {

// what path.rs:89-101 has in hand after `scene.bvh.intersect`: the SurfaceInteraction and its Bsdf
struct PathVertex {
    Surface sf;
    Material mat;
    Frame fr;
    V3 wo;
};

// `prim`: leaf-order slot of the primitive that was hit (what the render-loop traversal kernels report)
__device__ __forceinline__ void vertex_setup(const DevScene& sc, uint32_t prim, V3 o, V3 d, PathVertex& v) {
    v.sf = hit_surface_prim(sc, prim, o, d);
    v.mat = sc.materials[v.sf.material];
    if (v.mat.tex) {  // matte.rs:29-30: reflectance = kd.evaluate(si); no lobe when black
        RGB kd = texture_eval(sc, v.mat.tex - 1u, v.sf.u, v.sf.v);
        v.mat.a[0] = kd.r;
        v.mat.a[1] = kd.g;
        v.mat.a[2] = kd.b;
        if (is_black(kd)) v.mat.kind = MK_BLACK;
    }
    v.fr = make_frame(v.sf.n, v.sf.ns, v.sf.dpdus);
    v.wo = -d;
}

// next-event estimation towards light l (path.rs:102-119): draws its two sampler dimensions, and when the light
// contributes returns the contribution f * li * clamp(ns . l) / pdf and the shadow ray of its VisibilityTester
struct NeeSample {
    bool want;
    RGB contrib;
    V3 so, sd;
    int al;  // the sampled area light (its own surface does not occlude, bvh.rs:269-280) or -1
};
__device__ __forceinline__ NeeSample vertex_light(const DevScene& sc, const RenderParams& prm, SamplerState& st, unsigned l, const PathVertex& v) {
    NeeSample r;
    r.want = false;
    r.contrib = RGB{0, 0, 0};
    r.so = V3{0, 0, 0};
    r.sd = V3{0, 0, 1};
    r.al = -1;
    float ux, uy;
    sampler_get_2d(prm.sampler, st, ux, uy);
    LightSample ls = sample_light(sc.lights[l], (int)l, v.sf.p, ux, uy);
    if (!is_black(ls.li)) {
        RGB f = bsdf_f(v.mat, v.fr, v.sf.wo, ls.l);  // path.rs:105 uses si.wo
        if (ls.has_vis && !is_black(f)) {
            r.contrib = f * ls.li * rclamp(dot_nv(v.sf.ns, ls.l), 0.0f, 1.0f) / ls.pdf;
            // VisibilityTester::ray = p0.spawn_ray_to(p1), interaction.rs:44-59
            V3 offset = v.sf.n * 0.001f;
            r.so = dot(ls.p1 - v.sf.p, v.sf.n) > 0.0f ? v.sf.p + offset : v.sf.p - offset;
            r.sd = ls.p1 - r.so;
            r.al = ls.area_light;
            r.want = true;
        }
    }
    return r;
}

// kind bits of the pending term: 1 = miss, 2 = emission term present, 4 = indirect clamp applies
#define YK_PEND_MISS 1u
#define YK_PEND_EMISSION 2u
#define YK_PEND_CLAMP 4u
// pend[i].w = kind << 29 | (sample slot - first slot of the batch); batches hold at most 2^29 paths (yk_context_set_option)
#define YK_PEND_KIND_SHIFT 29
#define YK_PEND_SID_MASK 0x1fffffffu

// path.rs:155-160: incoming_radiance += beta * scene.background; break
__device__ __forceinline__ RGB vertex_miss_term(const DevScene& sc, RGB beta) { return beta * RGB{sc.background[0], sc.background[1], sc.background[2]}; }

// the rest of the vertex after the light loop (path.rs:121-169): emission term, BSDF sample, throughput, Russian roulette.
// In: beta / bounces / specular_bounce as they entered the vertex.  Out: the pending term and its kind bits; `alive` and
// the continuation (origin, direction, updated beta / bounces / specular flag) — the continuation state is written
// whenever the BSDF sample was usable, also when roulette or max_depth then end the path, as the wavefront stores it.
struct VertexEnd {
    RGB term;
    unsigned kind;
    bool alive, sampled;
    V3 no, wi;
};
__device__ __forceinline__ VertexEnd vertex_finish(const DevScene& sc, const RenderParams& prm, SamplerState& st, const PathVertex& v, RGB& beta, unsigned& bounces,
                                                    bool& specular_bounce) {
    VertexEnd e;
    e.term = RGB{0, 0, 0};
    e.kind = 0;
    e.alive = false;
    e.sampled = false;
    e.no = V3{0, 0, 0};
    e.wi = V3{0, 0, 1};
    if (bounces == 0 || specular_bounce) {  // path.rs:121-123
        RGB le = RGB{0, 0, 0};
        if (v.sf.area_light >= 0) {
            const DevLight& L = sc.lights[v.sf.area_light];
            le = dot_nv(v.sf.n, v.wo) > 0.0f ? RGB{L.i[0], L.i[1], L.i[2]} : RGB{0, 0, 0};  // rectangular_light.rs:75-81
        }
        e.term = beta * le;
        e.kind |= YK_PEND_EMISSION;
    }
    if (bounces > 0 && prm.has_clamp) e.kind |= YK_PEND_CLAMP;
    // path.rs:131-145
    float ux, uy;
    sampler_get_2d(prm.sampler, st, ux, uy);
    BsdfSample bs = bsdf_sample_f(v.mat, v.fr, v.wo, ux, uy);
    if (!(is_black(bs.f) || bs.pdf == 0.0f)) {
        specular_bounce = (bs.type & BX_SPECULAR) != 0;
        beta = beta * (bs.f * fabsf(dot_nv(bs.wi, v.sf.ns)) / bs.pdf);
        e.no = spawn_origin(v.sf.p, v.sf.n, bs.wi);
        e.wi = bs.wi;
        e.alive = true;
        e.sampled = true;
        // Russian roulette, path.rs:162-169
        if (bounces > 3) {
            float q = rmax(1.0f - beta.g, 0.05f);
            if (sampler_get_1d(prm.sampler, st) < q)
                e.alive = false;
            else
                beta = beta * (RGB{1.0f, 1.0f, 1.0f} / (1.0f - q));
        }
        bounces += 1;
        if (!(bounces < prm.max_depth)) e.alive = false;  // while bounces < max_depth
    }
    return e;
}

// `incoming_radiance += beta * radiance` of one vertex (path.rs:102-129), the fold a vertex ends with: `radiance` holds
// the unoccluded light contributions summed in light order; `beta` is the throughput the vertex was ENTERED with.
__device__ __forceinline__ RGB vertex_accumulate(const RenderParams& prm, RGB L, RGB beta, RGB radiance, RGB term, unsigned kind) {
    if (kind & YK_PEND_MISS) return L + term;
    if (kind & YK_PEND_EMISSION) radiance = radiance + term;
    if (kind & YK_PEND_CLAMP) radiance = rgb_min(radiance, RGB{1.0f, 1.0f, 1.0f} * prm.clamp);
    return L + beta * radiance;
}

}  // namespace yk
