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
#include "yk_shade.h"
#include "yk_wave.h"

namespace yk {

// ------------------------------------------------------------------ pixel table
// chunk-local pixel index -> pixel coordinates, tile-major / row-major in tile
// (the order Integrator::render visits them, integrators/mod.rs:145)
__global__ void k_pixel_table(const yk_tile* tiles, const uint32_t* tile_offset, uint32_t n_tiles, uint32_t n_pixels,
                              uint32_t* pixel_xy, const uint16_t* tile_sample, uint32_t* pixel_sample) {
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
    if (tile_sample) pixel_sample[i] = tile_sample[lo];  // FilmTile.sample of the pixel's tile (accumulating film)
}

// the same for a chunk of ONE tile — the reference's per-tile Integrator::render call: the tile travels as a kernel argument
// (no upload, no host synchronisation for the staging buffer)
__global__ void k_pixel_table_one(yk_tile t, uint32_t n_pixels, uint32_t* pixel_xy, uint32_t tile_sample, uint32_t* pixel_sample) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    uint32_t w = (uint32_t)t.x1 - t.x0;
    pixel_xy[i] = (t.x0 + i % w) | ((t.y0 + i / w) << 16);
    if (pixel_sample) pixel_sample[i] = tile_sample;
}

// per-pixel part of the camera samples' sampler start (yk_rng.h, PixelSampler): (stream.lo, stream.hi, hash0, -)
__global__ void k_pixel_sampler(SamplerCfg cfg, const uint32_t* pixel_xy, uint32_t n_pixels, uint4* pixel_aux) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    const uint32_t xy = pixel_xy[i];
    const PixelSampler ps = pixel_sampler(cfg, xy & 0xffffu, xy >> 16);
    pixel_aux[i] = make_uint4((unsigned)ps.stream, (unsigned)(ps.stream >> 32), ps.hash0, 0u);
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

// `pixel_sample` != null: the accumulating film (integrators/mod.rs:146-161) — prm.spe passes of ONE sample per
// pixel whose global index is the tile's FilmTile.sample; otherwise all spp samples.
__global__ void k_raygen(DevCamera cam, RenderParams prm, const uint32_t* pixel_xy, const uint32_t* pixel_sample, uint64_t work0, uint32_t n,
                         PathBuffers out, float4* sample_buf, unsigned* count, float4* lean_origin, const uint4* pixel_aux) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (cancel_raised(prm.cancel)) return;  // *count stays zero (the batch's control block was cleared): every later kernel of the batch finds an empty queue
    if (i == 0) *count = n;
    if (i >= n) return;
    const uint32_t w = (uint32_t)(work0 + i);  // a chunk holds fewer than 2^32 samples (it indexes sample_buf)
    uint32_t pix, ks;
    split_sample_id(w, prm.spe, pix, ks);
    const uint32_t s = (pixel_sample ? pixel_sample[pix] : prm.sample_base) + ks;
    uint32_t xy = pixel_xy[pix];
    uint32_t px = xy & 0xffffu, py = xy >> 16;
    SamplerState st;
    float ux, uy;
    if (pixel_aux) {  // the pixel's share of the work was done once per pixel (k_pixel_sampler)
        const uint4 a = pixel_aux[pix];
        PixelSampler ps;
        ps.stream = (u64)a.x | ((u64)a.y << 32);
        ps.hash0 = a.z;
        st = sampler_start_camera(prm.sampler, ps, px, py, s, ux, uy);
    } else {
        st = sampler_start(prm.sampler, px, py, s, 0);
        sampler_get_2d(prm.sampler, st, ux, uy);
    }
    V3 o, d;
    camera_ray(cam, (float)px + ux, (float)py + uy, o, d);
    if (lean_origin) {  // yk_device.h, YK_CTRL_CAM_O: one origin for all, throughput one, two sampler dimensions drawn
        if (i == 0) *lean_origin = make_float4(o.x, o.y, o.z, __uint_as_float(0u));
    } else {
        out.rayO[i] = make_float4(o.x, o.y, o.z, __uint_as_float(0u));
        out.thru[i] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(st.dimension));
    }
    out.rayD[i] = make_float4(d.x, d.y, d.z, __uint_as_float((uint32_t)w));
    out.rngs[i] = make_uint4((unsigned)st.rng.state, (unsigned)(st.rng.state >> 32), (unsigned)st.rng.inc, (unsigned)(st.rng.inc >> 32));
    // The Path integrator's first accumulate pass writes every sample of the batch (every camera ray hits or misses) and
    // starts from zero itself; only where no such pass follows does the slot have to be cleared here.
    if (prm.integrator != YK_INTEGRATOR_PATH || prm.max_depth == 0) sample_buf[w] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// Integrator::li entry: caller-supplied rays (yk_li)
__global__ void k_raygen_user(RenderParams prm, const float* ray_o, const float* ray_d, const uint16_t* pixel, const uint32_t* sample_index,
                              uint32_t dimension, uint32_t n, PathBuffers out, float4* sample_buf, uint32_t* pixel_xy, unsigned* count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *count = n;
    if (i >= n) return;
    uint32_t px = pixel[2 * i], py = pixel[2 * i + 1];
    pixel_xy[i] = px | (py << 16);
    SamplerState st = sampler_start(prm.sampler, px, py, sample_index[i], dimension);
    if (prm.sampler.kind == 1) st.dimension = dimension;  // caller already consumed `dimension` draws
    out.rayO[i] = make_float4(ray_o[3 * i], ray_o[3 * i + 1], ray_o[3 * i + 2], __uint_as_float(0u));
    out.rayD[i] = make_float4(ray_d[3 * i], ray_d[3 * i + 1], ray_d[3 * i + 2], __uint_as_float(i));
    out.thru[i] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(st.dimension));
    out.rngs[i] = make_uint4((unsigned)st.rng.state, (unsigned)(st.rng.state >> 32), (unsigned)st.rng.inc, (unsigned)(st.rng.inc >> 32));
    if (prm.integrator != YK_INTEGRATOR_PATH || prm.max_depth == 0) sample_buf[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// ------------------------------------------------------------------ shade
// Path::li_internal for one vertex of every active path (path.rs:89-169).
//
// Stream compaction: survivors and shadow rays are first appended to LDS staging
// buffers (wave ballot + one LDS atomic per wave) and flushed to HBM in dense runs
// of >= BLOCK entries with ONE global atomic per flush.  A global atomic per wave
// per append saturated the counter word (~90 M atomics/s, MI355X_MICROARCH.md
// "dequeue") and made this kernel atomic-bound (profiles/r01_a_*: 0.178 s/frame,
// 79 % of wave cycles waiting).
#ifndef SHADE_CAP
#define SHADE_CAP 335  // staged continuation paths per block (64 B each).  With SHADE_CAPQ, the sort window's order table and the round-3
                       // interruption flag the block's LDS is 53248 B: three blocks per CU (LDS is handed out in 1280-byte granules on gfx950:
                       // 42 granules = 53760 B x 3 fit 160 KB, one granule more does not).  335, not 336: the flag word would have made it
                       // 53312 B — inside the same 42 granules on paper, but only 53248 and 53296 are measured sizes
#endif
#ifndef SHADE_CAPQ
#define SHADE_CAPQ 768  // staged shadow rays per block (36 B each): a whole iteration of 256 paths x 3 lights fits
#endif
#ifndef SHADE_WIN
#define SHADE_WIN 8  // at most this many iterations (of 256 paths) are sorted by material kind together (the launch passes the actual number)
#endif
#ifndef SHADE_MIN_WAVES
#define SHADE_MIN_WAVES 3  // waves per SIMD the register allocator must leave room for (blocks of 256 threads)
#endif
template <int CAP, int CAPQ> struct ShadeStaging {
    float4 pO[CAP], pD[CAP], pT[CAP];
    uint4 pR[CAP];
    float4 qO[CAPQ], qD[CAPQ];
    unsigned qS[CAPQ];
    unsigned fill_p, fill_q, gbase;
    unsigned q_delta;  // class of the staged shadow rays: 1 = towards a point/spot/distant light
    unsigned cancel;   // the render was interrupted (yk_device.h, CancelRef): block-uniform copy of what thread 0 read
    unsigned bucket[8];                  // material-kind histogram of the window's paths
    unsigned short order[SHADE_WIN * 256];  // sorted position -> offset of the path inside the window
};

// every thread of the block calls this (converged); returns the staging position
__device__ __forceinline__ unsigned block_append(bool want, unsigned* lds_fill) {
    unsigned long long mask = __ballot(want);
    unsigned total = (unsigned)__popcll(mask);
    unsigned prefix = (unsigned)__popcll(mask & ((1ull << lane_id()) - 1ull));
    unsigned base = 0;
    if (total && lane_id() == 0) base = atomicAdd(lds_fill, total);
    base = __shfl(base, 0);
    return base + prefix;
}

__device__ __forceinline__ unsigned valid_light_kind(const DevScene& sc, unsigned l) { return sc.lights[l].kind; }

// Diagnostic builds only (tools/build_variant.sh prof -DYK_SHADE_PROFILE): per-wave cycle stamps around the stages of an
// iteration, summed into a device array (tools/micro/shade_profile.h).  The product build sees empty macros.
#ifdef YK_SHADE_PROFILE
#include "../../tools/micro/shade_profile.h"
#else
#define YK_PROF_DECL
#define YK_PROF_STAMP(k)
#define YK_PROF_FLUSH
#endif

template <int BLOCK, int CAP, int CAPQ>
__global__ __launch_bounds__(BLOCK, SHADE_MIN_WAVES) void k_shade(DevScene sc, RenderParams prm, const uint32_t* pixel_xy, const uint32_t* sample_index_tab,
                                                 PathBuffers cur, PathBuffers nxt,
                                                 const int* hit_tri, float4* pend, float4* shO, float4* shD, float4* shC, unsigned char* vis,
                                                 unsigned* shq, float4* shO2, float4* shD2, unsigned* shq2, unsigned* bc, unsigned split_delta, unsigned reorder,
                                                 unsigned win_max, unsigned block_slots, unsigned sid_base, const float4* __restrict__ lean_origin) {
    // `bc`: this bounce's words of the control block (yk_device.h); the next bounce's follow it
    // iterations per window: a full queue sorts win_max (<= SHADE_WIN) x 256 paths together; a queue too short to give every
    // resident block (`block_slots` of them on the device) a full window takes shorter ones, down to one iteration
    const unsigned win_iters = min(win_max, max(1u, bc[0] / (BLOCK * block_slots)));
    static_assert(sizeof(ShadeStaging<CAP, CAPQ>) <= 53296, "k_shade: more LDS than the measured 53296 bytes per block (42 granules = 53760 cost the third block per CU)");
    __shared__ ShadeStaging<CAP, CAPQ> stg;
    const unsigned nl = sc.n_lights;
    unsigned* next_count = bc + YK_CTRL_STRIDE;
    unsigned* shq_count = bc + YK_CTRL_SHQ;
    if (threadIdx.x == 0) {
        stg.fill_p = 0;
        stg.fill_q = 0;
        stg.q_delta = 0;
        stg.cancel = cancel_raised(prm.cancel) ? 1u : 0u;
    }
    __syncthreads();
    // an interrupted render: the queue counts as empty for the WHOLE block (the barriers below need block-uniform control flow,
    // so the word is read by one thread and shared through LDS) and nothing is appended for the next bounce; a block lives for
    // one or two windows of a long launch (the grid has 256 blocks per CU), so this is also the launch's look "per window".
    // (readfirstlane: an LDS read is per-lane to the compiler, the queue length has to stay a scalar)
    const unsigned n = __builtin_amdgcn_readfirstlane((int)stg.cancel) ? 0u : bc[0];
    // flush helpers (block-uniform control flow)
    // Shadow rays go to one of two queues: rays towards an area light leave a surface patch
    // in scattered directions, rays towards a point / spot / distant light converge on one
    // target and suit the wave-packet any-hit kernel.  The staging buffer only ever holds
    // rays of one class (it is flushed when the class of the light changes).
    auto flush_q = [&](unsigned fill, unsigned delta) {
        if (threadIdx.x == 0) stg.gbase = atomicAdd(delta ? bc + YK_CTRL_SHQ2 : shq_count, fill);
        __syncthreads();
        const unsigned gb = stg.gbase;
        float4* dO = delta ? shO2 : shO;
        float4* dD = delta ? shD2 : shD;
        unsigned* dS = delta ? shq2 : shq;
        for (unsigned k = threadIdx.x; k < fill; k += BLOCK) {
            dO[gb + k] = stg.qO[k];
            dD[gb + k] = stg.qD[k];
            dS[gb + k] = stg.qS[k];
        }
        __syncthreads();
        if (threadIdx.x == 0) stg.fill_q = 0;
        __syncthreads();
    };
    auto flush_p = [&](unsigned fill) {
        if (threadIdx.x == 0) stg.gbase = atomicAdd(next_count, fill);
        __syncthreads();
        const unsigned gb = stg.gbase;
        for (unsigned k = threadIdx.x; k < fill; k += BLOCK) {
            nxt.rayO[gb + k] = stg.pO[k];
            nxt.rayD[gb + k] = stg.pD[k];
            nxt.thru[gb + k] = stg.pT[k];
            nxt.rngs[gb + k] = stg.pR[k];
        }
        __syncthreads();
        if (threadIdx.x == 0) stg.fill_p = 0;
        __syncthreads();
    };
    // When a whole iteration's shadow rays fit the staging buffer and they all go to one queue,
    // the flush decisions are taken once per iteration instead of once per light (the barriers
    // of the per-light decisions cost more than the math between them).
    const bool per_iter = !split_delta && BLOCK * nl <= (unsigned)CAPQ;
    // all lanes stay in the loop together so the ballots below see whole waves
    const unsigned WIN = win_iters * BLOCK;  // paths per window
    const unsigned n_win = (n + WIN - 1) / WIN;
    YK_PROF_DECL
    for (unsigned w = blockIdx.x; w < n_win; w += gridDim.x) {
      const unsigned wbase = w * WIN;
      unsigned n_sub = win_iters;
      YK_PROF_STAMP(0)
      if (reorder) {
        // After the first bounce the 64 paths of a wave hit surfaces of five material kinds and every wave runs every
        // kind's BSDF code (PMC: 3750 VALU instructions per wave and vertex at 55 % lane utilisation against 2740 at
        // 91 % for the camera rays' vertices).  The paths of a window — SHADE_WIN iterations of the block — are therefore
        // dealt to the lanes sorted by the material kind of what they hit (the traversal kernel left the kind in the hit
        // word): with 4 x 256 paths and six keys most waves see ONE kind.  Which lane shades which path has no
        // influence on any result.  Counting sort: per key one ballot per wave and one LDS atomic per wave and key.
        if (threadIdx.x < 8) stg.bucket[threadIdx.x] = 0;
        __syncthreads();
        unsigned key[SHADE_WIN], rank[SHADE_WIN];
#pragma unroll
        for (int k = 0; k < SHADE_WIN; ++k) {
            const unsigned i0 = wbase + k * BLOCK + threadIdx.x;
            const bool in = (unsigned)k < win_iters && i0 < n;
            const int t = in ? hit_tri[i0] : -1;
            key[k] = !in ? 7u : (t < 0 ? 6u : ((unsigned)t >> YK_HIT_KIND_SHIFT));
        }
#pragma unroll
        for (int k = 0; k < SHADE_WIN; ++k) {
            rank[k] = 0;
            if ((unsigned)k >= win_iters) continue;  // block-uniform
            for (unsigned q = 0; q < 8; ++q) {  // wave-uniform loop: every lane takes part in every ballot
                const unsigned long long m = __ballot(key[k] == q);
                if (m == 0ull) continue;
                unsigned b0 = 0;
                if (lane_id() == 0) b0 = atomicAdd(&stg.bucket[q], (unsigned)__popcll(m));
                b0 = __shfl(b0, 0);
                if (key[k] == q) rank[k] = b0 + (unsigned)__popcll(m & ((1ull << lane_id()) - 1ull));
            }
        }
        __syncthreads();
        unsigned first[8];
        {
            unsigned acc = 0;
            for (unsigned q = 0; q < 8; ++q) {
                first[q] = acc;
                acc += stg.bucket[q];
            }
            n_sub = (acc - stg.bucket[7] + BLOCK - 1) / BLOCK;  // the invalid entries sort last: trailing iterations without any path are skipped
        }
#pragma unroll
        for (int k = 0; k < SHADE_WIN; ++k)
            if ((unsigned)k < win_iters) stg.order[first[key[k]] + rank[k]] = (unsigned short)(k * BLOCK + threadIdx.x);
        __syncthreads();
      } else {
        const unsigned left = n - wbase;  // w < n_win: at least one path
        n_sub = left >= WIN ? win_iters : (left + BLOCK - 1) / BLOCK;
      }
      for (unsigned sub = 0; sub < n_sub; ++sub) {
        const unsigned i = wbase + (reorder ? (unsigned)stg.order[sub * BLOCK + threadIdx.x] : sub * BLOCK + threadIdx.x);
        const bool valid = i < n;
        bool alive = false;
        YK_PROF_STAMP(1)
        if (per_iter) {
            // ONE block-wide decision per iteration: room for every shadow ray (BLOCK x nl) and every
            // continuation (BLOCK) this iteration can stage, so the appends below need no barriers
            __syncthreads();
            const unsigned fq = stg.fill_q, fp = stg.fill_p, staged = stg.q_delta;
            __syncthreads();
            if (fq && fq + BLOCK * nl > CAPQ) flush_q(fq, staged);
            if (fp + BLOCK > CAP) flush_p(fp);
        }
        float4 nO = make_float4(0, 0, 0, 0), nD = nO, nT = nO;
        uint4 nR = make_uint4(0, 0, 0, 0);
        V3 o = V3{0, 0, 0}, d = V3{0, 0, 1};
        RGB beta = RGB{0, 0, 0};
        unsigned flags = 0, sid = 0, bounces = 0;
        bool specular_bounce = false, hit = false;
        SamplerState st;
        st.rng.state = 0;
        st.rng.inc = 1;
        st.px = st.py = st.sample_index = st.dimension = 0;
        PathVertex v;  // lanes without a hit never read it; the defaults keep their registers defined
        v.sf.p = v.sf.n = v.sf.ns = v.sf.dpdus = v.sf.wo = V3{0, 0, 1};
        v.sf.material = 0;
        v.sf.area_light = -1;
        v.sf.u = v.sf.v = 0.0f;
        v.mat.kind = MK_BLACK;
        v.fr.s = v.fr.t = v.fr.n = v.fr.ng = V3{0, 0, 1};
        v.wo = V3{0, 0, 1};
        if (valid) {
            // lean camera bounce (yk_device.h, YK_CTRL_CAM_O): the shared origin, throughput one, the camera sample's two dimensions
            float4 a, c;
            const float4 b = cur.rayD[i];
            if (lean_origin) {  // kernel argument: a wave-uniform branch
                a = *lean_origin;
                c = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(2u));
            } else {
                a = cur.rayO[i];
                c = cur.thru[i];
            }
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
            uint32_t pix, ks;  // entry of the pixel table and sample within it (yk_device.h: RenderParams::spe)
            split_sample_id(sid, prm.spe, pix, ks);
            uint32_t xy = pixel_xy[pix];
            st.px = xy & 0xffffu;
            st.py = xy >> 16;
            st.sample_index = (sample_index_tab ? sample_index_tab[pix] : prm.sample_base) + ks;
            int tri = hit_tri[i];
            hit = tri >= 0;
            if (hit) vertex_setup(sc, (uint32_t)tri & YK_HIT_PRIM_MASK, o, d, v);  // the leaf-order slot reported by the render-loop trace kernels
        }
        YK_PROF_STAMP(2)
        // ---- next-event estimation over ALL lights (path.rs:102-119); two sampler
        // dimensions are consumed per light whether or not it contributes.
        // verdict bytes of a path: up to four lights share one word (written once after the loop, read once by k_accumulate)
        const unsigned vs = YK_VIS_STRIDE(nl);
        unsigned vword = 0u;
        for (unsigned l = 0; l < nl; ++l) {
            const unsigned slot = i * nl + l, vslot = i * vs + l;  // contribution slot / verdict byte (what the shadow queue carries)
            NeeSample ne;
            ne.want = false;
            ne.contrib = RGB{0, 0, 0};
            ne.so = V3{0, 0, 0};
            ne.sd = V3{0, 0, 1};
            ne.al = -1;
            if (valid && hit) ne = vertex_light(sc, prm, st, l, v);
            const bool want = ne.want;
            if (vs == 4u)
                vword |= (want ? 1u : 0u) << (8u * l);
            else if (valid)
                vis[vslot] = want ? 1 : 0;
            // shadow rays are appended densely (coalesced for the any-hit kernel); the
            // contribution stays at its (path, light) slot for `accumulate`
            if (!per_iter) {
                __syncthreads();
                const unsigned f = stg.fill_q, staged = stg.q_delta;
                const unsigned delta = (split_delta && valid_light_kind(sc, l) != YK_LIGHT_RECT) ? 1u : 0u;
                __syncthreads();  // every wave has read the same fill before any wave appends again
                if (f && (staged != delta || f + BLOCK > CAPQ)) flush_q(f, staged);
                if (threadIdx.x == 0) stg.q_delta = delta;
            }
            unsigned q = block_append(want, &stg.fill_q);
            if (want) {
                stg.qO[q] = make_float4(ne.so.x, ne.so.y, ne.so.z, 0.9999f);
                stg.qD[q] = make_float4(ne.sd.x, ne.sd.y, ne.sd.z, __uint_as_float((unsigned)ne.al));
                stg.qS[q] = vslot;
                shC[slot] = make_float4(ne.contrib.r, ne.contrib.g, ne.contrib.b, 0.0f);
            }
        }
        if (vs == 4u && valid) reinterpret_cast<unsigned*>(vis)[i] = vword;
        YK_PROF_STAMP(3)
        if (valid) {
            unsigned kind = 0;
            RGB term = RGB{0, 0, 0};
            if (!hit) {
                term = vertex_miss_term(sc, beta);
                kind = YK_PEND_MISS;
            } else {
                VertexEnd e = vertex_finish(sc, prm, st, v, beta, bounces, specular_bounce);
                term = e.term;
                kind = e.kind;
                alive = e.alive;
                if (e.sampled) {
                    nO = make_float4(e.no.x, e.no.y, e.no.z, __uint_as_float((bounces & 0xffu) | (specular_bounce ? 0x100u : 0u)));
                    nD = make_float4(e.wi.x, e.wi.y, e.wi.z, __uint_as_float(sid));
                    nT = make_float4(beta.r, beta.g, beta.b, __uint_as_float(st.dimension));
                    nR = make_uint4((unsigned)st.rng.state, (unsigned)(st.rng.state >> 32), (unsigned)st.rng.inc, (unsigned)(st.rng.inc >> 32));
                }
            }
            // the sample's slot travels with the term (as its offset in the batch, < 2^29): k_accumulate reads no path state for it
            pend[i] = make_float4(term.r, term.g, term.b, __uint_as_float((kind << YK_PEND_KIND_SHIFT) | (sid - sid_base)));
        }
        YK_PROF_STAMP(4)
        // ---- stream compaction of the survivors into the other buffer
        if (!per_iter) {
            __syncthreads();
            const unsigned f = stg.fill_p;
            __syncthreads();
            if (f + BLOCK > CAP) flush_p(f);
        }
        unsigned j = block_append(alive, &stg.fill_p);
        if (alive) {
            stg.pO[j] = nO;
            stg.pD[j] = nD;
            stg.pT[j] = nT;
            stg.pR[j] = nR;
        }
        YK_PROF_STAMP(5)
      }
      if (reorder) __syncthreads();  // the next window's sort overwrites `order`
    }
    YK_PROF_FLUSH
    __syncthreads();
    {
        const unsigned fq = stg.fill_q, fp = stg.fill_p;
        if (fq) flush_q(fq, stg.q_delta);
        if (fp) flush_p(fp);
    }
}

// ------------------------------------------------------------------ accumulate
// radiance = fold over lights (in light order) of the unoccluded contributions,
// + beta*Le, clamp, then incoming_radiance += beta * radiance   (path.rs:102-129)
// `first`: the camera bounce — the sample's radiance so far is zero (raygen does not clear the slot: 0 + x == x bit for bit, a
// stored +0 included), so the slot is written without being read.
__global__ void k_accumulate(RenderParams prm, PathBuffers cur, const float4* pend, const float4* shC, const unsigned char* vis, unsigned nl,
                             float4* sample_buf, const unsigned* bc, unsigned first, unsigned sid_base) {
    const unsigned n = cancel_raised(prm.cancel) ? 0u : bc[0];  // an interrupted k_shade left pending terms unwritten: nothing of this bounce is read
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 p = pend[i];
        const unsigned kind = __float_as_uint(p.w) >> YK_PEND_KIND_SHIFT;
        const unsigned sid = sid_base + (__float_as_uint(p.w) & YK_PEND_SID_MASK);
        RGB beta = RGB{1.0f, 1.0f, 1.0f};  // the throughput the vertex was entered with (one at the camera's); a miss adds its term as it is
        if (!first && !(kind & YK_PEND_MISS)) {
            const float4 c = cur.thru[i];
            beta = RGB{c.x, c.y, c.z};
        }
        RGB L = RGB{0.0f, 0.0f, 0.0f};
        if (!first) {
            const float4 acc = sample_buf[sid];
            L = RGB{acc.x, acc.y, acc.z};
        }
        RGB radiance = RGB{0.0f, 0.0f, 0.0f};
        if (!(kind & YK_PEND_MISS)) {
            const unsigned vs = YK_VIS_STRIDE(nl);
            const unsigned vword = vs == 4u ? reinterpret_cast<const unsigned*>(vis)[i] : 0u;  // the verdicts of up to four lights in one word
            for (unsigned l = 0; l < nl; ++l) {
                const unsigned slot = i * nl + l;
                const unsigned verdict = vs == 4u ? (vword >> (8u * l)) & 255u : (unsigned)vis[i * vs + l];
                if (verdict == 1u) {
                    float4 ct = shC[slot];
                    radiance = radiance + RGB{ct.x, ct.y, ct.z};
                }
            }
        }
        L = vertex_accumulate(prm, L, beta, radiance, RGB{p.x, p.y, p.z}, kind);
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
    // the sum keeps the reference's sample order (integrators/mod.rs:172); the loads of a group
    // of eight are issued together so the serial adds do not wait for one load each
    uint32_t k = 0;
    for (; k + 8 <= spp; k += 8) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = s[k + j];
#pragma unroll
        for (int j = 0; j < 8; ++j) color = color + RGB{v[j].x, v[j].y, v[j].z};
    }
    for (; k < spp; ++k) {
        float4 v = s[k];
        color = color + RGB{v.x, v.y, v.z};
    }
    color = color / (float)spp;
    out_rgb[3 * (size_t)p + 0] = color.r;
    out_rgb[3 * (size_t)p + 1] = color.g;
    out_rgb[3 * (size_t)p + 2] = color.b;
}

// Accumulating film, `n_passes` passes rendered at once: the raw value of every (pass, pixel),
// pass-major — what n_passes calls of Integrator::render(accumulating = true) with
// FilmTile.sample, sample + 1, ... would have written (integrators/mod.rs:146-161).
__global__ void k_resolve_passes(const float4* sample_buf, uint32_t n_pixels, uint32_t n_passes, float* out_rgb, size_t pass_stride) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_pixels * n_passes) return;
    const uint32_t p = (uint32_t)(i / n_passes), k = (uint32_t)(i % n_passes);
    const float4 v = sample_buf[i];
    float* o = out_rgb + (size_t)k * pass_stride + 3 * (size_t)p;
    o[0] = v.x;
    o[1] = v.y;
    o[2] = v.z;
}

// Film::update_tile on the device (film.rs:236-278): tile-major -> row-major film
__global__ void k_film_scatter(const uint32_t* pixel_xy, uint32_t n_pixels, const float* tile_rgb, uint32_t res_x, float* film_rgb, int accumulate,
                               uint32_t n_passes, size_t pass_stride) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pixels) return;
    uint32_t xy = pixel_xy[p];
    size_t dst = (size_t)(xy >> 16) * res_x + (xy & 0xffffu);
    if (accumulate) {  // film.rs:260-272: *fc += c, one pass after the other
        float r = film_rgb[3 * dst + 0], g = film_rgb[3 * dst + 1], b = film_rgb[3 * dst + 2];
        for (uint32_t k = 0; k < n_passes; ++k) {
            const float* t = tile_rgb + (size_t)k * pass_stride + 3 * (size_t)p;
            r += t[0];
            g += t[1];
            b += t[2];
        }
        film_rgb[3 * dst + 0] = r;
        film_rgb[3 * dst + 1] = g;
        film_rgb[3 * dst + 2] = b;
        return;
    }
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
        Surface sf = hit_surface_prim(sc, (uint32_t)tri & YK_HIT_PRIM_MASK, o, d);  // the normals integrators trace with the render-loop kernel: leaf-order slots
        V3 nn = integrator == YK_INTEGRATOR_GEOMETRY_NORMALS ? sf.n : sf.ns;
        c = RGB{nn.x, nn.y, nn.z} / 2.0f + 0.5f;
    }
    sample_buf[sid] = make_float4(c.r, c.g, c.b, 0.0f);
}

// ------------------------------------------------------------------ unit-test kernels
__global__ void k_device_math(int fn, size_t n, const float* a, const float* b, float* out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (fn >= 11 && fn <= 27) {  // vector functions of yk_math.h / yk_geom.h on triples: a, b = n/3 packed V3, thread 3k handles triple k
        if (i % 3) return;
        V3 va = V3{a[i], a[i + 1], a[i + 2]}, vb = b ? V3{b[i], b[i + 1], b[i + 2]} : V3{0.0f, 0.0f, 0.0f};
        V3 r = V3{0.0f, 0.0f, 0.0f};
        switch (fn) {
            case 11: r.x = dot(va, vb); break;
            case 12: r = cross(va, vb); break;
            case 13: r.x = length(va); break;
            case 14: r = normalize(va); break;
            case 15: r.x = (float)max_dimension(va); break;
            case 16: r = vabs(va); break;
            case 17: r.x = dot_nv(va, vb); break;
            case 18: {  // the permutation Triangle::intersect derives from the ray direction (triangle.rs:60-66)
                RayTri rt = ray_tri_setup(va);
                r = V3{(float)rt.kx, (float)rt.ky, (float)rt.kz};
                break;
            }
            case 19: r = V3{rmin(va.x, vb.x), rmin(va.y, vb.y), rmin(va.z, vb.z)}; break;
            case 20: r = V3{rmax(va.x, vb.x), rmax(va.y, vb.y), rmax(va.z, vb.z)}; break;
            case 21: r = faceforward_v(va, vb); break;
            // the operator traits of Vec3 / Point3 / Normal as the kernels use them (yk_math.h)
            case 22: r = va + vb; break;
            case 23: r = va - vb; break;
            case 24: r = va * vb.x; break;
            case 25: r = va / vb.x; break;
            case 26: r = -va; break;
            case 27: r.x = len_sqr(va); break;
            default: break;
        }
        out[i] = r.x;
        out[i + 1] = r.y;
        out[i + 2] = r.z;
        return;
    }
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
        case 28:  // det_sincosf (the reduction shared by both results): its sine ...
        case 29: {  // ... and its cosine
            float sn, cs;
            det_sincosf(x, sn, cs);
            r = fn == 28 ? sn : cs;
            break;
        }
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

// Light::sample_li + VisibilityTester::ray for n surface points (stage entry yk_light_sample)
__global__ void k_light_test(DevLight L, int index, size_t n, const float* p, const float* ng, const float* u, float* out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 sp = ld3(p, (uint32_t)i), nn = ld3(ng, (uint32_t)i);
    LightSample ls = sample_light(L, index, sp, u[2 * i], u[2 * i + 1]);
    // p0.spawn_ray_to(p1), interaction.rs:44-59 — as vertex_light does
    V3 offset = nn * 0.001f;
    V3 so = dot(ls.p1 - sp, nn) > 0.0f ? sp + offset : sp - offset;
    V3 sd = ls.p1 - so;
    float* o = out + 18 * i;
    o[0] = ls.l.x; o[1] = ls.l.y; o[2] = ls.l.z;
    o[3] = ls.li.r; o[4] = ls.li.g; o[5] = ls.li.b;
    o[6] = ls.pdf;
    o[7] = ls.has_vis ? 1.0f : 0.0f;
    o[8] = (float)ls.area_light;
    o[9] = ls.p1.x; o[10] = ls.p1.y; o[11] = ls.p1.z;
    o[12] = so.x; o[13] = so.y; o[14] = so.z;
    o[15] = sd.x; o[16] = sd.y; o[17] = sd.z;
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

#ifdef YK_SHADE_PROFILE
}  // namespace yk
extern "C" int yk_debug_shade_profile(unsigned long long* out8, int reset) {
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipDeviceSynchronize();
    int rc = (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(yk::yk_shade_prof), sizeof(zero));
    if (reset) rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(yk::yk_shade_prof), zero, sizeof(zero));
    return rc;
}
namespace yk {
#endif

// ------------------------------------------------------------------ launchers
static inline unsigned blocks_for(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

void launch_pixel_table(hipStream_t s, const yk_tile* tiles, const uint32_t* tile_offset, uint32_t n_tiles, uint32_t n_pixels, uint32_t* pixel_xy,
                        const uint16_t* tile_sample, uint32_t* pixel_sample) {
    if (!n_pixels) return;
    hipLaunchKernelGGL(k_pixel_table, dim3(blocks_for(n_pixels, 256)), dim3(256), 0, s, tiles, tile_offset, n_tiles, n_pixels, pixel_xy, tile_sample,
                       pixel_sample);
}
void launch_pixel_table_one(hipStream_t s, const yk_tile& tile, uint32_t n_pixels, uint32_t* pixel_xy, uint32_t tile_sample, uint32_t* pixel_sample) {
    if (!n_pixels) return;
    hipLaunchKernelGGL(k_pixel_table_one, dim3(blocks_for(n_pixels, 256)), dim3(256), 0, s, tile, n_pixels, pixel_xy, tile_sample, pixel_sample);
}
void launch_pixel_sampler(hipStream_t s, const SamplerCfg& cfg, const uint32_t* pixel_xy, uint32_t n_pixels, uint4* pixel_aux) {
    if (!n_pixels) return;
    hipLaunchKernelGGL(k_pixel_sampler, dim3(blocks_for(n_pixels, 256)), dim3(256), 0, s, cfg, pixel_xy, n_pixels, pixel_aux);
}
void launch_raygen(hipStream_t s, const DevCamera& cam, const RenderParams& prm, const uint32_t* pixel_xy, const uint32_t* pixel_sample, uint64_t work0,
                   uint32_t n, PathBuffers out, float4* sample_buf, unsigned* count, float4* lean_origin, const uint4* pixel_aux) {
    hipLaunchKernelGGL(k_raygen, dim3(blocks_for(n, 256)), dim3(256), 0, s, cam, prm, pixel_xy, pixel_sample, work0, n, out, sample_buf, count, lean_origin, pixel_aux);
}
void launch_raygen_user(hipStream_t s, const RenderParams& prm, const float* o, const float* d, const uint16_t* pixel, const uint32_t* sample_index,
                        uint32_t dimension, uint32_t n, PathBuffers out, float4* sample_buf, uint32_t* pixel_xy, unsigned* ctrl) {
    hipLaunchKernelGGL(k_raygen_user, dim3(blocks_for(n, 256)), dim3(256), 0, s, prm, o, d, pixel, sample_index, dimension, n, out, sample_buf,
                       pixel_xy, ctrl);
}
void launch_shade(hipStream_t s, unsigned grid, const DevScene& sc, const RenderParams& prm, const uint32_t* pixel_xy, const uint32_t* sample_index_tab,
                  PathBuffers cur, PathBuffers nxt,
                  const int* hit_tri, float4* pend, float4* shO, float4* shD, float4* shC, unsigned char* vis, unsigned* shq, float4* shO2,
                  float4* shD2, unsigned* shq2, unsigned* bc, unsigned split_delta, unsigned reorder, unsigned block_slots, unsigned sid_base, const float4* lean_origin) {
    hipLaunchKernelGGL((k_shade<256, SHADE_CAP, SHADE_CAPQ>), dim3(grid), dim3(256), 0, s, sc, prm, pixel_xy, sample_index_tab, cur, nxt, hit_tri, pend, shO, shD, shC, vis, shq,
                       shO2, shD2, shq2, bc, split_delta, reorder, (unsigned)SHADE_WIN, block_slots, sid_base, lean_origin);
}
void launch_accumulate(hipStream_t s, unsigned grid, const RenderParams& prm, PathBuffers cur, const float4* pend, const float4* shC,
                       const unsigned char* vis, unsigned nl, float4* sample_buf, const unsigned* bc, unsigned first, unsigned sid_base) {
    hipLaunchKernelGGL(k_accumulate, dim3(grid), dim3(256), 0, s, prm, cur, pend, shC, vis, nl, sample_buf, bc, first, sid_base);
}
void launch_resolve(hipStream_t s, const float4* sample_buf, uint32_t n_pixels, uint32_t spp, float* out_rgb) {
    if (!n_pixels) return;
    hipLaunchKernelGGL(k_resolve, dim3(blocks_for(n_pixels, 256)), dim3(256), 0, s, sample_buf, n_pixels, spp, out_rgb);
}
void launch_resolve_passes(hipStream_t s, const float4* sample_buf, uint32_t n_pixels, uint32_t n_passes, float* out_rgb, size_t pass_stride) {
    if (!n_pixels) return;
    hipLaunchKernelGGL(k_resolve_passes, dim3(blocks_for((size_t)n_pixels * n_passes, 256)), dim3(256), 0, s, sample_buf, n_pixels, n_passes, out_rgb, pass_stride);
}
void launch_film_scatter(hipStream_t s, const uint32_t* pixel_xy, uint32_t n_pixels, const float* tile_rgb, uint32_t res_x, float* film_rgb, int accumulate,
                         uint32_t n_passes, size_t pass_stride) {
    if (!n_pixels) return;
    hipLaunchKernelGGL(k_film_scatter, dim3(blocks_for(n_pixels, 256)), dim3(256), 0, s, pixel_xy, n_pixels, tile_rgb, res_x, film_rgb, accumulate, n_passes, pass_stride);
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
void launch_light_test(hipStream_t s, const DevLight& L, int index, size_t n, const float* p, const float* ng, const float* u, float* out) {
    hipLaunchKernelGGL(k_light_test, dim3(blocks_for(n, 256)), dim3(256), 0, s, L, index, n, p, ng, u, out);
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
