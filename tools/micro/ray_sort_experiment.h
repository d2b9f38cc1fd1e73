// ray_sort_experiment.h — TIMING EXPERIMENT, not part of the product build (tools/build_variant.sh sort -DYK_EXPERIMENT_SORT).
//
// continuation rays run at 3.7-4.6 Gray/s against 17.6 for the coherent camera rays.  How much of that gap
// does ORDER close?  After k_shade of bounce b the queue of bounce b + 1 (and the area-light shadow queue of bounce b) is sorted
// by a key built from the ray's origin cell (Morton code of the origin quantised in the scene's bounds) and its direction
// octant, with rocPRIM's radix sort — the best order a sort can give, at whatever it costs — and the records are permuted.  Per-ray
// results do not depend on the queue's order (the sample id travels with the ray), so images stay bit-identical.  The queue
// length is read back (a host synchronisation): the experiment measures kernel times, not the frame.
//   YK_SORT_BOUNCES=k   sort the queues entering bounces 1 .. k (default 0 = off)
//   YK_SORT_SHADOW=1    also sort the area-light shadow queue of bounces 0 .. k-1... (of every bounce < k)
//   YK_SORT_BITS=b      bits per axis of the Morton code (default 9)
//   YK_SORT_MODE=m      0: morton << 3 | octant   1: octant << 3b | morton   2: morton only   3: octant only
#pragma once
#include <rocprim/rocprim.hpp>

namespace yk_exp {

__device__ __forceinline__ unsigned spread3(unsigned x) {  // 10 bits -> every third bit
    x &= 0x3ffu;
    x = (x | (x << 16)) & 0x030000ffu;
    x = (x | (x << 8)) & 0x0300f00fu;
    x = (x | (x << 4)) & 0x030c30c3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

__device__ __forceinline__ unsigned ray_key(float4 o, float4 d, float3 lo, float3 scale, unsigned bits, unsigned mode) {
    const float m = (float)((1u << bits) - 1u);
    const unsigned qx = (unsigned)fminf(fmaxf((o.x - lo.x) * scale.x, 0.0f), m);
    const unsigned qy = (unsigned)fminf(fmaxf((o.y - lo.y) * scale.y, 0.0f), m);
    const unsigned qz = (unsigned)fminf(fmaxf((o.z - lo.z) * scale.z, 0.0f), m);
    const unsigned mort = (spread3(qx) | (spread3(qy) << 1) | (spread3(qz) << 2)) & ((1u << (3u * bits)) - 1u);
    const unsigned oct = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
    if (mode == 0) return (mort << 3) | oct;
    if (mode == 1) return (oct << (3u * bits)) | mort;
    if (mode == 2) return mort;
    return oct;
}

__global__ void k_keys(const float4* __restrict__ rayO, const float4* __restrict__ rayD, unsigned n, float3 lo, float3 scale, unsigned bits, unsigned mode,
                       unsigned* keys, unsigned* vals) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = ray_key(rayO[i], rayD[i], lo, scale, bits, mode);
    vals[i] = i;
}

__global__ void k_permute_paths(const unsigned* __restrict__ vals, unsigned n, yk::PathBuffers src, yk::PathBuffers dst) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned j = vals[i];
    dst.rayO[i] = src.rayO[j];
    dst.rayD[i] = src.rayD[j];
    dst.thru[i] = src.thru[j];
    dst.rngs[i] = src.rngs[j];
}

__global__ void k_permute_shadow(const unsigned* __restrict__ vals, unsigned n, const float4* sO, const float4* sD, const unsigned* sq, float4* dO, float4* dD,
                                 unsigned* dq) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned j = vals[i];
    dO[i] = sO[j];
    dD[i] = sD[j];
    dq[i] = sq[j];
}

struct Sorter {
    DevBuf keys[2], vals[2], temp, spare[4], shO, shD, shq;
    int bounces = 0, shadow = 0, bits = 9, mode = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    Sorter() {
        if (const char* s = std::getenv("YK_SORT_BOUNCES")) bounces = std::atoi(s);
        if (const char* s = std::getenv("YK_SORT_SHADOW")) shadow = std::atoi(s);
        if (const char* s = std::getenv("YK_SORT_BITS")) bits = std::min(std::max(std::atoi(s), 1), 9);
        if (const char* s = std::getenv("YK_SORT_MODE")) mode = std::atoi(s);
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
    }
    DevBuf xcd_heads;
    // YK_EXPERIMENT_XCD: eight zeroed queue heads for the next traversal launch on `st` (slot 0 closest-hit, 1 any-hit)
    unsigned* heads(hipStream_t st, int slot) {
        (void)xcd_heads.ensure(2 * 64);
        unsigned* h = xcd_heads.as<unsigned>() + 16 * slot;
        (void)hipMemsetAsync(h, 0, 64, st);
        return h;
    }
    unsigned end_bit() const { return mode == 3 ? 3u : (mode == 2 ? 3u * bits : 3u * bits + 3u); }
    // sorts (keys, vals) of n rays; returns the sorted index array
    const unsigned* sort(hipStream_t st, const float4* rayO, const float4* rayD, unsigned n, const yk::DevScene& sc) {
        (void)keys[0].ensure((size_t)n * 4);
        (void)keys[1].ensure((size_t)n * 4);
        (void)vals[0].ensure((size_t)n * 4);
        (void)vals[1].ensure((size_t)n * 4);
        const float m = (float)(1u << bits);
        const float3 lo = make_float3(sc.root_bmin[0], sc.root_bmin[1], sc.root_bmin[2]);
        const float3 scale = make_float3(m / (sc.root_bmax[0] - sc.root_bmin[0]), m / (sc.root_bmax[1] - sc.root_bmin[1]), m / (sc.root_bmax[2] - sc.root_bmin[2]));
        hipLaunchKernelGGL(k_keys, dim3((n + 255) / 256), dim3(256), 0, st, rayO, rayD, n, lo, scale, (unsigned)bits, (unsigned)mode, keys[0].as<unsigned>(),
                           vals[0].as<unsigned>());
        size_t bytes = 0;
        (void)rocprim::radix_sort_pairs(nullptr, bytes, keys[0].as<unsigned>(), keys[1].as<unsigned>(), vals[0].as<unsigned>(), vals[1].as<unsigned>(), (size_t)n, 0u,
                                        end_bit(), st);
        (void)temp.ensure(bytes);
        (void)rocprim::radix_sort_pairs(temp.p, bytes, keys[0].as<unsigned>(), keys[1].as<unsigned>(), vals[0].as<unsigned>(), vals[1].as<unsigned>(), (size_t)n, 0u,
                                        end_bit(), st);
        return vals[1].as<unsigned>();
    }
};

inline Sorter& sorter() {
    static Sorter s;
    return s;
}

}  // namespace yk_exp

// Expanded inside run_bounces (yk_render.cpp) after launch_shade of bounce b: uses its locals ctx, ws, st, bc, b, prm, pn, ds, cur.
#define YK_SORT_AFTER_SHADE \
        if (yk_exp::sorter().bounces > 0) { /* order the queues k_shade just wrote (synchronises: timing experiment) */ \
            yk_exp::Sorter& S = yk_exp::sorter(); \
            unsigned h[YK_CTRL_STRIDE + 1]; \
            (void)hipStreamSynchronize(st); \
            (void)hipMemcpy(h, bc, sizeof(h), hipMemcpyDeviceToHost); \
            const unsigned n_next = h[YK_CTRL_STRIDE], n_sh = h[YK_CTRL_SHQ]; \
            float ms_paths = 0.0f, ms_sh = 0.0f; \
            if ((int)(b + 1) <= S.bounces && b + 1 < prm.max_depth && n_next > 1) { \
                for (int k = 0; k < 4; ++k) (void)S.spare[k].ensure(ws.path[cur ^ 1u][k].bytes); \
                (void)hipEventRecord(S.e0, st); \
                const unsigned* order = S.sort(st, pn.rayO, pn.rayD, n_next, ds); \
                PathBuffers sp; \
                sp.rayO = S.spare[0].as<float4>(); \
                sp.rayD = S.spare[1].as<float4>(); \
                sp.thru = S.spare[2].as<float4>(); \
                sp.rngs = S.spare[3].as<uint4>(); \
                hipLaunchKernelGGL(yk_exp::k_permute_paths, dim3((n_next + 255) / 256), dim3(256), 0, st, order, n_next, pn, sp); \
                (void)hipEventRecord(S.e1, st); \
                (void)hipStreamSynchronize(st); \
                (void)hipEventElapsedTime(&ms_paths, S.e0, S.e1); \
                for (int k = 0; k < 4; ++k) std::swap(ws.path[cur ^ 1u][k], S.spare[k]); \
            } \
            if (S.shadow && (int)b < S.bounces && n_sh > 1) { \
                (void)S.shO.ensure(ws.shO.bytes); \
                (void)S.shD.ensure(ws.shD.bytes); \
                (void)S.shq.ensure(ws.shq.bytes); \
                (void)hipEventRecord(S.e0, st); \
                const unsigned* order = S.sort(st, ws.shO.as<float4>(), ws.shD.as<float4>(), n_sh, ds); \
                hipLaunchKernelGGL(yk_exp::k_permute_shadow, dim3((n_sh + 255) / 256), dim3(256), 0, st, order, n_sh, ws.shO.as<float4>(), ws.shD.as<float4>(), \
                                   ws.shq.as<unsigned>(), S.shO.as<float4>(), S.shD.as<float4>(), S.shq.as<unsigned>()); \
                (void)hipEventRecord(S.e1, st); \
                (void)hipStreamSynchronize(st); \
                (void)hipEventElapsedTime(&ms_sh, S.e0, S.e1); \
                std::swap(ws.shO, S.shO); \
                std::swap(ws.shD, S.shD); \
                std::swap(ws.shq, S.shq); \
            } \
            std::fprintf(stderr, "  sort after shade %u: next queue %u rays %.3f ms | area-light shadow queue %u rays %.3f ms\n", b, n_next, ms_paths, n_sh, ms_sh); \
        }
