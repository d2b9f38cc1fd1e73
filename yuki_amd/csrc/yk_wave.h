// yk_wave.h — wave64 helpers shared by the kernels
#pragma once
#include "yk_device.h"

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


}  // namespace yk
