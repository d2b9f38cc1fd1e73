// shade_profile.h — tools only: where does a k_shade wave spend its lifetime?  Included by yk_kernels.hip in
// -DYK_SHADE_PROFILE builds (tools/build_variant.sh prof -DYK_SHADE_PROFILE); never part of the product.
// Stamps: 0 iteration start | 1 after the material sort | 2 after state load + vertex_setup | 3 after the light loop
// (incl. its staging appends and flush decisions) | 4 after vertex_finish | 5 after the survivor compaction.
// yk_shade_prof[k] = shader cycles between stamp k-1 and k summed over all waves (k = 1..5), [0] = iterations x waves,
// [6] = barrier-free check: cycles of the stretch 1->2 .. 3->4 only.  Read with tools/shade_profile.py.
#pragma once
__device__ unsigned long long yk_shade_prof[8];
#define YK_PROF_DECL unsigned long long prof_t[6] = {0, 0, 0, 0, 0, 0}, prof_acc[6] = {0, 0, 0, 0, 0, 0};
#define YK_PROF_STAMP(k)                                            \
    prof_t[k] = __builtin_amdgcn_s_memtime();                       \
    if (k > 0) prof_acc[k] += prof_t[k] - prof_t[k - 1];            \
    else prof_acc[0] += 1;
#define YK_PROF_FLUSH                                                                      \
    if ((threadIdx.x & 63u) == 0u)                                                         \
        for (int k = 0; k < 6; ++k) atomicAdd(&yk_shade_prof[k], prof_acc[k]);
extern "C" int yk_debug_shade_profile(unsigned long long* out8, int reset);
