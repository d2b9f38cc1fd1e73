// trace_experiments.h — tools only (never part of the product build): hooks for the sensitivity builds of the persistent
// traversal kernels.  tools/build_variant.sh <name> -DYK_TRACE_EXPERIMENTS -DTRACE_X_LOADS   re-issues the node's four
// 16-byte loads; -DTRACE_X_VALU=<n> adds n dependent VALU instructions per node step (+41 % vs +5 %).
#pragma once
// (included from inside namespace yk by yk_trace.hip)
__device__ __forceinline__ void yk_experiment_node(const void* p, float sink) {
#ifdef TRACE_X_LOADS
    float4 d0, d1, d2, d3;
    asm volatile(
        "global_load_dwordx4 %0, %4, off\n global_load_dwordx4 %1, %4, off offset:16\n global_load_dwordx4 %2, %4, off offset:32\n"
        "global_load_dwordx4 %3, %4, off offset:48\n s_waitcnt vmcnt(0)"
        : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
        : "v"(p)
        : "memory");
#endif
#ifdef TRACE_X_VALU
#pragma unroll
    for (int k = 0; k < TRACE_X_VALU; ++k) asm volatile("v_add_f32 %0, %0, %0" : "+v"(sink));
#endif
}
#define YK_EXPERIMENT_NODE(p, nb) yk_experiment_node((const void*)(p), (nb).lo0.x)

