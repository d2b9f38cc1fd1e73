// gather_bench: how fast can a wave fetch 64 random 64-byte records (one per lane)?
//   mode 0: every lane issues 4 x global_load_dwordx4 on its own record (what traversal does)
//   mode 1: quad-cooperative — in instruction k lane l fetches piece (l%4) of the record of lane
//           16k + l/4, so the four lanes of a quad hit one 64-byte record
//   mode 2: as 1, then exchanged through LDS (ds_write_b128 / ds_read_b128) so that every lane
//           ends up with its own record
//   modes 3-7 (mode 0 with part of the wave masked off; what does a partially filled gather cost?):
//     3: lanes 0-31   4: even lanes   5: lanes 0-15   6: one lane per quad (lane % 4 == 0)   7: a random half, new every iteration
//   modes 8-10 (mode 0 with lanes SHARING records: does the TA merge equal addresses?):
//     8: the whole wave fetches one record   9: groups of 8 lanes share a record   10: pairs   11: groups of 4   12: groups of 16   13: groups of 32
// Table size is a parameter (fits L2 / Infinity Cache / HBM).  Prints records/s (records actually fetched).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned rng(unsigned& s) {
    s = s * 1664525u + 1013904223u;
    return s >> 8;
}

template <int MODE> __global__ __launch_bounds__(256, 6) void k(const float4* __restrict__ table, unsigned n_rec, unsigned iters, float* out) {
    __shared__ float4 stage[MODE == 2 ? 1024 : 1];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (unsigned it = 0; it < iters; ++it) {
        unsigned idx = rng(s) % n_rec;
        float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;
        if (MODE >= 8) {
            const unsigned share = MODE == 8 ? 0u : MODE == 9 ? (lane & ~7u) : MODE == 10 ? (lane & ~1u) : MODE == 11 ? (lane & ~3u) : MODE == 12 ? (lane & ~15u) : (lane & ~31u);
            idx = __shfl(idx, (int)share);
            const float4* p = table + (size_t)idx * 4;
            r0 = p[0]; r1 = p[1]; r2 = p[2]; r3 = p[3];
        } else if (MODE >= 3) {
            const bool on = MODE == 3 ? lane < 32u : MODE == 4 ? (lane & 1u) == 0u : MODE == 5 ? lane < 16u : MODE == 6 ? (lane & 3u) == 0u : ((idx >> 13) & 1u) != 0u;
            if (on) {
                const float4* p = table + (size_t)idx * 4;
                r0 = p[0]; r1 = p[1]; r2 = p[2]; r3 = p[3];
            }
        } else if (MODE == 0) {
            const float4* p = table + (size_t)idx * 4;
            r0 = p[0]; r1 = p[1]; r2 = p[2]; r3 = p[3];
        } else {
            unsigned i0 = __shfl(idx, (int)(lane >> 2)), i1 = __shfl(idx, (int)(16 + (lane >> 2))), i2 = __shfl(idx, (int)(32 + (lane >> 2))),
                     i3 = __shfl(idx, (int)(48 + (lane >> 2)));
            const unsigned piece = lane & 3u;
            r0 = table[(size_t)i0 * 4 + piece];
            r1 = table[(size_t)i1 * 4 + piece];
            r2 = table[(size_t)i2 * 4 + piece];
            r3 = table[(size_t)i3 * 4 + piece];
            if (MODE == 2) {
                float4* st = stage + wave * 256;
                st[lane] = r0; st[64 + lane] = r1; st[128 + lane] = r2; st[192 + lane] = r3;
                __builtin_amdgcn_wave_barrier();
                r0 = st[4 * lane]; r1 = st[4 * lane + 1]; r2 = st[4 * lane + 2]; r3 = st[4 * lane + 3];
                __builtin_amdgcn_wave_barrier();
            }
        }
        acc += r0.x + r1.y + r2.z + r3.w;
        s ^= __float_as_uint(acc) & 1u;  // dependent chain like a traversal step
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
    size_t kb = argc > 1 ? atoi(argv[1]) : 65536; size_t mb = kb / 1024;
    unsigned iters = argc > 2 ? atoi(argv[2]) : 2000;
    unsigned n_rec = (unsigned)(kb * 1024 / 64);
    float4* table;
    float* out;
    hipMalloc(&table, (size_t)n_rec * 64);
    hipMemset(table, 0, (size_t)n_rec * 64);
    const int grid = 256 * 6;
    hipMalloc(&out, grid * 256 * 4);
    const double active[14] = {1.0, 1.0, 1.0, 0.5, 0.5, 0.25, 0.25, 0.5, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0};
    for (int mode = 0; mode < 14; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t a, b;
            hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, table, n_rec, iters, out);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, table, n_rec, iters, out);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, table, n_rec, iters, out);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, table, n_rec, iters, out);
            if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, table, n_rec, iters, out);
            if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(grid), dim3(256), 0, 0, table, n_rec, iters, out);
            if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(grid), dim3(256), 0, 0, table, n_rec, iters, out);
            if (mode == 7) hipLaunchKernelGGL(k<7>, dim3(grid), dim3(256), 0, 0, table, n_rec, iters, out);
            if (mode == 8) hipLaunchKernelGGL(k<8>, dim3(grid), dim3(256), 0, 0, table, n_rec, iters, out);
            if (mode == 9) hipLaunchKernelGGL(k<9>, dim3(grid), dim3(256), 0, 0, table, n_rec, iters, out);
            if (mode == 10) hipLaunchKernelGGL(k<10>, dim3(grid), dim3(256), 0, 0, table, n_rec, iters, out);
            if (mode == 11) hipLaunchKernelGGL(k<11>, dim3(grid), dim3(256), 0, 0, table, n_rec, iters, out);
            if (mode == 12) hipLaunchKernelGGL(k<12>, dim3(grid), dim3(256), 0, 0, table, n_rec, iters, out);
            if (mode == 13) hipLaunchKernelGGL(k<13>, dim3(grid), dim3(256), 0, 0, table, n_rec, iters, out);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (rep) printf("table %zu KB (%zu MB) mode %d: %.3f ms  %.2f G records/s  %.2f TB/s  %.2f G wave-instr/s\n", kb, mb, mode, ms, active[mode] * grid * 256 * iters / ms * 1e-6,
                            active[mode] * grid * 256 * iters * 64 / ms * 1e-9, (double)grid * 4 * iters * 4 / ms * 1e-6);
        }
    }
    return 0;
}
