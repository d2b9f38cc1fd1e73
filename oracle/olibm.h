// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Single-precision transcendental functions: the platform libm of the reference, restated.
//
// The reference calls Rust's f32::{sin,cos,tan,ln,atan2,acos}, which lower to whatever libm
// the platform links; the reference itself does not pin their results.
// A path tracer's branches (hemisphere tests, Russian roulette) amplify a 1-ulp difference
// into a different path, so the oracle and the HIP kernels must use the *same* function —
// and to be the reference's image rather than "an" image it has to be the function the
// reference calls.  On x86-64 Linux that is glibc; namespace glibc below restates glibc
// 2.35's sinf, cosf, tanf, logf, expf, acosf, atanf and atan2f operation by operation (published
// sources: sysdeps/ieee754/flt-32/{s_sinf,s_cosf,s_tanf,k_tanf,e_logf,e_expf,e_acosf,s_atanf,
// e_atan2f}.c, sincosf.h; order and fusing read off the instructions of libm.so.6's x86-64
// build, FMA variants where glibc dispatches to them).  glibc is not under /root/reference
// — it is the reference's platform dependency — so the pin is the platform's own binary:
// tools/micro/glibc_libm_check.cpp compares every function with it for ALL 2^32 arguments
// (atan2f: every argument against 24 special partners + 2^32 random pairs); result in
// profiles/r03_glibc_libm_check.txt: identical, NaNs as a class.  tests/test_oracle_libm.py
// repeats a strided subset on every run.  On another libm (musl, macOS, a glibc before
// 2.28) the reference's own images differ from these in the last bits of a few samples —
// tools/libm_sensitivity.py measures by how much.
//
// No implicit FMA anywhere: compile with -ffp-contract=off; the fused operations of glibc's
// FMA builds are explicit std::fma calls.
//
// Build flavour -DORC_HOST_LIBM (liboracle_hostlibm.so, tools/libm_sensitivity.py): the six
// functions call the platform's sinf / cosf / tanf / logf / atan2f / acosf instead
// (call sites: sampling/mod.rs:62-87, trowbridge_reitz.rs:23-30,60-74, camera.rs:52-102,
// sphere.rs:38-119).  On glibc 2.35 it must render the oracle's images bit for bit (asserted
// in tests/test_oracle_sensitivity.py); on another platform it measures that platform's
// distance.  It is never the parity oracle.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {
namespace lm {

// ---------------------------------------------------------------------------------------------------------------
// sinf / cosf (s_sinf.c, s_cosf.c, sincosf.h — the single-precision routines of ARM's optimized-routines that glibc ships since 2.28).
// x86-64 glibc selects its FMA build at run time (`__sinf_fma`, ifunc) on every CPU with FMA3 + AVX2; which `a + b*c` are fused below
// is that build's: the polynomial helpers compile to exactly the fused multiply-adds the source's expressions suggest, the reduction
// `x - n*hpi` is one vfnmadd.  Constants: __sincosf_table, __inv_pio4.
namespace glibc {

// { sign[4], hpi_inv * 2^24, hpi, c0, c1, s1, c2, s2, c3, s3, c4 }; the second table serves the quadrants whose result is negated
static const double sincosf_table[2][14] = {
    {1.0, -1.0, -1.0, 1.0, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2, -0x1.555545995a603p-3, 0x1.55553e1068f19p-5,
     0x1.1107605230bc4p-7, -0x1.6c087e89a359dp-10, -0x1.994eb3774cf24p-13, 0x1.99343027bf8c3p-16},
    {1.0, -1.0, -1.0, 1.0, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.555545995a603p-3, -0x1.55553e1068f19p-5,
     0x1.1107605230bc4p-7, 0x1.6c087e89a359dp-10, -0x1.994eb3774cf24p-13, -0x1.99343027bf8c3p-16}};
// 4/pi as a bit string, 32 bits at every 8-bit offset (reduce_large)
static const uint32_t inv_pio4[24] = {0xa2,       0xa2f9,     0xa2f983,   0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529,
                                      0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1, 0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0,
                                      0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};

// sinf_poly, even quadrant: x + x^3 (s1 + x^2 (s2 + x^2 s3)) in this association; xs = x * sign
inline float poly_sin(double xs, double x2, const double* p) {
    double s1 = std::fma(x2, p[12], p[10]);
    double x3 = x2 * xs;
    double x7 = x2 * x3;
    double s = std::fma(x3, p[8], xs);
    return (float)std::fma(s1, x7, s);
}
// sinf_poly, odd quadrant: c0 + c1 x^2 + c2 x^4 + x^6 (c3 + c4 x^2)
inline float poly_cos(double x2, const double* p) {
    double x4 = x2 * x2;
    double c1 = std::fma(x2, p[7], p[6]);
    double c2 = std::fma(x2, p[13], p[11]);
    double x6 = x2 * x4;
    double c = std::fma(x4, p[9], c1);
    return (float)std::fma(c2, x6, c);
}
// reduce_fast: |x| < 120; n = round(x / (pi/2)) through the 2^24-scaled product, x - n * pi/2 in one fused step
inline double reduce_fast(double x, int& n) {
    double r = x * sincosf_table[0][4];
    n = ((int32_t)r + 0x800000) >> 24;
    return std::fma(-(double)n, sincosf_table[0][5], x);
}
// reduce_large: 120 <= |x| < inf, the argument's 24 mantissa bits times 96 bits of 4/pi
inline double reduce_large(uint32_t xi, int& n) {
    const uint32_t* arr = &inv_pio4[(xi >> 26) & 15];
    int shift = (int)((xi >> 23) & 7);
    xi = (xi & 0xffffff) | 0x800000;
    xi <<= shift;
    uint64_t res0 = (uint64_t)(uint32_t)(xi * arr[0]);
    uint64_t res1 = (uint64_t)xi * arr[4];
    uint64_t res2 = (uint64_t)xi * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    uint64_t nn = (res0 + (1ULL << 61)) >> 62;
    res0 -= nn << 62;
    n = (int)nn;
    return (double)(int64_t)res0 * 0x1.921FB54442D18p-62;
}
inline uint32_t abstop12(float y) {
    uint32_t b;
    std::memcpy(&b, &y, 4);
    return (b >> 20) & 0x7ff;
}

inline float sinf(float y) {
    double x = (double)y;
    const uint32_t top = abstop12(y);
    const double* p = sincosf_table[0];
    int n;
    if (top < 0x3f4) {  // |y| < pi/4
        double x2 = x * x;
        if (top < 0x398) return y;  // |y| < 2^-12
        return poly_sin(x, x2, p);
    }
    uint32_t xi;
    std::memcpy(&xi, &y, 4);
    int sign = 0;
    if (top < 0x42f) {  // |y| < 120
        x = reduce_fast(x, n);
    } else if (top < 0x7f8) {
        sign = (int)(xi >> 31);
        x = reduce_large(xi, n);
    } else {
        return y - y;  // inf, NaN: __math_invalidf
    }
    double s = p[(n + sign) & 3];
    if ((n + sign) & 2) p = sincosf_table[1];
    return (n & 1) ? poly_cos(x * x, p) : poly_sin(x * s, x * x, p);
}

inline float cosf(float y) {
    double x = (double)y;
    const uint32_t top = abstop12(y);
    const double* p = sincosf_table[0];
    int n;
    if (top < 0x3f4) {
        double x2 = x * x;
        if (top < 0x398) return 1.0f;
        return poly_cos(x2, p);
    }
    uint32_t xi;
    std::memcpy(&xi, &y, 4);
    int sign = 0;
    if (top < 0x42f) {
        x = reduce_fast(x, n);
    } else if (top < 0x7f8) {
        sign = (int)(xi >> 31);
        x = reduce_large(xi, n);
    } else {
        return y - y;
    }
    double s = p[(n + sign) & 3];
    if ((n + sign) & 2) p = sincosf_table[1];
    return (n & 1) ? poly_sin(x * s, x * x, p) : poly_cos(x * x, p);  // sinf_poly(.., n ^ 1)
}


inline float from_bits(uint32_t b) {
    float f;
    std::memcpy(&f, &b, 4);
    return f;
}
inline uint32_t to_bits(float f) {
    uint32_t b;
    std::memcpy(&b, &f, 4);
    return b;
}

// ---------------------------------------------------------------------------------------------------------------
// logf (sysdeps/ieee754/flt-32/e_logf.c + e_logf_data.c, the same ARM routine family; x86-64 dispatches to `__logf_fma`): a 16-entry
// table of 1/c and log(c), a cubic in binary64, one rounding to binary32.  Fused exactly where the FMA build fuses:
// r = fma(z, invc, -1), y0 = fma(k, ln2, logc), the cubic as two fma, the sum as one.
static const double logf_tab[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2}, {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},
    {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3}, {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4}, {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5},
    {0x1p+0, 0x0p+0},                              {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},   {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},
    {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
inline float logf(float xf) {
    uint32_t ix = to_bits(xf);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {  // zero, subnormal, negative, inf, NaN
        if (ix * 2u == 0u) return -1.0f / 0.0f;
        if (ix == 0x7f800000u) return xf;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return (xf - xf) / 0.0f;
        ix = to_bits(xf * 0x1p23f) - (23u << 23);
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const int k = (int32_t)tmp >> 23;
    const double z = (double)from_bits(ix - (tmp & 0xff800000u));
    const double r = std::fma(z, logf_tab[i][0], -1.0);
    const double y0 = std::fma((double)k, 0x1.62e42fefa39efp-1, logf_tab[i][1]);
    const double r2 = r * r;
    double y = std::fma(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
    y = std::fma(-0x1.00ea348b88334p-2, r2, y);
    return (float)std::fma(y, r2, y0 + r);
}

// ---------------------------------------------------------------------------------------------------------------
// acosf, atanf, atan2f, tanf: in glibc 2.35 these are still the binary32 fdlibm routines (e_acosf.c, s_atanf.c, e_atan2f.c,
// k_tanf.c; built for baseline x86-64, so no operation is fused), tanf with 2.35's argument reduction (s_tanf.c: the reduce_fast /
// reduce_large of sincosf.h, unfused here, the reduced argument split into a binary32 head and tail).  Plain binary32 + - * / sqrt
// in the order the instructions of libm.so.6 perform them.
inline float acosf(float x) {
    const float pi = from_bits(0x40490fdau), pio2_hi = from_bits(0x3fc90fdau), pio2_lo = from_bits(0x33a22168u);
    const float p0 = from_bits(0x3e2aaaabu), p1 = from_bits(0xbea6b090u), p2 = from_bits(0x3e4e0aa8u), p3 = from_bits(0xbd241146u),
                p4 = from_bits(0x3a4f7f04u), p5 = from_bits(0x3811ef08u);
    const float q1 = from_bits(0xc019d139u), q2 = from_bits(0x4001572du), q3 = from_bits(0xbf303361u), q4 = from_bits(0x3d9dc62eu);
    const int32_t hx = (int32_t)to_bits(x), ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;
    if (ix > 0x3f800000) return (x - x) / (x - x);
    if (ix < 0x3f000000) {  // |x| < 0.5
        if (ix <= 0x32800000) return pio2_hi + pio2_lo;
        const float z = x * x;
        const float p = z * (p0 + z * (p1 + z * (p2 + z * (p3 + z * (p4 + z * p5)))));
        const float q = 1.0f + z * (q1 + z * (q2 + z * (q3 + z * q4)));
        const float r = p / q;
        return pio2_hi - (x - (pio2_lo - r * x));
    }
    if (hx < 0) {  // x < -0.5
        const float z = (1.0f + x) * 0.5f;
        const float p = z * (p0 + z * (p1 + z * (p2 + z * (p3 + z * (p4 + z * p5)))));
        const float q = 1.0f + z * (q1 + z * (q2 + z * (q3 + z * q4)));
        const float s = std::sqrt(z);
        const float r = p / q;
        const float w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    }
    const float z = (1.0f - x) * 0.5f;  // x > 0.5
    const float s = std::sqrt(z);
    const float df = from_bits(to_bits(s) & 0xfffff000u);
    const float c = (z - df * df) / (s + df);
    const float p = z * (p0 + z * (p1 + z * (p2 + z * (p3 + z * (p4 + z * p5)))));
    const float q = 1.0f + z * (q1 + z * (q2 + z * (q3 + z * q4)));
    const float r = p / q;
    const float w = r * s + c;
    return 2.0f * (df + w);
}

inline float atanf(float x) {
    static const uint32_t hi_bits[4] = {0x3eed6338u, 0x3f490fdau, 0x3f7b985eu, 0x3fc90fdau};
    static const uint32_t lo_bits[4] = {0x31ac3769u, 0x33222168u, 0x33140fb4u, 0x33a22168u};
    const float a0 = from_bits(0x3eaaaaabu), a1 = from_bits(0xbe4ccccdu), a2 = from_bits(0x3e124925u), a3 = from_bits(0xbde38e38u),
                a4 = from_bits(0x3dba2e6eu), a5 = from_bits(0xbd9d8795u), a6 = from_bits(0x3d886b35u), a7 = from_bits(0xbd6ef16bu),
                a8 = from_bits(0x3d4bda59u), a9 = from_bits(0xbd15a221u), a10 = from_bits(0x3c8569d7u);
    const int32_t hx = (int32_t)to_bits(x), ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {  // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? from_bits(hi_bits[3]) + from_bits(lo_bits[3]) : -from_bits(hi_bits[3]) - from_bits(lo_bits[3]);
    }
    if (ix < 0x3ee00000) {  // |x| < 7/16
        if (ix < 0x31000000) return x;
        id = -1;
    } else {
        x = std::fabs(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) {
                id = 0;
                x = (2.0f * x - 1.0f) / (2.0f + x);
            } else {
                id = 1;
                x = (x - 1.0f) / (x + 1.0f);
            }
        } else if (ix < 0x401c0000) {
            id = 2;
            x = (x - 1.5f) / (1.0f + 1.5f * x);
        } else {
            id = 3;
            x = -1.0f / x;
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
    const float s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = from_bits(hi_bits[id]) - ((x * (s1 + s2) - from_bits(lo_bits[id])) - x);
    return hx < 0 ? -r : r;
}

inline float atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = from_bits(0x3f490fdbu), pi_o_2 = from_bits(0x3fc90fdbu), pi = from_bits(0x40490fdbu),
                pi_lo = from_bits(0xb3bbbd2eu);
    const int32_t hx = (int32_t)to_bits(x), hy = (int32_t)to_bits(y), ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return atanf(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);  // 2 * sign(x) + sign(y)
    if (iy == 0) {
        if (m < 2) return y;
        return m == 2 ? pi + tiny : -pi - tiny;
    }
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
                case 0: return pi_o_4 + tiny;
                case 1: return -pi_o_4 - tiny;
                case 2: return 3.0f * pi_o_4 + tiny;
                default: return -3.0f * pi_o_4 - tiny;
            }
        }
        switch (m) {
            case 0: return 0.0f;
            case 1: return -0.0f;
            case 2: return pi + tiny;
            default: return -pi - tiny;
        }
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60)
        z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60)
        z = 0.0f;
    else
        z = atanf(std::fabs(y / x));
    switch (m) {
        case 0: return z;
        case 1: return from_bits(to_bits(z) ^ 0x80000000u);
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

// __kernel_tanf: tan(x + y) for |x + y| <= pi/4 (iy = 1) or -1 / tan (iy = -1)
inline float kernel_tanf(float x, float y, int iy) {
    const float pio4 = from_bits(0x3f490fdau), pio4lo = from_bits(0x33222168u);
    const float t0 = from_bits(0x3eaaaaabu), t1 = from_bits(0x3e088889u), t2 = from_bits(0x3d5d0dd1u), t3 = from_bits(0x3cb327a4u),
                t4 = from_bits(0x3c11371fu), t5 = from_bits(0x3b6b6916u), t6 = from_bits(0x3abede48u), t7 = from_bits(0x3a1a26c8u),
                t8 = from_bits(0x398137b9u), t9 = from_bits(0x38a3f445u), t10 = from_bits(0x3895c07au), t11 = from_bits(0xb79bae5fu),
                t12 = from_bits(0x37d95384u);
    const int32_t hx = (int32_t)to_bits(x), ix = hx & 0x7fffffff;
    if (ix < 0x39000000) {  // |x| < 2^-13: (int)x == 0
        if ((ix | (iy + 1)) == 0) return 1.0f / std::fabs(x);
        return iy == 1 ? x : -1.0f / x;
    }
    if (ix >= 0x3f2ca140) {  // |x| >= 0.6744
        if (hx < 0) {
            x = -x;
            y = -y;
        }
        const float z = pio4 - x;
        const float w = pio4lo - y;
        x = z + w;
        y = 0.0f;
        if (std::fabs(x) < 0x1p-13f) return (float)((1 - ((hx >> 30) & 2)) * iy) * (1.0f - (float)(2 * iy) * x);
    }
    const float z = x * x;
    float w = z * z;
    float r = t1 + w * (t3 + w * (t5 + w * (t7 + w * (t9 + w * t11))));
    float v = z * (t2 + w * (t4 + w * (t6 + w * (t8 + w * (t10 + w * t12)))));
    float s = z * x;
    r = y + z * (s * (r + v) + y);
    r += t0 * s;
    w = x + r;
    if (ix >= 0x3f2ca140) {
        v = (float)iy;
        return (float)(1 - ((hx >> 30) & 2)) * (v - 2.0f * (x - (w * w / (w + v) - r)));
    }
    if (iy == 1) return w;
    const float zz = from_bits(to_bits(w) & 0xfffff000u);  // -1 / (x + r), accurately
    v = r - (zz - x);
    const float a = -1.0f / w;
    const float t = from_bits(to_bits(a) & 0xfffff000u);
    s = 1.0f + t * zz;
    return t + a * (s + t * v);
}

inline float tanf(float x) {
    const int32_t hx = (int32_t)to_bits(x), ix = hx & 0x7fffffff;
    if (ix <= 0x3f490fda) return kernel_tanf(x, 0.0f, 1);  // |x| <= pi/4
    if (ix >= 0x7f800000) return x - x;
    double dx = (double)x;
    int n;
    if (abstop12(x) < 0x42f) {  // |x| < 120: reduce_fast, NOT fused in this function
        const double r = dx * sincosf_table[0][4];
        n = ((int32_t)r + 0x800000) >> 24;
        dx = dx - (double)n * sincosf_table[0][5];
    } else {
        dx = reduce_large((uint32_t)hx, n);
        if (hx < 0) dx = -dx;
    }
    const float y0 = (float)dx;
    const float y1 = (float)(dx - (double)y0);
    return kernel_tanf(y0, y1, 1 - ((n & 1) << 1));
}

// ---------------------------------------------------------------------------------------------------------------
// expf (e_expf.c + e_exp2f_data.c, `__expf_fma`; the pbrt loader's CIE fits, pbrt/cie.rs:8-20): 2^(k/32) from a table, a cubic in
// binary64.  Fused where the FMA build fuses: kd = fma(InvLn2N, x, SHIFT) and r = fma(InvLn2N, x, -kd) — the product x * InvLn2N
// is never rounded on its own.
static const uint64_t exp2f_tab[32] = {
    0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b, 0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1,
    0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d, 0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585,
    0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13, 0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
    0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487, 0x3fef7c97337b9b5f, 0x3fefa4afa2a490da, 0x3fefd0765b6e4540};
inline float expf(float x) {
    const uint32_t ix = to_bits(x), abstop = (ix >> 20) & 0x7ffu;
    const double xd = (double)x;
    if (abstop >= 0x42bu) {  // |x| >= 88 or NaN
        if (ix == 0xff800000u) return 0.0f;
        if (abstop >= 0x7f8u) return x + x;
        if (x > 0x1.62e42ep6f) return 0x1p97f * 0x1p97f;            // overflow
        if (x < -0x1.9fe368p6f) return 0x1p-95f * 0x1p-95f;         // underflow to +0
        if (x < -0x1.9d1d9ep6f) return 0x1.4p-75f * 0x1.4p-75f;     // the smallest denormal
    }
    const double shift = 0x1.8p+52, inv_ln2n = 0x1.71547652b82fep+5;
    double kd = std::fma(inv_ln2n, xd, shift);
    uint64_t ki;
    std::memcpy(&ki, &kd, 8);
    kd -= shift;
    const double r = std::fma(inv_ln2n, xd, -kd);
    const uint64_t t = exp2f_tab[ki & 31u] + (ki << 47);
    double s;
    std::memcpy(&s, &t, 8);
    const double z = std::fma(0x1.c6af84b912394p-20, r, 0x1.ebfce50fac4f3p-13);
    const double r2 = r * r;
    double y = std::fma(0x1.62e42ff0c52d6p-6, r, 1.0);
    y = std::fma(z, r2, y);
    return (float)(y * s);
}

}  // namespace glibc

inline float sinf_(float xf) {
#if defined(ORC_HOST_LIBM) && (!defined(ORC_HOST_LIBM_ONLY) || ((ORC_HOST_LIBM_ONLY) & 1))
    return ::sinf(xf);  // sensitivity flavour: the platform libm, what Rust's f32 methods call on Linux
#endif
    return glibc::sinf(xf);  // glibc's algorithm restated: equal to the line above for all 2^32 arguments on glibc 2.35 / x86-64 with FMA
}

inline float cosf_(float xf) {
#if defined(ORC_HOST_LIBM) && (!defined(ORC_HOST_LIBM_ONLY) || ((ORC_HOST_LIBM_ONLY) & 1))
    return ::cosf(xf);
#endif
    return glibc::cosf(xf);
}

inline float tanf_(float xf) {
#if defined(ORC_HOST_LIBM) && (!defined(ORC_HOST_LIBM_ONLY) || ((ORC_HOST_LIBM_ONLY) & 2))
    return ::tanf(xf);
#endif
    return glibc::tanf(xf);
}

inline float logf_(float xf) {
#if defined(ORC_HOST_LIBM) && (!defined(ORC_HOST_LIBM_ONLY) || ((ORC_HOST_LIBM_ONLY) & 4))
    return ::logf(xf);
#endif
    return glibc::logf(xf);
}

inline float expf_(float xf) {
#if defined(ORC_HOST_LIBM) && (!defined(ORC_HOST_LIBM_ONLY) || ((ORC_HOST_LIBM_ONLY) & 16))
    return ::expf(xf);
#endif
    return glibc::expf(xf);
}

inline float atan2f_(float yf, float xf) {
#if defined(ORC_HOST_LIBM) && (!defined(ORC_HOST_LIBM_ONLY) || ((ORC_HOST_LIBM_ONLY) & 8))
    return ::atan2f(yf, xf);
#endif
    return glibc::atan2f(yf, xf);
}

inline float acosf_(float xf) {
#if defined(ORC_HOST_LIBM) && (!defined(ORC_HOST_LIBM_ONLY) || ((ORC_HOST_LIBM_ONLY) & 8))
    return ::acosf(xf);
#endif
    return glibc::acosf(xf);
}

}  // namespace lm
}  // namespace orc
