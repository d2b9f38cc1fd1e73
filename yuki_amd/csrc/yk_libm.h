// yk_libm.h — the transcendental functions the Path hot path needs, bit for bit the
// functions the reference calls, on the host and on gfx950.
//
// The reference calls Rust's f32::{sin,cos,tan,ln,acos,atan2} (call sites:
// sampling/mod.rs:86, bsdfs/mod.rs:280, trowbridge_reitz.rs:24, camera.rs:64,
// spot_light.rs:33-34, sphere.rs:90,102,111), i.e. the platform libm — on x86-64 Linux
// glibc's sinf / cosf / tanf / logf / acosf / atan2f.  A one-ulp difference in a sampled
// direction flips hemisphere / Russian-roulette branches and changes a whole path, so
// "some correct libm" is not enough: these are glibc 2.35's algorithms themselves, every
// operation in the order (and with the fusing) of its x86-64 build — binary64 multiply /
// fused multiply-add / conversions for sin, cos, log and the reduction of tan; binary32
// + - * / sqrt (correctly rounded on x86-64 and on CDNA4, -ffp-contract=off keeps them
// unfused) for acos, atan, atan2 and the tan kernel.  oracle/olibm.h holds the same
// restatement for the CPU oracle and is checked against the platform's functions for all
// 2^32 arguments (tools/micro/glibc_libm_check.cpp, profiles/r03_glibc_libm_check.txt);
// the device instance of this file is compared with the oracle's on the GPU
// (tests/test_gpu_stages.py).  A bounce needs two or three of these calls: negligible
// next to BVH traversal.
#pragma once
#include "yk_math.h"

namespace yk {

// ---------------------------------------------------------------------------------------------------------------
// sinf / cosf: glibc's own algorithm (s_sinf.c / s_cosf.c / sincosf.h: the single-precision routines glibc ships since 2.28), in the
// operation order of its x86-64 FMA build — the function Rust's f32::sin / f32::cos resolve to on every Linux machine with FMA3.
// Exact arithmetic only (binary64 multiply, fused multiply-add, conversions; integer arithmetic in the large-argument reduction):
// the same bits on the host and on gfx950.  The oracle's copy (oracle/olibm.h) equals the platform's functions for all 2^32
// arguments (profiles/r03_glibc_sincos_check.txt); this one equals the oracle's for all 2^32 arguments on the device
// (tests/test_gpu_stages.py).  The second coefficient table of glibc (quadrants whose result is negated) is the first with the
// cosine coefficients negated — negation is exact, so `neg ? -c : c` is that table.
YK_HD float gl_poly_sin(double xs, double x2) {
    double s1 = fma(x2, -0x1.994eb3774cf24p-13, 0x1.1107605230bc4p-7);
    double x3 = x2 * xs;
    double x7 = x2 * x3;
    double s = fma(x3, -0x1.555545995a603p-3, xs);
    return (float)fma(s1, x7, s);
}
YK_HD float gl_poly_cos(double x2, bool neg) {
    const double c0 = neg ? -0x1p0 : 0x1p0, c1 = neg ? 0x1.ffffffd0c621cp-2 : -0x1.ffffffd0c621cp-2, c2 = neg ? -0x1.55553e1068f19p-5 : 0x1.55553e1068f19p-5,
                 c3 = neg ? 0x1.6c087e89a359dp-10 : -0x1.6c087e89a359dp-10, c4 = neg ? -0x1.99343027bf8c3p-16 : 0x1.99343027bf8c3p-16;
    double x4 = x2 * x2;
    double a = fma(x2, c1, c0);
    double b = fma(x2, c4, c3);
    double x6 = x2 * x4;
    double c = fma(x4, c2, a);
    return (float)fma(b, x6, c);
}
// reduce_fast (|x| < 120): n = round(x / (pi/2)) through the 2^24-scaled product; x - n * pi/2 in one fused step
YK_HD double gl_reduce_fast(double x, int& n) {
    double r = x * 0x1.45F306DC9C883p+23;
    n = ((int)r + 0x800000) >> 24;
    return fma(-(double)n, 0x1.921FB54442D18p0, x);
}
// __inv_pio4[k]: 32 bits of 4/pi ending at byte k of its bit string (a2 f9 83 6e 4e 44 15 29 fc 27 57 d1 f5 34 dd c0 db 62 95 99 3c 43 90 41)
YK_HD unsigned gl_inv_pio4(int k) {
    const unsigned long long w0 = 0xa2f9836e4e441529ull, w1 = 0xfc2757d1f534ddc0ull, w2 = 0xdb6295993c439041ull;
    unsigned v = 0;
    for (int j = 3; j >= 0; --j) {
        const int i = k - j;  // byte index, most significant first
        unsigned byte = 0;
        if (i >= 0) {
            const unsigned long long w = i < 8 ? w0 : (i < 16 ? w1 : w2);
            byte = (unsigned)(w >> (8 * (7 - (i & 7)))) & 0xffu;
        }
        v = (v << 8) | byte;
    }
    return v;
}
// reduce_large (120 <= |x| < inf): the 24 mantissa bits times 96 bits of 4/pi
YK_HD double gl_reduce_large(unsigned xi, int& n) {
    const int k = (int)((xi >> 26) & 15u);
    const int shift = (int)((xi >> 23) & 7u);
    xi = (xi & 0xffffffu) | 0x800000u;
    xi <<= shift;
    unsigned long long res0 = (unsigned long long)(unsigned)(xi * gl_inv_pio4(k));
    const unsigned long long res1 = (unsigned long long)xi * gl_inv_pio4(k + 4);
    const unsigned long long res2 = (unsigned long long)xi * gl_inv_pio4(k + 8);
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    const unsigned long long nn = (res0 + (1ull << 61)) >> 62;
    res0 -= nn << 62;
    n = (int)nn;
    return (double)(long long)res0 * 0x1.921FB54442D18p-62;
}
// both functions: quadrant n, sign of the reduced argument, which polynomial
YK_HD float gl_sincosf(float y, int want_cos) {
    union {
        float f;
        unsigned u;
    } cv;
    cv.f = y;
    const unsigned xi = cv.u, top = (xi >> 20) & 0x7ffu;
    double x = (double)y;
    if (top < 0x3f4u) {  // |y| < pi/4
        if (top < 0x398u) return want_cos ? 1.0f : y;  // |y| < 2^-12
        return want_cos ? gl_poly_cos(x * x, false) : gl_poly_sin(x, x * x);
    }
    int n, sign = 0;
    if (top < 0x42fu) {
        x = gl_reduce_fast(x, n);
    } else if (top < 0x7f8u) {
        sign = (int)(xi >> 31);
        x = gl_reduce_large(xi, n);
    } else {
        return y - y;  // inf, NaN
    }
    const int q = (n + sign) & 3;
    const double s = (q == 1 || q == 2) ? -1.0 : 1.0;
    const bool neg = ((n + sign) & 2) != 0;
    const bool use_cos = ((n ^ want_cos) & 1) != 0;
    return use_cos ? gl_poly_cos(x * x, neg) : gl_poly_sin(x * s, x * x);
}

// sinf(y) and cosf(y) of the same argument with the reduction done once: the two functions reduce y identically (same quadrant n,
// same remainder), so sharing it changes no bit — what differs per function is only which polynomial serves which quadrant.
// Below 120 there is ONE path for all lanes: glibc's shortcut for |y| < pi/4 is its general path with n = 0 (reduce_fast then
// returns y itself: fma(-0, pi/2, y)), so taking the general path there gives the same bits without the lanes of a wave parting
// ways at pi/4 — the disk sample's theta straddles it.  |y| < 2^-12 keeps glibc's exact answers (y and 1).
YK_HD void gl_sincosf_pair(float y, float& sin_out, float& cos_out) {
    const unsigned xi = __builtin_bit_cast(unsigned, y), top = (xi >> 20) & 0x7ffu;
    double x = (double)y;
    int n, sign = 0;
    if (top < 0x42fu) {
        x = gl_reduce_fast(x, n);
    } else if (top < 0x7f8u) {
        sign = (int)(xi >> 31);
        x = gl_reduce_large(xi, n);
    } else {
        sin_out = cos_out = y - y;
        return;
    }
    const int q = (n + sign) & 3;
    const double s = (q == 1 || q == 2) ? -1.0 : 1.0;
    const bool neg = ((n + sign) & 2) != 0;
    const double x2 = x * x;
    const float by_sin = gl_poly_sin(x * s, x2), by_cos = gl_poly_cos(x2, neg);
    // sinf takes the cosine polynomial in odd quadrants; cosf is sinf one quadrant on (sinf_poly(.., n ^ 1)); signs and table follow n + sign in both
    const bool tiny = top < 0x398u;
    sin_out = tiny ? y : ((n & 1) ? by_cos : by_sin);
    cos_out = tiny ? 1.0f : ((n & 1) ? by_sin : by_cos);
}

YK_HD float det_sinf(float xf) {
#if defined(YK_ABLATE_LIBM) && defined(__HIP_DEVICE_COMPILE__)  // timing builds only: the hardware approximations instead
    return __sinf(xf);
#endif
    return gl_sincosf(xf, 0);
}

// sin and cos of one angle (the disk sample's theta, the microfacet normal's phi)
YK_HD void det_sincosf(float xf, float& s, float& c) {
#if defined(YK_ABLATE_LIBM) && defined(__HIP_DEVICE_COMPILE__)
    s = __sinf(xf);
    c = __cosf(xf);
    return;
#endif
    gl_sincosf_pair(xf, s, c);
}

YK_HD float det_cosf(float xf) {
#if defined(YK_ABLATE_LIBM) && defined(__HIP_DEVICE_COMPILE__)  // timing builds only: the hardware approximations instead
    return __cosf(xf);
#endif
    return gl_sincosf(xf, 1);
}

YK_HD float gl_from_bits(unsigned b) { return __builtin_bit_cast(float, b); }
YK_HD unsigned gl_to_bits(float f) { return __builtin_bit_cast(unsigned, f); }

// logf (e_logf.c of glibc >= 2.27, the `__logf_fma` build): 16-entry table of 1/c and log c, a cubic in binary64, one rounding
YK_HD void gl_logf_entry(int i, double& invc, double& logc) {
    switch (i) {
        case 0: invc = 0x1.661ec79f8f3bep+0; logc = -0x1.57bf7808caadep-2; break;
        case 1: invc = 0x1.571ed4aaf883dp+0; logc = -0x1.2bef0a7c06ddbp-2; break;
        case 2: invc = 0x1.49539f0f010bp+0; logc = -0x1.01eae7f513a67p-2; break;
        case 3: invc = 0x1.3c995b0b80385p+0; logc = -0x1.b31d8a68224e9p-3; break;
        case 4: invc = 0x1.30d190c8864a5p+0; logc = -0x1.6574f0ac07758p-3; break;
        case 5: invc = 0x1.25e227b0b8eap+0; logc = -0x1.1aa2bc79c81p-3; break;
        case 6: invc = 0x1.1bb4a4a1a343fp+0; logc = -0x1.a4e76ce8c0e5ep-4; break;
        case 7: invc = 0x1.12358f08ae5bap+0; logc = -0x1.1973c5a611cccp-4; break;
        case 8: invc = 0x1.0953f419900a7p+0; logc = -0x1.252f438e10c1ep-5; break;
        case 9: invc = 0x1p+0; logc = 0x0p+0; break;
        case 10: invc = 0x1.e608cfd9a47acp-1; logc = 0x1.aa5aa5df25984p-5; break;
        case 11: invc = 0x1.ca4b31f026aap-1; logc = 0x1.c5e53aa362eb4p-4; break;
        case 12: invc = 0x1.b2036576afce6p-1; logc = 0x1.526e57720db08p-3; break;
        case 13: invc = 0x1.9c2d163a1aa2dp-1; logc = 0x1.bc2860d22477p-3; break;
        case 14: invc = 0x1.886e6037841edp-1; logc = 0x1.1058bc8a07ee1p-2; break;
        default: invc = 0x1.767dcf5534862p-1; logc = 0x1.4043057b6ee09p-2; break;
    }
}
YK_HD float gl_logf(float xf) {
    unsigned ix = gl_to_bits(xf);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {  // zero, subnormal, negative, inf, NaN
        if (ix * 2u == 0u) return -1.0f / 0.0f;
        if (ix == 0x7f800000u) return xf;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return (xf - xf) / 0.0f;
        ix = gl_to_bits(xf * 0x1p23f) - (23u << 23);
    }
    const unsigned tmp = ix - 0x3f330000u;
    const int k = (int)tmp >> 23;
    double invc, logc;
    gl_logf_entry((int)((tmp >> 19) & 15u), invc, logc);
    const double z = (double)gl_from_bits(ix - (tmp & 0xff800000u));
    const double r = fma(z, invc, -1.0);
    const double y0 = fma((double)k, 0x1.62e42fefa39efp-1, logc);
    const double r2 = r * r;
    double y = fma(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
    y = fma(-0x1.00ea348b88334p-2, r2, y);
    return (float)fma(y, r2, y0 + r);
}

// acosf (e_acosf.c, binary32 fdlibm): rational p/q on [0, 0.5], sqrt form with a split root above
YK_HD float gl_acos_ratio(float z) {
    const float p = z * (gl_from_bits(0x3e2aaaabu) +
                         z * (gl_from_bits(0xbea6b090u) +
                              z * (gl_from_bits(0x3e4e0aa8u) + z * (gl_from_bits(0xbd241146u) + z * (gl_from_bits(0x3a4f7f04u) + z * gl_from_bits(0x3811ef08u))))));
    const float q = 1.0f + z * (gl_from_bits(0xc019d139u) + z * (gl_from_bits(0x4001572du) + z * (gl_from_bits(0xbf303361u) + z * gl_from_bits(0x3d9dc62eu))));
    return p / q;
}
YK_HD float gl_acosf(float x) {
    const float pi = gl_from_bits(0x40490fdau), pio2_hi = gl_from_bits(0x3fc90fdau), pio2_lo = gl_from_bits(0x33a22168u);
    const int hx = (int)gl_to_bits(x), ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;
    if (ix > 0x3f800000) return (x - x) / (x - x);
    if (ix < 0x3f000000) {
        if (ix <= 0x32800000) return pio2_hi + pio2_lo;
        const float r = gl_acos_ratio(x * x);
        return pio2_hi - (x - (pio2_lo - r * x));
    }
    if (hx < 0) {
        const float z = (1.0f + x) * 0.5f;
        const float r = gl_acos_ratio(z);
        const float s = sqrtf(z);
        const float w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    }
    const float z = (1.0f - x) * 0.5f;
    const float s = sqrtf(z);
    const float df = gl_from_bits(gl_to_bits(s) & 0xfffff000u);
    const float c = (z - df * df) / (s + df);
    const float r = gl_acos_ratio(z);
    const float w = r * s + c;
    return 2.0f * (df + w);
}

// atanf (s_atanf.c): four breakpoints, odd / even halves of an 11-term polynomial
YK_HD float gl_atanf(float x) {
    const int hx = (int)gl_to_bits(x), ix = hx & 0x7fffffff;
    float hi = 0.0f, lo = 0.0f;
    bool reduced = true;
    if (ix >= 0x4c000000) {  // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? gl_from_bits(0x3fc90fdau) + gl_from_bits(0x33a22168u) : -gl_from_bits(0x3fc90fdau) - gl_from_bits(0x33a22168u);
    }
    if (ix < 0x3ee00000) {
        if (ix < 0x31000000) return x;
        reduced = false;
    } else {
        x = fabsf(x);
        if (ix < 0x3f300000) {
            hi = gl_from_bits(0x3eed6338u);
            lo = gl_from_bits(0x31ac3769u);
            x = (2.0f * x - 1.0f) / (2.0f + x);
        } else if (ix < 0x3f980000) {
            hi = gl_from_bits(0x3f490fdau);
            lo = gl_from_bits(0x33222168u);
            x = (x - 1.0f) / (x + 1.0f);
        } else if (ix < 0x401c0000) {
            hi = gl_from_bits(0x3f7b985eu);
            lo = gl_from_bits(0x33140fb4u);
            x = (x - 1.5f) / (1.0f + 1.5f * x);
        } else {
            hi = gl_from_bits(0x3fc90fdau);
            lo = gl_from_bits(0x33a22168u);
            x = -1.0f / x;
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (gl_from_bits(0x3eaaaaabu) +
                          w * (gl_from_bits(0x3e124925u) +
                               w * (gl_from_bits(0x3dba2e6eu) + w * (gl_from_bits(0x3d886b35u) + w * (gl_from_bits(0x3d4bda59u) + w * gl_from_bits(0x3c8569d7u))))));
    const float s2 = w * (gl_from_bits(0xbe4ccccdu) +
                          w * (gl_from_bits(0xbde38e38u) + w * (gl_from_bits(0xbd9d8795u) + w * (gl_from_bits(0xbd6ef16bu) + w * gl_from_bits(0xbd15a221u)))));
    if (!reduced) return x - x * (s1 + s2);
    const float r = hi - ((x * (s1 + s2) - lo) - x);
    return hx < 0 ? -r : r;
}

// atan2f (e_atan2f.c): the special cases, then atanf(|y / x|) moved to the quadrant
YK_HD float gl_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = gl_from_bits(0x3f490fdbu), pi_o_2 = gl_from_bits(0x3fc90fdbu), pi = gl_from_bits(0x40490fdbu),
                pi_lo = gl_from_bits(0xb3bbbd2eu);
    const int hx = (int)gl_to_bits(x), hy = (int)gl_to_bits(y), ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return gl_atanf(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) {
        if (m < 2) return y;
        return m == 2 ? pi + tiny : -pi - tiny;
    }
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            if (m == 0) return pi_o_4 + tiny;
            if (m == 1) return -pi_o_4 - tiny;
            if (m == 2) return 3.0f * pi_o_4 + tiny;
            return -3.0f * pi_o_4 - tiny;
        }
        if (m == 0) return 0.0f;
        if (m == 1) return -0.0f;
        if (m == 2) return pi + tiny;
        return -pi - tiny;
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60)
        z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60)
        z = 0.0f;
    else
        z = gl_atanf(fabsf(y / x));
    if (m == 0) return z;
    if (m == 1) return gl_from_bits(gl_to_bits(z) ^ 0x80000000u);
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}

// __kernel_tanf (k_tanf.c): tan(x + y) on |x + y| <= pi/4 for iy = 1, -1 / tan(x + y) for iy = -1
YK_HD float gl_kernel_tanf(float x, float y, int iy) {
    const float pio4 = gl_from_bits(0x3f490fdau), pio4lo = gl_from_bits(0x33222168u);
    const int hx = (int)gl_to_bits(x), ix = hx & 0x7fffffff;
    if (ix < 0x39000000) {  // |x| < 2^-13
        if ((ix | (iy + 1)) == 0) return 1.0f / fabsf(x);
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
        if (fabsf(x) < 0x1p-13f) return (float)((1 - ((hx >> 30) & 2)) * iy) * (1.0f - (float)(2 * iy) * x);
    }
    const float z = x * x;
    float w = z * z;
    float r = gl_from_bits(0x3e088889u) +
              w * (gl_from_bits(0x3cb327a4u) +
                   w * (gl_from_bits(0x3b6b6916u) + w * (gl_from_bits(0x3a1a26c8u) + w * (gl_from_bits(0x38a3f445u) + w * gl_from_bits(0xb79bae5fu)))));
    float v = z * (gl_from_bits(0x3d5d0dd1u) +
                   w * (gl_from_bits(0x3c11371fu) +
                        w * (gl_from_bits(0x3abede48u) + w * (gl_from_bits(0x398137b9u) + w * (gl_from_bits(0x3895c07au) + w * gl_from_bits(0x37d95384u))))));
    float s = z * x;
    r = y + z * (s * (r + v) + y);
    r += gl_from_bits(0x3eaaaaabu) * s;
    w = x + r;
    if (ix >= 0x3f2ca140) {
        v = (float)iy;
        return (float)(1 - ((hx >> 30) & 2)) * (v - 2.0f * (x - (w * w / (w + v) - r)));
    }
    if (iy == 1) return w;
    const float zz = gl_from_bits(gl_to_bits(w) & 0xfffff000u);
    v = r - (zz - x);
    const float a = -1.0f / w;
    const float t = gl_from_bits(gl_to_bits(a) & 0xfffff000u);
    s = 1.0f + t * zz;
    return t + a * (s + t * v);
}

// tanf (s_tanf.c of glibc 2.35): sincosf.h's reductions, unfused in this function, the remainder split into a binary32 head and tail
YK_HD float gl_tanf(float x) {
    const int hx = (int)gl_to_bits(x), ix = hx & 0x7fffffff;
    if (ix <= 0x3f490fda) return gl_kernel_tanf(x, 0.0f, 1);
    if (ix >= 0x7f800000) return x - x;
    double dx = (double)x;
    int n;
    if ((((unsigned)hx >> 20) & 0x7ffu) < 0x42fu) {  // |x| < 120
        const double r = dx * 0x1.45F306DC9C883p+23;
        n = ((int)r + 0x800000) >> 24;
        dx = dx - (double)n * 0x1.921FB54442D18p0;
    } else {
        dx = gl_reduce_large((unsigned)hx, n);
        if (hx < 0) dx = -dx;
    }
    const float y0 = (float)dx;
    const float y1 = (float)(dx - (double)y0);
    return gl_kernel_tanf(y0, y1, 1 - ((n & 1) << 1));
}

// expf (e_expf.c, the `__expf_fma` build; host side only — the pbrt loader's CIE fits): 2^(k/32) from a table, a cubic in binary64
YK_HD unsigned long long gl_exp2f_entry(unsigned i) {
    const unsigned long long tab[32] = {
    0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b, 0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1,
    0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d, 0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585,
    0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13, 0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
    0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487, 0x3fef7c97337b9b5f, 0x3fefa4afa2a490da, 0x3fefd0765b6e4540};
    return tab[i & 31u];
}
YK_HD float gl_expf(float x) {
    const unsigned ix = gl_to_bits(x), abstop = (ix >> 20) & 0x7ffu;
    const double xd = (double)x;
    if (abstop >= 0x42bu) {
        if (ix == 0xff800000u) return 0.0f;
        if (abstop >= 0x7f8u) return x + x;
        if (x > 0x1.62e42ep6f) return 0x1p97f * 0x1p97f;
        if (x < -0x1.9fe368p6f) return 0x1p-95f * 0x1p-95f;
        if (x < -0x1.9d1d9ep6f) return 0x1.4p-75f * 0x1.4p-75f;
    }
    const double shift = 0x1.8p+52, inv_ln2n = 0x1.71547652b82fep+5;
    double kd = fma(inv_ln2n, xd, shift);
    const unsigned long long ki = __builtin_bit_cast(unsigned long long, kd);
    kd -= shift;
    const double r = fma(inv_ln2n, xd, -kd);
    const double s = __builtin_bit_cast(double, gl_exp2f_entry((unsigned)ki) + (ki << 47));
    const double z = fma(0x1.c6af84b912394p-20, r, 0x1.ebfce50fac4f3p-13);
    const double r2 = r * r;
    double y = fma(0x1.62e42ff0c52d6p-6, r, 1.0);
    y = fma(z, r2, y);
    return (float)(y * s);
}

YK_HD float det_tanf(float xf) {
#if defined(YK_ABLATE_LIBM) && defined(__HIP_DEVICE_COMPILE__)  // timing builds only: the hardware approximations instead
    return __tanf(xf);
#endif
    return gl_tanf(xf);
}

YK_HD float det_logf(float xf) {
#if defined(YK_ABLATE_LIBM) && defined(__HIP_DEVICE_COMPILE__)  // timing builds only: the hardware approximations instead
    return __logf(xf);
#endif
    return gl_logf(xf);
}

YK_HD float det_atan2f(float yf, float xf) { return gl_atan2f(yf, xf); }

YK_HD float det_acosf(float xf) { return gl_acosf(xf); }

YK_HD float det_expf(float xf) { return gl_expf(xf); }

}  // namespace yk
