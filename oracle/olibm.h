// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Deterministic single-precision transcendental functions.
//
// The reference calls Rust's f32::{sin,cos,tan,ln,atan2,acos}, which lower to
// whatever libm the platform links (glibc, musl, MSVCRT ...): their results are
// NOT pinned by the reference ("parity unpinned").  A
// path tracer's branches (hemisphere tests, Russian roulette) amplify a 1-ulp
// difference into a different path, so the oracle and the HIP kernels must use
// the *same* function.  Both therefore implement this fixed recipe: evaluate in
// binary64 with the classic fdlibm minimax kernels using only +,-,*,/ and sqrt
// (all correctly rounded on the host and on gfx950), then round once to
// binary32.  The result is within 0.5 ulp + 2^-29 of the true value, i.e. a
// legitimate libm; tests/test_oracle_libm.py bounds the distance to glibc.
//
// No FMA anywhere: compile with -ffp-contract=off.
//
// Build flavour -DORC_HOST_LIBM (liboracle_hostlibm.so, tools/libm_sensitivity.py): the six
// functions call the platform's sinf / cosf / tanf / logf / atan2f / acosf instead — on
// x86-64 Linux with glibc that is what Rust's f32::{sin,cos,tan,ln,atan2,acos} resolve to
// (call sites: sampling/mod.rs:62-87, trowbridge_reitz.rs:23-30,60-74, camera.rs:52-102,
// sphere.rs:38-119).  It measures how far the fixed recipe moves an IMAGE from the one a
// glibc build of the reference would produce; it is never the parity oracle.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {
namespace lm {

inline double k_sin(double r) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = r * r;
    double p = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return r + (r * z) * (S1 + z * p);
}

inline double k_cos(double r) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = r * r;
    double p = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    return (1.0 - 0.5 * z) + z * p;
}

// r = x - n*(pi/2), n = nearest integer; three-part Cody-Waite in binary64.
// Accurate for |x| up to ~1e6 (the hot path feeds |x| <= 2*pi).
inline double reduce_pio2(double x, int64_t& n) {
    const double INV_PIO2 = 6.36619772367581382433e-01;
    const double P1 = 1.57079632673412561417e+00;   // first 33 bits of pi/2
    const double P2 = 6.07710050630396597660e-11;   // next 33 bits
    const double P3 = 2.02226624879595063154e-21;   // remainder
    double fn = std::floor(x * INV_PIO2 + 0.5);
    n = (int64_t)fn;
    double r = x - fn * P1;
    r = r - fn * P2;
    r = r - fn * P3;
    return r;
}

inline float sinf_(float xf) {
#ifdef ORC_HOST_LIBM
    return ::sinf(xf);  // sensitivity flavour: the platform libm, what Rust's f32 methods call on Linux
#endif
    double x = (double)xf;
    if (!(std::fabs(x) < 1.0e300)) return xf - xf;  // inf/NaN -> NaN
    int64_t n;
    double r = reduce_pio2(x, n);
    double v;
    switch (n & 3) {
        case 0: v = k_sin(r); break;
        case 1: v = k_cos(r); break;
        case 2: v = -k_sin(r); break;
        default: v = -k_cos(r); break;
    }
    return (float)v;
}

inline float cosf_(float xf) {
#ifdef ORC_HOST_LIBM
    return ::cosf(xf);  // sensitivity flavour: the platform libm, what Rust's f32 methods call on Linux
#endif
    double x = (double)xf;
    if (!(std::fabs(x) < 1.0e300)) return xf - xf;
    int64_t n;
    double r = reduce_pio2(x, n);
    double v;
    switch (n & 3) {
        case 0: v = k_cos(r); break;
        case 1: v = -k_sin(r); break;
        case 2: v = -k_cos(r); break;
        default: v = k_sin(r); break;
    }
    return (float)v;
}

inline float tanf_(float xf) {
#ifdef ORC_HOST_LIBM
    return ::tanf(xf);  // sensitivity flavour: the platform libm, what Rust's f32 methods call on Linux
#endif
    double x = (double)xf;
    if (!(std::fabs(x) < 1.0e300)) return xf - xf;
    int64_t n;
    double r = reduce_pio2(x, n);
    double s = k_sin(r), c = k_cos(r);
    double v = (n & 1) ? -(c / s) : (s / c);
    return (float)v;
}

// natural log of a positive finite binary32 value, evaluated in binary64
inline float logf_(float xf) {
#ifdef ORC_HOST_LIBM
    return ::logf(xf);  // sensitivity flavour: the platform libm, what Rust's f32 methods call on Linux
#endif
    if (xf != xf) return xf;
    if (xf < 0.0f) return (xf - xf) / 0.0f;                   // NaN
    if (xf == 0.0f) return -1.0f / 0.0f;                      // -inf (Rust ln(0) = -inf)
    if (xf > 3.0e38f && xf + xf == xf) return xf;             // +inf
    const double LN2 = 6.93147180559945286227e-01;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    double x = (double)xf;  // exact; subnormal floats become normal doubles
    uint64_t bits;
    std::memcpy(&bits, &x, 8);
    int64_t e = (int64_t)((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m;
    std::memcpy(&m, &bits, 8);  // m in [1,2)
    if (m > 1.41421356237309514547) {
        m = m * 0.5;
        e += 1;
    }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double R = z * (Lg1 + z * (Lg2 + z * (Lg3 + z * (Lg4 + z * (Lg5 + z * (Lg6 + z * Lg7))))));
    double lg = 2.0 * s + s * R;
    return (float)((double)e * LN2 + lg);
}

// atan on [0, inf) in binary64, fdlibm breakpoints
inline double k_atan_pos(double x) {
    const double atanhi[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01,
                              1.57079632679489655800e+00};
    const double atanlo[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17,
                              6.12323399573676603587e-17};
    const double aT[11] = {3.33333333333329318027e-01,  -1.99999999998764832476e-01, 1.42857142725034663711e-01,
                           -1.11111104054623557880e-01, 9.09088713343650656196e-02,  -7.69187620504482999495e-02,
                           6.66107313738753120669e-02,  -5.83357013379057348645e-02, 4.97687799461593236017e-02,
                           -3.65315727442169155270e-02, 1.62858201153657823623e-02};
    int id;
    if (x < 0.4375) {
        id = -1;
    } else if (x < 1.1875) {
        if (x < 0.6875) {
            id = 0;
            x = (2.0 * x - 1.0) / (2.0 + x);
        } else {
            id = 1;
            x = (x - 1.0) / (x + 1.0);
        }
    } else {
        if (x < 2.4375) {
            id = 2;
            x = (x - 1.5) / (1.0 + 1.5 * x);
        } else {
            id = 3;
            x = -1.0 / x;
        }
    }
    double z = x * x;
    double w = z * z;
    double s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    double s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    return atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
}

inline float atan2f_(float yf, float xf) {
#ifdef ORC_HOST_LIBM
    return ::atan2f(yf, xf);  // sensitivity flavour: the platform libm, what Rust's f32 methods call on Linux
#endif
    const double PI = 3.14159265358979311600e+00, PIO2 = 1.57079632679489655800e+00;
    if (xf != xf || yf != yf) return xf + yf;
    double y = (double)yf, x = (double)xf;
    if (y == 0.0) {
        bool xneg = std::signbit(xf);
        double v = xneg ? PI : 0.0;
        return (float)(std::signbit(yf) ? -v : v);
    }
    if (x == 0.0) return (float)(y > 0.0 ? PIO2 : -PIO2);
    double ax = std::fabs(x), ay = std::fabs(y);
    double a;
    if (ax > 1.0e300 && ay > 1.0e300)
        a = 7.85398163397448278999e-01;
    else if (ay > 1.0e300)
        a = PIO2;
    else if (ax > 1.0e300)
        a = 0.0;
    else
        a = k_atan_pos(ay / ax);
    if (x < 0.0) a = PI - a;
    return (float)(y < 0.0 ? -a : a);
}

inline float acosf_(float xf) {
#ifdef ORC_HOST_LIBM
    return ::acosf(xf);  // sensitivity flavour: the platform libm, what Rust's f32 methods call on Linux
#endif
    if (xf != xf) return xf;
    double x = (double)xf;
    if (x > 1.0 || x < -1.0) return (xf - xf) / (xf - xf);  // NaN
    // acos(x) = 2*atan2(sqrt(1-x), sqrt(1+x))
    double a = std::sqrt(1.0 - x), b = std::sqrt(1.0 + x);
    double t;
    if (b == 0.0)
        t = 1.57079632679489655800e+00;
    else
        t = k_atan_pos(a / b);
    return (float)(2.0 * t);
}

}  // namespace lm
}  // namespace orc
