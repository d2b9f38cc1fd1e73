// yk_libm.h — the transcendental functions the Path hot path needs, as one fixed
// recipe that runs identically on the host and on gfx950.
//
// The reference calls Rust's f32::{sin,cos,tan,ln,acos,atan2} (call sites:
// sampling/mod.rs:86, bsdfs/mod.rs:280, trowbridge_reitz.rs:24, camera.rs:64,
// spot_light.rs:33-34, sphere.rs:90,102,111), i.e. the platform libm, whose
// last-bit behaviour is not defined by the reference.  A one-ulp difference in a
// sampled direction flips hemisphere / Russian-roulette branches and changes a
// whole path, so this library fixes the function instead: evaluate in binary64
// with the fdlibm minimax kernels using only + - * / sqrt (each correctly
// rounded on x86-64 and on CDNA4; -ffp-contract=off keeps them unfused) and
// round once to binary32.  f64 vector throughput on MI355X is half the f32
// rate, and a bounce needs two or three of these calls: negligible next to BVH
// traversal.
#pragma once
#include "yk_math.h"

namespace yk {

YK_HD double poly_sin(double r) {
    double z = r * r;
    double p = 8.33333333332248946124e-03 +
               z * (-1.98412698298579493134e-04 +
                    z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    return r + (r * z) * (-1.66666666666666324348e-01 + z * p);
}

YK_HD double poly_cos(double r) {
    double z = r * r;
    double p =
        z * (4.16666666666666019037e-02 +
             z * (-1.38888888888741095749e-03 +
                  z * (2.48015872894767294178e-05 +
                       z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    return (1.0 - 0.5 * z) + z * p;
}

// x = n*(pi/2) + r with |r| <= pi/4 (+eps); Cody-Waite, pi/2 split in three
YK_HD double quadrant_reduce(double x, int& quadrant) {
    double fn = floor(x * 6.36619772367581382433e-01 + 0.5);
    double r = x - fn * 1.57079632673412561417e+00;
    r = r - fn * 6.07710050630396597660e-11;
    r = r - fn * 2.02226624879595063154e-21;
    quadrant = (int)((long long)fn & 3);
    return r;
}

YK_HD float det_sinf(float xf) {
#if defined(YK_ABLATE_LIBM) && defined(__HIP_DEVICE_COMPILE__)  // timing builds only: the hardware approximations instead of the f64 recipe
    return __sinf(xf);
#endif
    double x = (double)xf;
    if (!(fabs(x) < 1.0e300)) return xf - xf;
    int q;
    double r = quadrant_reduce(x, q);
    double v = (q & 1) ? poly_cos(r) : poly_sin(r);
    return (float)((q & 2) ? -v : v);
}

YK_HD float det_cosf(float xf) {
#if defined(YK_ABLATE_LIBM) && defined(__HIP_DEVICE_COMPILE__)  // timing builds only: the hardware approximations instead of the f64 recipe
    return __cosf(xf);
#endif
    double x = (double)xf;
    if (!(fabs(x) < 1.0e300)) return xf - xf;
    int q;
    double r = quadrant_reduce(x, q);
    double v = (q & 1) ? poly_sin(r) : poly_cos(r);
    return (float)(((q + 1) & 2) ? -v : v);
}

YK_HD float det_tanf(float xf) {
#if defined(YK_ABLATE_LIBM) && defined(__HIP_DEVICE_COMPILE__)  // timing builds only: the hardware approximations instead of the f64 recipe
    return __tanf(xf);
#endif
    double x = (double)xf;
    if (!(fabs(x) < 1.0e300)) return xf - xf;
    int q;
    double r = quadrant_reduce(x, q);
    double s = poly_sin(r), c = poly_cos(r);
    return (float)((q & 1) ? -(c / s) : (s / c));
}

YK_HD float det_logf(float xf) {
#if defined(YK_ABLATE_LIBM) && defined(__HIP_DEVICE_COMPILE__)  // timing builds only: the hardware approximations instead of the f64 recipe
    return __logf(xf);
#endif
    if (xf != xf) return xf;
    if (xf < 0.0f) return (xf - xf) / 0.0f;
    if (xf == 0.0f) return -1.0f / 0.0f;
    if (xf > 3.0e38f && xf + xf == xf) return xf;
    double x = (double)xf;
    unsigned long long bits = (unsigned long long)__builtin_bit_cast(unsigned long long, x);
    long long e = (long long)((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = __builtin_bit_cast(double, bits);
    if (m > 1.41421356237309514547) {
        m = m * 0.5;
        e += 1;
    }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double R = z * (6.666666666666735130e-01 +
                    z * (3.999999999940941908e-01 +
                         z * (2.857142874366239149e-01 +
                              z * (2.222219843214978396e-01 +
                                   z * (1.818357216161805012e-01 + z * (1.531383769920937332e-01 + z * 1.479819860511658591e-01))))));
    double lg = 2.0 * s + s * R;
    return (float)((double)e * 6.93147180559945286227e-01 + lg);
}

// atan for x >= 0 in binary64
YK_HD double atan_nonneg(double x) {
    double hi = 0.0, lo = 0.0;
    int reduced = 1;
    if (x < 0.4375) {
        reduced = 0;
    } else if (x < 0.6875) {
        hi = 4.63647609000806093515e-01;
        lo = 2.26987774529616870924e-17;
        x = (2.0 * x - 1.0) / (2.0 + x);
    } else if (x < 1.1875) {
        hi = 7.85398163397448278999e-01;
        lo = 3.06161699786838301793e-17;
        x = (x - 1.0) / (x + 1.0);
    } else if (x < 2.4375) {
        hi = 9.82793723247329054082e-01;
        lo = 1.39033110312309984516e-17;
        x = (x - 1.5) / (1.0 + 1.5 * x);
    } else {
        hi = 1.57079632679489655800e+00;
        lo = 6.12323399573676603587e-17;
        x = -1.0 / x;
    }
    double z = x * x;
    double w = z * z;
    double s1 = z * (3.33333333333329318027e-01 +
                     w * (1.42857142725034663711e-01 +
                          w * (9.09088713343650656196e-02 +
                               w * (6.66107313738753120669e-02 + w * (4.97687799461593236017e-02 + w * 1.62858201153657823623e-02)))));
    double s2 = w * (-1.99999999998764832476e-01 +
                     w * (-1.11111104054623557880e-01 +
                          w * (-7.69187620504482999495e-02 + w * (-5.83357013379057348645e-02 + w * -3.65315727442169155270e-02))));
    if (!reduced) return x - x * (s1 + s2);
    return hi - ((x * (s1 + s2) - lo) - x);
}

YK_HD float det_atan2f(float yf, float xf) {
    const double PI = 3.14159265358979311600e+00, PIO2 = 1.57079632679489655800e+00;
    if (xf != xf || yf != yf) return xf + yf;
    double y = (double)yf, x = (double)xf;
    if (y == 0.0) {
        double v = __builtin_signbit(xf) ? PI : 0.0;
        return (float)(__builtin_signbit(yf) ? -v : v);
    }
    if (x == 0.0) return (float)(y > 0.0 ? PIO2 : -PIO2);
    double ax = fabs(x), ay = fabs(y), a;
    if (ax > 1.0e300 && ay > 1.0e300)
        a = 7.85398163397448278999e-01;
    else if (ay > 1.0e300)
        a = PIO2;
    else if (ax > 1.0e300)
        a = 0.0;
    else
        a = atan_nonneg(ay / ax);
    if (x < 0.0) a = PI - a;
    return (float)(y < 0.0 ? -a : a);
}

YK_HD float det_acosf(float xf) {
    if (xf != xf) return xf;
    double x = (double)xf;
    if (x > 1.0 || x < -1.0) return (xf - xf) / (xf - xf);
    double a = sqrt(1.0 - x), b = sqrt(1.0 + x);
    double t = (b == 0.0) ? 1.57079632679489655800e+00 : atan_nonneg(a / b);
    return (float)(2.0 * t);
}

}  // namespace yk
