// yk_math.h — POD math shared by the host code (camera, lights, BVH builder) and
// the gfx950 kernels.  Every function states the reference arithmetic it must
// reproduce bit for bit: the operation ORDER matters (no FMA contraction — the
// library is built with -ffp-contract=off — no reassociation), f64 islands are
// kept where the reference has them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define YK_HD __host__ __device__ __forceinline__

namespace yk {

struct V3 {
    float x, y, z;
};
struct RGB {
    float r, g, b;
};

YK_HD V3 v3(float x, float y, float z) { return V3{x, y, z}; }
YK_HD RGB rgb(float r, float g, float b) { return RGB{r, g, b}; }

// Rust f32::min/max semantics (math/common.rs:60-84): a NaN operand is dropped.
// Select form: also fixes which of +0/-0 is returned, so host-built data (BVH
// bounds) is bit-identical everywhere.
YK_HD float rmin(float a, float b) { return (a != a) ? b : ((b != b) ? a : (b < a ? b : a)); }
YK_HD float rmax(float a, float b) { return (a != a) ? b : ((b != b) ? a : (b > a ? b : a)); }
// Same NaN-dropping semantics via v_min_f32/v_max_f32 (IEEE minNum/maxNum); the
// sign of a zero result is unspecified, so use only where the result feeds
// comparisons (the slab test).
YK_HD float fmin_nan(float a, float b) { return fminf(a, b); }
YK_HD float fmax_nan(float a, float b) { return fmaxf(a, b); }
// f32::clamp: NaN stays NaN
YK_HD float rclamp(float v, float lo, float hi) {
    float r = v;
    if (r < lo) r = lo;
    if (r > hi) r = hi;
    return r;
}

YK_HD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
YK_HD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
YK_HD V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
YK_HD V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
#if defined(YK_ABLATE_DIV3) && defined(__HIP_DEVICE_COMPILE__)  // timing builds only (NOT exact): what sharing the denominator's work among three divisions could return at most
YK_HD V3 operator/(V3 a, float s) {
    const float r = 1.0f / s;
    return V3{a.x * r, a.y * r, a.z * r};
}
#else
YK_HD V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
#endif

// yuki_derive/src/impl_vec_like.rs:188-197: ((0 + x*x') + y*y') + z*z'
YK_HD float dot(V3 a, V3 b) { return ((0.0f + a.x * b.x) + a.y * b.y) + a.z * b.z; }
// math/vector.rs:228-230 and math/normal.rs:57-59: no leading zero
YK_HD float dot_nv(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
YK_HD float len_sqr(V3 a) { return dot(a, a); }
// impl_vec_like.rs:223-228: sqrt taken in f64 then narrowed.  That equals the
// correctly rounded f32 sqrt (double rounding is innocuous for sqrt: 53 >= 2*24+2),
// which is what sqrtf is on the host and — under hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt — on gfx950 (tests/test_gpu_stages.py
// checks both forms bit for bit), at a fraction of the f64 cost.
YK_HD float length(V3 a) { return sqrtf(len_sqr(a)); }
// impl_vec_like.rs:231-235: component-wise DIVISION
YK_HD V3 normalize(V3 a) { return a / length(a); }
// math/vector.rs:236-255: cross product evaluated in f64
YK_HD V3 cross(V3 a, V3 b) {
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return V3{(float)((ay * bz) - (az * by)), (float)((az * bx) - (ax * bz)), (float)((ax * by) - (ay * bx))};
}
YK_HD V3 vabs(V3 a) { return V3{fabsf(a.x), fabsf(a.y), fabsf(a.z)}; }
// math/vector.rs:181-195
YK_HD int max_dimension(V3 a) {
    if (a.x > a.y) return a.x > a.z ? 0 : 2;
    return a.y > a.z ? 1 : 2;
}
YK_HD float comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
// math/normal.rs:62-77
YK_HD V3 faceforward_v(V3 n, V3 v) { return dot_nv(n, v) < 0.0f ? V3{-n.x, -n.y, -n.z} : n; }
YK_HD V3 faceforward_n(V3 n, V3 m) { return dot(n, m) < 0.0f ? V3{-n.x, -n.y, -n.z} : n; }

// math/mod.rs:26-34 (second branch: divide by y*y + z + z, no sqrt — sic)
YK_HD void coordinate_system(V3 v, V3& v1, V3& v2) {
    if (fabsf(v.x) > fabsf(v.y))
        v1 = V3{-v.z, 0.0f, v.x} / sqrtf(v.x * v.x + v.z * v.z);
    else
        v1 = V3{0.0f, v.z, -v.y} / (v.y * v.y + v.z + v.z);
    v2 = cross(v, v1);
}

YK_HD RGB operator+(RGB a, RGB b) { return RGB{a.r + b.r, a.g + b.g, a.b + b.b}; }
YK_HD RGB operator-(RGB a, RGB b) { return RGB{a.r - b.r, a.g - b.g, a.b - b.b}; }
YK_HD RGB operator*(RGB a, RGB b) { return RGB{a.r * b.r, a.g * b.g, a.b * b.b}; }
YK_HD RGB operator/(RGB a, RGB b) { return RGB{a.r / b.r, a.g / b.g, a.b / b.b}; }
YK_HD RGB operator+(RGB a, float s) { return RGB{a.r + s, a.g + s, a.b + s}; }
YK_HD RGB operator-(RGB a, float s) { return RGB{a.r - s, a.g - s, a.b - s}; }
YK_HD RGB operator*(RGB a, float s) { return RGB{a.r * s, a.g * s, a.b * s}; }
#if defined(YK_ABLATE_DIV3) && defined(__HIP_DEVICE_COMPILE__)
YK_HD RGB operator/(RGB a, float s) {
    const float r = 1.0f / s;
    return RGB{a.r * r, a.g * r, a.b * r};
}
#else
YK_HD RGB operator/(RGB a, float s) { return RGB{a.r / s, a.g / s, a.b / s}; }
#endif
YK_HD bool is_black(RGB a) { return a.r == 0.0f && a.g == 0.0f && a.b == 0.0f; }
YK_HD RGB rgb_min(RGB a, RGB b) { return RGB{rmin(a.r, b.r), rmin(a.g, b.g), rmin(a.b, b.b)}; }
YK_HD RGB rgb_sqrt(RGB a) { return RGB{sqrtf(a.r), sqrtf(a.g), sqrtf(a.b)}; }

// row-major 4x4 (math/matrix.rs) applied as in math/transform.rs:105-167
struct M44 {
    float m[16];
};
YK_HD V3 xf_vector(const float* m, V3 v) {
    return V3{m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z};
}
YK_HD V3 xf_point(const float* m, V3 p) {
    float xp = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    float yp = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    float zp = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    float wp = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    if (wp == 1.0f) return V3{xp, yp, zp};
    return V3{xp, yp, zp} / wp;
}
// normals use the inverse matrix, transposed through the accesses
YK_HD V3 xf_normal(const float* mi, V3 n) {
    return V3{mi[0] * n.x + mi[4] * n.y + mi[8] * n.z, mi[1] * n.x + mi[5] * n.y + mi[9] * n.z, mi[2] * n.x + mi[6] * n.y + mi[10] * n.z};
}

}  // namespace yk
