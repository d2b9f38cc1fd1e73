// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of yuki/src/interaction.rs, yuki/src/shapes/{triangle,mesh,
// sphere}.rs.  Parity unpinned by the reference (it has no tests for shapes).
#pragma once
#include <vector>

#include "olibm.h"
#include "omath.h"

namespace orc {

// yuki/src/interaction.rs:7-60
struct Interaction {
    Point3f p;
    Normalf n;
    Interaction() : p(), n(0.0f, 0.0f, 1.0f) {}
    Interaction(Point3f p_, Normalf n_) : p(p_), n(n_) {}
    // interaction.rs:27-40: fixed 1e-3 offset along the geometric normal
    Rayf spawn_ray(const Vec3f& d) const {
        Vec3f nv(n);
        Vec3f offset = nv * 0.001f;
        Point3f o = d.dot(nv) > 0.0f ? p + offset : p - offset;
        return Rayf(o, d, std::numeric_limits<float>::infinity());
    }
    // interaction.rs:44-59: direction not normalised, t_max 0.9999
    Rayf spawn_ray_to(const Interaction& other) const {
        Vec3f nv(n);
        Vec3f offset = nv * 0.001f;
        Point3f o = (other.p - p).dot(nv) > 0.0f ? p + offset : p - offset;
        Vec3f d = other.p - o;
        return Rayf(o, d, 0.9999f);
    }
};

struct ShadingGeometry {
    Normalf n;
    Vec3f dpdu, dpdv;
};

// yuki/src/interaction.rs:81-139
struct SurfaceInteraction {
    Point3f p;
    Normalf n;
    Point2f uv;
    Vec3f dpdu, dpdv;
    ShadingGeometry shading;
    Vec3f wo;
    bool swaps_handedness;
    int area_light;  // index into scene lights, -1 = none

    SurfaceInteraction() : swaps_handedness(false), area_light(-1) {}
    // interaction.rs:95-124
    SurfaceInteraction(Point3f p_, Vec3f wo_, Point2f uv_, Vec3f dpdu_, Vec3f dpdv_, bool swaps, int al)
        : p(p_), uv(uv_), dpdu(dpdu_), dpdv(dpdv_), wo(wo_), swaps_handedness(swaps), area_light(al) {
        Normalf nn(dpdu_.cross(dpdv_).normalized());
        n = swaps ? -nn : nn;
        shading.n = n;
        shading.dpdu = dpdu_;
        shading.dpdv = dpdv_;
    }
    // interaction.rs:126-132
    void set_shading_geometry(const Vec3f& dpdus, const Vec3f& dpdvs) {
        shading.n = Normalf(dpdus.cross(dpdvs)).normalized();
        n = n.faceforward_n(shading.n);
        shading.dpdu = dpdus;
        shading.dpdv = dpdvs;
    }
};

// interaction.rs:141-164 (Transform * SurfaceInteraction, used by Sphere)
inline SurfaceInteraction transform_si(const Transformf& t, const SurfaceInteraction& o) {
    Normalf n = t.apply(o.n).normalized();
    ShadingGeometry sh;
    sh.n = t.apply(o.shading.n).normalized();
    sh.dpdu = t.apply(o.shading.dpdu);
    sh.dpdv = t.apply(o.shading.dpdv);
    sh.n = sh.n.faceforward_n(n);
    SurfaceInteraction r;
    r.p = t.apply(o.p);
    r.n = n;
    r.uv = o.uv;
    r.dpdu = t.apply(o.dpdu);
    r.dpdv = t.apply(o.dpdv);
    r.wo = t.apply(o.wo).normalized();
    r.shading = sh;
    r.area_light = o.area_light;
    r.swaps_handedness = o.swaps_handedness;
    r.shading.n = r.shading.n.faceforward_n(r.n);
    return r;
}

// yuki/src/shapes/mesh.rs — points/normals already in world space
struct Mesh {
    bool has_normals, has_uvs, swaps_handedness;
};

enum ShapeKind { SHAPE_TRIANGLE = 0, SHAPE_SPHERE = 1 };

struct Shape {
    int kind;
    // triangle
    uint32_t v[3];
    uint32_t mesh;
    // sphere
    Transformf object_to_world, world_to_object;
    float radius;
    bool sphere_swaps;
    // common
    int material;
    int area_light;
    uint32_t source_index;  // index in the caller's shape order (triangles then spheres)
};

struct Hit {
    float t;
    SurfaceInteraction si;
    const Shape* shape;
};

struct Geometry {
    const float* points;   // 3 per vertex, world space
    const float* normals;  // 3 per vertex or null
    const float* uvs;      // 2 per vertex or null
    std::vector<Mesh> meshes;
    Point3f P(uint32_t i) const { return Point3f(points[3 * i], points[3 * i + 1], points[3 * i + 2]); }
    Normalf N(uint32_t i) const { return Normalf(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]); }
    Point2f UV(uint32_t i) const { return Point2f(uvs[2 * i], uvs[2 * i + 1]); }
};

// yuki/src/shapes/triangle.rs:49-227
inline bool triangle_intersect(const Geometry& g, const Shape& s, const Rayf& ray, Hit& out) {
    Point3f p0 = g.P(s.v[0]), p1 = g.P(s.v[1]), p2 = g.P(s.v[2]);

    Vec3f p0t = p0 - ray.o, p1t = p1 - ray.o, p2t = p2 - ray.o;
    int kz = ray.d.abs().max_dimension();
    int kx = kz < 2 ? kz + 1 : 0;
    int ky = kx < 2 ? kx + 1 : 0;
    p0t = p0t.permuted(kx, ky, kz);
    p1t = p1t.permuted(kx, ky, kz);
    p2t = p2t.permuted(kx, ky, kz);
    Vec3f d = ray.d.permuted(kx, ky, kz);

    float sx = -d.x / d.z;
    float sy = -d.y / d.z;
    float sz = 1.0f / d.z;
    p0t.x += sx * p0t.z;
    p0t.y += sy * p0t.z;
    p1t.x += sx * p1t.z;
    p1t.y += sy * p1t.z;
    p2t.x += sx * p2t.z;
    p2t.y += sy * p2t.z;

    float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
        double e0_64 = (double)p1t.x * (double)p2t.y - (double)p1t.y * (double)p2t.x;
        double e1_64 = (double)p2t.x * (double)p0t.y - (double)p2t.y * (double)p0t.x;
        double e2_64 = (double)p0t.x * (double)p1t.y - (double)p0t.y * (double)p1t.x;
        e0 = (float)e0_64;
        e1 = (float)e1_64;
        e2 = (float)e2_64;
    }

    if ((e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f)) return false;

    float det = e0 + e1 + e2;
    if (det == 0.0f) return false;

    float p0z_scaled = p0t.z * sz;
    float p1z_scaled = p1t.z * sz;
    float p2z_scaled = p2t.z * sz;
    float t_scaled = e0 * p0z_scaled + e1 * p1z_scaled + e2 * p2z_scaled;

    if ((det < 0.0f && (t_scaled >= 0.0f || t_scaled < ray.t_max * det)) ||
        (det > 0.0f && (t_scaled <= 0.0f || t_scaled > ray.t_max * det)))
        return false;

    float inv_det = 1.0f / det;
    float b0 = e0 * inv_det;
    float b1 = e1 * inv_det;
    float b2 = e2 * inv_det;
    float t = t_scaled * inv_det;

    const Mesh& mesh = g.meshes[s.mesh];
    Point2f uvs[3];
    if (!mesh.has_uvs) {
        uvs[0] = Point2f(0.0f, 0.0f);
        uvs[1] = Point2f(1.0f, 0.0f);
        uvs[2] = Point2f(1.0f, 1.0f);
    } else {
        uvs[0] = g.UV(s.v[0]);
        uvs[1] = g.UV(s.v[1]);
        uvs[2] = g.UV(s.v[2]);
    }

    Vec2<float> duv02 = uvs[0] - uvs[2];
    Vec2<float> duv12 = uvs[1] - uvs[2];
    Vec3f dp02 = p0 - p2;
    Vec3f dp12 = p1 - p2;

    float uv_det = duv02[0] * duv12[1] - duv02[1] * duv12[0];
    Vec3f dpdu, dpdv;
    if (uv_det == 0.0f) {
        Vec3f n = (p2 - p0).cross(p1 - p0).normalized();
        coordinate_system(n, dpdu, dpdv);
    } else {
        float inv_uv_det = 1.0f / uv_det;
        dpdu = (dp02 * duv12[1] - dp12 * duv02[1]) * inv_uv_det;
        dpdv = ((-dp02) * duv12[0] + dp12 * duv02[0]) * inv_uv_det;
    }

    Point3f p_hit = p0 * b0 + p1 * b1 + p2 * b2;
    Point2f uv_hit = uvs[0] * b0 + uvs[1] * b1 + uvs[2] * b2;
    SurfaceInteraction si(p_hit, -ray.d, uv_hit, dpdu, dpdv, mesh.swaps_handedness, s.area_light);

    Normalf n(dp02.cross(dp12).normalized());
    if (mesh.swaps_handedness) {
        si.n = -n;
        si.shading.n = -n;
    } else {
        si.n = n;
        si.shading.n = n;
    }

    if (mesh.has_normals) {
        Normalf n0 = g.N(s.v[0]), n1 = g.N(s.v[1]), n2 = g.N(s.v[2]);
        Vec3f ns;
        {
            Vec3f nn = Vec3f(n0 * b0 + n1 * b1 + n2 * b2).normalized();
            if (nn.len_sqr() > 0.0f)
                ns = nn.normalized();
            else
                ns = Vec3f(si.n);
        }
        Vec3f ss = si.dpdu.normalized();
        Vec3f ts = ss.cross(ns);
        if (ts.len_sqr() > 0.0f) {
            ts = ts.normalized();
            ss = ts.cross(ns);
        } else {
            coordinate_system(ns, ss, ts);
        }
        si.set_shading_geometry(ss, ts);
    }

    out.t = t;
    out.si = si;
    out.shape = &s;
    return true;
}

// yuki/src/shapes/sphere.rs:38-119 (libm calls go through olibm.h)
inline bool sphere_intersect(const Shape& s, const Rayf& ray, Hit& out) {
    Rayf r = s.world_to_object.apply(ray);

    float a = r.d.x * r.d.x + r.d.y * r.d.y + r.d.z * r.d.z;
    float b = 2.0f * (r.d.x * r.o.x + r.d.y * r.o.y + r.d.z * r.o.z);
    float c = r.o.x * r.o.x + r.o.y * r.o.y + r.o.z * r.o.z - s.radius * s.radius;

    float discrim = b * b - 4.0f * a * c;
    if (discrim < 0.0f) return false;
    float rd = std::sqrt(discrim);

    float q = b < 0.0f ? -0.5f * (b - rd) : -0.5f * (b + rd);

    float t0 = q / a;
    float t1 = c / q;
    if (t0 > t1) {
        float tmp = t0;
        t0 = t1;
        t1 = tmp;
    }
    if (t0 > r.t_max || t1 <= 0.0f) return false;
    float t = t0;
    if (t <= 0.0f) {
        t = t1;
        if (t > r.t_max) return false;
    }

    Point3f p = r.point(t);
    p = p * (s.radius / p.dist(Point3f()));
    if (p.x == 0.0f && p.y == 0.0f) p.x = 1e-5f * s.radius;

    const float PI = 3.14159265358979323846f;
    float phi = lm::atan2f_(p.y, p.x);
    if (phi < 0.0f) phi += 2.0f * PI;

    float phi_max = 2.0f * PI;
    float theta_min = PI;
    float theta_max = 0.0f;
    float u = phi / phi_max;
    float theta = lm::acosf_(rclamp(p.z / s.radius, -1.0f, 1.0f));
    float v = (theta - theta_min) / (theta_max - theta_min);

    float z_radius = std::sqrt(p.x * p.x + p.y * p.y);
    float inv_z_radius = 1.0f / z_radius;
    float cos_phi = p.x * inv_z_radius;
    float sin_phi = p.y * inv_z_radius;
    Vec3f dpdu(-phi_max * p.y, phi_max * p.x, 0.0f);
    Vec3f dpdv = Vec3f(p.z * cos_phi, p.z * sin_phi, -s.radius * lm::sinf_(theta)) * (theta_max - theta_min);

    SurfaceInteraction si_obj(p, -ray.d, Point2f(u, v), dpdu, dpdv, s.sphere_swaps, -1);
    out.si = transform_si(s.object_to_world, si_obj);
    out.t = t;
    out.shape = &s;
    return true;
}

inline bool shape_intersect(const Geometry& g, const Shape& s, const Rayf& ray, Hit& out) {
    return s.kind == SHAPE_TRIANGLE ? triangle_intersect(g, s, ray, out) : sphere_intersect(s, ray, out);
}

// triangle.rs:229-235 / sphere.rs:121-123
inline Bounds3f shape_world_bound(const Geometry& g, const Shape& s) {
    if (s.kind == SHAPE_TRIANGLE) return Bounds3f(g.P(s.v[0]), g.P(s.v[1])).union_p(g.P(s.v[2]));
    return s.object_to_world.apply(
        Bounds3f(Point3f(-s.radius, -s.radius, -s.radius), Point3f(s.radius, s.radius, s.radius)));
}

}  // namespace orc
