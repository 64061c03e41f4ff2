// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.hpp).
// orc_shapes.hpp: src/interaction.rs, src/shapes/{mod,sphere,triangle}.rs, src/geometry/bounds.rs,
// src/sampling.rs (warping functions used by the shapes).
#pragma once
#include "orc_transform.hpp"
#include <vector>

namespace orc {

// ---- Bounds3f: src/geometry/bounds.rs:101-233
struct Bounds3 {
    Vec3 min, max;
    static Bounds3 empty() {   // Point3::max_value / min_value (num::Bounded: f32::MAX / f32::MIN)
        Bounds3 b;
        const Float M = std::numeric_limits<Float>::max();
        b.min = Vec3(M, M, M); b.max = Vec3(-M, -M, -M);
        return b;
    }
    Bounds3 join(const Bounds3& o) const {
        Bounds3 b;
        b.min = Vec3(fmin_(min.x, o.min.x), fmin_(min.y, o.min.y), fmin_(min.z, o.min.z));
        b.max = Vec3(fmax_(max.x, o.max.x), fmax_(max.y, o.max.y), fmax_(max.z, o.max.z));
        return b;
    }
    Bounds3 join_point(Vec3 p) const {
        Bounds3 b;
        b.min = Vec3(fmin_(min.x, p.x), fmin_(min.y, p.y), fmin_(min.z, p.z));
        b.max = Vec3(fmax_(max.x, p.x), fmax_(max.y, p.y), fmax_(max.z, p.z));
        return b;
    }
    Vec3 diagonal() const { return max - min; }
    Vec3 centroid() const { return min + (diagonal() / 2.0f); }            // :160-162
    int maximum_extent() const {                                           // :168-177
        Vec3 d = diagonal();
        if (d.x > d.y && d.x > d.z) return 0;
        else if (d.y > d.z) return 1;
        return 2;
    }
    bool is_point() const { return max == min; }
    // bounding_sphere: :208-212
    void bounding_sphere(Vec3* center, Float* radius) const {
        *center = Vec3(0, 0, 0) + ((min + max) / 2.0f);
        *radius = distance(*center, max);
    }
    // intersect_test: :214-233
    bool intersect_test(const Ray& ray, Float* t0_out = nullptr, Float* t1_out = nullptr) const {
        Float t0 = 0.0f, t1 = ray.t_max;
        for (int i = 0; i < 3; i++) {
            Float inv_ray_dir = 1.0f / ray.dir[i];
            Float t_near = (min[i] - ray.origin[i]) * inv_ray_dir;
            Float t_far = (max[i] - ray.origin[i]) * inv_ray_dir;
            if (t_near > t_far) std::swap(t_near, t_far);
            t_far *= 1.0f + 2.0f * gamma(3);
            t0 = fmax_(t0, t_near);
            t1 = fmin_(t1, t_far);
            if (t0 > t1) return false;
        }
        if (t0_out) *t0_out = t0;
        if (t1_out) *t1_out = t1;
        return true;
    }
};
// Bounds3f::transform: :270-277 (iter_corners order :183-194)
inline Bounds3 tf_bounds(const Transform& tf, const Bounds3& b) {
    Bounds3 r = Bounds3::empty();
    Vec3 c[8] = {Vec3(b.min.x, b.min.y, b.min.z), Vec3(b.min.x, b.min.y, b.max.z), Vec3(b.min.x, b.max.y, b.min.z),
                 Vec3(b.min.x, b.max.y, b.max.z), Vec3(b.max.x, b.min.y, b.min.z), Vec3(b.max.x, b.min.y, b.max.z),
                 Vec3(b.max.x, b.max.y, b.min.z), Vec3(b.max.x, b.max.y, b.max.z)};
    for (int i = 0; i < 8; i++) r = r.join_point(tf_point(tf, c[i]));
    return r;
}

// ---- sampling warps: src/sampling.rs:5-57
inline Vec2 concentric_sample_disk(Vec2 u) {
    Vec2 uo(2.0f * u.x - 1.0f, 2.0f * u.y - 1.0f);
    if (uo.x == 0.0f && uo.y == 0.0f) return Vec2(0.0f, 0.0f);
    Float theta, r;
    if (fabsf(uo.x) > fabsf(uo.y)) { theta = FRAC_PI_4 * (uo.y / uo.x); r = uo.x; }
    else { theta = FRAC_PI_2 - FRAC_PI_4 * (uo.x / uo.y); r = uo.y; }
    return Vec2(r * m_cos(theta), r * m_sin(theta));
}
inline Vec3 cosine_sample_hemisphere(Vec2 u) {
    Vec2 d = concentric_sample_disk(u);
    Float z = sqrtf(fmax_(0.0f, 1.0f - d.x * d.x - d.y * d.y));
    return Vec3(d.x, d.y, z);
}
inline Vec3 uniform_sample_sphere(Vec2 u) {
    Float z = 1.0f - 2.0f * u.x;
    Float r = sqrtf(fmax_(1.0f - z * z, 0.0f));
    Float phi = 2.0f * PI * u.y;
    return Vec3(r * m_cos(phi), r * m_sin(phi), z);
}
inline Vec2 uniform_sample_triangle(Vec2 u) {
    Float su0 = sqrtf(u.x);
    return Vec2(1.0f - su0, u.y * su0);
}
inline Float power_heuristic(uint32_t nf, Float f_pdf, uint32_t ng, Float g_pdf) {
    Float f = (Float)nf * f_pdf, g = (Float)ng * g_pdf;
    return (f * f) / (f * f + g * g);
}

// ---- SurfaceHit / SurfaceInteraction: src/interaction.rs
static const Float SHADOW_EPSILON = 0.0001f;

struct SurfaceHit {
    Vec3 p, p_err; Float time; Vec3 n;
    Ray spawn_ray(Vec3 dir) const {                                       // :22-30
        Ray r; r.origin = offset_ray_origin(p, p_err, n, dir); r.dir = dir; r.t_max = INF; r.time = time; return r;
    }
    Ray spawn_ray_to_hit(const SurfaceHit& to) const {                    // :48-58
        Vec3 origin = offset_ray_origin(p, p_err, n, to.p - p);
        Vec3 target = offset_ray_origin(to.p, to.p_err, to.n, origin - to.p);
        Ray r; r.origin = origin; r.dir = target - origin; r.t_max = 1.0f - SHADOW_EPSILON; r.time = time; return r;
    }
};
struct DiffGeom { Vec3 dpdu, dpdv, dndu, dndv; };
struct TextureDifferentials { Vec3 dpdx, dpdy; Float dudx = 0, dvdx = 0, dudy = 0, dvdy = 0; };

struct SurfaceInteraction {
    SurfaceHit hit; Vec2 uv; Vec3 wo; DiffGeom geom; Vec3 shading_n; DiffGeom shading_geom;
    TextureDifferentials tex_diffs;
    int prim = -1;   // index into the BVH-ordered primitive array (Option<&dyn Primitive>)
    static SurfaceInteraction make(Vec3 p, Vec3 p_err, Float time, Vec2 uv, Vec3 wo, Vec3 n, DiffGeom g) {  // :84-107
        SurfaceInteraction s; s.hit.p = p; s.hit.p_err = p_err; s.hit.time = time; s.hit.n = n;
        s.uv = uv; s.wo = wo; s.geom = g; s.shading_n = n; s.shading_geom = g; return s;
    }
    // compute_tex_differentials: :124-173 (None -> default zeros)
    void compute_tex_differentials(const RayDifferential& ray) {
        tex_diffs = TextureDifferentials();
        if (!ray.has_diff) return;
        Vec3 n = hit.n;
        const Differential& diff = ray.diff;
        Float d = dot(n, hit.p);
        Float tx = -(dot(n, diff.rx_origin) - d) / dot(n, diff.rx_dir);
        Vec3 px = diff.rx_origin + tx * diff.rx_dir;
        Float ty = -(dot(n, diff.ry_origin) - d) / dot(n, diff.ry_dir);
        Vec3 py = diff.ry_origin + ty * diff.ry_dir;
        Vec3 dpdx = px - hit.p, dpdy = py - hit.p;
        int d0, d1;
        if (fabsf(n.x) > fabsf(n.y) && fabsf(n.x) > fabsf(n.z)) { d0 = 1; d1 = 2; }
        else if (fabsf(n.y) > fabsf(n.z)) { d0 = 0; d1 = 2; }
        else { d0 = 0; d1 = 1; }
        Vec3 dpdu = geom.dpdu, dpdv = geom.dpdv;
        // A = from_cols((dpdu[d0], dpdu[d1]), (dpdv[d0], dpdv[d1]))
        Float dudx, dvdx, dudy, dvdy;
        if (!solve_linear_system_2x2(dpdu[d0], dpdu[d1], dpdv[d0], dpdv[d1], dpdx[d0], dpdx[d1], &dudx, &dvdx)) return;
        if (!solve_linear_system_2x2(dpdu[d0], dpdu[d1], dpdv[d0], dpdv[d1], dpdy[d0], dpdy[d1], &dudy, &dvdy)) return;
        tex_diffs.dpdx = dpdx; tex_diffs.dpdy = dpdy;
        tex_diffs.dudx = dudx; tex_diffs.dvdx = dvdx; tex_diffs.dudy = dudy; tex_diffs.dvdy = dvdy;
    }
};

// SurfaceHit::transform: src/geometry/transform.rs:340-346; SurfaceInteraction::transform: :369-385
inline SurfaceHit tf_surface_hit(const Transform& tf, const SurfaceHit& h) {
    SurfaceHit o;
    o.p = tf_point_err_to_err(tf, h.p, h.p_err, &o.p_err);
    o.n = normalize(tf_normal(tf, h.n));
    o.time = h.time;
    return o;
}
inline DiffGeom tf_diff_geom(const Transform& tf, const DiffGeom& g) {
    DiffGeom o; o.dpdu = tf_vector(tf, g.dpdu); o.dpdv = tf_vector(tf, g.dpdv);
    o.dndu = tf_normal(tf, g.dndu); o.dndv = tf_normal(tf, g.dndv); return o;
}
inline SurfaceInteraction tf_surface_interaction(const Transform& tf, const SurfaceInteraction& s) {
    SurfaceInteraction o;
    o.hit = tf_surface_hit(tf, s.hit);
    o.uv = s.uv;
    o.wo = normalize(tf_vector(tf, s.wo));
    o.geom = tf_diff_geom(tf, s.geom);
    o.shading_n = normalize(tf_normal(tf, s.shading_n));
    o.shading_geom = tf_diff_geom(tf, s.shading_geom);
    o.tex_diffs = s.tex_diffs;
    o.tex_diffs.dpdx = tf_vector(tf, s.tex_diffs.dpdx);
    o.tex_diffs.dpdy = tf_vector(tf, s.tex_diffs.dpdy);
    o.prim = s.prim;
    return o;
}

// ---- Shape interface: src/shapes/mod.rs:10-68
struct Shape {
    virtual ~Shape() {}
    virtual Bounds3 world_bound() const = 0;
    virtual Float area() const = 0;
    virtual bool intersect(const Ray& ray, Float* t_hit, SurfaceInteraction* si) const = 0;
    virtual bool intersect_test(const Ray& ray) const { Float t; SurfaceInteraction si; return intersect(ray, &t, &si); }  // :35-37
    virtual SurfaceHit sample(Vec2 u) const = 0;
    Float pdf_from_ref(const SurfaceHit& reference, Vec3 wi) const {      // :55-66
        Ray ray = reference.spawn_ray(wi);
        Float t; SurfaceInteraction isect;
        if (intersect(ray, &t, &isect)) {
            return distance_sq(reference.p, isect.hit.p) / (abs_dot(isect.hit.n, -wi) * area());
        }
        return 0.0f;
    }
};

// ---- Sphere: src/shapes/sphere.rs
struct Sphere : Shape {
    Transform object_to_world, world_to_object;
    bool reverse_orientation;
    Float radius, z_min, z_max, theta_min, theta_max, phi_max;

    static Sphere make(const Transform& o2w, const Transform& w2o, bool rev, Float radius, Float z_min, Float z_max, Float phi_max) {  // :30-50
        Sphere s; s.object_to_world = o2w; s.world_to_object = w2o; s.reverse_orientation = rev; s.radius = radius;
        s.z_min = clampf(fmin_(z_min, z_max), -radius, radius);
        s.z_max = clampf(fmax_(z_min, z_max), -radius, radius);
        s.theta_min = m_acos(clampf(z_min / radius, -1.0f, 1.0f));
        s.theta_max = m_acos(clampf(z_max / radius, -1.0f, 1.0f));
        s.phi_max = to_radians(clampf(phi_max, 0.0f, 360.0f));
        return s;
    }
    Bounds3 world_bound() const override {                                // :62-64 + shapes/mod.rs:13-15
        Bounds3 ob; ob.min = Vec3(-radius, -radius, z_min); ob.max = Vec3(radius, radius, z_max);
        return tf_bounds(object_to_world, ob);   // bounds3f! == with_bounds (src/macros.rs)
    }
    Float area() const override { return phi_max * radius * (z_max - z_min); }  // :77-79

    bool clipped(Vec3 p_hit, Float phi) const {
        return (z_min > -radius && p_hit.z < z_min) || (z_max < radius && p_hit.z > z_max) || phi > phi_max;
    }
    bool intersect(const Ray& world_ray, Float* t_hit_out, SurfaceInteraction* si) const override {  // :83-200
        Vec3 origin_err, dir_err;
        Ray ray = tf_ray_exact_to_err(world_to_object, world_ray, &origin_err, &dir_err);
        EFloat ox = EFloat::with_err(ray.origin.x, origin_err.x);
        EFloat oy = EFloat::with_err(ray.origin.y, origin_err.y);
        EFloat oz = EFloat::with_err(ray.origin.z, origin_err.z);
        EFloat dirx = EFloat::with_err(ray.dir.x, dir_err.x);
        EFloat diry = EFloat::with_err(ray.dir.y, dir_err.y);
        EFloat dirz = EFloat::with_err(ray.dir.z, dir_err.z);
        EFloat a = dirx * dirx + diry * diry + dirz * dirz;
        EFloat b = 2.0f * (dirx * ox + diry * oy + dirz * oz);
        EFloat c = ox * ox + oy * oy + oz * oz - EFloat(radius) * EFloat(radius);
        EFloat t0, t1;
        if (!quadratic(a, b, c, &t0, &t1)) return false;
        if (t0.upper_bound() > ray.t_max || t1.lower_bound() <= 0.0f) return false;
        EFloat t_shape_hit = t0;
        bool is_t1 = false;
        if (t_shape_hit.lower_bound() <= 0.0f) {
            t_shape_hit = t1; is_t1 = true;
            if (t_shape_hit.upper_bound() > ray.t_max) return false;
        }
        Vec3 p_hit = ray.at(t_shape_hit.v);
        p_hit = p_hit * (radius / distance(p_hit, Vec3(0, 0, 0)));
        if (p_hit.x == 0.0f && p_hit.y == 0.0f) p_hit.x = 1.0e-5f * radius;
        Float phi = m_atan2(p_hit.y, p_hit.x);
        if (phi < 0.0f) phi += 2.0f * PI;
        if (clipped(p_hit, phi)) {
            // `t_shape_hit == t1` compares the EFloat values (PartialEq on .v, err_float.rs:91-95)
            (void)is_t1;
            if (t_shape_hit.v == t1.v) return false;
            if (t1.upper_bound() > ray.t_max) return false;
            t_shape_hit = t1;
            p_hit = ray.at(t_shape_hit.v);
            p_hit = p_hit * (radius / distance(p_hit, Vec3(0, 0, 0)));
            if (p_hit.x == 0.0f && p_hit.y == 0.0f) p_hit.x = 1.0e-5f * radius;
            phi = m_atan2(p_hit.y, p_hit.x);
            if (phi < 0.0f) phi += 2.0f * PI;
            if (clipped(p_hit, phi)) return false;
        }
        Float u = phi / phi_max;
        Float theta = m_acos(clampf(p_hit.z / radius, -1.0f, 1.0f));
        Float v = (theta - theta_min) / (theta_max - theta_min);
        Float z_radius = sqrtf(p_hit.x * p_hit.x + p_hit.y * p_hit.y);
        Float inv_z_radius = 1.0f / z_radius;
        Float cos_phi = p_hit.x * inv_z_radius;
        Float sin_phi = p_hit.y * inv_z_radius;
        Vec3 dpdu(-phi_max * p_hit.y, phi_max * p_hit.x, 0.0f);
        Vec3 dpdv = (theta_max - theta_min) * Vec3(p_hit.z * cos_phi, p_hit.z * sin_phi, -radius * m_sin(theta));
        Vec3 d2pduu = (-phi_max * phi_max) * Vec3(p_hit.x, p_hit.y, 0.0f);
        Vec3 d2pduv = (theta_max - theta_min) * p_hit.z * phi_max * Vec3(-sin_phi, cos_phi, 0.0f);
        Vec3 d2pdvv = -(theta_max - theta_min) * (theta_max - theta_min) * Vec3(p_hit.x, p_hit.y, p_hit.z);
        Float E = dot(dpdu, dpdu), F = dot(dpdu, dpdv), G = dot(dpdv, dpdv);
        Vec3 N = normalize(cross(dpdu, dpdv));
        Float e = dot(N, d2pduu), f = dot(N, d2pduv), g = dot(N, d2pdvv);
        Float invEGF2 = 1.0f / (E * G - F * F);
        Vec3 dndu = (f * F - e * G) * invEGF2 * dpdu + (e * F - f * E) * invEGF2 * dpdv;
        Vec3 dndv = (g * F - f * G) * invEGF2 * dpdu + (f * F - g * E) * invEGF2 * dpdv;
        Vec3 p_err = gamma(5) * vabs(p_hit);
        if (reverse_orientation) N = N * -1.0f;
        DiffGeom dg; dg.dpdu = dpdu; dg.dpdv = dpdv; dg.dndu = dndu; dg.dndv = dndv;
        SurfaceInteraction interact = SurfaceInteraction::make(p_hit, p_err, ray.time, Vec2(u, v), -ray.dir, N, dg);
        *si = tf_surface_interaction(object_to_world, interact);
        *t_hit_out = t_shape_hit.v;
        return true;
    }
    SurfaceHit sample(Vec2 u) const override {                            // :202-218
        Vec3 p_obj = Vec3(0, 0, 0) + radius * uniform_sample_sphere(u);
        Vec3 n = normalize(tf_normal(object_to_world, p_obj));
        if (reverse_orientation) n = n * -1.0f;
        p_obj = p_obj * (radius / distance(p_obj, Vec3(0, 0, 0)));
        Vec3 p_obj_err = gamma(5) * vabs(p_obj);
        SurfaceHit h;
        h.p = tf_point_err_to_err(object_to_world, p_obj, p_obj_err, &h.p_err);
        h.time = 0.0f; h.n = n;
        return h;
    }
};

// ---- TriangleMesh / Triangle: src/shapes/triangle.rs
struct TriangleMesh {
    std::vector<uint32_t> vertex_indices;   // global: into the shared vertex pool
    const Float* P = nullptr;               // world-space positions (pool)
    const Float* N = nullptr;               // world-space normals or nullptr
    const Float* UV = nullptr;
    const Float* S = nullptr;               // world-space shading tangents or nullptr (triangle.rs:19, :53-58)
    bool has_normals = false, has_uvs = false, flip_normals = false, reverse_orientation = false, has_tangents = false;
    Vec3 tangent(uint32_t i) const { return Vec3(S[3 * i], S[3 * i + 1], S[3 * i + 2]); }
    Vec3 vertex(uint32_t i) const { return Vec3(P[3 * i], P[3 * i + 1], P[3 * i + 2]); }
    Vec3 normal(uint32_t i) const { return Vec3(N[3 * i], N[3 * i + 1], N[3 * i + 2]); }
    Vec2 uv(uint32_t i) const { return Vec2(UV[2 * i], UV[2 * i + 1]); }
};

inline bool sign_differs(Float v1, Float v2, Float v3) {                  // :428-434
    return is_sign_positive(v1) != is_sign_positive(v2) || is_sign_positive(v2) != is_sign_positive(v3);
}

struct Triangle : Shape {
    const TriangleMesh* mesh; uint32_t v[3];
    Bounds3 world_bound() const override {                                // :152-158
        return Bounds3::empty().join_point(mesh->vertex(v[0])).join_point(mesh->vertex(v[1])).join_point(mesh->vertex(v[2]));
    }
    Float area() const override {                                         // :171-174
        Vec3 p0 = mesh->vertex(v[0]), p1 = mesh->vertex(v[1]), p2 = mesh->vertex(v[2]);
        return 0.5f * magnitude(cross(p1 - p0, p2 - p0));
    }
    void get_uvs(Vec2 uv[3]) const {                                      // :131-144
        if (mesh->has_uvs) { uv[0] = mesh->uv(v[0]); uv[1] = mesh->uv(v[1]); uv[2] = mesh->uv(v[2]); }
        else { uv[0] = Vec2(0, 0); uv[1] = Vec2(1, 0); uv[2] = Vec2(1, 1); }
    }
    bool intersect(const Ray& ray, Float* t_out, SurfaceInteraction* si) const override {  // :176-393
        Vec3 p0 = mesh->vertex(v[0]), p1 = mesh->vertex(v[1]), p2 = mesh->vertex(v[2]);
        Vec3 p0t = p0 - ray.origin, p1t = p1 - ray.origin, p2t = p2 - ray.origin;
        int kz = max_dimension(vabs(ray.dir));
        int kx = (kz + 1) % 3, ky = (kx + 1) % 3;
        Vec3 dir = permute(ray.dir, kx, ky, kz);
        p0t = permute(p0t, kx, ky, kz); p1t = permute(p1t, kx, ky, kz); p2t = permute(p2t, kx, ky, kz);
        Float shear_x = -dir.x / dir.z, shear_y = -dir.y / dir.z, shear_z = 1.0f / dir.z;
        p0t.x += shear_x * p0t.z; p0t.y += shear_y * p0t.z;
        p1t.x += shear_x * p1t.z; p1t.y += shear_y * p1t.z;
        p2t.x += shear_x * p2t.z; p2t.y += shear_y * p2t.z;
        Float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
        Float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
        Float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
        if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
            e0 = (Float)((double)p1t.x * (double)p2t.y - (double)p1t.y * (double)p2t.x);
            e1 = (Float)((double)p2t.x * (double)p0t.y - (double)p2t.y * (double)p0t.x);
            e2 = (Float)((double)p0t.x * (double)p1t.y - (double)p0t.y * (double)p1t.x);
        }
        if (sign_differs(e0, e1, e2)) return false;
        Float det = e0 + e1 + e2;
        if (det == 0.0f) return false;
        p0t.z *= shear_z; p1t.z *= shear_z; p2t.z *= shear_z;
        Float t_scaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
        if ((det < 0.0f && (t_scaled >= 0.0f || t_scaled < ray.t_max * det)) ||
            (det > 0.0f && (t_scaled <= 0.0f || t_scaled > ray.t_max * det)))
            return false;
        Float inv_det = 1.0f / det;
        Float b0 = e0 * inv_det, b1 = e1 * inv_det, b2 = e2 * inv_det;
        Float t = t_scaled * inv_det;
        Float max_zt = fmax_(fmax_(fabsf(p0t.z), fabsf(p1t.z)), fabsf(p2t.z));
        Float delta_z = gamma(3) * max_zt;
        Float max_xt = fmax_(fmax_(fabsf(p0t.x), fabsf(p1t.x)), fabsf(p2t.x));
        Float max_yt = fmax_(fmax_(fabsf(p0t.y), fabsf(p1t.y)), fabsf(p2t.y));
        Float delta_x = gamma(5) * (max_xt + max_zt);
        Float delta_y = gamma(5) * (max_yt + max_zt);
        Float delta_e = 2.0f * (gamma(2) * max_xt * max_yt + delta_y * max_xt + delta_x * max_yt);
        Float max_e = fmax_(fmax_(fabsf(e0), fabsf(e1)), fabsf(e2));
        Float delta_t = 3.0f * (gamma(3) * max_e * max_zt + delta_e * max_zt + delta_z * max_e) * fabsf(inv_det);
        if (t <= delta_t) return false;

        Vec2 uv[3]; get_uvs(uv);
        Vec2 duv02(uv[0].x - uv[2].x, uv[0].y - uv[2].y), duv12(uv[1].x - uv[2].x, uv[1].y - uv[2].y);
        Vec3 dp02 = p0 - p2, dp12 = p1 - p2;
        Float determinant = duv02.x * duv12.y - duv02.y * duv12.x;
        bool degenerate_uv = fabsf(determinant) < 1.0e-8f;
        Vec3 dpdu, dpdv;
        Float uv_inv_det = 0.0f;
        if (degenerate_uv) {
            Vec3 ng = cross(p2 - p0, p1 - p0);
            if (magnitude2(ng) == 0.0f) return false;
            coordinate_system(normalize(ng), &dpdu, &dpdv);
        } else {
            uv_inv_det = 1.0f / determinant;
            dpdu = (duv12.y * dp02 - duv02.y * dp12) * uv_inv_det;
            dpdv = (-duv12.x * dp02 + duv02.x * dp12) * uv_inv_det;
        }
        Float x_abs_sum = fabsf(b0 * p0.x) + fabsf(b1 * p1.x) + fabsf(b2 * p2.x);
        Float y_abs_sum = fabsf(b0 * p0.y) + fabsf(b1 * p1.y) + fabsf(b2 * p2.y);
        Float z_abs_sum = fabsf(b0 * p0.z) + fabsf(b1 * p1.z) + fabsf(b2 * p2.z);
        Vec3 p_err = gamma(7) * Vec3(x_abs_sum, y_abs_sum, z_abs_sum);
        Vec3 p_hit = b0 * p0 + b1 * p1 + b2 * p2;
        Vec2 uv_hit((b0 * uv[0].x + b1 * uv[1].x) + b2 * uv[2].x, (b0 * uv[0].y + b1 * uv[1].y) + b2 * uv[2].y);
        DiffGeom dg; dg.dpdu = dpdu; dg.dpdv = dpdv; dg.dndu = Vec3(); dg.dndv = Vec3();
        Vec3 geom_normal = normalize(cross(dp02, dp12));
        SurfaceInteraction isect = SurfaceInteraction::make(p_hit, p_err, ray.time, uv_hit, -ray.dir, geom_normal, dg);
        if (mesh->flip_normals) { isect.hit.n = isect.hit.n * -1.0f; isect.shading_n = isect.shading_n * -1.0f; }
        if (mesh->has_normals || mesh->has_tangents) {   // triangle.rs:332
            Vec3 n0, n1, n2;
            if (mesh->has_normals) { n0 = mesh->normal(v[0]); n1 = mesh->normal(v[1]); n2 = mesh->normal(v[2]); }
            // compute shading normal :334-338
            Vec3 ns = mesh->has_normals ? normalize(b0 * n0 + b1 * n1 + b2 * n2) : isect.hit.n;
            // compute shading tangent :340-345
            Vec3 ss = mesh->has_tangents ? normalize(b0 * mesh->tangent(v[0]) + b1 * mesh->tangent(v[1]) + b2 * mesh->tangent(v[2])) : normalize(isect.geom.dpdu);
            Vec3 ts = cross(ns, ss);
            if (magnitude2(ts) > 0.0f) { ts = normalize(ts); ss = cross(ts, ns); }
            else coordinate_system(ns, &ts, &ss);   // NB: (v2, v3) is bound as (ts, ss) in the reference, :343-349
            Vec3 dndu, dndv;
            Vec3 dn1 = n0 - n2, dn2 = n1 - n2;
            if (!mesh->has_normals) { dndu = Vec3(); dndv = Vec3(); }      // :367-369
            else if (degenerate_uv) {
                Vec3 dn = cross(n2 - n0, n1 - n0);
                if (magnitude2(dn) == 0.0f) { dndu = Vec3(); dndv = Vec3(); }
                else coordinate_system(dn, &dndu, &dndv);
            } else {
                // NB: `inv_det` here is the BARYCENTRIC 1/det (the uv one is scoped to its block), :361-363
                dndu = (duv12.y * dn1 - duv02.y * dn2) * inv_det;
                dndv = (-duv12.x * dn1 + duv02.x * dn2) * inv_det;
            }
            DiffGeom sg; sg.dpdu = ss; sg.dpdv = ts; sg.dndu = dndu; sg.dndv = dndv;
            isect.shading_geom = sg;
            isect.shading_n = ns;
            isect.hit.n = faceforward(isect.hit.n, isect.shading_n);
        }
        *t_out = t; *si = isect;
        return true;
    }
    SurfaceHit sample(Vec2 u) const override {                            // :395-420
        Vec2 b = uniform_sample_triangle(u);
        Vec3 p0 = mesh->vertex(v[0]), p1 = mesh->vertex(v[1]), p2 = mesh->vertex(v[2]);
        Vec3 sample_p = b.x * p0 + b.y * p1 + (1.0f - b.x - b.y) * p2;
        Vec3 n = normalize(cross(p1 - p0, p2 - p0));
        Vec3 sample_n;
        if (mesh->has_normals) {
            Vec3 ns = normalize(b.x * mesh->normal(v[0]) + b.y * mesh->normal(v[1]) + (1.0f - b.x - b.y) * mesh->normal(v[2]));
            sample_n = faceforward(n, ns);
        } else if (mesh->flip_normals) sample_n = n * -1.0f;
        else sample_n = n;
        Vec3 p_abs_sum = vabs(b.x * p0) + vabs(b.y * p1) + vabs((1.0f - b.x - b.y) * p2);
        SurfaceHit h; h.p = Vec3(0, 0, 0) + sample_p; h.p_err = gamma(6) * p_abs_sum; h.time = 0.0f; h.n = sample_n;
        return h;
    }
};

}  // namespace orc
