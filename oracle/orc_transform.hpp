// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.hpp).
// orc_transform.hpp: src/geometry/transform.rs restated, plus the cgmath 0.17.0 Matrix4 operations it
// calls (column-major; products and inverse restated from the published cgmath source -- bit-level
// parity unpinned).
#pragma once
#include "orc_math.hpp"

namespace orc {

struct Mat4 {
    Float a[16];  // a[c*4 + r] == cgmath m[c][r]
    Float at(int c, int r) const { return a[c * 4 + r]; }
    Float& at(int c, int r) { return a[c * 4 + r]; }
    static Mat4 identity() {
        Mat4 m; for (int i = 0; i < 16; i++) m.a[i] = (i % 5 == 0) ? 1.0f : 0.0f; return m;
    }
    // Matrix4::new(c0r0, c0r1, c0r2, c0r3, c1r0, ...)
    static Mat4 from_flat(const Float* f) { Mat4 m; for (int i = 0; i < 16; i++) m.a[i] = f[i]; return m; }
};

// cgmath: Matrix4 * Matrix4: column j = a*rhs[j][0] + b*rhs[j][1] + c*rhs[j][2] + d*rhs[j][3]
inline Mat4 mul(const Mat4& l, const Mat4& r) {
    Mat4 o;
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++)
            o.at(j, i) = ((l.at(0, i) * r.at(j, 0) + l.at(1, i) * r.at(j, 1)) + l.at(2, i) * r.at(j, 2)) + l.at(3, i) * r.at(j, 3);
    return o;
}

// cgmath 0.17 det_sub_proc_unsafe(m, x, y, z) -> Vector4
inline void det_sub_proc(const Mat4& m, int x, int y, int z, Float out[4]) {
    const Float* s = m.a;
    Float a[4] = {s[4 + x], s[12 + x], s[x], s[8 + x]};
    Float b[4] = {s[8 + y], s[8 + y], s[4 + y], s[4 + y]};
    Float c[4] = {s[12 + z], s[z], s[12 + z], s[z]};
    Float d[4] = {s[8 + x], s[8 + x], s[4 + x], s[4 + x]};
    Float e[4] = {s[12 + y], s[y], s[12 + y], s[y]};
    Float f[4] = {s[4 + z], s[12 + z], s[z], s[8 + z]};
    Float g[4] = {s[12 + x], s[x], s[12 + x], s[x]};
    Float h[4] = {s[4 + y], s[12 + y], s[y], s[8 + y]};
    Float i[4] = {s[8 + z], s[8 + z], s[4 + z], s[4 + z]};
    for (int k = 0; k < 4; k++) {
        Float t = a[k] * (b[k] * c[k]);
        t += d[k] * (e[k] * f[k]);
        t += g[k] * (h[k] * i[k]);
        t -= a[k] * (e[k] * i[k]);
        t -= d[k] * (h[k] * c[k]);
        t -= g[k] * (b[k] * f[k]);
        out[k] = t;
    }
}
inline Float dot4(const Float a[4], Float b0, Float b1, Float b2, Float b3) {
    return ((a[0] * b0 + a[1] * b1) + a[2] * b2) + a[3] * b3;   // Vector4::dot: sum of element-wise products
}
inline Float determinant(const Mat4& m) {
    Float tmp[4]; det_sub_proc(m, 1, 2, 3, tmp);
    return dot4(tmp, m.at(0, 0), m.at(1, 0), m.at(2, 0), m.at(3, 0));
}
inline bool invert(const Mat4& m, Mat4* out) {
    Float t0[4]; det_sub_proc(m, 1, 2, 3, t0);
    Float det = dot4(t0, m.at(0, 0), m.at(1, 0), m.at(2, 0), m.at(3, 0));
    if (det == 0.0f) return false;
    Float inv_det = 1.0f / det;
    Float t1[4], t2[4], t3[4];
    det_sub_proc(m, 0, 3, 2, t1); det_sub_proc(m, 0, 1, 3, t2); det_sub_proc(m, 0, 2, 1, t3);
    for (int k = 0; k < 4; k++) {
        out->at(0, k) = t0[k] * inv_det; out->at(1, k) = t1[k] * inv_det;
        out->at(2, k) = t2[k] * inv_det; out->at(3, k) = t3[k] * inv_det;
    }
    return true;
}

// cgmath Matrix4 * Vector4, then truncate / from_homogeneous
inline Vec3 mat_transform_vector(const Mat4& m, Vec3 v) {   // (m * v.extend(0)).truncate()
    Vec3 o;
    for (int i = 0; i < 3; i++) o[i] = ((m.at(0, i) * v.x + m.at(1, i) * v.y) + m.at(2, i) * v.z) + m.at(3, i) * 0.0f;
    return o;
}
inline Vec3 mat_transform_point(const Mat4& m, Vec3 p) {    // Point3::from_homogeneous(m * p.to_homogeneous())
    Float h[4];
    for (int i = 0; i < 4; i++) h[i] = ((m.at(0, i) * p.x + m.at(1, i) * p.y) + m.at(2, i) * p.z) + m.at(3, i) * 1.0f;
    Float e = 1.0f / h[3];
    return Vec3(h[0] * e, h[1] * e, h[2] * e);
}

// ---- Transform: src/geometry/transform.rs:7-149
struct Transform {
    Mat4 t, invt;
    static Transform identity() { Transform r; r.t = Mat4::identity(); r.invt = Mat4::identity(); return r; }
    static Transform make(const Mat4& m, const Mat4& mi) { Transform r; r.t = m; r.invt = mi; return r; }
    static bool from_mat(const Mat4& m, Transform* out) {          // :23-26
        Mat4 mi; if (!invert(m, &mi)) return false; *out = make(m, mi); return true;
    }
    Transform inverse() const { return make(invt, t); }            // :117-119
    bool swaps_handedness() const { return determinant(t) < 0.0f; }  // :121-123
};
inline Transform operator*(const Transform& a, const Transform& b) {  // :143-149
    return Transform::make(mul(a.t, b.t), mul(b.invt, a.invt));
}
inline Transform tf_translate(Vec3 d) {                           // :62-66 (Matrix4::from_translation)
    Mat4 m = Mat4::identity(), mi = Mat4::identity();
    m.at(3, 0) = d.x; m.at(3, 1) = d.y; m.at(3, 2) = d.z;
    mi.at(3, 0) = -d.x; mi.at(3, 1) = -d.y; mi.at(3, 2) = -d.z;
    return Transform::make(m, mi);
}
inline Transform tf_scale(Float sx, Float sy, Float sz) {         // :68-72
    Mat4 m = Mat4::identity(), mi = Mat4::identity();
    m.at(0, 0) = sx; m.at(1, 1) = sy; m.at(2, 2) = sz;
    mi.at(0, 0) = 1.0f / sx; mi.at(1, 1) = 1.0f / sy; mi.at(2, 2) = 1.0f / sz;
    return Transform::make(m, mi);
}
inline Float deg_to_rad_cgmath(Float deg) { return deg * (Float)(3.14159265358979323846 / 180.0); }  // Rad::from(Deg)
inline bool tf_rotate(Float angle_deg, Vec3 axis, Transform* out) {   // :74-78, Matrix4::from_axis_angle
    Vec3 a = normalize(axis);
    Float ang = deg_to_rad_cgmath(angle_deg);
    Float s = m_sin(ang), c = m_cos(ang);
    Float omc = 1.0f - c;
    Float f[16] = {
        omc * a.x * a.x + c,       omc * a.x * a.y + s * a.z, omc * a.x * a.z - s * a.y, 0.0f,
        omc * a.x * a.y - s * a.z, omc * a.y * a.y + c,       omc * a.y * a.z + s * a.x, 0.0f,
        omc * a.x * a.z + s * a.y, omc * a.y * a.z - s * a.x, omc * a.z * a.z + c,       0.0f,
        0.0f, 0.0f, 0.0f, 1.0f};
    return Transform::from_mat(Mat4::from_flat(f), out);
}
inline bool tf_look_at(Vec3 pos, Vec3 look, Vec3 up, Transform* out) {  // :41-56
    Vec3 dir = normalize(look - pos);
    Vec3 right = normalize(cross(normalize(up), dir));
    Vec3 new_up = cross(dir, right);
    Float f[16] = {right.x, right.y, right.z, 0.0f, new_up.x, new_up.y, new_up.z, 0.0f,
                   dir.x, dir.y, dir.z, 0.0f, pos.x, pos.y, pos.z, 1.0f};
    Mat4 mat = Mat4::from_flat(f), minv;
    if (!invert(mat, &minv)) return false;
    *out = Transform::make(minv, mat);
    return true;
}
inline bool tf_perspective(Float fov, Float n, Float f, Transform* out) {  // :105-115
    Float m[16] = {1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f,
                   0.0f, 0.0f, f / (f - n), 1.0f, 0.0f, 0.0f, -f * n / (f - n), 0.0f};
    Float inv_tan_ang = 1.0f / m_tan(to_radians(fov) / 2.0f);
    Transform p;
    if (!Transform::from_mat(Mat4::from_flat(m), &p)) return false;
    *out = tf_scale(inv_tan_ang, inv_tan_ang, 1.0f) * p;
    return true;
}

// transform_normal: :125-131
inline Vec3 tf_normal(const Transform& tf, Vec3 n) {
    const Mat4& i = tf.invt;
    Float x = i.at(0, 0) * n.x + i.at(1, 0) * n.y + i.at(2, 0) * n.z;
    Float y = i.at(0, 1) * n.x + i.at(1, 1) * n.y + i.at(2, 1) * n.z;
    Float z = i.at(0, 2) * n.x + i.at(1, 2) * n.y + i.at(2, 2) * n.z;
    return Vec3(x, y, z);
}
inline Vec3 tf_vector(const Transform& tf, Vec3 v) { return mat_transform_vector(tf.t, v); }  // :176-178
inline Vec3 tf_point(const Transform& tf, Vec3 p) { return mat_transform_point(tf.t, p); }    // :223-225

// Vec3f::tf_exact_to_err: :183-197
inline Vec3 tf_vector_exact_to_err(const Transform& tf, Vec3 v, Vec3* err) {
    Vec3 vt = mat_transform_vector(tf.t, v);
    const Mat4& m = tf.t;
    Float xs = fabsf(m.at(0, 0) * v.x) + fabsf(m.at(1, 0) * v.y) + fabsf(m.at(2, 0) * v.z);
    Float ys = fabsf(m.at(0, 1) * v.x) + fabsf(m.at(1, 1) * v.y) + fabsf(m.at(2, 1) * v.z);
    Float zs = fabsf(m.at(0, 2) * v.x) + fabsf(m.at(1, 2) * v.y) + fabsf(m.at(2, 2) * v.z);
    *err = Vec3(xs, ys, zs) * gamma(3);
    return vt;
}
// Point3f::tf_exact_to_err: :230-244
inline Vec3 tf_point_exact_to_err(const Transform& tf, Vec3 p, Vec3* err) {
    Vec3 pt = mat_transform_point(tf.t, p);
    const Mat4& m = tf.t;
    Float xs = fabsf(m.at(0, 0) * p.x) + fabsf(m.at(1, 0) * p.y) + fabsf(m.at(2, 0) * p.z) + fabsf(m.at(3, 0));
    Float ys = fabsf(m.at(0, 1) * p.x) + fabsf(m.at(1, 1) * p.y) + fabsf(m.at(2, 1) * p.z) + fabsf(m.at(3, 1));
    Float zs = fabsf(m.at(0, 2) * p.x) + fabsf(m.at(1, 2) * p.y) + fabsf(m.at(2, 2) * p.z) + fabsf(m.at(3, 2));
    *err = Vec3(xs, ys, zs) * gamma(3);
    return pt;
}
// Point3f::tf_err_to_err: :246-267 (NB the z row takes abs of products, x/y rows abs of the matrix entry)
inline Vec3 tf_point_err_to_err(const Transform& tf, Vec3 p, Vec3 perr, Vec3* err) {
    Vec3 pt = mat_transform_point(tf.t, p);
    const Mat4& m = tf.t;
    const Float g3 = gamma(3);
    Float xerr = (g3 + 1.0f) * (fabsf(m.at(0, 0)) * perr.x + fabsf(m.at(1, 0)) * perr.y + fabsf(m.at(2, 0)) * perr.z) +
                 g3 * (fabsf(m.at(0, 0) * p.x) + fabsf(m.at(1, 0) * p.y) + fabsf(m.at(2, 0) * p.z) + fabsf(m.at(3, 0)));
    Float yerr = (g3 + 1.0f) * (fabsf(m.at(0, 1)) * perr.x + fabsf(m.at(1, 1)) * perr.y + fabsf(m.at(2, 1)) * perr.z) +
                 g3 * (fabsf(m.at(0, 1) * p.x) + fabsf(m.at(1, 1) * p.y) + fabsf(m.at(2, 1) * p.z) + fabsf(m.at(3, 1)));
    Float zerr = (g3 + 1.0f) * (fabsf(m.at(0, 2) * perr.x) + fabsf(m.at(1, 2) * perr.y) + fabsf(m.at(2, 2) * perr.z)) +
                 g3 * (fabsf(m.at(0, 2) * p.x) + fabsf(m.at(1, 2) * p.y) + fabsf(m.at(2, 2) * p.z) + fabsf(m.at(3, 2)));
    *err = Vec3(xerr, yerr, zerr);
    return pt;
}
// Vec3f::tf_err_to_err: :200-220
inline Vec3 tf_vector_err_to_err(const Transform& tf, Vec3 v, Vec3 verr, Vec3* err) {
    Vec3 vt = mat_transform_vector(tf.t, v);
    const Mat4& m = tf.t;
    const Float g3 = gamma(3);
    Float e[3];
    for (int r = 0; r < 3; r++)
        e[r] = (g3 + 1.0f) * (fabsf(m.at(0, r) * verr.x) + fabsf(m.at(1, r) * verr.y) + fabsf(m.at(2, r) * verr.z)) +
               g3 * (fabsf(m.at(0, r) * v.x) + fabsf(m.at(1, r) * v.y) + fabsf(m.at(2, r) * v.z));
    *err = Vec3(e[0], e[1], e[2]);
    return vt;
}

// Ray::tf_exact_to_err: :287-300
inline Ray tf_ray_exact_to_err(const Transform& tf, const Ray& r, Vec3* o_err, Vec3* d_err) {
    Vec3 ot = tf_point_exact_to_err(tf, r.origin, o_err);
    Vec3 dir_t = tf_vector_exact_to_err(tf, r.dir, d_err);
    Float tmax = r.t_max;
    Float len_sq = magnitude2(dir_t);
    if (len_sq > 0.0f) {
        Float dt = dot(vabs(dir_t), *o_err) / len_sq;
        ot = ot + dir_t * dt;
        tmax -= dt;
    }
    Ray out; out.origin = ot; out.dir = dir_t; out.t_max = tmax; out.time = r.time;
    return out;
}
// Ray::transform: :307-322
inline Ray tf_ray(const Transform& tf, const Ray& r) {
    Vec3 o_err;
    Vec3 ot = tf_point_exact_to_err(tf, r.origin, &o_err);
    Vec3 dir = tf_vector(tf, r.dir);
    Float t_max = r.t_max;
    Float len_sq = magnitude2(dir);
    if (len_sq > 0.0f) {
        Float dt = dot(vabs(dir), o_err) / len_sq;
        ot = ot + dir * dt;
        t_max -= dt;
    }
    Ray out; out.origin = ot; out.dir = dir; out.t_max = t_max; out.time = r.time;
    return out;
}
// RayDifferential::transform: :325-338
inline RayDifferential tf_ray_diff(const Transform& tf, const RayDifferential& rd) {
    RayDifferential o;
    o.ray = tf_ray(tf, rd.ray);
    o.has_diff = rd.has_diff;
    if (rd.has_diff) {
        o.diff.rx_origin = tf_point(tf, rd.diff.rx_origin);
        o.diff.ry_origin = tf_point(tf, rd.diff.ry_origin);
        o.diff.rx_dir = tf_vector(tf, rd.diff.rx_dir);
        o.diff.ry_dir = tf_vector(tf, rd.diff.ry_dir);
    }
    return o;
}

}  // namespace orc
