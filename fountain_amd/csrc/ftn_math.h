/*
 * ftn_math.h -- f32 vector / error-bound / spectrum arithmetic shared by the host constructors and the
 * gfx950 kernels of the fountain path-tracing core.  Every function states the reference expression it
 * evaluates; operation order follows the reference (and cgmath 0.17 where the reference delegates to it) so
 * that results are bit-identical to a CPU evaluation: build with -ffp-contract=off.
 *
 * Reference: src/math.rs, src/err_float.rs, src/geometry/mod.rs, src/spectrum/mod.rs.
 */
#ifndef FTN_MATH_H
#define FTN_MATH_H

#include "detmath.h"

#define FTN_INF (__builtin_huge_valf())
#define FTN_PI 3.14159265358979323846264338327950288f
#define FTN_INV_PI 0.318309886183790671537767526745028724f
#define FTN_PI_2 1.57079632679489661923132169163975144f
#define FTN_PI_4 0.785398163397448309615660845819875721f
#define FTN_EPS_HALF 5.9604644775390625e-08f /* f32::EPSILON * 0.5  (err_float.rs:5) */

namespace ftn {

using ftn_det::f2u;
using ftn_det::u2f;

/* gamma(n) = n*eps / (1 - n*eps)  (err_float.rs:7-10) */
FTN_HD constexpr float gamma_n(int n) { return ((float)n * FTN_EPS_HALF) / (1.0f - (float)n * FTN_EPS_HALF); }

/* Rust f32::max / min (NaN-ignoring) and f32::clamp */
FTN_HD float fmax_(float a, float b) { return fmaxf(a, b); }
FTN_HD float fmin_(float a, float b) { return fminf(a, b); }
FTN_HD float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
FTN_HD bool sign_pos(float v) { return (f2u(v) >> 31) == 0; }
FTN_HD bool is_inf(float v) { return (f2u(v) & 0x7fffffffu) == 0x7f800000u; }
FTN_HD bool is_nan(float v) { return v != v; }
/* `as i32` / `as usize` casts: saturating, NaN -> 0 */
FTN_HD int f2i_sat(float v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}
FTN_HD long long f2usize(float v) {
    if (v != v || v <= 0.0f) return 0;
    if (v >= 9.2e18f) return 0x7fffffffffffffffLL;
    return (long long)v;
}

/* next_float_up / next_float_down exactly as err_float.rs:12-30 (including the -0.0 >= 0.0 quirk) */
FTN_HD float next_up(float v) {
    if (v == FTN_INF) return v;
    if (v == -0.0f) v = 0.0f;
    uint32_t b = f2u(v);
    b = (v >= 0.0f) ? b + 1u : b - 1u;
    return u2f(b);
}
FTN_HD float next_down(float v) {
    if (v == -FTN_INF) return v;
    if (v == 0.0f) v = -0.0f;
    uint32_t b = f2u(v);
    b = (v >= 0.0f) ? b - 1u : b + 1u;
    return u2f(b);
}

struct V3 {
    float x, y, z;
    FTN_HD V3() : x(0.0f), y(0.0f), z(0.0f) {}
    FTN_HD V3(float a, float b, float c) : x(a), y(b), z(c) {}
    FTN_HD float get(int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    FTN_HD void set(int i, float v) { if (i == 0) x = v; else if (i == 1) y = v; else z = v; }
};
FTN_HD V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
FTN_HD V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
FTN_HD V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
FTN_HD V3 operator*(V3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
FTN_HD V3 operator*(float s, V3 a) { return V3(s * a.x, s * a.y, s * a.z); }
FTN_HD V3 operator/(V3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
FTN_HD bool veq(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
FTN_HD float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }   /* cgmath: sum of element products */
FTN_HD V3 cross(V3 a, V3 b) { return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
FTN_HD float len2(V3 a) { return dot(a, a); }
FTN_HD float len(V3 a) { return sqrtf(dot(a, a)); }
FTN_HD V3 normalize(V3 a) { return a * (1.0f / len(a)); }                      /* cgmath: v * (1/|v|) */
FTN_HD V3 vabs(V3 a) { return V3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
FTN_HD float abs_dot(V3 a, V3 b) { return fabsf(dot(a, b)); }                 /* math.rs:32-34 */
FTN_HD V3 faceforward(V3 a, V3 b) { return dot(a, b) < 0.0f ? -a : a; }       /* geometry/mod.rs:64-70 */

struct V2 { float x, y; FTN_HD V2() : x(0.0f), y(0.0f) {} FTN_HD V2(float a, float b) : x(a), y(b) {} };

/* geometry/mod.rs:45-62 */
FTN_HD int max_dimension(V3 v) { return v.x > v.y ? (v.x > v.z ? 0 : 2) : (v.y > v.z ? 1 : 2); }
FTN_HD void coordinate_system(V3 v1, V3* v2, V3* v3) {
    if (fabsf(v1.x) > fabsf(v1.y)) *v2 = normalize(V3(-v1.z, 0.0f, v1.x));
    else *v2 = normalize(V3(0.0f, v1.z, -v1.y));
    *v3 = cross(v1, *v2);
}
/* offset_ray_origin: geometry/mod.rs:72-85 */
FTN_HD V3 offset_ray_origin(V3 p, V3 p_err, V3 n, V3 dir) {
    float d = dot(vabs(n), p_err);
    V3 off = d * n;
    if (dot(dir, n) < 0.0f) off = -off;
    V3 po = p + off;
    if (off.x > 0.0f) po.x = next_up(po.x); else if (off.x < 0.0f) po.x = next_down(po.x);
    if (off.y > 0.0f) po.y = next_up(po.y); else if (off.y < 0.0f) po.y = next_down(po.y);
    if (off.z > 0.0f) po.z = next_up(po.z); else if (off.z < 0.0f) po.z = next_down(po.z);
    return po;
}
/* geometry/mod.rs:23-34 */
FTN_HD float spherical_theta(V3 v) { return ftn_det::acosf_det(clampf(v.z, -1.0f, 1.0f)); }
FTN_HD float spherical_phi(V3 v) { float p = ftn_det::atan2f_det(v.y, v.x); return p < 0.0f ? p + (2.0f * FTN_PI) : p; }
/* math.rs:74-80 */
FTN_HD V3 spherical_direction(float st, float ct, float phi) { float s, c; ftn_det::sincosf_det(phi, &s, &c); return V3(st * c, st * s, ct); }

/* ---- Spectrum (RGB f32): spectrum/mod.rs */
struct Rgb {
    float r, g, b;
    FTN_HD Rgb() : r(0.0f), g(0.0f), b(0.0f) {}
    FTN_HD explicit Rgb(float v) : r(v), g(v), b(v) {}
    FTN_HD Rgb(float x, float y, float z) : r(x), g(y), b(z) {}
    FTN_HD bool is_black() const { return r == 0.0f && g == 0.0f && b == 0.0f; }
    FTN_HD bool has_nans() const { return r != r || g != g || b != b; }
    FTN_HD float max_component() const { float m = r; if (!(g < m)) m = g; if (!(b < m)) m = b; return m; }   /* max_by(total_cmp) */
    FTN_HD float luminance() const { return r * 0.212671f + g * 0.715160f + b * 0.072169f; }
};
#define FTN_RGB_OP(op) \
    FTN_HD Rgb operator op(Rgb a, Rgb b) { return Rgb(a.r op b.r, a.g op b.g, a.b op b.b); } \
    FTN_HD Rgb operator op(Rgb a, float s) { return Rgb(a.r op s, a.g op s, a.b op s); } \
    FTN_HD Rgb operator op(float s, Rgb a) { return Rgb(s op a.r, s op a.g, s op a.b); }
FTN_RGB_OP(+) FTN_RGB_OP(-) FTN_RGB_OP(*) FTN_RGB_OP(/)
#undef FTN_RGB_OP
FTN_HD Rgb rgb_sqrt(Rgb a) { return Rgb(sqrtf(a.r), sqrtf(a.g), sqrtf(a.b)); }
FTN_HD Rgb clamp_positive(Rgb a) { return Rgb(clampf(a.r, 0.0f, FTN_INF), clampf(a.g, 0.0f, FTN_INF), clampf(a.b, 0.0f, FTN_INF)); }
/* spectrum/mod.rs:28-43 */
FTN_HD void xyz_to_rgb(const float xyz[3], float rgb[3]) {
    rgb[0] = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];
    rgb[1] = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
    rgb[2] = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
}
FTN_HD void rgb_to_xyz(Rgb c, float xyz[3]) {
    xyz[0] = 0.412453f * c.r + 0.357580f * c.g + 0.180423f * c.b;
    xyz[1] = 0.212671f * c.r + 0.715160f * c.g + 0.072169f * c.b;
    xyz[2] = 0.019334f * c.r + 0.119193f * c.g + 0.950227f * c.b;
}

/* ---- EFloat interval arithmetic: err_float.rs:33-225 */
struct EF {
    float v, lo, hi;
    FTN_HD EF() : v(0.0f), lo(0.0f), hi(0.0f) {}
    FTN_HD explicit EF(float a) : v(a), lo(a), hi(a) {}
    FTN_HD EF(float a, float l, float h) : v(a), lo(l), hi(h) {}
};
FTN_HD EF ef_err(float v, float err) { return err == 0.0f ? EF(v) : EF(v, next_down(v - err), next_up(v + err)); }
FTN_HD EF operator+(EF a, EF b) { return EF(a.v + b.v, next_down(a.lo + b.lo), next_up(a.hi + b.hi)); }
FTN_HD EF operator-(EF a, EF b) { return EF(a.v - b.v, next_down(a.lo - b.lo), next_up(a.hi - b.hi)); }   /* sic, :116-125 */
FTN_HD EF operator*(EF a, EF b) {
    float p1 = a.lo * b.lo, p2 = a.hi * b.lo, p3 = a.lo * b.hi, p4 = a.hi * b.hi;
    return EF(a.v * b.v, next_down(fmin_(fmin_(p1, p2), fmin_(p3, p4))), next_up(fmax_(fmax_(p1, p2), fmax_(p3, p4))));
}
FTN_HD EF operator/(EF a, EF b) {
    float v = a.v / b.v;
    if (b.lo < 0.0f && b.hi > 0.0f) return EF(v, -FTN_INF, FTN_INF);
    float d1 = a.lo / b.lo, d2 = a.hi / b.lo, d3 = a.lo / b.hi, d4 = a.hi / b.hi;
    return EF(v, next_down(fmin_(fmin_(d1, d2), fmin_(d3, d4))), next_up(fmax_(fmax_(d1, d2), fmax_(d3, d4))));
}
FTN_HD EF ef_neg(EF a) { return EF(-a.v, -a.hi, -a.lo); }
/* quadratic: math.rs:36-53 (f64 discriminant) */
FTN_HD bool quadratic(EF a, EF b, EF c, EF* t0, EF* t1) {
    double discrim = (double)b.v * (double)b.v - (4.0 * (double)a.v * (double)c.v);
    if (discrim < 0.0) return false;
    double rd = sqrt(discrim);
    EF root = ef_err((float)rd, FTN_EPS_HALF * (float)rd);
    EF q = (b.v < 0.0f) ? (EF(-0.5f) * (b - root)) : (EF(-0.5f) * (b + root));
    EF r0 = q / a, r1 = c / q;
    if (r0.v > r1.v) { *t0 = r1; *t1 = r0; } else { *t0 = r0; *t1 = r1; }
    return true;
}

/* ---- 4x4 column-major matrices (cgmath Matrix4): m[c*4+r] */
struct M4 { float a[16]; };
FTN_HD V3 m4_vector(const float* m, V3 v) {   /* (m * v.extend(0)).truncate(), transform.rs:177 */
    return V3(((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12] * 0.0f,
              ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13] * 0.0f,
              ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14] * 0.0f);
}
FTN_HD V3 m4_point(const float* m, V3 p) {    /* Point3::from_homogeneous(m * p.to_homogeneous()), transform.rs:224 */
    float hx = ((m[0] * p.x + m[4] * p.y) + m[8] * p.z) + m[12] * 1.0f;
    float hy = ((m[1] * p.x + m[5] * p.y) + m[9] * p.z) + m[13] * 1.0f;
    float hz = ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14] * 1.0f;
    float hw = ((m[3] * p.x + m[7] * p.y) + m[11] * p.z) + m[15] * 1.0f;
    float e = 1.0f / hw;
    return V3(hx * e, hy * e, hz * e);
}
FTN_HD V3 m4_normal(const float* inv, V3 n) {  /* transform_normal with the inverse, transform.rs:125-131 */
    return V3(inv[0] * n.x + inv[4] * n.y + inv[8] * n.z, inv[1] * n.x + inv[5] * n.y + inv[9] * n.z, inv[2] * n.x + inv[6] * n.y + inv[10] * n.z);
}
/* Point3f::tf_exact_to_err: transform.rs:230-244 */
FTN_HD V3 m4_point_exact_to_err(const float* m, V3 p, V3* err) {
    V3 pt = m4_point(m, p);
    float xs = fabsf(m[0] * p.x) + fabsf(m[4] * p.y) + fabsf(m[8] * p.z) + fabsf(m[12]);
    float ys = fabsf(m[1] * p.x) + fabsf(m[5] * p.y) + fabsf(m[9] * p.z) + fabsf(m[13]);
    float zs = fabsf(m[2] * p.x) + fabsf(m[6] * p.y) + fabsf(m[10] * p.z) + fabsf(m[14]);
    *err = V3(xs, ys, zs) * gamma_n(3);
    return pt;
}
/* Vec3f::tf_exact_to_err: transform.rs:183-197 */
FTN_HD V3 m4_vector_exact_to_err(const float* m, V3 v, V3* err) {
    V3 vt = m4_vector(m, v);
    float xs = fabsf(m[0] * v.x) + fabsf(m[4] * v.y) + fabsf(m[8] * v.z);
    float ys = fabsf(m[1] * v.x) + fabsf(m[5] * v.y) + fabsf(m[9] * v.z);
    float zs = fabsf(m[2] * v.x) + fabsf(m[6] * v.y) + fabsf(m[10] * v.z);
    *err = V3(xs, ys, zs) * gamma_n(3);
    return vt;
}
/* Point3f::tf_err_to_err: transform.rs:246-267 (x/y rows: |m|*err, z row: |m*err|) */
FTN_HD V3 m4_point_err_to_err(const float* m, V3 p, V3 pe, V3* err) {
    V3 pt = m4_point(m, p);
    const float g3 = gamma_n(3);
    float xe = (g3 + 1.0f) * (fabsf(m[0]) * pe.x + fabsf(m[4]) * pe.y + fabsf(m[8]) * pe.z) +
               g3 * (fabsf(m[0] * p.x) + fabsf(m[4] * p.y) + fabsf(m[8] * p.z) + fabsf(m[12]));
    float ye = (g3 + 1.0f) * (fabsf(m[1]) * pe.x + fabsf(m[5]) * pe.y + fabsf(m[9]) * pe.z) +
               g3 * (fabsf(m[1] * p.x) + fabsf(m[5] * p.y) + fabsf(m[9] * p.z) + fabsf(m[13]));
    float ze = (g3 + 1.0f) * (fabsf(m[2] * pe.x) + fabsf(m[6] * pe.y) + fabsf(m[10] * pe.z)) +
               g3 * (fabsf(m[2] * p.x) + fabsf(m[6] * p.y) + fabsf(m[10] * p.z) + fabsf(m[14]));
    *err = V3(xe, ye, ze);
    return pt;
}

/* ---- sampling warps: sampling.rs:5-57 */
FTN_HD V2 concentric_sample_disk(V2 u) {
    float ox = 2.0f * u.x - 1.0f, oy = 2.0f * u.y - 1.0f;
    if (ox == 0.0f && oy == 0.0f) return V2(0.0f, 0.0f);
    float theta, r;
    if (fabsf(ox) > fabsf(oy)) { theta = FTN_PI_4 * (oy / ox); r = ox; }
    else { theta = FTN_PI_2 - FTN_PI_4 * (ox / oy); r = oy; }
    float s, c; ftn_det::sincosf_det(theta, &s, &c);
    return V2(r * c, r * s);
}
FTN_HD V3 cosine_sample_hemisphere(V2 u) {
    V2 d = concentric_sample_disk(u);
    return V3(d.x, d.y, sqrtf(fmax_(0.0f, 1.0f - d.x * d.x - d.y * d.y)));
}
FTN_HD V3 uniform_sample_sphere(V2 u) {
    float z = 1.0f - 2.0f * u.x;
    float r = sqrtf(fmax_(1.0f - z * z, 0.0f));
    float phi = 2.0f * FTN_PI * u.y;
    float s, c; ftn_det::sincosf_det(phi, &s, &c);
    return V3(r * c, r * s, z);
}
FTN_HD V2 uniform_sample_triangle(V2 u) { float s = sqrtf(u.x); return V2(1.0f - s, u.y * s); }
FTN_HD float power_heuristic(float f_pdf, float g_pdf) {   /* nf = ng = 1 (sampling.rs:53-57) */
    float f = 1.0f * f_pdf, g = 1.0f * g_pdf;
    return (f * f) / (f * f + g * g);
}

}  // namespace ftn
#endif
