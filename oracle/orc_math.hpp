// ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement of akofke/fountain's path-tracing hot path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this code; the product
// (fountain_amd/) never links, imports or executes anything under oracle/.
//
// Parity status: the Rust reference cannot be built here (no cargo/rustc), so this
// restatement is pinned by the reference's own known-answer tests (tests/test_oracle_kat.py lists each
// with its reference file:line).  Arithmetic that lives in un-vendored crates (cgmath 0.17.0,
// rand_xoshiro 0.2.0, rand 0.6.5) is restated from the published algorithms: "parity unpinned" at bit
// level for those, as SURVEY.md 8(c) records.
//
// orc_math.hpp: L1 math substrate.
//   src/math.rs, src/err_float.rs, src/geometry/mod.rs, src/spectrum/mod.rs, cgmath vector/matrix ops.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <algorithm>

#ifdef ORC_DETMATH
#include "../fountain_amd/csrc/detmath.h"
#endif

namespace orc {

typedef float Float;  // src/math.rs:10

static const Float INF = std::numeric_limits<Float>::infinity();
static const Float PI = 3.14159265358979323846264338327950288f;        // std::f32::consts::PI
static const Float FRAC_1_PI = 0.318309886183790671537767526745028724f;
static const Float FRAC_PI_2 = 1.57079632679489661923132169163975144f;
static const Float FRAC_PI_4 = 0.785398163397448309615660845819875721f;

// ---- transcendental functions: Rust f32::sin etc. lower to the platform libm.
#ifdef ORC_DETMATH
inline Float m_sin(Float x) { return ftn_det::sinf_det(x); }
inline Float m_cos(Float x) { return ftn_det::cosf_det(x); }
inline Float m_tan(Float x) { return ftn_det::tanf_det(x); }
inline Float m_acos(Float x) { return ftn_det::acosf_det(x); }
inline Float m_atan(Float x) { return ftn_det::atanf_det(x); }
inline Float m_atan2(Float y, Float x) { return ftn_det::atan2f_det(y, x); }
inline Float m_ln(Float x) { return ftn_det::logf_det(x); }
inline Float m_log2(Float x) { return ftn_det::log2f_det(x); }
inline Float m_pow(Float x, Float y) { return ftn_det::powf_det(x, y); }
#else
inline Float m_sin(Float x) { return sinf(x); }
inline Float m_cos(Float x) { return cosf(x); }
inline Float m_tan(Float x) { return tanf(x); }
inline Float m_acos(Float x) { return acosf(x); }
inline Float m_atan(Float x) { return atanf(x); }
inline Float m_atan2(Float y, Float x) { return atan2f(y, x); }
inline Float m_ln(Float x) { return logf(x); }
inline Float m_log2(Float x) { return log2f(x); }
inline Float m_pow(Float x, Float y) { return powf(x, y); }
#endif

// ---- Rust float semantics helpers
inline Float fmax_(Float a, Float b) { return fmaxf(a, b); }  // f32::max: NaN-ignoring
inline Float fmin_(Float a, Float b) { return fminf(a, b); }
// f32::clamp(min,max): if self < min {min} else if self > max {max} else {self}
inline Float clampf(Float v, Float lo, Float hi) { return v < lo ? lo : (v > hi ? hi : v); }
// `as usize` / `as i32` from f32: saturating, NaN -> 0
inline int64_t f2usize(Float v) {
    if (!(v == v)) return 0;
    if (v <= 0.0f) return 0;
    if (v >= 9.2e18f) return INT64_MAX;
    return (int64_t)v;
}
inline int32_t f2i32(Float v) {
    if (!(v == v)) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}
inline bool is_sign_positive(Float v) { uint32_t u; memcpy(&u, &v, 4); return (u >> 31) == 0; }
inline Float to_radians(Float deg) { return deg * (PI / 180.0f); }  // f32::to_radians

// ---- src/err_float.rs:5-30
static const Float MACHINE_EPSILON = std::numeric_limits<Float>::epsilon() * 0.5f;
inline constexpr Float gamma(int n) {
    return ((Float)n * (std::numeric_limits<Float>::epsilon() * 0.5f)) /
           (1.0f - (Float)n * (std::numeric_limits<Float>::epsilon() * 0.5f));
}
inline Float next_float_up(Float v) {
    if (v == INF) return v;
    if (v == -0.0f) v = 0.0f;
    uint32_t bits; memcpy(&bits, &v, 4);
    bits = (v >= 0.0f) ? bits + 1 : bits - 1;
    Float r; memcpy(&r, &bits, 4); return r;
}
inline Float next_float_down(Float v) {
    if (v == -INF) return v;
    if (v == 0.0f) v = -0.0f;
    uint32_t bits; memcpy(&bits, &v, 4);
    bits = (v >= 0.0f) ? bits - 1 : bits + 1;   // NB: -0.0 >= 0.0 is true, as in the reference
    Float r; memcpy(&r, &bits, 4); return r;
}

// ---- cgmath Vector3 / Point3 (cgmath 0.17: dot = (x*x + y*y) + z*z; normalize = v * (1/|v|))
struct Vec3 {
    Float x, y, z;
    Vec3() : x(0), y(0), z(0) {}
    Vec3(Float x_, Float y_, Float z_) : x(x_), y(y_), z(z_) {}
    Float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    Float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 operator-(Vec3 a) { return Vec3(-a.x, -a.y, -a.z); }
inline Vec3 operator*(Vec3 a, Float s) { return Vec3(a.x * s, a.y * s, a.z * s); }
inline Vec3 operator*(Float s, Vec3 a) { return Vec3(s * a.x, s * a.y, s * a.z); }
inline Vec3 operator/(Vec3 a, Float s) { return Vec3(a.x / s, a.y / s, a.z / s); }
inline bool operator==(Vec3 a, Vec3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline Float dot(Vec3 a, Vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b) {
    return Vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline Float magnitude2(Vec3 a) { return dot(a, a); }
inline Float magnitude(Vec3 a) { return sqrtf(dot(a, a)); }
inline Vec3 normalize(Vec3 a) { return a * (1.0f / magnitude(a)); }
inline Vec3 vabs(Vec3 a) { return Vec3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
inline Float abs_dot(Vec3 a, Vec3 b) { return fabsf(dot(a, b)); }          // src/math.rs:32-34
inline Float distance(Vec3 a, Vec3 b) { return magnitude(a - b); }          // src/geometry/mod.rs:15-17
inline Float distance_sq(Vec3 a, Vec3 b) { return magnitude2(a - b); }      // :19-21

struct Vec2 { Float x, y; Vec2() : x(0), y(0) {} Vec2(Float a, Float b) : x(a), y(b) {} Float operator[](int i) const { return i ? y : x; } };

// src/geometry/mod.rs:23-85
inline Float spherical_theta(Vec3 v) { return m_acos(clampf(v.z, -1.0f, 1.0f)); }
inline Float spherical_phi(Vec3 v) {
    Float p = m_atan2(v.y, v.x);
    return p < 0.0f ? p + (2.0f * PI) : p;
}
inline int max_dimension(Vec3 v) {
    if (v.x > v.y) return v.x > v.z ? 0 : 2;
    return v.y > v.z ? 1 : 2;
}
inline Vec3 permute(Vec3 v, int ix, int iy, int iz) { return Vec3(v[ix], v[iy], v[iz]); }
inline void coordinate_system(Vec3 v1, Vec3* v2, Vec3* v3) {
    if (fabsf(v1.x) > fabsf(v1.y)) *v2 = normalize(Vec3(-v1.z, 0.0f, v1.x));
    else *v2 = normalize(Vec3(0.0f, v1.z, -v1.y));
    *v3 = cross(v1, *v2);
}
inline Vec3 faceforward(Vec3 v1, Vec3 v2) { return dot(v1, v2) < 0.0f ? -v1 : v1; }
// offset_ray_origin: src/geometry/mod.rs:72-85
inline Vec3 offset_ray_origin(Vec3 p, Vec3 p_err, Vec3 n, Vec3 dir) {
    Float d = dot(vabs(n), p_err);
    Vec3 offset = d * n;
    if (dot(dir, n) < 0.0f) offset = -offset;
    Vec3 po = p + offset;
    for (int i = 0; i < 3; i++) {
        if (offset[i] > 0.0f) po[i] = next_float_up(po[i]);
        else if (offset[i] < 0.0f) po[i] = next_float_down(po[i]);
    }
    return po;
}

// src/math.rs:74-80
inline Vec3 spherical_direction(Float sin_theta, Float cos_theta, Float phi) {
    return Vec3(sin_theta * m_cos(phi), sin_theta * m_sin(phi), cos_theta);
}
// src/math.rs:56-72 (A column-major: a00=A[0][0], a01=A[0][1] (col 0,row 1), ...)
inline bool solve_linear_system_2x2(Float a00, Float a01, Float a10, Float a11, Float b0, Float b1,
                                    Float* x0, Float* x1) {
    Float det = a00 * a11 - a10 * a01;   // cgmath Matrix2::determinant: self[0][0]*self[1][1] - self[1][0]*self[0][1]
    if (fabsf(det) < 1.0e-10f) return false;
    *x0 = (a11 * b0 - a10 * b1) / det;
    *x1 = (a00 * b1 - a01 * b0) / det;
    if (std::isnan(*x0) || std::isnan(*x1)) return false;
    return true;
}

// ---- Ray: src/geometry/mod.rs:87-107
struct Ray {
    Vec3 origin, dir;
    Float t_max, time;
    Ray() : t_max(INF), time(0) {}
    Ray(Vec3 o, Vec3 d) : origin(o), dir(d), t_max(INF), time(0) {}
    Vec3 at(Float t) const { return origin + (dir * t); }
};
struct Differential { Vec3 rx_origin, ry_origin, rx_dir, ry_dir; };
struct RayDifferential {
    Ray ray; bool has_diff; Differential diff;
    // src/geometry/mod.rs:125-133
    void scale_differentials(Float s) {
        if (has_diff) {
            diff.rx_origin = ray.origin + (diff.rx_origin - ray.origin) * s;
            diff.ry_origin = ray.origin + (diff.ry_origin - ray.origin) * s;
            diff.rx_dir = ray.dir + (diff.rx_dir - ray.dir) * s;
            diff.ry_dir = ray.dir + (diff.ry_dir - ray.dir) * s;
        }
    }
};

// ---- EFloat: src/err_float.rs:33-225
struct EFloat {
    Float v, low, high;
    EFloat() : v(0), low(0), high(0) {}
    explicit EFloat(Float v_) : v(v_), low(v_), high(v_) {}
    static EFloat with_err(Float v, Float err) {
        EFloat e;
        if (err == 0.0f) { e.v = v; e.low = v; e.high = v; return e; }
        e.v = v; e.low = next_float_down(v - err); e.high = next_float_up(v + err);
        return e;
    }
    static EFloat with_bounds(Float v, Float lo, Float hi) { EFloat e; e.v = v; e.low = lo; e.high = hi; return e; }
    Float upper_bound() const { return high; }
    Float lower_bound() const { return low; }
};
inline EFloat operator+(EFloat a, EFloat b) {
    return EFloat::with_bounds(a.v + b.v, next_float_down(a.low + b.low), next_float_up(a.high + b.high));
}
inline EFloat operator-(EFloat a, EFloat b) {
    // NB reference: low = down(self.low - rhs.low), high = up(self.high - rhs.high)  (err_float.rs:116-125)
    return EFloat::with_bounds(a.v - b.v, next_float_down(a.low - b.low), next_float_up(a.high - b.high));
}
inline EFloat operator*(EFloat a, EFloat b) {
    Float p1 = a.low * b.low, p2 = a.high * b.low, p3 = a.low * b.high, p4 = a.high * b.high;
    Float lo = next_float_down(fmin_(fmin_(p1, p2), fmin_(p3, p4)));
    Float hi = next_float_up(fmax_(fmax_(p1, p2), fmax_(p3, p4)));
    return EFloat::with_bounds(a.v * b.v, lo, hi);
}
inline EFloat operator/(EFloat a, EFloat b) {
    Float v = a.v / b.v;
    if (b.low < 0.0f && b.high > 0.0f) return EFloat::with_bounds(v, -INF, INF);
    Float d1 = a.low / b.low, d2 = a.high / b.low, d3 = a.low / b.high, d4 = a.high / b.high;
    Float lo = next_float_down(fmin_(fmin_(d1, d2), fmin_(d3, d4)));
    Float hi = next_float_up(fmax_(fmax_(d1, d2), fmax_(d3, d4)));
    return EFloat::with_bounds(v, lo, hi);
}
inline EFloat operator-(EFloat a) { return EFloat::with_bounds(-a.v, -a.high, -a.low); }
inline EFloat operator*(Float s, EFloat a) { return EFloat(s) * a; }

// quadratic: src/math.rs:36-53
inline bool quadratic(EFloat a, EFloat b, EFloat c, EFloat* t0, EFloat* t1) {
    double discrim = (double)b.v * (double)b.v - (4.0 * (double)a.v * (double)c.v);
    if (discrim < 0.0) return false;
    double root_discrim_d = sqrt(discrim);
    EFloat root_discrim = EFloat::with_err((Float)root_discrim_d, MACHINE_EPSILON * (Float)root_discrim_d);
    EFloat q = (b.v < 0.0f) ? (-0.5f * (b - root_discrim)) : (-0.5f * (b + root_discrim));
    EFloat r0 = q / a;
    EFloat r1 = c / q;
    if (r0.v > r1.v) { *t0 = r1; *t1 = r0; } else { *t0 = r0; *t1 = r1; }
    return true;
}

// ---- Spectrum = CoefficientSpectrum<3>: src/spectrum/mod.rs
struct Spectrum {
    Float c[3];
    Spectrum() { c[0] = c[1] = c[2] = 0.0f; }
    explicit Spectrum(Float v) { c[0] = c[1] = c[2] = v; }
    Spectrum(Float r, Float g, Float b) { c[0] = r; c[1] = g; c[2] = b; }
    Float operator[](int i) const { return c[i]; }
    Float& operator[](int i) { return c[i]; }
    bool is_black() const { return c[0] == 0.0f && c[1] == 0.0f && c[2] == 0.0f; }
    bool has_nans() const { return std::isnan(c[0]) || std::isnan(c[1]) || std::isnan(c[2]); }
    // max_by(total_cmp): last maximum wins; identical to fmax for non-NaN inputs
    Float max_component_value() const {
        Float m = c[0];
        if (!(c[1] < m)) m = c[1];
        if (!(c[2] < m)) m = c[2];
        return m;
    }
    Float luminance() const { return c[0] * 0.212671f + c[1] * 0.715160f + c[2] * 0.072169f; }
    Spectrum clamp_positive() const { return Spectrum(clampf(c[0], 0.0f, INF), clampf(c[1], 0.0f, INF), clampf(c[2], 0.0f, INF)); }
    Spectrum sqrt() const { return Spectrum(sqrtf(c[0]), sqrtf(c[1]), sqrtf(c[2])); }
};
#define ORC_SPEC_OP(op) \
    inline Spectrum operator op(Spectrum a, Spectrum b) { return Spectrum(a.c[0] op b.c[0], a.c[1] op b.c[1], a.c[2] op b.c[2]); } \
    inline Spectrum operator op(Spectrum a, Float s) { return Spectrum(a.c[0] op s, a.c[1] op s, a.c[2] op s); } \
    inline Spectrum operator op(Float s, Spectrum a) { return Spectrum(s op a.c[0], s op a.c[1], s op a.c[2]); }
ORC_SPEC_OP(+) ORC_SPEC_OP(-) ORC_SPEC_OP(*) ORC_SPEC_OP(/)
#undef ORC_SPEC_OP
inline Spectrum& operator+=(Spectrum& a, Spectrum b) { a = a + b; return a; }
inline Spectrum& operator*=(Spectrum& a, Spectrum b) { a = a * b; return a; }
inline Spectrum& operator/=(Spectrum& a, Float s) { a = a / s; return a; }

// src/spectrum/mod.rs:28-43
inline void xyz_to_rgb(const Float xyz[3], Float rgb[3]) {
    rgb[0] = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];
    rgb[1] = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
    rgb[2] = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
}
inline void rgb_to_xyz(const Float rgb[3], Float xyz[3]) {
    xyz[0] = 0.412453f * rgb[0] + 0.357580f * rgb[1] + 0.180423f * rgb[2];
    xyz[1] = 0.212671f * rgb[0] + 0.715160f * rgb[1] + 0.072169f * rgb[2];
    xyz[2] = 0.019334f * rgb[0] + 0.119193f * rgb[1] + 0.950227f * rgb[2];
}

}  // namespace orc
