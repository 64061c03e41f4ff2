/*
 * ftn_host.cpp -- host side of libfountain_hip.so: the C ABI of include/fountain_hip.h.
 *
 *   - exact f32 restatements of the reference's host-only steps on the path: Transform algebra
 *     (src/geometry/transform.rs), Sphere::new, PerspectiveCamera::new, Film::new / sample_bounds / tiles,
 *     BVH::build (src/bvh.rs:27-158), Scene::new (src/scene/mod.rs:32-49), Distribution2D for env lights;
 *   - scene flattening into the HBM layout of ftn_device.h;
 *   - the render driver (SamplerIntegrator::render_parallel, src/integrator/mod.rs:218-227).
 * There is no CPU fallback: every compute entry point needs a HIP device and fails with FTN_ERR_NO_DEVICE without one.
 */
#include "ftn_kernels.h"
#include "ftn_wavefront.h"
#include "ftn_texture.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <thread>
#include <memory>
#include <sched.h>
#include <atomic>
#include <limits>
#include <functional>
#include <cstdlib>

using namespace ftn;

static_assert(sizeof(ftn_transform) == 128 && sizeof(ftn_pixel) == 16 && sizeof(ftn_bvh_node) == 32 && sizeof(ftn_prim) == 16, "ABI");
static_assert(sizeof(ftn_mesh) == 20 && sizeof(ftn_sphere) == 288 && sizeof(ftn_material) == 48 && sizeof(ftn_light) == 160, "ABI");
static_assert(sizeof(ftn_envmap) == 16 && sizeof(ftn_camera_desc) == 296 && sizeof(ftn_film_desc) == 32 && sizeof(ftn_sampler_desc) == 24, "ABI");
static_assert(sizeof(ftn_integrator_desc) == 16 && sizeof(ftn_tile_range) == 16 && sizeof(ftn_render_options) == 16 && sizeof(ftn_stats) == 152, "ABI");

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? FTN_ERR_OUT_OF_MEMORY : FTN_ERR_NO_DEVICE, std::string(#expr ": ") + hipGetErrorString(e_)); } while (0)

/* ================================================================== Transform algebra (cgmath 0.17 Matrix4 semantics) */
static void m4_identity(float* m) { for (int i = 0; i < 16; i++) m[i] = (i % 5 == 0) ? 1.0f : 0.0f; }
static void m4_mul(const float* l, const float* r, float* o) {   /* column j = l0*r[j][0] + l1*r[j][1] + l2*r[j][2] + l3*r[j][3] */
    float t[16];
    for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++)
        t[j * 4 + i] = ((l[i] * r[j * 4] + l[4 + i] * r[j * 4 + 1]) + l[8 + i] * r[j * 4 + 2]) + l[12 + i] * r[j * 4 + 3];
    memcpy(o, t, sizeof(t));
}
/* cgmath det_sub_proc_unsafe(m, x, y, z) */
static void m4_det_sub(const float* s, int x, int y, int z, float out[4]) {
    const float a[4] = {s[4 + x], s[12 + x], s[x], s[8 + x]}, b[4] = {s[8 + y], s[8 + y], s[4 + y], s[4 + y]}, c[4] = {s[12 + z], s[z], s[12 + z], s[z]};
    const float d[4] = {s[8 + x], s[8 + x], s[4 + x], s[4 + x]}, e[4] = {s[12 + y], s[y], s[12 + y], s[y]}, f[4] = {s[4 + z], s[12 + z], s[z], s[8 + z]};
    const float g[4] = {s[12 + x], s[x], s[12 + x], s[x]}, h[4] = {s[4 + y], s[12 + y], s[y], s[8 + y]}, i[4] = {s[8 + z], s[8 + z], s[4 + z], s[4 + z]};
    for (int k = 0; k < 4; k++) {
        float t = a[k] * (b[k] * c[k]);
        t += d[k] * (e[k] * f[k]); t += g[k] * (h[k] * i[k]);
        t -= a[k] * (e[k] * i[k]); t -= d[k] * (h[k] * c[k]); t -= g[k] * (b[k] * f[k]);
        out[k] = t;
    }
}
static float m4_det(const float* m) { float t[4]; m4_det_sub(m, 1, 2, 3, t); return ((t[0] * m[0] + t[1] * m[4]) + t[2] * m[8]) + t[3] * m[12]; }
static bool m4_invert(const float* m, float* o) {
    float t0[4], t1[4], t2[4], t3[4];
    m4_det_sub(m, 1, 2, 3, t0);
    float det = ((t0[0] * m[0] + t0[1] * m[4]) + t0[2] * m[8]) + t0[3] * m[12];
    if (det == 0.0f) return false;
    float inv_det = 1.0f / det;
    m4_det_sub(m, 0, 3, 2, t1); m4_det_sub(m, 0, 1, 3, t2); m4_det_sub(m, 0, 2, 1, t3);
    for (int k = 0; k < 4; k++) { o[k] = t0[k] * inv_det; o[4 + k] = t1[k] * inv_det; o[8 + k] = t2[k] * inv_det; o[12 + k] = t3[k] * inv_det; }
    return true;
}
static void tf_mul(const ftn_transform* a, const ftn_transform* b, ftn_transform* o) {   /* transform.rs:143-149 */
    ftn_transform r; m4_mul(a->m, b->m, r.m); m4_mul(b->inv, a->inv, r.inv); *o = r;
}

extern "C" {

int ftn_transform_identity(ftn_transform* out) { m4_identity(out->m); m4_identity(out->inv); return FTN_OK; }
int ftn_transform_translate(const float d[3], ftn_transform* out) {              /* :62-66 */
    m4_identity(out->m); m4_identity(out->inv);
    for (int i = 0; i < 3; i++) { out->m[12 + i] = d[i]; out->inv[12 + i] = -d[i]; }
    return FTN_OK;
}
int ftn_transform_scale(float sx, float sy, float sz, ftn_transform* out) {      /* :68-72 */
    m4_identity(out->m); m4_identity(out->inv);
    out->m[0] = sx; out->m[5] = sy; out->m[10] = sz;
    out->inv[0] = 1.0f / sx; out->inv[5] = 1.0f / sy; out->inv[10] = 1.0f / sz;
    return FTN_OK;
}
int ftn_transform_from_flat(const float m[16], ftn_transform* out) {             /* :23-39 */
    ftn_transform r; memcpy(r.m, m, 64);
    if (!m4_invert(r.m, r.inv)) return fail(FTN_ERR_INVALID_ARGUMENT, "Could not invert matrix");
    *out = r; return FTN_OK;
}
int ftn_transform_rotate(float angle_deg, const float axis[3], ftn_transform* out) {   /* :74-78; Matrix4::from_axis_angle */
    V3 a = normalize(V3(axis[0], axis[1], axis[2]));
    float ang = angle_deg * (float)(3.14159265358979323846 / 180.0);             /* Rad::from(Deg) */
    float s = ftn_det::sinf_det(ang), c = ftn_det::cosf_det(ang), omc = 1.0f - c;
    float f[16] = {omc * a.x * a.x + c,       omc * a.x * a.y + s * a.z, omc * a.x * a.z - s * a.y, 0.0f,
                   omc * a.x * a.y - s * a.z, omc * a.y * a.y + c,       omc * a.y * a.z + s * a.x, 0.0f,
                   omc * a.x * a.z + s * a.y, omc * a.y * a.z - s * a.x, omc * a.z * a.z + c,       0.0f,
                   0.0f, 0.0f, 0.0f, 1.0f};
    return ftn_transform_from_flat(f, out);
}
int ftn_transform_look_at(const float p[3], const float l[3], const float u[3], ftn_transform* out) {   /* :41-56 */
    V3 pos(p[0], p[1], p[2]);
    V3 dir = normalize(V3(l[0], l[1], l[2]) - pos);
    V3 right = normalize(cross(normalize(V3(u[0], u[1], u[2])), dir));
    V3 new_up = cross(dir, right);
    float mat[16] = {right.x, right.y, right.z, 0.0f, new_up.x, new_up.y, new_up.z, 0.0f, dir.x, dir.y, dir.z, 0.0f, pos.x, pos.y, pos.z, 1.0f};
    ftn_transform r; memcpy(r.inv, mat, 64);
    if (!m4_invert(mat, r.m)) return fail(FTN_ERR_INVALID_ARGUMENT, "Could not invert matrix");
    *out = r; return FTN_OK;
}
int ftn_transform_mul(const ftn_transform* a, const ftn_transform* b, ftn_transform* out) { tf_mul(a, b, out); return FTN_OK; }
int ftn_transform_inverse(const ftn_transform* a, ftn_transform* out) { ftn_transform r; memcpy(r.m, a->inv, 64); memcpy(r.inv, a->m, 64); *out = r; return FTN_OK; }
int ftn_transform_perspective(float fov, float n, float f, ftn_transform* out) {   /* :105-115 */
    float m[16] = {1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, f / (f - n), 1.0f, 0.0f, 0.0f, -f * n / (f - n), 0.0f};
    float inv_tan_ang = 1.0f / ftn_det::tanf_det((fov * (FTN_PI / 180.0f)) / 2.0f);   /* f32::to_radians */
    ftn_transform p, s;
    int rc = ftn_transform_from_flat(m, &p); if (rc) return rc;
    ftn_transform_scale(inv_tan_ang, inv_tan_ang, 1.0f, &s);
    tf_mul(&s, &p, out);
    return FTN_OK;
}
int ftn_transform_point(const ftn_transform* t, const float p[3], float o[3]) { V3 r = m4_point(t->m, V3(p[0], p[1], p[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; return FTN_OK; }
int ftn_transform_vector(const ftn_transform* t, const float p[3], float o[3]) { V3 r = m4_vector(t->m, V3(p[0], p[1], p[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; return FTN_OK; }
int ftn_transform_normal(const ftn_transform* t, const float p[3], float o[3]) { V3 r = m4_normal(t->inv, V3(p[0], p[1], p[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; return FTN_OK; }
int ftn_transform_swaps_handedness(const ftn_transform* t) { return m4_det(t->m) < 0.0f ? 1 : 0; }
int ftn_transform_points(const ftn_transform* t, size_t n, const float* in, float* out) {
    for (size_t i = 0; i < n; i++) { V3 r = m4_point(t->m, V3(in[3 * i], in[3 * i + 1], in[3 * i + 2])); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; }
    return FTN_OK;
}
int ftn_transform_normals(const ftn_transform* t, size_t n, const float* in, float* out) {
    for (size_t i = 0; i < n; i++) { V3 r = m4_normal(t->inv, V3(in[3 * i], in[3 * i + 1], in[3 * i + 2])); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; }
    return FTN_OK;
}

/* ================================================================== Sphere::new / PerspectiveCamera::new / Film */
int ftn_sphere_init(const ftn_transform* o2w, const ftn_transform* w2o, int rev, float radius, float z_min, float z_max, float phi_max_deg, ftn_sphere* out) {
    memset(out, 0, sizeof(*out));                                                 /* sphere.rs:30-50 */
    out->object_to_world = *o2w; out->world_to_object = *w2o; out->reverse_orientation = rev ? 1u : 0u;
    out->radius = radius;
    out->z_min = clampf(fmin_(z_min, z_max), -radius, radius);
    out->z_max = clampf(fmax_(z_min, z_max), -radius, radius);
    out->theta_min = ftn_det::acosf_det(clampf(z_min / radius, -1.0f, 1.0f));
    out->theta_max = ftn_det::acosf_det(clampf(z_max / radius, -1.0f, 1.0f));
    out->phi_max = clampf(phi_max_deg, 0.0f, 360.0f) * (FTN_PI / 180.0f);
    return FTN_OK;
}
int ftn_camera_perspective(const ftn_transform* c2w, const int32_t res[2], const float sw[4], const float sh[2], float lens_radius, float focal_dist,
                           float fov, ftn_camera_desc* out) {                   /* camera/mod.rs:51-69, 85-114 */
    memset(out, 0, sizeof(*out));
    ftn_transform persp, s1, s2, tr, s2r, s2r_inv, pinv, r2c;
    int rc = ftn_transform_perspective(fov, 1.0e-2f, 1000.0f, &persp); if (rc) return rc;
    ftn_transform_scale((float)res[0], (float)res[1], 1.0f, &s1);
    ftn_transform_scale(1.0f / (sw[2] - sw[0]), 1.0f / (sw[1] - sw[3]), 1.0f, &s2);
    const float d[3] = {-sw[0], -sw[3], 0.0f};
    ftn_transform_translate(d, &tr);
    ftn_transform t12; tf_mul(&s1, &s2, &t12); tf_mul(&t12, &tr, &s2r);           /* screen_to_raster */
    ftn_transform_inverse(&s2r, &s2r_inv);                                       /* raster_to_screen */
    ftn_transform_inverse(&persp, &pinv);
    tf_mul(&pinv, &s2r_inv, &r2c);                                               /* raster_to_camera */
    out->camera_to_world = *c2w; out->raster_to_camera = r2c;
    out->shutter_open = sh[0]; out->shutter_close = sh[1]; out->lens_radius = lens_radius; out->focal_dist = focal_dist;
    V3 o = m4_point(r2c.m, V3(0.0f, 0.0f, 0.0f));
    V3 dx = m4_point(r2c.m, V3(1.0f, 0.0f, 0.0f)) - o, dy = m4_point(r2c.m, V3(0.0f, 1.0f, 0.0f)) - o;
    out->dx_camera[0] = dx.x; out->dx_camera[1] = dx.y; out->dx_camera[2] = dx.z;
    out->dy_camera[0] = dy.x; out->dy_camera[1] = dy.y; out->dy_camera[2] = dy.z;
    return FTN_OK;
}
int ftn_film_init(const int32_t res[2], const float cw[4], ftn_film_desc* out) {   /* film.rs:43-58 */
    out->full_resolution[0] = res[0]; out->full_resolution[1] = res[1];
    out->crop[0] = f2i_sat(ceilf((float)res[0] * cw[0])); out->crop[1] = f2i_sat(ceilf((float)res[1] * cw[1]));
    out->crop[2] = f2i_sat(ceilf((float)res[0] * cw[2])); out->crop[3] = f2i_sat(ceilf((float)res[1] * cw[3]));
    out->filter_radius[0] = 0.5f; out->filter_radius[1] = 0.5f;
    return FTN_OK;
}
int ftn_film_sample_bounds(const ftn_film_desc* f, int32_t o[4]) {                 /* film.rs:86-93 */
    o[0] = f2i_sat(floorf((float)f->crop[0] + 0.5f - f->filter_radius[0])); o[1] = f2i_sat(floorf((float)f->crop[1] + 0.5f - f->filter_radius[1]));
    o[2] = f2i_sat(ceilf((float)f->crop[2] - 0.5f + f->filter_radius[0])); o[3] = f2i_sat(ceilf((float)f->crop[3] - 0.5f + f->filter_radius[1]));
    return FTN_OK;
}
static void list_tiles(const ftn_film_desc* f, std::vector<DTile>* tiles) {        /* bounds.rs:85-97, integrator/mod.rs:182-185 */
    int32_t sb[4]; ftn_film_sample_bounds(f, sb);
    for (int y = sb[1]; y < sb[3]; y += 16) for (int x = sb[0]; x < sb[2]; x += 16) {
        DTile t; t.valid_off = 0; t._pad = 0; t.x0 = x; t.y0 = y; t.x1 = std::min(x + 16, sb[2]); t.y1 = std::min(y + 16, sb[3]);
        t.tile_id = (unsigned long long)(long long)(t.y0 * sb[2] + t.x0);
        tiles->push_back(t);
    }
}
int ftn_film_tile_count(const ftn_film_desc* f, uint32_t* out) { std::vector<DTile> t; list_tiles(f, &t); *out = (uint32_t)t.size(); return FTN_OK; }
int ftn_film_resolve(const ftn_pixel* p, size_t n, float* rgb_out) {               /* film.rs:195-210 (host buffers; output stage) */
    for (size_t i = 0; i < n; i++) {
        float rgb[3]; xyz_to_rgb(p[i].xyz, rgb);
        if (p[i].filter_weight_sum != 0.0f) { float inv = 1.0f / p[i].filter_weight_sum; for (int c = 0; c < 3; c++) rgb[c] = fmax_(0.0f, rgb[c] * inv); }
        rgb_out[3 * i] = rgb[0]; rgb_out[3 * i + 1] = rgb[1]; rgb_out[3 * i + 2] = rgb[2];
    }
    return FTN_OK;
}

}  /* extern "C" */

/* ================================================================== BVH::build (bvh.rs:27-158), written directly in flattened DFS order */
namespace {

struct Aabb { float lo[3], hi[3]; };
static const float kFmax = 3.402823466e+38f;
static Aabb aabb_empty() { Aabb b; for (int i = 0; i < 3; i++) { b.lo[i] = kFmax; b.hi[i] = -kFmax; } return b; }
static void aabb_join_point(Aabb& b, V3 p) { b.lo[0] = fmin_(b.lo[0], p.x); b.lo[1] = fmin_(b.lo[1], p.y); b.lo[2] = fmin_(b.lo[2], p.z); b.hi[0] = fmax_(b.hi[0], p.x); b.hi[1] = fmax_(b.hi[1], p.y); b.hi[2] = fmax_(b.hi[2], p.z); }

/* ------------------------------------------------------------------ host threads for the per-element passes of ftn_scene_create
 * (this process's share of the host: bvh_default_threads; FTN_BVH_THREADS overrides; at most 32).  f(begin, end) over [0, n) in
 * contiguous blocks: every pass that uses it writes each element from that element's inputs alone, so the result does not depend on
 * the number of threads. */
static int bvh_default_threads();
static int host_threads() {
    int n = bvh_default_threads();
    if (const char* e = getenv("FTN_BVH_THREADS")) n = atoi(e);
    return std::max(1, std::min(n, 32));
}
template <class F> static void parallel_for(size_t n, F f) {
    const int nt = (int)std::min<size_t>((size_t)host_threads(), n / 65536 + 1);
    if (nt <= 1) { f((size_t)0, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back([&, t]() { f(n * (size_t)t / (size_t)nt, n * (size_t)(t + 1) / (size_t)nt); });
    for (auto& x : th) x.join();
}

struct BvhBuilder {
    const std::vector<Aabb>& bounds;
    std::vector<float> own_centroid; std::vector<uint32_t> own_order;
    std::vector<float>& centroid;                                    /* 3 per prim */
    std::vector<uint32_t>& order;                                    /* permutation being partitioned */
    std::vector<ftn_bvh_node> nodes; std::vector<uint32_t> leaf_order; uint32_t max_depth = 0;
    size_t leaf_base = 0;                                            /* global position of this builder's first leaf primitive */
    /* sub-builder over a range of a parent's arrays (parallel build) */
    BvhBuilder(const std::vector<Aabb>& b, std::vector<float>& c, std::vector<uint32_t>& o, size_t base) : bounds(b), centroid(c), order(o), leaf_base(base) {}
    explicit BvhBuilder(const std::vector<Aabb>& b) : bounds(b), centroid(own_centroid), order(own_order) {
        size_t n = b.size(); centroid.resize(3 * n); order.resize(n);
        parallel_for(n, [&](size_t i0, size_t i1) { for (size_t i = i0; i < i1; i++) {
            order[i] = (uint32_t)i;
            for (int k = 0; k < 3; k++) centroid[3 * i + k] = b[i].lo[k] + ((b[i].hi[k] - b[i].lo[k]) / 2.0f);   /* Bounds3::centroid bounds.rs:160-162 */
        } });
    }
    void build(size_t lo, size_t hi, uint32_t depth) {               /* recursive_build :66-120 + flatten_tree :133-158 */
        if (depth > max_depth) max_depth = depth;
        Aabb nb = aabb_empty(), cb = aabb_empty();
        for (size_t i = lo; i < hi; i++) {
            const uint32_t p = order[i]; const Aabb& b = bounds[p];
            for (int k = 0; k < 3; k++) { nb.lo[k] = fmin_(nb.lo[k], b.lo[k]); nb.hi[k] = fmax_(nb.hi[k], b.hi[k]); cb.lo[k] = fmin_(cb.lo[k], centroid[3 * p + k]); cb.hi[k] = fmax_(cb.hi[k], centroid[3 * p + k]); }
        }
        ftn_bvh_node node; memset(&node, 0, sizeof(node));
        for (int k = 0; k < 3; k++) { node.bmin[k] = nb.lo[k]; node.bmax[k] = nb.hi[k]; }
        const size_t n = hi - lo;
        const bool is_point = cb.lo[0] == cb.hi[0] && cb.lo[1] == cb.hi[1] && cb.lo[2] == cb.hi[2];
        if (n == 1 || is_point) {
            node.is_leaf = 1; node.idx = (uint32_t)(leaf_base + leaf_order.size()); node.n_prims = (uint16_t)n;
            for (size_t i = lo; i < hi; i++) leaf_order.push_back(order[i]);
            nodes.push_back(node);
            return;
        }
        const float dx = cb.hi[0] - cb.lo[0], dy = cb.hi[1] - cb.lo[1], dz = cb.hi[2] - cb.lo[2];
        const int ax = (dx > dy && dx > dz) ? 0 : (dy > dz ? 1 : 2);                 /* maximum_extent bounds.rs:168-177 */
        const float mid = (cb.lo[ax] + cb.hi[ax]) / 2.0f;
        uint32_t* first = order.data() + lo; uint32_t* last = order.data() + hi;
        uint32_t* split = std::partition(first, last, [&](uint32_t p) { return centroid[3 * p + ax] < mid; });
        size_t m = (size_t)(split - first);
        if (m == 0 || m == n) {                                                    /* partition_equal_counts :122-131 */
            m = n / 2;
            std::nth_element(first, first + m, last, [&](uint32_t a, uint32_t b) { return centroid[3 * a + ax] < centroid[3 * b + ax]; });
        }
        node.is_leaf = 0; node.axis = (uint8_t)ax; node.idx = 0;
        const size_t my = nodes.size();
        nodes.push_back(node);
        build(lo, lo + m, depth + 1);
        nodes[my].idx = (uint32_t)nodes.size();                                    /* second_child_idx = my + first_subtree_len + 1 */
        build(lo + m, hi, depth + 1);
    }
};

/* Parallel driver for BvhBuilder: the top of the tree is split sequentially (same partitions as the serial build); every range
 * of at most `grain` primitives becomes a task that builds its subtree into a private BvhBuilder-style buffer; the buffers are then
 * stitched in DFS order with index fix-ups.  The result is identical to the serial build, node for node. */
struct ParallelBvh {
    BvhBuilder& B; size_t grain;
    struct Task { size_t lo, hi; uint32_t depth; std::vector<ftn_bvh_node> nodes; std::vector<uint32_t> leaf_order; uint32_t max_depth = 0; };
    struct Top { ftn_bvh_node node; int child[2]; bool is_task[2]; };
    std::vector<Task> tasks; std::vector<Top> tops;
    ParallelBvh(BvhBuilder& b, size_t g) : B(b), grain(g) {}
    /* returns (index, is_task) */
    std::pair<int, bool> split(size_t lo, size_t hi, uint32_t depth) {
        if (hi - lo <= grain) { Task t; t.lo = lo; t.hi = hi; t.depth = depth; tasks.push_back(std::move(t)); return {(int)tasks.size() - 1, true}; }
        Aabb nb = aabb_empty(), cb = aabb_empty();
        for (size_t i = lo; i < hi; i++) {
            const uint32_t p = B.order[i]; const Aabb& b = B.bounds[p];
            for (int k = 0; k < 3; k++) { nb.lo[k] = fmin_(nb.lo[k], b.lo[k]); nb.hi[k] = fmax_(nb.hi[k], b.hi[k]); cb.lo[k] = fmin_(cb.lo[k], B.centroid[3 * p + k]); cb.hi[k] = fmax_(cb.hi[k], B.centroid[3 * p + k]); }
        }
        const bool is_point = cb.lo[0] == cb.hi[0] && cb.lo[1] == cb.hi[1] && cb.lo[2] == cb.hi[2];
        if (is_point) { Task t; t.lo = lo; t.hi = hi; t.depth = depth; tasks.push_back(std::move(t)); return {(int)tasks.size() - 1, true}; }   /* becomes one leaf */
        Top tp; memset(&tp.node, 0, sizeof(tp.node));
        for (int k = 0; k < 3; k++) { tp.node.bmin[k] = nb.lo[k]; tp.node.bmax[k] = nb.hi[k]; }
        const float dx = cb.hi[0] - cb.lo[0], dy = cb.hi[1] - cb.lo[1], dz = cb.hi[2] - cb.lo[2];
        const int ax = (dx > dy && dx > dz) ? 0 : (dy > dz ? 1 : 2);
        const float mid = (cb.lo[ax] + cb.hi[ax]) / 2.0f;
        uint32_t* first = B.order.data() + lo; uint32_t* last = B.order.data() + hi;
        uint32_t* sp = std::partition(first, last, [&](uint32_t p) { return B.centroid[3 * p + ax] < mid; });
        size_t m = (size_t)(sp - first), n = hi - lo;
        if (m == 0 || m == n) { m = n / 2; std::nth_element(first, first + m, last, [&](uint32_t a, uint32_t b) { return B.centroid[3 * a + ax] < B.centroid[3 * b + ax]; }); }
        tp.node.is_leaf = 0; tp.node.axis = (uint8_t)ax;
        const int me = (int)tops.size();
        tops.push_back(tp);
        auto c0 = split(lo, lo + m, depth + 1);
        auto c1 = split(lo + m, hi, depth + 1);
        tops[me].child[0] = c0.first; tops[me].is_task[0] = c0.second; tops[me].child[1] = c1.first; tops[me].is_task[1] = c1.second;
        return {me, false};
    }
    void run_tasks(int n_threads) {
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (;;) {
                size_t k = next.fetch_add(1); if (k >= tasks.size()) break;
                Task& t = tasks[k];
                BvhBuilder sub(B.bounds, B.centroid, B.order, t.lo);      /* shares bounds / centroids / the order array (disjoint ranges) */
                sub.build(t.lo, t.hi, t.depth);
                t.nodes.swap(sub.nodes); t.leaf_order.swap(sub.leaf_order); t.max_depth = sub.max_depth;
            }
        };
        std::vector<std::thread> th;
        for (int i = 0; i < n_threads; i++) th.emplace_back(worker);
        for (auto& t : th) t.join();
    }
    /* stitch: offsets by one sequential DFS over the (small) top tree, then the copies in parallel */
    std::vector<size_t> task_off;
    size_t assign(int idx, bool is_task, size_t off) {
        if (is_task) { task_off[idx] = off; return off + tasks[idx].nodes.size(); }
        ftn_bvh_node& n = tops[idx].node;
        const size_t my = off;
        off = assign(tops[idx].child[0], tops[idx].is_task[0], off + 1);
        n.idx = (uint32_t)off;                                   /* second_child_idx */
        off = assign(tops[idx].child[1], tops[idx].is_task[1], off);
        top_off[idx] = my;
        return off;
    }
    std::vector<size_t> top_off;
    void stitch(std::pair<int, bool> root, int n_threads) {
        task_off.assign(tasks.size(), 0); top_off.assign(tops.size(), 0);
        const size_t total = assign(root.first, root.second, 0);
        B.nodes.resize(total); B.leaf_order.resize(B.order.size());
        for (size_t i = 0; i < tops.size(); i++) B.nodes[top_off[i]] = tops[i].node;
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (;;) {
                size_t k = next.fetch_add(1); if (k >= tasks.size()) break;
                Task& t = tasks[k];
                const uint32_t base = (uint32_t)task_off[k];
                ftn_bvh_node* dst = B.nodes.data() + task_off[k];
                for (size_t i = 0; i < t.nodes.size(); i++) { ftn_bvh_node n = t.nodes[i]; if (!n.is_leaf) n.idx += base; dst[i] = n; }
                memcpy(B.leaf_order.data() + t.lo, t.leaf_order.data(), t.leaf_order.size() * sizeof(uint32_t));
                std::vector<ftn_bvh_node>().swap(t.nodes); std::vector<uint32_t>().swap(t.leaf_order);
            }
        };
        std::vector<std::thread> th;
        for (int i = 0; i < n_threads; i++) th.emplace_back(worker);
        for (auto& t : th) t.join();
        for (const Task& t : tasks) if (t.max_depth > B.max_depth) B.max_depth = t.max_depth;
    }
};

/* ------------------------------------------------------------------ four-box records (DScene::quad): the SAME tree, two levels per record
 * One 128-byte record (= one cache line) per interior node R reached on even collapsed levels: the node records of R's grandchildren
 * side by side, slots [A0, A1, B0, B1] with A = R's first child (bvh.rs: the next node) and B = its second child.  A child that is a
 * leaf takes its pair's first slot itself (the second one is empty).  The traversal kernels (ftn_trace4.hip) test the four boxes of a
 * record in one step and never test A's and B's own boxes:
 *   - Bounds3f::intersect_test (bounds.rs:214-233) is monotone in the box -- every operation in it is a correctly rounded, monotone f32
 *     operation -- so a ray whose plane distances are all numbers and that enters a child's box enters its parent's box, for the same
 *     or any larger t_max: skipping the parent's test changes nothing the reference would have visited.  A NaN (0 * inf: a zero direction
 *     component with the origin on a bounding plane) drops a constraint from ONE of the two tests, so for such rays the skip is not
 *     exact: the kernels hand every ray with a zero / non-finite 1/d component, or overflowing plane distances, to the reference-order
 *     kernel through the exception queue (ftn_trace4.hip, point 5; ray_is_exceptional);
 *   - the reference's visiting order (near child first by dir_is_neg[split_axis], bvh.rs:187-196) of the four grandchildren follows from
 *     the three split axes of R, A and B, which the record carries.
 * Slot layout = the 32-byte node record of ftn_device.h: {min.x, max.x, min.y, max.y} {min.z, max.z, bits(link), bits(meta)};
 * link = the traversal stack's entry for the child: byte offset of an interior child's own record, or first primitive | bit 31 for a
 * leaf (its primitives run to the one flagged GF_LEAF_END in `geom`).  Slot 0's meta carries the three one-hot split axes that order
 * the visit, one per byte: byte 0 A's (orders slots 0 / 1), byte 1 R's (orders the pairs), byte 2 B's (orders slots 2 / 3) -- tested
 * against the ray's dir_is_neg bits repeated in the same bytes.  An empty slot (bit 25 of its meta, for tools) holds the box
 * x = y = [0, 0], z = [+inf, +inf], which no ray that passes ray_is_exceptional enters: the kernels need no test for it.
 * Returns the records in DFS order (slot order) and an upper bound of the traversal stack (entries pending at any time). */
struct QuadBvh { std::vector<float> rec; /* 32 floats per record */ uint32_t n_records = 0, stack_bound = 0; bool ok = false; };
static void build_quads(const std::vector<ftn_bvh_node>& nodes, QuadBvh* out) {
    out->rec.clear(); out->n_records = 0; out->stack_bound = 0; out->ok = false;
    if (nodes.empty() || nodes[0].is_leaf) { out->ok = true; return; }          /* no interior node: the kernels test the root's leaf directly */
    /* pass 1: which interior nodes own a record, in DFS slot order (explicit stack: the tree can be 64 deep, the record tree 32) */
    std::vector<uint32_t> owner_id(nodes.size(), 0xffffffffu), owners;
    {
        std::vector<uint32_t> todo{0u};
        while (!todo.empty()) {
            const uint32_t r = todo.back(); todo.pop_back();
            owner_id[r] = (uint32_t)owners.size(); owners.push_back(r);
            uint32_t gc[4]; int ng = 0;
            const uint32_t ch[2] = {r + 1u, nodes[r].idx};
            for (int k = 0; k < 2; k++) if (!nodes[ch[k]].is_leaf) { gc[ng++] = ch[k] + 1u; gc[ng++] = nodes[ch[k]].idx; }
            for (int k = ng - 1; k >= 0; k--) if (!nodes[gc[k]].is_leaf) todo.push_back(gc[k]);      /* reversed: slot 0's subtree is numbered first */
        }
    }
    if (owners.size() >= (1u << 24)) return;                                   /* links are 31-bit byte offsets: 2^24 records of 128 bytes */
    out->n_records = (uint32_t)owners.size();
    out->rec.assign((size_t)32 * owners.size(), 0.0f);
    std::vector<uint32_t> bound(owners.size(), 0);
    const float kInf = std::numeric_limits<float>::infinity();
    auto put = [&](float* slot, const ftn_bvh_node* c, uint32_t c_index, uint32_t axis_bits) {
        uint32_t link = 0, meta = axis_bits;
        if (!c) { slot[0] = 0.0f; slot[1] = 0.0f; slot[2] = 0.0f; slot[3] = 0.0f; slot[4] = kInf; slot[5] = kInf; link = 0xffffffffu; meta |= 1u << 25; }      /* empty: fails every slab test */
        else {
            slot[0] = c->bmin[0]; slot[1] = c->bmax[0]; slot[2] = c->bmin[1]; slot[3] = c->bmax[1]; slot[4] = c->bmin[2]; slot[5] = c->bmax[2];
            if (c->is_leaf) { link = c->idx | 0x80000000u; meta |= 1u << 24; }
            else link = owner_id[c_index] * 128u;
        }
        slot[6] = ftn_det::u2f(link); slot[7] = ftn_det::u2f(meta);
    };
    /* the records: each from the nodes alone (host threads) */
    parallel_for(owners.size(), [&](size_t q0, size_t q1) {
        for (size_t q = q0; q < q1; q++) {
            const uint32_t r = owners[q];
            const uint32_t ch[2] = {r + 1u, nodes[r].idx};
            float* R = &out->rec[(size_t)32 * q];
            for (int k = 0; k < 2; k++) {
                const ftn_bvh_node& c = nodes[ch[k]];
                /* slot 0's meta: one-hot axes of A (byte 0), R (byte 1), B (byte 2); a leaf child has no axis (its pair has one member) */
                const uint32_t axes = k == 0 ? ((c.is_leaf ? 0u : (1u << c.axis)) | ((1u << nodes[r].axis) << 8) | ((nodes[ch[1]].is_leaf ? 0u : (1u << nodes[ch[1]].axis)) << 16)) : 0u;
                if (c.is_leaf) { put(R + 16 * k, &c, ch[k], axes); put(R + 16 * k + 8, nullptr, 0, 0u); }
                else { const uint32_t g0 = ch[k] + 1u, g1 = c.idx; put(R + 16 * k, &nodes[g0], g0, axes); put(R + 16 * k + 8, &nodes[g1], g1, 0u); }
            }
        }
    });
    /* the stack bound, bottom up */
    for (size_t q = owners.size(); q-- > 0;) {                                  /* children have larger ids: their bounds are known */
        const uint32_t r = owners[q];
        const uint32_t ch[2] = {r + 1u, nodes[r].idx};
        uint32_t valid = 0, deepest = 0;
        for (int k = 0; k < 2; k++) {
            const ftn_bvh_node& c = nodes[ch[k]];
            if (c.is_leaf) valid += 1;
            else {
                const uint32_t g0 = ch[k] + 1u, g1 = c.idx; valid += 2;
                if (!nodes[g0].is_leaf) deepest = std::max(deepest, bound[owner_id[g0]]);
                if (!nodes[g1].is_leaf) deepest = std::max(deepest, bound[owner_id[g1]]);
            }
        }
        bound[q] = (valid - 1u) + deepest;
    }
    out->stack_bound = bound[0];
    out->ok = true;
}

/* ------------------------------------------------------------------ eight-box occlusion records (DScene::oct): a SECOND tree over the same leaves, for
 * Scene::intersect_test only (k_wf_trace8_any, ftn_trace8.hip).
 * intersect_test (bvh.rs:217-266) never shrinks t_max and returns a boolean: a ray is occluded iff SOME leaf's own box passes the slab test
 * and one of its primitives passes its test (the boxes of the leaf's ancestors pass whenever the leaf's does -- the monotonicity argument of
 * build_quads).  Which interior boxes are tested on the way is therefore free, as long as every leaf whose box the ray enters is reached:
 * interior boxes may be LARGER than the reference's.  A record holds up to eight children of a collapsed subtree (an interior node's
 * children, the largest one replaced by its own children until eight are there) with their boxes quantised to 8 bits per plane on a
 * per-record grid (origin = the subtree's min corner, a power-of-two step per axis), rounded outwards: 96 bytes decide up to three
 * levels -- a walk fetches ~40 % fewer records than four-box records and half the bytes.  Leaves are tested exactly in the kernel: a
 * single triangle's leaf box is the min / max of its vertices (checked here against the reference's node), any other leaf carries its
 * exact box in `xbox`.
 * Record (32 words): [0..2] origin, [3] step exponents (biased, one byte per axis), [4..11] child links -- byte offset of the child's
 * record | 0, leaf: bit 31 | first primitive, leaf with an explicit box: bit 31 | bit 30 | index into xbox, empty: 0xffffffff --,
 * [12..23] planes, one byte per child: lo.x[8] hi.x[8] lo.y[8] hi.y[8] lo.z[8] hi.z[8]; [24..31] unused (one 128-byte line per record). */
struct OctBvh { std::vector<uint32_t> rec; std::vector<float4> xbox; uint32_t n_records = 0, stack_bound = 0; bool ok = false; };
static void build_octs(const std::vector<ftn_bvh_node>& nodes, const std::vector<float4>& geom, OctBvh* out) {
    out->rec.clear(); out->xbox.clear(); out->n_records = 0; out->stack_bound = 0; out->ok = false;
    if (nodes.empty() || nodes[0].is_leaf) return;                                /* (a single leaf: the four-box path handles it) */
    auto area = [&](uint32_t i) { const ftn_bvh_node& n = nodes[i]; const double dx = (double)n.bmax[0] - n.bmin[0], dy = (double)n.bmax[1] - n.bmin[1], dz = (double)n.bmax[2] - n.bmin[2]; return 2.0 * (dx * dy + dy * dz + dz * dx); };
    struct Rec { uint32_t root; uint32_t child[8]; int n; };
    std::vector<Rec> recs; std::vector<uint32_t> rec_of(nodes.size(), 0xffffffffu);
    {   /* pass 1: collapse, records numbered in DFS order */
        std::vector<uint32_t> todo{0u};
        while (!todo.empty()) {
            const uint32_t r = todo.back(); todo.pop_back();
            Rec R; R.root = r; R.n = 2; R.child[0] = r + 1u; R.child[1] = nodes[r].idx;
            for (;;) {
                if (R.n == 8) break;
                int best = -1; double ba = -1.0;
                for (int k = 0; k < R.n; k++) if (!nodes[R.child[k]].is_leaf) { const double a = area(R.child[k]); if (a > ba) { ba = a; best = k; } }
                if (best < 0) break;
                const uint32_t c = R.child[best];
                R.child[best] = c + 1u; R.child[R.n++] = nodes[c].idx;
            }
            rec_of[r] = (uint32_t)recs.size(); recs.push_back(R);
            for (int k = R.n - 1; k >= 0; k--) if (!nodes[R.child[k]].is_leaf) todo.push_back(R.child[k]);
        }
    }
    if (recs.size() >= (1u << 24)) return;                                       /* links are byte offsets below 2^31 */
    out->rec.assign((size_t)32 * recs.size(), 0u);
    std::vector<uint32_t> bound(recs.size(), 0);
    /* which leaf children carry an explicit box (bit k of xmask[q]): every leaf that is not a single triangle whose node box is the min / max
     * of its vertices.  Their xbox slots are numbered in the order records are finished (last record first, children in slot order) */
    auto leaf_is_implicit = [&](const ftn_bvh_node& c) {
        if (c.n_prims != 1 || geom.empty()) return false;
        const float4 g0 = geom[(size_t)FTN_GS * c.idx], g1 = geom[(size_t)FTN_GS * c.idx + 1], g2 = geom[(size_t)FTN_GS * c.idx + 2];
        if (ftn_det::f2u(g0.w) & GF_KIND_SPHERE) return false;
        const float lo[3] = {std::min(std::min(g0.x, g1.x), g2.x), std::min(std::min(g0.y, g1.y), g2.y), std::min(std::min(g0.z, g1.z), g2.z)};
        const float hi[3] = {std::max(std::max(g0.x, g1.x), g2.x), std::max(std::max(g0.y, g1.y), g2.y), std::max(std::max(g0.z, g1.z), g2.z)};
        for (int a = 0; a < 3; a++) if (ftn_det::f2u(lo[a]) != ftn_det::f2u(c.bmin[a]) || ftn_det::f2u(hi[a]) != ftn_det::f2u(c.bmax[a])) return false;
        return true;
    };
    std::vector<uint8_t> xmask(recs.size(), 0);
    parallel_for(recs.size(), [&](size_t q0, size_t q1) {
        for (size_t q = q0; q < q1; q++) {
            const Rec& R = recs[q]; uint8_t m = 0;
            for (int k = 0; k < R.n; k++) { const ftn_bvh_node& c = nodes[R.child[k]]; if (c.is_leaf && !leaf_is_implicit(c)) m |= (uint8_t)(1u << k); }
            xmask[q] = m;
        }
    });
    std::vector<uint32_t> xbase(recs.size(), 0);
    { uint32_t run = 0; for (size_t q = recs.size(); q-- > 0;) { xbase[q] = run; run += (uint32_t)__builtin_popcount(xmask[q]); } out->xbox.assign(2 * (size_t)run, make_float4(0.0f, 0.0f, 0.0f, 0.0f)); }
    /* the records: each from the nodes alone (host threads) */
    parallel_for(recs.size(), [&](size_t q0, size_t q1) {
    for (size_t q = q0; q < q1; q++) {
        const Rec& R = recs[q];
        const ftn_bvh_node& root = nodes[R.root];
        uint32_t* W = &out->rec[(size_t)32 * q];
        uint32_t ebits[3]; float step[3];
        for (int a = 0; a < 3; a++) {
            /* the smallest power-of-two step that covers the subtree's extent with 255 steps -- one more if the outward rounding of a child
             * plane would not fit */
            const double ext = (double)root.bmax[a] - (double)root.bmin[a];
            int e = ext > 0.0 ? (int)ceil(log2(ext / 255.0)) : -126;
            if (e < -126) e = -126;
            for (;; e++) {
                bool fits = true;
                const float st = ldexpf(1.0f, e);
                for (int k = 0; k < R.n && fits; k++) {
                    const ftn_bvh_node& c = nodes[R.child[k]];
                    const double qh = ceil(((double)c.bmax[a] - (double)root.bmin[a]) / (double)st);
                    if (qh > 254.0) fits = false;                                 /* (one step of slack for the adjustment below) */
                }
                if (fits || e >= 126) break;
            }
            step[a] = ldexpf(1.0f, e); ebits[a] = (uint32_t)(e + 127);
            memcpy(&W[a], &root.bmin[a], 4);
        }
        W[3] = ebits[0] | (ebits[1] << 8) | (ebits[2] << 16);
        uint32_t xi = xbase[q];
        for (int k = 0; k < 8; k++) {
            uint32_t link = 0xffffffffu; uint32_t qb[6] = {255u, 0u, 255u, 0u, 255u, 0u};
            if (k < R.n) {
                const uint32_t ci = R.child[k]; const ftn_bvh_node& c = nodes[ci];
                for (int a = 0; a < 3; a++) {
                    /* decoded plane = origin + q * step, exactly as the kernel's bounds (real-number inequality: lo_dec <= lo, hi_dec >= hi) */
                    const double o = (double)root.bmin[a], stp = (double)step[a];
                    double ql = floor(((double)c.bmin[a] - o) / stp), qh = ceil(((double)c.bmax[a] - o) / stp);
                    while (ql > 0.0 && o + ql * stp > (double)c.bmin[a]) ql -= 1.0;
                    while (o + qh * stp < (double)c.bmax[a]) qh += 1.0;
                    if (ql < 0.0) ql = 0.0;
                    if (qh > 255.0) qh = 255.0;                                   /* (cannot happen: the step was chosen with slack) */
                    qb[2 * a] = (uint32_t)ql; qb[2 * a + 1] = (uint32_t)qh;
                }
                if (!c.is_leaf) link = rec_of[ci] * 128u;
                else if (!((xmask[q] >> k) & 1u)) link = 0x80000000u | c.idx;         /* a single triangle whose node box is the min / max of its vertices */
                else {
                    link = 0xc0000000u | xi;
                    out->xbox[2 * (size_t)xi] = make_float4(c.bmin[0], c.bmin[1], c.bmin[2], ftn_det::u2f(c.idx));
                    out->xbox[2 * (size_t)xi + 1] = make_float4(c.bmax[0], c.bmax[1], c.bmax[2], 0.0f);
                    xi++;
                }
            }
            W[4 + k] = link;
            for (int j = 0; j < 6; j++) W[12 + 2 * j + (k >> 2)] |= qb[j] << (8 * (k & 3));
        }
    }
    });
    /* the stack bound, bottom up */
    for (size_t q = recs.size(); q-- > 0;) {
        const Rec& R = recs[q]; uint32_t deepest = 0;
        for (int k = 0; k < R.n; k++) if (!nodes[R.child[k]].is_leaf) deepest = std::max(deepest, bound[rec_of[R.child[k]]]);
        bound[q] = (uint32_t)(R.n - 1) + deepest;
    }
    if (out->xbox.size() / 2 >= (1u << 30) || nodes.size() >= (1u << 30)) return;
    out->n_records = (uint32_t)recs.size(); out->stack_bound = bound[0]; out->ok = true;
}

static int bvh_default_threads() {
    int cores = (int)std::thread::hardware_concurrency();
#ifdef __linux__
    cpu_set_t set; CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) cores = c; }
#endif
    int ranks = 1;
    if (const char* e = getenv("LOCAL_WORLD_SIZE")) { const int r = atoi(e); if (r > 1) ranks = r; }
    return std::max(1, cores / ranks);
}

struct HostScene {
    std::vector<ftn_bvh_node> nodes; std::vector<uint32_t> order; uint32_t max_depth = 0; Aabb world;
    std::vector<int32_t> light_kind, light_prim;
};

static Aabb prim_bounds(const ftn_scene_desc* d, const ftn_prim& p) {
    Aabb b = aabb_empty();
    if (p.shape_kind == FTN_SHAPE_TRIANGLE) {                                      /* Triangle::world_bound triangle.rs:152-158 */
        for (int k = 0; k < 3; k++) { uint32_t v = d->tri_indices[3 * (size_t)p.shape_index + k]; aabb_join_point(b, V3(d->P[3 * v], d->P[3 * v + 1], d->P[3 * v + 2])); }
    } else {                                                                       /* Shape::world_bound default + Bounds3f::transform, shapes/mod.rs:13-15 */
        const ftn_sphere& s = d->spheres[p.shape_index];
        const float lo[3] = {-s.radius, -s.radius, s.z_min}, hi[3] = {s.radius, s.radius, s.z_max};
        for (int c = 0; c < 8; c++) {                                              /* iter_corners order bounds.rs:183-194 */
            V3 q((c & 4) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 1) ? hi[2] : lo[2]);
            aabb_join_point(b, m4_point(s.object_to_world.m, q));
        }
    }
    return b;
}
static int validate_desc(const ftn_scene_desc* d) {
    if (!d) return fail(FTN_ERR_INVALID_ARGUMENT, "null scene description");
    if ((d->n_prims && !d->prims) || (d->n_triangles && (!d->tri_indices || !d->tri_mesh)) || (d->n_vertices && !d->P) || (d->n_meshes && !d->meshes) ||
        (d->n_spheres && !d->spheres) || (d->n_materials && !d->materials) || (d->n_area_emit && !d->area_emit) || (d->n_lights && !d->lights) || (d->n_envmaps && !d->envmaps))
        return fail(FTN_ERR_INVALID_ARGUMENT, "a count is non-zero but its array is NULL");
    for (uint32_t i = 0; i < d->n_materials; i++) if (d->materials[i].type > FTN_MAT_GLASS) return fail(FTN_ERR_INVALID_ARGUMENT, "unknown material type");
    for (uint32_t i = 0; i < d->n_meshes; i++) {
        if (d->meshes[i].has_normals && !d->N) return fail(FTN_ERR_INVALID_ARGUMENT, "a mesh has normals but N is NULL");
        if (d->meshes[i].has_uvs && !d->UV) return fail(FTN_ERR_INVALID_ARGUMENT, "a mesh has uvs but UV is NULL");
        if (d->meshes[i].has_tangents && !d->S) return fail(FTN_ERR_INVALID_ARGUMENT, "a mesh has tangents but S is NULL");
    }
    for (uint32_t i = 0; i < d->n_envmaps; i++) if (!d->envmaps[i].texels) return fail(FTN_ERR_INVALID_ARGUMENT, "environment map without texels");
    if (d->n_textures && d->textures) {                        /* textures */
        const int nt = (int)d->n_textures, ni = d->images ? (int)d->n_images : 0;
        for (int i = 0; i < nt; i++) {
            const ftn_texture& t = d->textures[i];
            if (t.kind > FTN_TEX_IMAGE) return fail(FTN_ERR_INVALID_ARGUMENT, "unknown texture kind");
            if (t.kind == FTN_TEX_CHECKERBOARD && (t.tex1 < 0 || t.tex1 >= nt || t.tex2 < 0 || t.tex2 >= nt)) return fail(FTN_ERR_INVALID_ARGUMENT, "checkerboard child texture out of range");
            if (t.kind == FTN_TEX_IMAGE && (t.image < 0 || t.image >= ni)) return fail(FTN_ERR_INVALID_ARGUMENT, "image index out of range");
        }
        for (int i = 0; i < ni; i++) { const ftn_image& im = d->images[i]; if (im.width == 0 || im.height == 0 || !im.texels || im.wrap > FTN_WRAP_CLAMP) return fail(FTN_ERR_INVALID_ARGUMENT, "bad image"); }
        if (d->material_textures) for (uint32_t i = 0; i < d->n_materials; i++) { const ftn_material_textures& mt = d->material_textures[i];
            for (int32_t id : {mt.a, mt.b, mt.s0, mt.s1, mt.s2}) if (id >= nt) return fail(FTN_ERR_INVALID_ARGUMENT, "material texture index out of range"); }
    }
    for (uint32_t i = 0; i < d->n_triangles; i++) {
        if (d->tri_mesh[i] >= d->n_meshes) return fail(FTN_ERR_INVALID_ARGUMENT, "triangle mesh id out of range");
        for (int k = 0; k < 3; k++) if (d->tri_indices[3 * (size_t)i + k] >= d->n_vertices) return fail(FTN_ERR_INVALID_ARGUMENT, "vertex index out of range");
    }
    for (uint32_t i = 0; i < d->n_prims; i++) {
        const ftn_prim& p = d->prims[i];
        if (p.shape_kind == FTN_SHAPE_TRIANGLE) { if (p.shape_index >= d->n_triangles) return fail(FTN_ERR_INVALID_ARGUMENT, "triangle index out of range"); }
        else if (p.shape_kind == FTN_SHAPE_SPHERE) { if (p.shape_index >= d->n_spheres) return fail(FTN_ERR_INVALID_ARGUMENT, "sphere index out of range"); }
        else return fail(FTN_ERR_INVALID_ARGUMENT, "unknown shape kind");
        if (p.material >= (int)d->n_materials || p.area_emit >= (int)d->n_area_emit) return fail(FTN_ERR_INVALID_ARGUMENT, "material / area light index out of range");
    }
    for (uint32_t i = 0; i < d->n_envmaps; i++) {
        uint32_t w = d->envmaps[i].width, h = d->envmaps[i].height;
        /* any size: compute_distribution's filter width 1 / max(w, h) puts the lookup at pyramid level floor(log2 max) - log2 max, which is
         * 0 for a power of two (lerp with weight exactly 0 on level 1) and negative otherwise (level 0 alone, mipmap.rs:247-249) -- level 1
         * never contributes, so no pyramid is needed for environment maps (infinite.rs:63-77) */
        if (w == 0 || h == 0) return fail(FTN_ERR_INVALID_ARGUMENT, "environment map without texels");
    }
    for (uint32_t i = 0; i < d->n_lights; i++) {
        if (d->lights[i].type > FTN_LIGHT_INFINITE) return fail(FTN_ERR_INVALID_ARGUMENT, "unknown light type");
        if (d->lights[i].type == FTN_LIGHT_INFINITE && (d->lights[i].envmap < 0 || d->lights[i].envmap >= (int)d->n_envmaps)) return fail(FTN_ERR_INVALID_ARGUMENT, "envmap index out of range");
    }
    return FTN_OK;
}
static int build_host_scene(const ftn_scene_desc* d, HostScene* hs) {
    int rc = validate_desc(d); if (rc) return rc;
    std::vector<Aabb> pb(d->n_prims);
    parallel_for(d->n_prims, [&](size_t i0, size_t i1) { for (size_t i = i0; i < i1; i++) pb[i] = prim_bounds(d, d->prims[i]); });
    hs->world = aabb_empty();
    if (d->n_prims) {
        BvhBuilder b(pb);
        /* build threads: this process's share of the host -- the cores it may run on (affinity mask), divided by the ranks a launcher
         * started on this node (LOCAL_WORLD_SIZE: an 8-rank job builds eight scenes side by side), at most 32; FTN_BVH_THREADS overrides */
        int n_threads = bvh_default_threads();
        if (const char* e = getenv("FTN_BVH_THREADS")) n_threads = atoi(e);
        if (n_threads > 32) n_threads = 32;
        if (n_threads > 1 && d->n_prims >= (1u << 16)) {
            const bool dbg = getenv("FTN_BVH_DEBUG") != nullptr;
            auto t0 = std::chrono::steady_clock::now();
            ParallelBvh pb2(b, std::max<size_t>(4096, d->n_prims / (size_t)(n_threads * 8)));
            auto root = pb2.split(0, d->n_prims, 0);
            auto t1 = std::chrono::steady_clock::now();
            pb2.run_tasks(n_threads);
            auto t2 = std::chrono::steady_clock::now();
            pb2.stitch(root, n_threads);
            auto t3 = std::chrono::steady_clock::now();
            if (dbg) fprintf(stderr, "[ftn bvh] %d threads, %zu tasks: top split %.2fs, subtrees %.2fs, stitch %.2fs\n", n_threads, pb2.tasks.size(),
                             std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count(), std::chrono::duration<double>(t3 - t2).count());
            /* depth of the top part */
            std::function<void(int, bool, uint32_t)> walk = [&](int i, bool t, uint32_t dep) { if (t) return; if (dep > b.max_depth) b.max_depth = dep; walk(pb2.tops[i].child[0], pb2.tops[i].is_task[0], dep + 1); walk(pb2.tops[i].child[1], pb2.tops[i].is_task[1], dep + 1); };
            walk(root.first, root.second, 0);
        } else b.build(0, d->n_prims, 0);
        hs->nodes.swap(b.nodes); hs->order.swap(b.leaf_order); hs->max_depth = b.max_depth;
        for (int k = 0; k < 3; k++) { hs->world.lo[k] = hs->nodes[0].bmin[k]; hs->world.hi[k] = hs->nodes[0].bmax[k]; }
    }
    /* Scene::new (scene/mod.rs:32-49): explicit lights first, then one area light per emissive primitive in BVH order */
    for (uint32_t i = 0; i < d->n_lights; i++) { hs->light_kind.push_back((int)d->lights[i].type); hs->light_prim.push_back(-1); }
    for (size_t i = 0; i < hs->order.size(); i++) if (d->prims[hs->order[i]].area_emit >= 0) { hs->light_kind.push_back((int)LK_AREA); hs->light_prim.push_back((int)i); }
    return FTN_OK;
}

/* Distribution1D::new (sampling.rs:84-107) into flat arrays */
static float dist1d_build(const float* f, size_t n, float* cdf) {
    cdf[0] = 0.0f;
    for (size_t i = 1; i < n + 1; i++) cdf[i] = cdf[i - 1] + (f[i - 1] / (float)n);
    float integral = cdf[n];
    if (integral == 0.0f) { for (size_t i = 1; i < n + 1; i++) cdf[i] = (float)i / (float)n; }
    else { for (size_t i = 1; i < n + 1; i++) cdf[i] /= integral; }
    return integral;
}

template <class T> struct DevBuf {
    T* p = nullptr; size_t n = 0;
    int upload(const T* src, size_t count) {
        n = count; if (!count) return FTN_OK;
        HIP_TRY(hipMalloc((void**)&p, count * sizeof(T)));
        HIP_TRY(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
        return FTN_OK;
    }
    int alloc_zero(size_t count) {
        n = count; if (!count) return FTN_OK;
        HIP_TRY(hipMalloc((void**)&p, count * sizeof(T)));
        HIP_TRY(hipMemset(p, 0, count * sizeof(T)));
        return FTN_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

}  // namespace

/* ------------------------------------------------------------------ MIPMap::<Spectrum>::new (mipmap.rs:78-145): level 0 = the image, every further
 * level = the previous one shrunk to max(1, w/2) x max(1, h/2) by the `resize` crate's Triangle filter (resize 0.4.3, un-vendored:
 * its published algorithm restated -- DESIGN.md "parity unpinned" note).  Separable: per output sample a coefficient line
 * (support 1.0 scaled by the shrink ratio, taps clamped to the image, normalised); rows first (H1 -> H2, into a transposed f32
 * buffer), then columns (W1 -> W2); plain f32 multiply-adds in tap order. */
namespace {
struct MipLevelHost { uint32_t w, h; std::vector<float> rgb; };
struct TapLine { size_t first; std::vector<float> w; };
static std::vector<TapLine> triangle_taps(size_t n_src, size_t n_dst) {
    const float ratio = (float)n_src / (float)n_dst;
    const float scale = ratio > 1.0f ? ratio : 1.0f;
    const float radius = ceilf(scale);                                        /* support (1.0) * scale */
    std::vector<TapLine> lines(n_dst);
    for (size_t j = 0; j < n_dst; j++) {
        const float centre = ((float)j + 0.5f) * ratio - 0.5f;
        long lo = (long)ceilf(centre - radius), hi = (long)floorf(centre + radius);
        lo = std::min<long>(std::max<long>(lo, 0), (long)n_src - 1); hi = std::min<long>(std::max<long>(hi, 0), (long)n_src - 1);
        float total = 0.0f;
        for (long k = lo; k <= hi; k++) total += fmaxf(1.0f - fabsf(((float)k - centre) / scale), 0.0f);
        lines[j].first = (size_t)lo;
        for (long k = lo; k <= hi; k++) lines[j].w.push_back(fmaxf(1.0f - fabsf(((float)k - centre) / scale), 0.0f) / total);
    }
    return lines;
}
static void shrink_triangle(uint32_t w1, uint32_t h1, uint32_t w2, uint32_t h2, const std::vector<float>& src, std::vector<float>* dst) {
    const std::vector<TapLine> across = triangle_taps(w1, w2), down = triangle_taps(h1, h2);
    std::vector<float> mid((size_t)w1 * h2 * 3);                              /* [x1][y2] */
    for (size_t x = 0; x < w1; x++)
        for (size_t y = 0; y < h2; y++) {
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f; const TapLine& L = down[y];
            for (size_t k = 0; k < L.w.size(); k++) { const float* q = &src[((L.first + k) * w1 + x) * 3]; a0 += q[0] * L.w[k]; a1 += q[1] * L.w[k]; a2 += q[2] * L.w[k]; }
            float* o = &mid[(x * h2 + y) * 3]; o[0] = a0; o[1] = a1; o[2] = a2;
        }
    dst->assign((size_t)w2 * h2 * 3, 0.0f);
    for (size_t y = 0; y < h2; y++)
        for (size_t x = 0; x < w2; x++) {
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f; const TapLine& L = across[x];
            for (size_t k = 0; k < L.w.size(); k++) { const float* q = &mid[((L.first + k) * h2 + y) * 3]; a0 += q[0] * L.w[k]; a1 += q[1] * L.w[k]; a2 += q[2] * L.w[k]; }
            float* o = &(*dst)[(y * w2 + x) * 3]; o[0] = a0; o[1] = a1; o[2] = a2;
        }
}
static void build_mip_pyramid(uint32_t w, uint32_t h, const float* rgb, std::vector<MipLevelHost>* out) {
    out->clear();
    MipLevelHost l0; l0.w = w; l0.h = h; l0.rgb.assign(rgb, rgb + (size_t)w * h * 3); out->push_back(l0);
    uint32_t big = std::max(w, h); int n_levels = 1; while (big >>= 1) n_levels++;          /* 1 + log2_usize(max(w, h)) */
    for (int l = 1; l < n_levels; l++) {
        const MipLevelHost& prev = out->back();
        MipLevelHost nx; nx.w = std::max<uint32_t>(1, prev.w / 2); nx.h = std::max<uint32_t>(1, prev.h / 2);
        shrink_triangle(prev.w, prev.h, nx.w, nx.h, prev.rgb, &nx.rgb);
        out->push_back(std::move(nx));
    }
}
}  // namespace

struct ftn_scene {
    int device = 0;
    HostScene host;
    DScene d; uint32_t stack_entries = 1;
    DevBuf<float4> nodes, geom, fat, srec, quad, oct_xbox; DevBuf<uint4> prim_info, oct; DevBuf<float> N, UV, T; DevBuf<DSphere> spheres; DevBuf<ftn_material> materials; DevBuf<DLight> lights;
    DevBuf<uint32_t> inf_lights; DevBuf<unsigned char> prim_class; std::vector<DevBuf<float>> misc; std::vector<DevBuf<float4>> misc4;
    DevBuf<ftn_texture> textures; DevBuf<ftn_material_textures> mtex; DevBuf<DImage> images; DevBuf<float4> texels;
    /* render work buffers (grow-only, reused across calls) */
    DevBuf<float4> accA, accB, accC; DevBuf<DTile> tiles; DevBuf<DevStats> stats; size_t acc_pixels = 0;
    bool spill_acc_dirty = true;       /* accB / accC may hold something other than zeros */
    WavefrontState* wf = nullptr;
    std::vector<DTile> sel; int32_t tile_key[10] = {0};
    ~ftn_scene() {
        nodes.release(); geom.release(); fat.release(); srec.release(); quad.release(); oct.release(); oct_xbox.release(); prim_info.release(); N.release(); UV.release(); T.release(); spheres.release(); materials.release(); lights.release(); inf_lights.release(); prim_class.release();
        for (auto& b : misc) b.release();
        for (auto& b : misc4) b.release();
        textures.release(); mtex.release(); images.release(); texels.release();
        accA.release(); accB.release(); accC.release(); tiles.release(); stats.release();
        wavefront_destroy(wf);
    }
};

/* FTN_SCENE_DEBUG: wall time of the phases of ftn_scene_create on stderr */
struct PhaseClock {
    bool on; std::chrono::steady_clock::time_point t;
    PhaseClock() : on(getenv("FTN_SCENE_DEBUG") != nullptr), t(std::chrono::steady_clock::now()) {}
    void mark(const char* what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[ftn scene] %-44s %.3f s\n", what, std::chrono::duration<double>(n - t).count()); t = n;
    }
};
static int upload_scene(const ftn_scene_desc* d, ftn_scene* sc) {
    const HostScene& hs = sc->host;
    PhaseClock clk;
    const size_t np = hs.order.size();
    std::vector<float4> nodes(2 * hs.nodes.size());
    parallel_for(hs.nodes.size(), [&](size_t i0, size_t i1) { for (size_t i = i0; i < i1; i++) {
        const ftn_bvh_node& n = hs.nodes[i];
        /* slab pairs first: the x and y (min, max) pairs, then the z pair with the two index words (see ftn_device.h) */
        nodes[2 * i] = make_float4(n.bmin[0], n.bmax[0], n.bmin[1], n.bmax[1]);
        /* interior: byte offset of the second child (the first child is the next record: +32); leaf: first primitive.
         * meta: n_prims | one-hot split axis << 16 (matches the ray's dir_is_neg bits) | leaf << 24 */
        const uint32_t link = n.is_leaf ? n.idx : n.idx * 32u;
        nodes[2 * i + 1] = make_float4(n.bmin[2], n.bmax[2], ftn_det::u2f(link), ftn_det::u2f((uint32_t)n.n_prims | ((1u << n.axis) << 16) | ((uint32_t)n.is_leaf << 24)));
    } });
    /* two-box records (see DScene::fat): one per interior node, numbered in DFS order */
    if (hs.nodes.size() >= (1u << 27)) return fail(FTN_ERR_UNSUPPORTED, "more than 2^27 BVH nodes (node links are 32-bit byte offsets)");
    /* only the legacy any-hit kernel (k_wf_trace_any2: FTN_TRACE4=0, or a scene without four-box records) reads them: 639 MB at 10 M
     * triangles that a default scene does not allocate.  A scene created without them still renders under FTN_TRACE4=0 (plain node walk). */
    auto env_is = [](const char* name, int value) { const char* v = getenv(name); return v && atoi(v) == value; };
    const bool want_fat = env_is("FTN_TRACE4", 0) || env_is("FTN_QUAD", 0) || env_is("FTN_FAT", 1);
    std::vector<uint32_t> fat_id(want_fat ? hs.nodes.size() : 0, 0xffffffffu);
    uint32_t n_fat = 0;
    if (want_fat) for (size_t i = 0; i < hs.nodes.size(); i++) if (!hs.nodes[i].is_leaf) fat_id[i] = n_fat++;
    std::vector<float4> fat(4 * (size_t)n_fat);
    std::vector<uint8_t> leaf_end(np, 0);
    for (size_t i = 0; i < hs.nodes.size(); i++) {
        const ftn_bvh_node& n = hs.nodes[i];
        if (n.is_leaf) { if (n.n_prims) leaf_end[n.idx + n.n_prims - 1] = 1; continue; }
        if (!want_fat) continue;
        const size_t ch[2] = {i + 1, (size_t)n.idx};
        for (int k = 0; k < 2; k++) {                            /* the two children's node records side by side (same layout as `nodes`) */
            const ftn_bvh_node& c = hs.nodes[ch[k]];
            const uint32_t link = c.is_leaf ? c.idx : fat_id[ch[k]] * 64u;            /* byte offset of the child's own two-box record, or first primitive */
            const uint32_t meta = (uint32_t)c.n_prims | (k == 0 ? ((1u << n.axis) << 16) : 0u) | ((uint32_t)c.is_leaf << 24);   /* record 0 carries THIS node's split axis */
            fat[4 * (size_t)fat_id[i] + 2 * k] = make_float4(c.bmin[0], c.bmax[0], c.bmin[1], c.bmax[1]);
            fat[4 * (size_t)fat_id[i] + 2 * k + 1] = make_float4(c.bmin[2], c.bmax[2], ftn_det::u2f(link), ftn_det::u2f(meta));
        }
    }
    std::vector<int> prim_light(np, -1);
    for (size_t l = 0; l < hs.light_prim.size(); l++) if (hs.light_prim[l] >= 0) prim_light[hs.light_prim[l]] = (int)l;
    std::vector<float4> geom((size_t)FTN_GS * np); std::vector<uint4> info(2 * np);
    parallel_for(np, [&](size_t i0, size_t i1) { for (size_t i = i0; i < i1; i++) {
        const ftn_prim& p = d->prims[hs.order[i]];
        uint32_t fl = 0;
        if (p.shape_kind == FTN_SHAPE_SPHERE) {
            fl = GF_KIND_SPHERE | (leaf_end[i] ? GF_LEAF_END : 0u);
            geom[FTN_GS * i] = make_float4(0, 0, 0, ftn_det::u2f(fl)); geom[FTN_GS * i + 1] = make_float4(0, 0, 0, ftn_det::u2f(p.shape_index)); geom[FTN_GS * i + 2] = make_float4(0, 0, 0, 0);
            info[2 * i + 1] = make_uint4(0, 0, 0, p.shape_index);
        } else {
            const ftn_mesh& m = d->meshes[d->tri_mesh[p.shape_index]];
            if (m.has_normals && d->N) fl |= GF_HAS_NORMALS;
            if (m.has_uvs && d->UV) fl |= GF_HAS_UVS;
            if (m.has_tangents && d->S) fl |= GF_HAS_TANGENTS;
            if (m.flip_normals) fl |= GF_FLIP;
            if (leaf_end[i]) fl |= GF_LEAF_END;
            const uint32_t* vi = d->tri_indices + 3 * (size_t)p.shape_index;
            const float* P = d->P;
            geom[FTN_GS * i] = make_float4(P[3 * vi[0]], P[3 * vi[0] + 1], P[3 * vi[0] + 2], ftn_det::u2f(fl));
            geom[FTN_GS * i + 1] = make_float4(P[3 * vi[1]], P[3 * vi[1] + 1], P[3 * vi[1] + 2], ftn_det::u2f(p.shape_index));
            geom[FTN_GS * i + 2] = make_float4(P[3 * vi[2]], P[3 * vi[2] + 1], P[3 * vi[2] + 2], 0.0f);
            info[2 * i + 1] = make_uint4(vi[0], vi[1], vi[2], p.shape_index);
        }
        info[2 * i] = make_uint4((uint32_t)p.material, (uint32_t)prim_light[i], fl, 0);
    } });
    clk.mark("node records, leaf-test records (host)");
    int rc;
    if ((rc = sc->nodes.upload(nodes.data(), nodes.size()))) return rc;
    clk.mark("upload nodes");
    /* two-box record links are 31-bit byte offsets: beyond 2^25 interior nodes the any-hit kernel falls back to the plain node walk */
    if (n_fat && n_fat < (1u << 25)) { if ((rc = sc->fat.upload(fat.data(), fat.size()))) return rc; }
    /* four-box records (DScene::quad): what the production traversal kernels walk.  FTN_QUAD=0: not built (the two-record kernels run) */
    uint32_t n_quads = 0, quad_bound = 0;
    if (!env_is("FTN_QUAD", 0)) {
        QuadBvh qb; build_quads(hs.nodes, &qb);
        if (qb.ok && qb.n_records) {
            if ((rc = sc->quad.upload(reinterpret_cast<const float4*>(qb.rec.data()), (size_t)8 * qb.n_records))) return rc;
            n_quads = qb.n_records; quad_bound = qb.stack_bound;
        }
    }
    clk.mark("four-box records (host + upload)");
    /* eight-box occlusion records (DScene::oct): triangle-only scenes with four-box records (the kernel hands its exceptional rays to the
     * same fallback).  FTN_OCT=0: not built (the four-box any-hit kernel traces the shadow rays) */
    uint32_t n_octs = 0, oct_bound = 0;
    if (!env_is("FTN_OCT", 0) && n_quads != 0 && d->n_spheres == 0) {
        OctBvh ob; build_octs(hs.nodes, geom, &ob);
        if (ob.ok && ob.n_records) {
            if ((rc = sc->oct.upload(reinterpret_cast<const uint4*>(ob.rec.data()), (size_t)8 * ob.n_records))) return rc;
            if (!ob.xbox.empty() && (rc = sc->oct_xbox.upload(ob.xbox.data(), ob.xbox.size()))) return rc;
            n_octs = ob.n_records; oct_bound = ob.stack_bound;
        }
    }
    clk.mark("eight-box records (host + upload)");
    /* shading records (DScene::srec): one 128-byte line per primitive with everything make_interaction reads.  FTN_SREC=0: not built.
     * (Building these three arrays on a worker while this thread uploads the previous one was measured: no faster -- the pageable
     * host-to-device copies and the builders want the same memory bandwidth.) */
    if (!env_is("FTN_SREC", 0) && np != 0 && (uint64_t)np * 128u <= (16ull << 30)) {
        std::unique_ptr<float4[]> rec(new float4[8 * np]);          /* (not value-initialised: every word is written below) */
        parallel_for(np, [&](size_t i0, size_t i1) { for (size_t i = i0; i < i1; i++) {
            float4* R = &rec[8 * i];
            for (int k = 3; k < 8; k++) R[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            const uint4 pi = info[2 * i], vi = info[2 * i + 1];
            R[0] = geom[FTN_GS * i]; R[1] = geom[FTN_GS * i + 1]; R[2] = geom[FTN_GS * i + 2];
            R[1].w = ftn_det::u2f(pi.x); R[2].w = ftn_det::u2f(pi.y); R[6].w = ftn_det::u2f(vi.w);
            const uint32_t fl = ftn_det::f2u(R[0].w);
            if (fl & GF_KIND_SPHERE) continue;
            const uint32_t v[3] = {vi.x, vi.y, vi.z};
            if (fl & GF_HAS_NORMALS) for (int k = 0; k < 3; k++) { R[3 + k].x = d->N[3 * (size_t)v[k]]; R[3 + k].y = d->N[3 * (size_t)v[k] + 1]; R[3 + k].z = d->N[3 * (size_t)v[k] + 2]; }
            if (fl & GF_HAS_UVS) {
                R[3].w = d->UV[2 * (size_t)v[0]]; R[4].w = d->UV[2 * (size_t)v[0] + 1]; R[5].w = d->UV[2 * (size_t)v[1]];
                R[6].x = d->UV[2 * (size_t)v[1] + 1]; R[6].y = d->UV[2 * (size_t)v[2]]; R[6].z = d->UV[2 * (size_t)v[2] + 1];
            }
        } });
        if ((rc = sc->srec.upload(rec.get(), 8 * np))) return rc;
    }
    clk.mark("shading records (host + upload)");
    /* leaf-test records: the shading records' first 48 bytes in triangle-only scenes (DScene::geom_stride = 8), the dense array otherwise */
    const bool geom_in_srec = sc->srec.p != nullptr && d->n_spheres == 0 && !env_is("FTN_GEOM", 1);
    if (!geom_in_srec && (rc = sc->geom.upload(geom.data(), geom.size()))) return rc;
    /* prim_info and the per-vertex normals / uvs: what the shading records replace (ftn_device.h: prim_mat_light / prim_normals / prim_uvs).
     * Resident only without records, or when a mesh has shading tangents (gathered through prim_info's vertex indices) */
    bool any_tangents = false;
    for (uint32_t i = 0; i < d->n_meshes; i++) if (d->meshes[i].has_tangents && d->S) any_tangents = true;
    if (!sc->srec.p || any_tangents || env_is("FTN_LEGACY_ARRAYS", 1)) {
        if ((rc = sc->prim_info.upload(info.data(), info.size()))) return rc;
        if (d->N && (rc = sc->N.upload(d->N, 3 * (size_t)d->n_vertices))) return rc;
        if (d->UV && (rc = sc->UV.upload(d->UV, 2 * (size_t)d->n_vertices))) return rc;
    }
    if (any_tangents && (rc = sc->T.upload(d->S, 3 * (size_t)d->n_vertices))) return rc;
    std::vector<DSphere> sph(d->n_spheres);
    for (uint32_t i = 0; i < d->n_spheres; i++) {
        const ftn_sphere& s = d->spheres[i]; DSphere& o = sph[i];
        memcpy(o.o2w, s.object_to_world.m, 64); memcpy(o.o2w_inv, s.object_to_world.inv, 64); memcpy(o.w2o, s.world_to_object.m, 64);
        o.radius = s.radius; o.z_min = s.z_min; o.z_max = s.z_max; o.theta_min = s.theta_min; o.theta_max = s.theta_max; o.phi_max = s.phi_max;
        o.reverse_orientation = s.reverse_orientation; o._pad = 0;
    }
    if ((rc = sc->spheres.upload(sph.data(), sph.size()))) return rc;
    /* materials: evaluate the constant-per-material parts of compute_scattering_functions once (textured materials stay raw and are
     * finalized per hit by material_resolve, ftn_texture.h) */
    std::vector<ftn_material> mats(d->materials, d->materials + d->n_materials);
    const bool has_tex = d->n_textures != 0 && d->textures != nullptr;
    for (size_t i = 0; i < mats.size(); i++) {
        bool textured = false;
        if (has_tex && d->material_textures) { const ftn_material_textures& mt = d->material_textures[i]; textured = (mt.a & mt.b & mt.s0 & mt.s1 & mt.s2) >= 0; }
        if (!textured) material_finalize(mats[i]);
    }
    if ((rc = sc->materials.upload(mats.data(), mats.size()))) return rc;
    clk.mark("leaf-test / attribute arrays, spheres, materials");
    {   /* shading class per primitive (DScene::prim_class) */
        std::vector<unsigned char> cls(np);
        for (size_t i = 0; i < np; i++) { const int m = (int)info[2 * i].x; cls[i] = (unsigned char)(m < 0 || (size_t)m >= mats.size() ? 7u : std::min<uint32_t>(2u + mats[(size_t)m].type, 7u)); }
        if ((rc = sc->prim_class.upload(cls.data(), cls.size()))) return rc;
    }
    /* lights */
    std::vector<DLight> lights(hs.light_kind.size());
    std::vector<uint32_t> inf;
    V3 wc((hs.world.lo[0] + hs.world.hi[0]) / 2.0f, (hs.world.lo[1] + hs.world.hi[1]) / 2.0f, (hs.world.lo[2] + hs.world.hi[2]) / 2.0f);   /* bounding_sphere bounds.rs:208-212 */
    wc = V3(0.0f, 0.0f, 0.0f) + wc;
    float wr = len(wc - V3(hs.world.hi[0], hs.world.hi[1], hs.world.hi[2]));
    for (size_t l = 0; l < lights.size(); l++) {
        DLight& L = lights[l]; memset(&L, 0, sizeof(L));
        L.kind = (uint32_t)hs.light_kind[l]; L.prim = hs.light_prim[l];
        if (L.kind == LK_AREA) {
            const ftn_prim& p = d->prims[hs.order[L.prim]];
            for (int k = 0; k < 3; k++) L.rgb[k] = d->area_emit[3 * p.area_emit + k];
            if (p.shape_kind == FTN_SHAPE_SPHERE) { const ftn_sphere& s = d->spheres[p.shape_index]; L.area = s.phi_max * s.radius * (s.z_max - s.z_min); }   /* sphere.rs:77-79 */
            else {                                                                /* triangle.rs:171-174 */
                const uint32_t* vi = d->tri_indices + 3 * (size_t)p.shape_index; const float* P = d->P;
                V3 p0(P[3 * vi[0]], P[3 * vi[0] + 1], P[3 * vi[0] + 2]), p1(P[3 * vi[1]], P[3 * vi[1] + 1], P[3 * vi[1] + 2]), p2(P[3 * vi[2]], P[3 * vi[2] + 1], P[3 * vi[2] + 2]);
                L.area = 0.5f * len(cross(p1 - p0, p2 - p0));
            }
            continue;
        }
        const ftn_light& src = d->lights[l];
        for (int k = 0; k < 3; k++) { L.rgb[k] = src.rgb[k]; L.v[k] = src.v[k]; }
        if (L.kind == LK_DISTANT || L.kind == LK_INFINITE) { L.world_center[0] = wc.x; L.world_center[1] = wc.y; L.world_center[2] = wc.z; L.world_radius = wr; }   /* preprocess */
        if (L.kind == LK_INFINITE) {
            inf.push_back((uint32_t)l);
            const ftn_envmap& e = d->envmaps[src.envmap];
            memcpy(L.l2w, src.light_to_world.m, 64); memcpy(L.w2l, src.light_to_world.inv, 64);
            L.env_w = e.width; L.env_h = e.height;
            /* compute_distribution infinite.rs:63-78: (height, width) = resolution() name swap (the distribution of a w x h map has h columns and
             * w rows, its rows filled from texture rows as if the map were h wide: reproduced as written); only pyramid level 0 is ever read */
            const uint32_t height = e.width, width = e.height;
            std::vector<float4> tex4((size_t)e.width * e.height);
            for (size_t k = 0; k < tex4.size(); k++) tex4[k] = make_float4(e.texels[3 * k], e.texels[3 * k + 1], e.texels[3 * k + 2], 0.0f);
            DLight hostL = L; hostL.texels = tex4.data();
            std::vector<float> img((size_t)width * height);
            for (uint32_t j = 0; j < height; j++) {
                float v = (float)j / (float)height;
                float sin_theta = ftn_det::sinf_det(FTN_PI * ((float)j + 0.5f) / (float)height);
                for (uint32_t i = 0; i < width; i++) {
                    float u = (float)i / (float)width;
                    Rgb tex = (e.width == 1 && e.height == 1) ? env_texel(hostL, 0, 0) : env_lookup(hostL, V2(u, v));
                    img[i + (size_t)j * width] = tex.luminance() * sin_theta;
                }
            }
            const uint32_t nu = width, nv = height;
            std::vector<float> ccdf((size_t)nv * (nu + 1)), cint(nv), mcdf(nv + 1);
            for (uint32_t v2 = 0; v2 < nv; v2++) cint[v2] = dist1d_build(&img[(size_t)v2 * nu], nu, &ccdf[(size_t)v2 * (nu + 1)]);
            float mint = dist1d_build(cint.data(), nv, mcdf.data());
            L.nu = nu; L.nv = nv; L.marg_integral = mint;
            DevBuf<float4> b0; DevBuf<float> b1, b2, b3, b4, b5;
            if ((rc = b0.upload(tex4.data(), tex4.size()))) return rc;
            if ((rc = b1.upload(img.data(), img.size()))) return rc;
            if ((rc = b2.upload(ccdf.data(), ccdf.size()))) return rc;
            if ((rc = b3.upload(cint.data(), cint.size()))) return rc;
            if ((rc = b4.upload(cint.data(), cint.size()))) return rc;             /* marginal func == conditional integrals */
            if ((rc = b5.upload(mcdf.data(), mcdf.size()))) return rc;
            L.texels = b0.p; L.cond_func = b1.p; L.cond_cdf = b2.p; L.cond_integral = b3.p; L.marg_func = b4.p; L.marg_cdf = b5.p;
            /* cell records (DLight::cells): square maps only -- compute_distribution's (height, width) swap makes the distribution's cell
             * (u, v) the texel (u, v) only then.  FTN_ENV_CELLS=0: not built */
            if (e.width == e.height && (uint64_t)e.width * e.height * 128u <= (2ull << 30) && !env_is("FTN_ENV_CELLS", 0)) {
                const int w = (int)e.width, h = (int)e.height;
                std::unique_ptr<float4[]> cells(new float4[8 * (size_t)w * h]);
                parallel_for((size_t)h, [&](size_t y0, size_t y1) { for (size_t cy = y0; cy < y1; cy++) for (int cx = 0; cx < w; cx++) {
                    float* c = reinterpret_cast<float*>(&cells[8 * (cy * (size_t)w + (size_t)cx)]);
                    for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
                        const int sx = ((cx + dx) % w + w) % w, sy = (((int)cy + dy) % h + h) % h;        /* env_texel's wrap */
                        const float4 t = tex4[(size_t)sy * w + sx];
                        float* o = c + 3 * ((dy + 1) * 3 + dx + 1);
                        o[0] = t.x; o[1] = t.y; o[2] = t.z;
                    }
                    c[27] = img[(size_t)cx + cy * (size_t)width];
                    c[28] = c[29] = c[30] = c[31] = 0.0f;
                } });
                DevBuf<float4> bc;
                if ((rc = bc.upload(cells.get(), 8 * (size_t)w * h))) return rc;
                L.cells = bc.p; sc->misc4.push_back(bc);
            }
            {   /* every 32nd CDF entry (DLight::cond_coarse / marg_coarse); only for CDFs that really are non-decreasing */
                bool monotone = true;
                auto check = [&](const float* c, uint32_t n) { for (uint32_t k = 0; k < n; k++) if (!(c[k] <= c[k + 1])) monotone = false; };
                for (uint32_t v2 = 0; v2 < nv; v2++) check(&ccdf[(size_t)v2 * (nu + 1)], nu);
                check(mcdf.data(), nv);
                if (monotone && !getenv("FTN_NO_COARSE_CDF")) {
                    const uint32_t nbu = (nu + 31u) / 32u, nbv = (nv + 31u) / 32u;
                    std::vector<float> cc((size_t)nv * (nbu + 1)), mc(nbv + 1);
                    for (uint32_t v2 = 0; v2 < nv; v2++) for (uint32_t k = 0; k <= nbu; k++) cc[(size_t)v2 * (nbu + 1) + k] = ccdf[(size_t)v2 * (nu + 1) + std::min(k * 32u, nu)];
                    for (uint32_t k = 0; k <= nbv; k++) mc[k] = mcdf[std::min(k * 32u, nv)];
                    DevBuf<float> b6, b7;
                    if ((rc = b6.upload(cc.data(), cc.size()))) return rc;
                    if ((rc = b7.upload(mc.data(), mc.size()))) return rc;
                    L.cond_coarse = b6.p; L.marg_coarse = b7.p;
                    sc->misc.push_back(b6); sc->misc.push_back(b7);
                }
            }
            sc->misc4.push_back(b0); sc->misc.push_back(b1); sc->misc.push_back(b2); sc->misc.push_back(b3); sc->misc.push_back(b4); sc->misc.push_back(b5);
        }
    }
    if ((rc = sc->lights.upload(lights.data(), lights.size()))) return rc;
    if ((rc = sc->inf_lights.upload(inf.data(), inf.size()))) return rc;
    clk.mark("classes, lights, environment tables, textures");
    DScene& D = sc->d; memset(&D, 0, sizeof(D));
    D.nodes = sc->nodes.p; D.geom = geom_in_srec ? sc->srec.p : sc->geom.p; D.geom_stride = geom_in_srec ? 8u : (uint32_t)FTN_GS; D.prim_info = sc->prim_info.p; D.N = sc->N.p; D.UV = sc->UV.p; D.T = sc->T.p; D.spheres = sc->spheres.p;
    D.materials = sc->materials.p; D.lights = sc->lights.p; D.inf_lights = sc->inf_lights.p;
    D.n_nodes = (uint32_t)hs.nodes.size(); D.n_prims = (uint32_t)np; D.n_lights = (uint32_t)lights.size(); D.n_inf_lights = (uint32_t)inf.size(); D.n_spheres = d->n_spheres;
    D.srec = sc->srec.p; D.prim_class = sc->prim_class.p;
    D.quad = sc->quad.p; D.n_quads = n_quads; D.quad_stack_bound = quad_bound;
    D.oct = sc->oct.p; D.oct_xbox = sc->oct_xbox.p; D.n_octs = n_octs; D.oct_stack_bound = oct_bound;
    D.fat = sc->fat.p; D.n_fat = n_fat; D.root_is_leaf = (!hs.nodes.empty() && hs.nodes[0].is_leaf) ? 1u : 0u;
    for (int k = 0; k < 3; k++) { D.root_lo[k] = hs.nodes.empty() ? 0.0f : hs.nodes[0].bmin[k]; D.root_hi[k] = hs.nodes.empty() ? 0.0f : hs.nodes[0].bmax[k]; }
    if (lights.size() == 1 && lights[0].kind == LK_INFINITE) { D.env_only = 1; D.env0 = lights[0]; }
    sc->stack_entries = std::max<uint32_t>(hs.max_depth, 1u);
    for (const ftn_material& m : mats) if (m.type < 32u) D.material_types |= 1u << m.type;
    if ((rc = sc->stats.alloc_zero(1))) return rc;
    if (has_tex) {                                               /* textures + MIP pyramids */
        const int nt = (int)d->n_textures, ni = d->images ? (int)d->n_images : 0;
        for (int i = 0; i < nt; i++) {
            const ftn_texture& t = d->textures[i];
            if (t.kind > FTN_TEX_IMAGE) return fail(FTN_ERR_INVALID_ARGUMENT, "unknown texture kind");
            if (t.kind == FTN_TEX_CHECKERBOARD && (t.tex1 < 0 || t.tex1 >= nt || t.tex2 < 0 || t.tex2 >= nt)) return fail(FTN_ERR_INVALID_ARGUMENT, "checkerboard child texture out of range");
            if (t.kind == FTN_TEX_IMAGE && (t.image < 0 || t.image >= ni)) return fail(FTN_ERR_INVALID_ARGUMENT, "image index out of range");
        }
        std::vector<ftn_material_textures> mt(d->n_materials);
        for (uint32_t i = 0; i < d->n_materials; i++) {
            if (d->material_textures) mt[i] = d->material_textures[i]; else { mt[i].a = mt[i].b = mt[i].s0 = mt[i].s1 = mt[i].s2 = -1; }
            for (int32_t id : {mt[i].a, mt[i].b, mt[i].s0, mt[i].s1, mt[i].s2}) if (id >= nt) return fail(FTN_ERR_INVALID_ARGUMENT, "material texture index out of range");
        }
        std::vector<DImage> imgs((size_t)ni); std::vector<float4> texels;
        for (int i = 0; i < ni; i++) {
            const ftn_image& im = d->images[i];
            if (im.width == 0 || im.height == 0 || !im.texels || im.wrap > FTN_WRAP_CLAMP) return fail(FTN_ERR_INVALID_ARGUMENT, "bad image");
            std::vector<MipLevelHost> pyr; build_mip_pyramid(im.width, im.height, im.texels, &pyr);
            if (pyr.size() > 16) return fail(FTN_ERR_UNSUPPORTED, "image larger than 32768 texels on a side");
            DImage& D2 = imgs[(size_t)i]; memset(&D2, 0, sizeof(D2));
            D2.w = im.width; D2.h = im.height; D2.wrap = im.wrap; D2.n_levels = (uint32_t)pyr.size();
            for (size_t l = 0; l < pyr.size(); l++) {
                D2.off[l] = (uint32_t)texels.size(); D2.lw[l] = pyr[l].w; D2.lh[l] = pyr[l].h;
                for (size_t k = 0; k < (size_t)pyr[l].w * pyr[l].h; k++) texels.push_back(make_float4(pyr[l].rgb[3 * k], pyr[l].rgb[3 * k + 1], pyr[l].rgb[3 * k + 2], 0.0f));
            }
        }
        if ((rc = sc->textures.upload(d->textures, (size_t)nt)) || (rc = sc->mtex.upload(mt.data(), mt.size())) || (rc = sc->images.upload(imgs.data(), imgs.size())) ||
            (rc = sc->texels.upload(texels.data(), texels.size()))) return rc;
        D.textures = sc->textures.p; D.mtex = sc->mtex.p; D.images = sc->images.p; D.texels = sc->texels.p; D.n_textures = (uint32_t)nt;
    }
    return FTN_OK;
}

static int set_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return fail(FTN_ERR_NO_DEVICE, "no HIP device available: the fountain HIP path needs an AMD GPU (there is no CPU fallback)");
    if (device >= 0) {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) { (void)hipGetLastError();      /* do not leave the error behind for the next call's hipGetLastError() */
                               return fail(FTN_ERR_NO_DEVICE, std::string("hipSetDevice(device): ") + hipGetErrorString(e)); }
    }
    return FTN_OK;
}

extern "C" {

const char* ftn_last_error(void) { return g_err.c_str(); }
const char* ftn_version(void) { return "fountain_hip 0.3 (gfx950)"; }
int ftn_abi_version(void) { return FTN_ABI_VERSION; }
int ftn_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

int ftn_bvh_build(const ftn_scene_desc* d, ftn_bvh_node* nodes_out, uint32_t* order_out, uint32_t* n_nodes_out, uint32_t* max_depth_out) {
    HostScene hs; int rc = build_host_scene(d, &hs); if (rc) return rc;
    if (nodes_out) memcpy(nodes_out, hs.nodes.data(), hs.nodes.size() * sizeof(ftn_bvh_node));
    if (order_out) memcpy(order_out, hs.order.data(), hs.order.size() * sizeof(uint32_t));
    if (n_nodes_out) *n_nodes_out = (uint32_t)hs.nodes.size();
    if (max_depth_out) *max_depth_out = hs.max_depth;
    return FTN_OK;
}

int ftn_bvh_quads(const ftn_bvh_node* nodes, uint32_t n_nodes, float* records_out, uint32_t* n_records_out, uint32_t* stack_bound_out) {
    if (!nodes && n_nodes) return fail(FTN_ERR_INVALID_ARGUMENT, "null nodes");
    std::vector<ftn_bvh_node> v(nodes, nodes + n_nodes);
    for (uint32_t i = 0; i < n_nodes; i++) if (!v[i].is_leaf && (v[i].idx <= i + 1 || v[i].idx >= n_nodes || v[i].axis > 2)) return fail(FTN_ERR_INVALID_ARGUMENT, "not a flattened BVH (bvh.rs:133-158)");
    QuadBvh qb; build_quads(v, &qb);
    if (!qb.ok) return fail(FTN_ERR_UNSUPPORTED, "more than 2^24 four-box records");
    if (records_out) memcpy(records_out, qb.rec.data(), qb.rec.size() * sizeof(float));
    if (n_records_out) *n_records_out = qb.n_records;
    if (stack_bound_out) *stack_bound_out = qb.stack_bound;
    return FTN_OK;
}

int ftn_bvh_octs(const ftn_bvh_node* nodes, uint32_t n_nodes, uint32_t* records_out, uint32_t* n_records_out, uint32_t* stack_bound_out, float* xbox_out, uint32_t* n_xbox_out) {
    if (!nodes && n_nodes) return fail(FTN_ERR_INVALID_ARGUMENT, "null nodes");
    std::vector<ftn_bvh_node> v(nodes, nodes + n_nodes);
    for (uint32_t i = 0; i < n_nodes; i++) if (!v[i].is_leaf && (v[i].idx <= i + 1 || v[i].idx >= n_nodes || v[i].axis > 2)) return fail(FTN_ERR_INVALID_ARGUMENT, "not a flattened BVH (bvh.rs:133-158)");
    OctBvh ob; build_octs(v, std::vector<float4>(), &ob);       /* (no vertex data here: every leaf gets an explicit box) */
    if (!ob.ok) return fail(FTN_ERR_UNSUPPORTED, "no interior node, or more than 2^24 eight-box records");
    if (records_out) memcpy(records_out, ob.rec.data(), ob.rec.size() * sizeof(uint32_t));
    if (n_records_out) *n_records_out = ob.n_records;
    if (stack_bound_out) *stack_bound_out = ob.stack_bound;
    if (xbox_out) memcpy(xbox_out, ob.xbox.data(), ob.xbox.size() * sizeof(float4));
    if (n_xbox_out) *n_xbox_out = (uint32_t)(ob.xbox.size() / 2);
    return FTN_OK;
}

int ftn_scene_create(const ftn_scene_desc* d, int device, ftn_scene** out) {
    if (!out) return fail(FTN_ERR_INVALID_ARGUMENT, "null output");
    int rc = set_device(device); if (rc) return rc;
    ftn_scene* sc = new ftn_scene();
    sc->device = device;
    PhaseClock clk;
    rc = build_host_scene(d, &sc->host);
    clk.mark("host scene: bounds, BVH, lights");
    if (!rc && sc->host.max_depth > 64) rc = fail(FTN_ERR_BVH_TOO_DEEP, "BVH deeper than the reference's 64-entry traversal stack (bvh.rs:168)");
    if (!rc) rc = upload_scene(d, sc);
    if (rc) { delete sc; return rc; }
    *out = sc; return FTN_OK;
}
void ftn_scene_destroy(ftn_scene* s) { delete s; }
int ftn_scene_info(const ftn_scene* s, uint32_t* n_nodes, uint32_t* n_prims, uint32_t* n_lights, uint32_t* max_depth, float wb[6]) {
    if (n_nodes) *n_nodes = (uint32_t)s->host.nodes.size();
    if (n_prims) *n_prims = (uint32_t)s->host.order.size();
    if (n_lights) *n_lights = (uint32_t)s->host.light_kind.size();
    if (max_depth) *max_depth = s->host.max_depth;
    if (wb) for (int i = 0; i < 3; i++) { wb[i] = s->host.world.lo[i]; wb[3 + i] = s->host.world.hi[i]; }
    return FTN_OK;
}
int ftn_scene_memory_info(const ftn_scene* s, ftn_scene_memory* m) {
    if (!s || !m) return fail(FTN_ERR_INVALID_ARGUMENT, "null scene / output");
    memset(m, 0, sizeof(*m));
    m->oct = s->oct.n * sizeof(uint4) + s->oct_xbox.n * sizeof(float4);
    m->nodes = s->nodes.n * sizeof(float4); m->quad = s->quad.n * sizeof(float4); m->fat = s->fat.n * sizeof(float4); m->geom = s->geom.n * sizeof(float4);
    m->srec = s->srec.n * sizeof(float4); m->indexed_attributes = s->prim_info.n * sizeof(uint4) + (s->N.n + s->UV.n + s->T.n) * sizeof(float);
    m->prim_class = s->prim_class.n;
    m->lights = s->lights.n * sizeof(DLight) + s->inf_lights.n * sizeof(uint32_t);
    for (const auto& b : s->misc) m->lights += b.n * sizeof(float);
    for (const auto& b : s->misc4) m->lights += b.n * sizeof(float4);
    m->textures = s->textures.n * sizeof(ftn_texture) + s->mtex.n * sizeof(ftn_material_textures) + s->images.n * sizeof(DImage) + s->texels.n * sizeof(float4);
    m->other = s->spheres.n * sizeof(DSphere) + s->materials.n * sizeof(ftn_material) + s->stats.n * sizeof(DevStats);
    m->total = m->oct + m->nodes + m->quad + m->fat + m->geom + m->srec + m->indexed_attributes + m->prim_class + m->lights + m->textures + m->other;
    return FTN_OK;
}
int ftn_scene_get_nodes(const ftn_scene* s, ftn_bvh_node* nodes, uint32_t* order) {
    if (nodes) memcpy(nodes, s->host.nodes.data(), s->host.nodes.size() * sizeof(ftn_bvh_node));
    if (order) memcpy(order, s->host.order.data(), s->host.order.size() * sizeof(uint32_t));
    return FTN_OK;
}
int ftn_scene_get_lights(const ftn_scene* s, int32_t* kind, int32_t* prim) {
    for (size_t i = 0; i < s->host.light_kind.size(); i++) { kind[i] = s->host.light_kind[i]; prim[i] = s->host.light_prim[i]; }
    return FTN_OK;
}

/* ------------------------------------------------------------------ batch intersection */
static void stats_out(const DevStats& ds, ftn_stats* st, double ms) {
    if (!st) return;
    memset(st, 0, sizeof(*st));
    st->rays_closest = ds.rays_closest; st->rays_any = ds.rays_any; st->nodes_visited = ds.nodes_visited; st->prims_tested = ds.prims_tested;
    st->camera_samples = ds.camera_samples; st->spill_samples = ds.spill_samples; st->kernel_ms = ms;
    st->nodes_visited_any = ds.nodes_any; st->prims_tested_any = ds.prims_any;
    st->quad_records = ds.quad_records; st->quad_records_any = ds.quad_records_any;
}
static int trace_batch(const ftn_scene* cs, const float* rays, size_t n, int mode, float* t_hit, int32_t* prim, float* bary, uint8_t* occ, float* out24, ftn_stats* st) {
    ftn_scene* s = const_cast<ftn_scene*>(cs);
    int rc = set_device(s->device); if (rc) return rc;
    DevBuf<float> d_rays, d_t, d_b, d_o; DevBuf<int> d_p; DevBuf<unsigned char> d_occ;
    auto cleanup = [&]() { d_rays.release(); d_t.release(); d_b.release(); d_o.release(); d_p.release(); d_occ.release(); };
    if ((rc = d_rays.upload(rays, 8 * n))) { cleanup(); return rc; }
    if (mode == 0) { if ((rc = d_t.alloc_zero(n)) || (rc = d_p.alloc_zero(n)) || (rc = d_b.alloc_zero(3 * n))) { cleanup(); return rc; } }
    else if (mode == 1) { if ((rc = d_occ.alloc_zero(n))) { cleanup(); return rc; } }
    else { if ((rc = d_o.alloc_zero(24 * n))) { cleanup(); return rc; } }
    HIP_TRY(hipMemset(s->stats.p, 0, sizeof(DevStats)));
    hipEvent_t e0, e1; HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, 0));
    if (getenv("FTN_BATCH_SIMPLE")) launch_trace_batch(s->d, d_rays.p, n, mode, d_t.p, d_p.p, d_b.p, d_occ.p, d_o.p, s->stats.p, s->stack_entries, st != nullptr, 0);   /* one lane per ray, plain loop */
    else if ((rc = wavefront_trace_batch(&s->wf, s->d, s->stack_entries, d_rays.p, n, mode, st != nullptr, d_t.p, d_p.p, d_b.p, d_occ.p, d_o.p, s->stats.p, 0))) { cleanup(); return fail(rc, wavefront_error()); }
    HIP_TRY(hipEventRecord(e1, 0));
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipGetLastError());
    float ms = 0.0f; (void)hipEventElapsedTime(&ms, e0, e1); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (t_hit) HIP_TRY(hipMemcpy(t_hit, d_t.p, n * 4, hipMemcpyDeviceToHost));
    if (prim) HIP_TRY(hipMemcpy(prim, d_p.p, n * 4, hipMemcpyDeviceToHost));
    if (bary) HIP_TRY(hipMemcpy(bary, d_b.p, 3 * n * 4, hipMemcpyDeviceToHost));
    if (occ) HIP_TRY(hipMemcpy(occ, d_occ.p, n, hipMemcpyDeviceToHost));
    if (out24) HIP_TRY(hipMemcpy(out24, d_o.p, 24 * n * 4, hipMemcpyDeviceToHost));
    DevStats ds; HIP_TRY(hipMemcpy(&ds, s->stats.p, sizeof(ds), hipMemcpyDeviceToHost));
    stats_out(ds, st, ms);
    cleanup();
    return FTN_OK;
}
int ftn_intersect(const ftn_scene* s, const float* rays, size_t n, float* t_hit, int32_t* prim, float* bary, ftn_stats* st) {
    return trace_batch(s, rays, n, 0, t_hit, prim, bary, nullptr, nullptr, st);
}
int ftn_intersect_test(const ftn_scene* s, const float* rays, size_t n, uint8_t* occluded, ftn_stats* st) {
    return trace_batch(s, rays, n, 1, nullptr, nullptr, nullptr, occluded, nullptr, st);
}
int ftn_intersect_full(const ftn_scene* s, const float* rays, size_t n, float* out24) {
    return trace_batch(s, rays, n, 2, nullptr, nullptr, nullptr, nullptr, out24, nullptr);
}

/* ------------------------------------------------------------------ render */
int ftn_render_device(const ftn_scene* cs, const ftn_camera_desc* cam, const ftn_film_desc* film, const ftn_sampler_desc* sd, const ftn_integrator_desc* id,
                      const ftn_tile_range* tr, const ftn_render_options* opt, void* device_pixels, void* stream_v, ftn_stats* st) {
    if (!cs || !cam || !film || !sd || !id || !device_pixels) return fail(FTN_ERR_INVALID_ARGUMENT, "null argument");
    ftn_scene* s = const_cast<ftn_scene*>(cs);
    /* the scene's arrays live on the device it was created on: a different device in the options would launch with foreign pointers */
    if (opt && opt->device >= 0 && s->device >= 0 && opt->device != s->device) return fail(FTN_ERR_INVALID_ARGUMENT, "ftn_render_options.device differs from the device the scene was created on");
    int rc = set_device(opt && opt->device >= 0 ? opt->device : s->device); if (rc) return rc;
    hipStream_t stream = (hipStream_t)stream_v;
    const bool indexed = sd->kind == FTN_SAMPLER_INDEXED;
    /* sample range [first_sample, first_sample + sample_count) must lie inside [0, samples_per_pixel] (64-bit: no wrap-around) */
    if (indexed && ((uint64_t)sd->first_sample > (uint64_t)sd->samples_per_pixel || (uint64_t)sd->first_sample + (uint64_t)sd->sample_count > (uint64_t)sd->samples_per_pixel))
        return fail(FTN_ERR_INVALID_ARGUMENT, "sample range outside [0, samples_per_pixel]");
    if (!indexed && sd->kind != FTN_SAMPLER_TILE_SERIAL) return fail(FTN_ERR_INVALID_ARGUMENT, "unknown sampler kind");
    if (!indexed && (sd->first_sample != 0 || (sd->sample_count != 0 && sd->sample_count != sd->samples_per_pixel)))
        return fail(FTN_ERR_INVALID_ARGUMENT, "sample ranges need FTN_SAMPLER_INDEXED");
    if (id->kind != FTN_INTEGRATOR_PATH && id->kind != FTN_INTEGRATOR_DIRECT_LIGHTING && id->kind != FTN_INTEGRATOR_WHITTED) return fail(FTN_ERR_INVALID_ARGUMENT, "unknown integrator kind");
    if (id->max_depth > 65535u) return fail(FTN_ERR_INVALID_ARGUMENT, "max_depth is a u16 in the reference (integrator/path.rs:14)");
    uint32_t pipeline = opt ? opt->pipeline : FTN_PIPELINE_AUTO;
    /* the wavefront pipeline renders the indexed sampler for PathIntegrator, DirectLightingIntegrator and WhittedIntegrator (the latter with at
     * most 32 lights: one bit per light in a path's pending-light word), and the reference's tile-serial sampler for PathIntegrator (one
     * path per tile in flight: wavefront_render_serial).  AUTO takes it for the tile-serial sampler once the call has enough tiles to fill
     * the queues (below, when the tile list is known); everything else is the megakernel's */
    const bool wf_ok = indexed ? (id->kind != FTN_INTEGRATOR_WHITTED || s->d.n_lights <= 32u) : id->kind == FTN_INTEGRATOR_PATH;
    const bool auto_pipeline = pipeline == FTN_PIPELINE_AUTO;
    if (auto_pipeline) pipeline = wf_ok ? FTN_PIPELINE_WAVEFRONT : FTN_PIPELINE_MEGAKERNEL;
    if (pipeline == FTN_PIPELINE_WAVEFRONT && !wf_ok)
        return fail(FTN_ERR_UNSUPPORTED, "the wavefront pipeline renders PathIntegrator with either sampler, DirectLightingIntegrator / WhittedIntegrator (up to 32 lights) with FTN_SAMPLER_INDEXED");
    const bool count = opt && opt->count_traffic;
    const bool count_production = opt && opt->count_traffic == 2;      /* tally the production configuration instead of the reference's walk */

    const uint32_t stride = tr && tr->stride ? tr->stride : 1, first = tr ? tr->first : 0, cnt = tr ? tr->count : 0;
    /* the tile list only depends on the film and the tile range: keep it (and its device copy) between calls */
    int32_t key[10] = {film->crop[0], film->crop[1], film->crop[2], film->crop[3], (int32_t)ftn_det::f2u(film->filter_radius[0]), (int32_t)ftn_det::f2u(film->filter_radius[1]),
                       (int32_t)first, (int32_t)stride, (int32_t)cnt, 1};
    const bool tiles_cached = memcmp(key, s->tile_key, sizeof(key)) == 0;
    if (!tiles_cached) {
        memset(s->tile_key, 0, sizeof(s->tile_key));      /* the host list is about to change: no key is valid until list AND device copy are in place */
        std::vector<DTile> all; list_tiles(film, &all);
        s->sel.clear();
        for (size_t i = first, k = 0; i < all.size() && (cnt == 0 || k < cnt); i += stride, k++) s->sel.push_back(all[i]);
        uint32_t off = 0; for (DTile& t : s->sel) { t.valid_off = off; t._pad = 0; off += (uint32_t)((t.x1 - t.x0) * (t.y1 - t.y0)); }
    }
    std::vector<DTile>& sel = s->sel;
    /* tile-serial: a round of the queue pipeline costs ~0.3 ms whatever it holds, a megakernel lane per tile diverges from its 63 neighbours:
     * the queues win once a call has a few thousand tiles (measured: profiles/r03), a handful of tiles is the megakernel's */
    if (auto_pipeline && !indexed && pipeline == FTN_PIPELINE_WAVEFRONT && sel.size() < 2048) pipeline = FTN_PIPELINE_MEGAKERNEL;

    RenderParams P; memset(&P, 0, sizeof(P));
    P.S = s->d;
    memcpy(P.C.c2w, cam->camera_to_world.m, 64); memcpy(P.C.r2c, cam->raster_to_camera.m, 64);
    P.C.shutter_open = cam->shutter_open; P.C.shutter_close = cam->shutter_close; P.C.lens_radius = cam->lens_radius; P.C.focal_dist = cam->focal_dist;
    for (int k = 0; k < 3; k++) { P.C.dx_camera[k] = cam->dx_camera[k]; P.C.dy_camera[k] = cam->dy_camera[k]; }
    for (int i = 0; i < 4; i++) P.crop[i] = film->crop[i];
    P.radius[0] = film->filter_radius[0]; P.radius[1] = film->filter_radius[1]; P.inv_radius[0] = 1.0f / P.radius[0]; P.inv_radius[1] = 1.0f / P.radius[1];
    P.sampler_kind = sd->kind; P.spp = sd->samples_per_pixel; P.seed = sd->seed;
    P.first_sample = indexed ? sd->first_sample : 0;
    P.last_sample = indexed ? sd->first_sample + (sd->sample_count ? sd->sample_count : (sd->samples_per_pixel - sd->first_sample)) : sd->samples_per_pixel;
    P.integrator_kind = id->kind; P.max_depth = id->max_depth; P.rr_threshold = id->rr_threshold;
    P.stack_entries = s->stack_entries;

    const size_t npix = (size_t)std::max(0, film->crop[2] - film->crop[0]) * (size_t)std::max(0, film->crop[3] - film->crop[1]);
    if (npix > s->acc_pixels) {
        s->accA.release(); s->accB.release(); s->accC.release();
        HIP_TRY(hipMalloc((void**)&s->accA.p, npix * sizeof(float4))); HIP_TRY(hipMalloc((void**)&s->accB.p, npix * sizeof(float4))); HIP_TRY(hipMalloc((void**)&s->accC.p, npix * sizeof(float4)));
        s->acc_pixels = npix; s->spill_acc_dirty = true;
    }
    if (sel.size() > s->tiles.n) { s->tiles.release(); HIP_TRY(hipMalloc((void**)&s->tiles.p, sel.size() * sizeof(DTile))); s->tiles.n = sel.size(); }
    HIP_TRY(hipMemsetAsync(s->accA.p, 0, npix * sizeof(float4), stream));
    if (s->spill_acc_dirty) {          /* else: still all zero from the last call (DevStats::bc_writes said nothing was added) */
        HIP_TRY(hipMemsetAsync(s->accB.p, 0, npix * sizeof(float4), stream));
        HIP_TRY(hipMemsetAsync(s->accC.p, 0, npix * sizeof(float4), stream));
    }
    s->spill_acc_dirty = true;         /* until this call has finished and reported otherwise */
    HIP_TRY(hipMemsetAsync(s->stats.p, 0, sizeof(DevStats), stream));
    if (!tiles_cached) {
        if (!sel.empty()) HIP_TRY(hipMemcpyAsync(s->tiles.p, sel.data(), sel.size() * sizeof(DTile), hipMemcpyHostToDevice, stream));
        memcpy(s->tile_key, key, sizeof(key));            /* (an empty selection is a valid cached state too) */
    }
    P.tiles = s->tiles.p; P.n_tiles = (uint32_t)sel.size();
    P.accA = s->accA.p; P.accB = s->accB.p; P.accC = s->accC.p; P.stats = s->stats.p;

    struct EventPair {                                    /* destroyed on every way out */
        hipEvent_t a = nullptr, b = nullptr;
        ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } ev;
    HIP_TRY(hipEventCreate(&ev.a)); HIP_TRY(hipEventCreate(&ev.b));
    HIP_TRY(hipEventRecord(ev.a, stream));
    WavefrontTimes wt; memset(&wt, 0, sizeof(wt));
    if (pipeline == FTN_PIPELINE_WAVEFRONT) { rc = wavefront_render(&s->wf, P, sel, count, stream, &wt, count_production); if (rc) return fail(rc, wavefront_error()); }
    else launch_render_mega(P, count, stream);
    launch_film_resolve(P, (ftn_pixel*)device_pixels, stream);
    HIP_TRY(hipEventRecord(ev.b, stream));
    HIP_TRY(hipEventSynchronize(ev.b));
    HIP_TRY(hipGetLastError());
    float ms = 0.0f; (void)hipEventElapsedTime(&ms, ev.a, ev.b);
    DevStats ds; HIP_TRY(hipMemcpy(&ds, s->stats.p, sizeof(ds), hipMemcpyDeviceToHost));
    s->spill_acc_dirty = ds.bc_writes != 0;
    ds.rays_closest += wt.mis_any_rays; ds.rays_any -= wt.mis_any_rays;       /* they are Scene::intersect calls in the reference's accounting */
    stats_out(ds, st, ms);
    if (ds.t4_occ[0] | ds.t4_occ[7]) { const char* dbg = getenv("FTN_WF_DEBUG"); if (dbg && atoi(dbg)) {     /* lane occupancy of the counting builds (experiments) */
        for (int k = 0; k < 2; k++) { const unsigned long long* o = &ds.t4_occ[7 * k]; if (!o[0]) continue;
            fprintf(stderr, "[wf] %s four-box trace: %llu control rounds; %llu record steps with %.1f of 64 lanes; %llu leaf steps with %.1f lanes; %llu refills of %.1f lanes\n", k ? "any-hit" : "closest-hit",
                    o[0], o[1], o[1] ? (double)o[2] / (double)o[1] : 0.0, o[3], o[3] ? (double)o[4] / (double)o[3] : 0.0, o[5], o[5] ? (double)o[6] / (double)o[5] : 0.0); } } }
    if (st) {
        st->trace_ms = wt.trace_ms; st->trace_launches = wt.trace_launches; st->mis_rays_any_hit = wt.mis_any_rays;
        st->any_ms = wt.any_ms; st->any_launches = wt.any_launches; st->shade_ms = wt.shade_ms; st->shade_launches = wt.shade_launches; st->sort_ms = wt.sort_ms;
    }
    if (ds.error == FTN_ERR_NAN_RADIANCE) return fail(FTN_ERR_NAN_RADIANCE, "NaN radiance value (integrator/mod.rs:285-287)");
    if (ds.error) return fail(ds.error, "unsupported material / integrator combination (e.g. specular glass: material/glass.rs:66)");
    return FTN_OK;
}

int ftn_test_math(int which, const float* x, const float* y, size_t n, float* out) {
    int rc = set_device(-1); if (rc) return rc;
    DevBuf<float> dx, dy, dout;
    if ((rc = dx.upload(x, n)) || (rc = dy.upload(y, n)) || (rc = dout.alloc_zero(n))) { dx.release(); dy.release(); dout.release(); return rc; }
    launch_test_math(which, dx.p, dy.p, n, dout.p, 0);
    hipError_t e = hipMemcpy(out, dout.p, n * sizeof(float), hipMemcpyDeviceToHost);
    dx.release(); dy.release(); dout.release();
    if (e != hipSuccess) return fail(FTN_ERR_NO_DEVICE, hipGetErrorString(e));
    return FTN_OK;
}

/* test hooks for the texture path: a pyramid level as built on the host; Texture::evaluate on the device for arrays of (uv, dudx, dvdx, dudy, dvdy) */
int ftn_test_mipmap_level(uint32_t width, uint32_t height, const float* texels, uint32_t level, uint32_t* level_w, uint32_t* level_h, float* rgb_out) {
    if (!texels || width == 0 || height == 0) return fail(FTN_ERR_INVALID_ARGUMENT, "bad image");
    std::vector<MipLevelHost> pyr; build_mip_pyramid(width, height, texels, &pyr);
    if (level >= pyr.size()) return fail(FTN_ERR_INVALID_ARGUMENT, "no such level");
    if (level_w) *level_w = pyr[level].w;
    if (level_h) *level_h = pyr[level].h;
    if (rgb_out) memcpy(rgb_out, pyr[level].rgb.data(), pyr[level].rgb.size() * sizeof(float));
    return FTN_OK;
}
int ftn_test_texture_eval(const ftn_scene* cs, int32_t texture, const float* uv_diffs6, size_t n, float* rgb_out) {
    if (!cs || !uv_diffs6 || !rgb_out) return fail(FTN_ERR_INVALID_ARGUMENT, "null argument");
    if (texture < 0 || (uint32_t)texture >= cs->d.n_textures) return fail(FTN_ERR_INVALID_ARGUMENT, "texture index out of range");
    int rc = set_device(cs->device); if (rc) return rc;
    DevBuf<float> din, dout;
    if ((rc = din.upload(uv_diffs6, 6 * n)) || (rc = dout.alloc_zero(3 * n))) { din.release(); dout.release(); return rc; }
    launch_test_texture_eval(cs->d, texture, din.p, n, dout.p, 0);
    hipError_t e = hipMemcpy(rgb_out, dout.p, 3 * n * sizeof(float), hipMemcpyDeviceToHost);
    din.release(); dout.release();
    if (e != hipSuccess) return fail(FTN_ERR_NO_DEVICE, hipGetErrorString(e));
    return FTN_OK;
}

int ftn_film_resolve_device(const void* device_pixels, size_t n, void* device_rgb_out, void* stream) {   /* film.rs:195-210 in HBM */
    if (!device_pixels || !device_rgb_out) return fail(FTN_ERR_INVALID_ARGUMENT, "null argument");
    if (ftn_device_count() <= 0) return fail(FTN_ERR_NO_DEVICE, "no HIP device");
    launch_spectrum_buffer((const ftn_pixel*)device_pixels, n, (float*)device_rgb_out, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FTN_ERR_INTERNAL, hipGetErrorString(e));
    return FTN_OK;
}

int ftn_render(const ftn_scene* cs, const ftn_camera_desc* cam, const ftn_film_desc* film, const ftn_sampler_desc* sd, const ftn_integrator_desc* id,
               const ftn_tile_range* tr, const ftn_render_options* opt, ftn_pixel* out_pixels, ftn_stats* st) {
    if (!cs || !film || !out_pixels) return fail(FTN_ERR_INVALID_ARGUMENT, "null argument");
    if (opt && opt->device >= 0 && cs->device >= 0 && opt->device != cs->device) return fail(FTN_ERR_INVALID_ARGUMENT, "ftn_render_options.device differs from the device the scene was created on");
    int rc = set_device(opt && opt->device >= 0 ? opt->device : cs->device); if (rc) return rc;
    const size_t npix = (size_t)std::max(0, film->crop[2] - film->crop[0]) * (size_t)std::max(0, film->crop[3] - film->crop[1]);
    DevBuf<ftn_pixel> dev;
    if ((rc = dev.alloc_zero(npix))) return rc;
    rc = ftn_render_device(cs, cam, film, sd, id, tr, opt, dev.p, nullptr, st);
    if (rc == FTN_OK || rc == FTN_ERR_NAN_RADIANCE) {
        std::vector<ftn_pixel> h(npix);
        if (hipMemcpy(h.data(), dev.p, npix * sizeof(ftn_pixel), hipMemcpyDeviceToHost) != hipSuccess) { dev.release(); return fail(FTN_ERR_NO_DEVICE, "copy back failed"); }
        for (size_t i = 0; i < npix; i++) {                                       /* merge_pixel.xyz[i] += xyz[i] (film.rs:127-130) */
            out_pixels[i].xyz[0] += h[i].xyz[0]; out_pixels[i].xyz[1] += h[i].xyz[1]; out_pixels[i].xyz[2] += h[i].xyz[2];
            out_pixels[i].filter_weight_sum += h[i].filter_weight_sum;
        }
    }
    dev.release();
    return rc;
}

}  /* extern "C" */
