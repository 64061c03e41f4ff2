/*
 * ftn_device.h -- gfx950 device library of the fountain path-tracing core: HBM data layout and the
 * per-lane routines (traversal, intersection, shading geometry, BSDFs, lights, sampler, camera, film).
 *
 * Layout in HBM (all built by ftn_scene_create, BVH primitive order):
 *   nodes      2 x float4 per LinearBVHNode: {min.x, max.x, min.y, max.y} {min.z, max.z, bits(link), bits(n_prims | onehot(axis)<<16 | leaf<<24)}
 *              link = BYTE offset of the second child's record (interior; the first child is the next record) or first primitive (leaf):
 *              traversal keeps byte offsets, so a node address is base + offset with no index arithmetic;
 *              (each axis' (min, max) pair sits in an even/odd register pair after the two dwordx4 loads, which is what the
 *               packed-f32 subtract / multiply of the slab test consume: no register shuffling per node visit)
 *   geom       3 x float4 per primitive    : triangle {p0, bits(flags)} {p1, bits(shape idx)} {p2, 0}
 *                                            sphere   {0,0,0, bits(flags)} {0,0,0, bits(sphere idx)} {..}
 *              (vertices pre-gathered, so a leaf test is one 48-byte read: the reference chases
 *               Box<dyn Primitive> -> Arc<Triangle> -> Arc<TriangleMesh> -> indices -> vertices)
 *   prim_info  2 x uint4 per primitive     : {material, light, flags, 0} {v0, v1, v2, shape idx}   (shading only)
 *   N, UV      per-vertex normals / uvs (shading only)
 * A closest hit is carried as {t, prim, b0, b1, b2} (20 bytes); the full SurfaceInteraction is recomputed from it
 * by the shading stage with the reference's expressions (src/shapes/triangle.rs:270-393), instead of being built
 * for every candidate hit as the reference does.
 */
#ifndef FTN_DEVICE_H
#define FTN_DEVICE_H

#include <hip/hip_runtime.h>
#include "ftn_math.h"
#include "../../include/fountain_hip.h"

namespace ftn {

#define FTN_DEV_NOINLINE __device__ inline   /* out-of-line variants were measured slower (see detmath.h) */

/* float4s per primitive in the HOST copy of the leaf-test records (the device stride is DScene::geom_stride) */
#define FTN_GS 3
enum : uint32_t { GF_KIND_SPHERE = 1u, GF_HAS_NORMALS = 2u, GF_HAS_UVS = 4u, GF_FLIP = 8u, GF_LEAF_END = 16u /* last primitive of its BVH leaf */, GF_HAS_TANGENTS = 32u /* per-vertex shading tangents in DScene::T */ };
enum : uint32_t { LK_POINT = 0, LK_DISTANT = 1, LK_INFINITE = 2, LK_AREA = 3 };

struct DSphere {
    float o2w[16], o2w_inv[16], w2o[16];
    float radius, z_min, z_max, theta_min, theta_max, phi_max;
    uint32_t reverse_orientation, _pad;
};

struct DLight {
    uint32_t kind; int32_t prim;            /* area: BVH-ordered primitive */
    float rgb[3];                           /* I | L | emit */
    float v[3];                             /* world_point | dir_to_light */
    float world_center[3]; float world_radius; float area;
    /* infinite */
    uint32_t env_w, env_h;
    const float4* texels;                   /* env_w*env_h texels, rgb + pad: one aligned 16-byte load each */
    const float* cond_func;                 /* [nv][nu]   */
    const float* cond_cdf;                  /* [nv][nu+1] */
    const float* cond_integral;             /* [nv]       */
    const float* marg_func;                 /* [nv]       */
    const float* marg_cdf;                  /* [nv+1]     */
    float marg_integral; uint32_t nu, nv, _pad;
    float l2w[16], w2l[16];
    /* every 32nd entry of each CDF, rows of (nu + 31) / 32 + 1 (conditional) and (nv + 31) / 32 + 1 (marginal) floats: 135 KB for a
     * 1024^2 map, cache resident -- find_interval looks there first and then inside ONE 32-entry block of the 4 MB table instead of
     * bisecting across it (NULL: a CDF was not monotone, e.g. NaN texels; the plain search is used) */
    const float* cond_coarse; const float* marg_coarse;
    /* Cell records (square maps; NULL: not built): one 128-byte line per cell (u, v) of the map with what every lookup around that cell
     * reads -- its 3 x 3 texel neighbourhood (wrapped like env_texel; 27 floats, texel (dx, dy) at 3 * ((dy + 1) * 3 + dx + 1)) and the
     * distribution's function value of the cell (float 27).  A light sample reads ONE line behind its CDF search (function value +
     * the four texels of Le) instead of three, the pdf of a BSDF-sampled direction and the Le of the same direction one bounce later
     * read the same line: env_cell_* below.  128 MB for a 1024^2 map. */
    const float4* cells;
};

/* one ftn_image with its MIP pyramid (mipmap.rs:78-145): level l is lw[l] x lh[l] float4 texels starting at texels[off[l]] */
struct DImage { uint32_t w, h, wrap, n_levels; uint32_t off[16], lw[16], lh[16]; };

struct DScene {
    const float4* nodes; const float4* geom; const uint4* prim_info;
    /* float4s between two primitives' leaf-test records {p0, flags} {p1, shape index} {p2, -}: 3 = the dense `geom` array; 8 = `geom` is the
     * shading-record array itself (its first 48 bytes are the same three vertices): triangle-only scenes with shading records keep no
     * separate copy -- a leaf test is a random 128-byte line either way (measured: no difference, profiles/r03) and 480 MB stay free */
    uint32_t geom_stride, _pad_gs;
    const float* N; const float* UV;
    const float* T;                         /* per-vertex shading tangents ("S", triangle.rs:341-347) or NULL; gathered through prim_info's vertex indices */
    const DSphere* spheres; const ftn_material* materials; const DLight* lights;
    uint32_t n_nodes, n_prims, n_lights, n_inf_lights, n_spheres, _pad;
    const uint32_t* inf_lights;             /* indices of infinite lights (environment_emitted_radiance sums all lights) */
    /* second view of the same BVH for the any-hit kernel (k_wf_trace_any2): one 64-byte record per INTERIOR node = its two children's
     * node records side by side, {min.x,max.x,min.y,max.y} {min.z,max.z,bits(link),bits(meta)} each; link = byte offset of an interior
     * child's own record / first primitive of a leaf child; meta = n_prims | leaf<<24, record 0 also carries onehot(this node's axis)<<16 */
    const float4* fat;
    float root_lo[3], root_hi[3];
    uint32_t root_is_leaf, n_fat;
    /* textures (NULL / 0 when every material parameter is a constant): see ftn_texture.h */
    const ftn_texture* textures; const ftn_material_textures* mtex; const DImage* images; const float4* texels;
    uint32_t n_textures;
    uint32_t material_types;                /* bit t set: some material has ftn_material.type == t */
    /* scenes lit by ONE light that is an InfiniteAreaLight (the usual environment-lit scene): its record also travels in the kernel
     * arguments, so the shading kernels specialised for it (k_wf_shade<.., ENV>) read it with scalar loads and drop the other kinds */
    uint32_t env_only, _pad2;
    DLight env0;
    /* shading records (NULL: not built): everything make_interaction needs about a primitive in ONE 128-byte line, instead of
     * geom (48 B) + prim_info (2 x 16 B) + three normals and three uvs gathered through the vertex indices (5-9 lines for a hit at a
     * random place of a large scene).  8 float4 per primitive: {p0, flags} {p1, material} {p2, light} {n0, uv0.x} {n1, uv0.y}
     * {n2, uv1.x} {uv1.y, uv2.x, uv2.y, shape index} {unused}.  The traversal kernels keep reading the dense `geom`. */
    const float4* srec;
    /* third view of the same BVH, what the production traversal kernels (ftn_trace4.hip) walk: one 128-byte record per two levels of
     * the tree = the node records of an interior node's four grandchildren side by side (built by build_quads, ftn_host.cpp; NULL: not
     * built).  quad_stack_bound: upper bound of the entries a walk can have pending. */
    const float4* quad;
    uint32_t n_quads, quad_stack_bound;
    /* a second tree over the same leaves for Scene::intersect_test only (build_octs, ftn_host.cpp; ftn_trace8.hip): 128-byte records of up
     * to eight children with outward-rounded 8-bit boxes, exact leaf boxes (oct_xbox: the leaves whose box is not their one triangle's) */
    const uint4* oct; const float4* oct_xbox;
    uint32_t n_octs, oct_stack_bound;
    /* shading class of every primitive (k_wf_classify, ftn_wavefront.hip): 2 + material type, 7 for a primitive without material.  One
     * byte per primitive -- cache resident where prim_info (32 bytes per primitive) is not.  Never NULL for a scene with primitives. */
    const unsigned char* prim_class;
};

struct DCamera {
    float c2w[16]; float r2c[16];
    float shutter_open, shutter_close, lens_radius, focal_dist;
    float dx_camera[3], dy_camera[3];       /* ray differentials only (camera/mod.rs:100-104) */
};

struct DRay { V3 o, d; float t_max, time; };
struct DHit { float t; int prim; float b0, b1, b2; };
struct DSurfHit { V3 p, p_err, n; float time; };
/* the part of SurfaceInteraction the integrator consumes */
struct DSI { DSurfHit hit; V3 wo, shading_n, s_dpdu; int prim; int mat, light; /* material / area light of the primitive, -1 = none */ };
/* the rest of SurfaceInteraction, only needed by textured materials: uv, geometric dpdu/dpdv, shading dndu/dndv */
struct DSIX { V2 uv; V3 dpdu, dpdv, dndu, dndv; };

/* ------------------------------------------------------------------ spawn rays: interaction.rs:22-58 */
__device__ inline DRay spawn_ray(const DSurfHit& h, V3 dir) {
    DRay r; r.o = offset_ray_origin(h.p, h.p_err, h.n, dir); r.d = dir; r.t_max = FTN_INF; r.time = h.time; return r;
}
__device__ inline DRay spawn_ray_to_hit(const DSurfHit& a, const DSurfHit& to) {
    V3 origin = offset_ray_origin(a.p, a.p_err, a.n, to.p - a.p);
    V3 target = offset_ray_origin(to.p, to.p_err, to.n, origin - to.p);
    DRay r; r.o = origin; r.d = target - origin; r.t_max = 1.0f - 0.0001f; r.time = a.time; return r;
}

/* ------------------------------------------------------------------ Bounds3f::intersect_test: bounds.rs:214-233
 * inv = 1/dir is the value the reference recomputes at every node. */
/* the same test on a node record as stored in HBM: a = {min.x, max.x, min.y, max.y}, b = {min.z, max.z, idx, meta} */
__device__ inline bool slab_test(float4 nlo, float4 nhi, V3 o, V3 inv, float t_max);
__device__ inline bool slab_test_node(float4 a, float4 b, V3 o, V3 inv, float t_max) {
    return slab_test(make_float4(a.x, a.z, b.x, 0.0f), make_float4(a.y, a.w, b.y, 0.0f), o, inv, t_max);
}
__device__ inline bool slab_test(float4 nlo, float4 nhi, V3 o, V3 inv, float t_max) {
    /* Branch-free form of the reference's loop with its three early `return None`s.  The running t0 only grows and t1 only
     * shrinks (an update happens only when the comparison is true, which also keeps NaNs from 0 * inf out of t0 / t1 -- exactly
     * what fmaxf / fminf do), so `t0 > t1` after any axis implies `t0 > t1` after the last one: testing once at the end returns
     * the same boolean as the three early exits.  (With the early exits the compiler sinks the loads of the y/z bounds behind the
     * x test: three dependent memory round trips per node instead of one.) */
    const float k = 1.0f + 2.0f * gamma_n(3);
    float tnx = (nlo.x - o.x) * inv.x, tfx = (nhi.x - o.x) * inv.x;
    float tny = (nlo.y - o.y) * inv.y, tfy = (nhi.y - o.y) * inv.y;
    float tnz = (nlo.z - o.z) * inv.z, tfz = (nhi.z - o.z) * inv.z;
    if (tnx > tfx) { float s = tnx; tnx = tfx; tfx = s; }
    if (tny > tfy) { float s = tny; tny = tfy; tfy = s; }
    if (tnz > tfz) { float s = tnz; tnz = tfz; tfz = s; }
    tfx *= k; tfy *= k; tfz *= k;
    const float t0 = fmax_(fmax_(fmax_(0.0f, tnx), tny), tnz);
    const float t1 = fmin_(fmin_(fmin_(t_max, tfx), tfy), tfz);
    return !(t0 > t1);
}
/* same test, also handing back the clipped interval [t0, t1] (the any-hit walk orders the two children by it) */
__device__ inline bool slab_test_node_iv(float4 a, float4 b, V3 o, V3 inv, float t_max, float* t0o, float* t1o) {
    const float k = 1.0f + 2.0f * gamma_n(3);
    float tnx = (a.x - o.x) * inv.x, tfx = (a.y - o.x) * inv.x;
    float tny = (a.z - o.y) * inv.y, tfy = (a.w - o.y) * inv.y;
    float tnz = (b.x - o.z) * inv.z, tfz = (b.y - o.z) * inv.z;
    if (tnx > tfx) { float s = tnx; tnx = tfx; tfx = s; }
    if (tny > tfy) { float s = tny; tny = tfy; tfy = s; }
    if (tnz > tfz) { float s = tnz; tnz = tfz; tfz = s; }
    tfx *= k; tfy *= k; tfz *= k;
    const float t0 = fmax_(fmax_(fmax_(0.0f, tnx), tny), tnz);
    const float t1 = fmin_(fmin_(fmin_(t_max, tfx), tfy), tfz);
    *t0o = t0; *t1o = t1;
    return !(t0 > t1);
}
/* keeps a loaded float4 whole: stops the compiler from splitting / sinking its dword loads behind later branches */
__device__ inline void pin4(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

/* ------------------------------------------------------------------ Triangle::intersect, hit-test part: triangle.rs:176-268 */
__device__ inline bool tri_hit(V3 o, V3 d, float t_max, V3 p0, V3 p1, V3 p2, float* t_out, float* b0o, float* b1o, float* b2o) {
    V3 p0t = p0 - o, p1t = p1 - o, p2t = p2 - o;
    int kz = max_dimension(vabs(d));
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    V3 dir(d.get(kx), d.get(ky), d.get(kz));
    p0t = V3(p0t.get(kx), p0t.get(ky), p0t.get(kz));
    p1t = V3(p1t.get(kx), p1t.get(ky), p1t.get(kz));
    p2t = V3(p2t.get(kx), p2t.get(ky), p2t.get(kz));
    float sx = -dir.x / dir.z, sy = -dir.y / dir.z, sz = 1.0f / dir.z;
    p0t.x += sx * p0t.z; p0t.y += sy * p0t.z;
    p1t.x += sx * p1t.z; p1t.y += sy * p1t.z;
    p2t.x += sx * p2t.z; p2t.y += sy * p2t.z;
    float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
        e0 = (float)((double)p1t.x * (double)p2t.y - (double)p1t.y * (double)p2t.x);
        e1 = (float)((double)p2t.x * (double)p0t.y - (double)p2t.y * (double)p0t.x);
        e2 = (float)((double)p0t.x * (double)p1t.y - (double)p0t.y * (double)p1t.x);
    }
    if (sign_pos(e0) != sign_pos(e1) || sign_pos(e1) != sign_pos(e2)) return false;   /* sign_differs :428-434 */
    float det = e0 + e1 + e2;
    if (det == 0.0f) return false;
    p0t.z *= sz; p1t.z *= sz; p2t.z *= sz;
    float t_scaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if ((det < 0.0f && (t_scaled >= 0.0f || t_scaled < t_max * det)) || (det > 0.0f && (t_scaled <= 0.0f || t_scaled > t_max * det)))
        return false;
    float inv_det = 1.0f / det;
    float b0 = e0 * inv_det, b1 = e1 * inv_det, b2 = e2 * inv_det;
    float t = t_scaled * inv_det;
    float max_zt = fmax_(fmax_(fabsf(p0t.z), fabsf(p1t.z)), fabsf(p2t.z));
    float delta_z = gamma_n(3) * max_zt;
    float max_xt = fmax_(fmax_(fabsf(p0t.x), fabsf(p1t.x)), fabsf(p2t.x));
    float max_yt = fmax_(fmax_(fabsf(p0t.y), fabsf(p1t.y)), fabsf(p2t.y));
    float delta_x = gamma_n(5) * (max_xt + max_zt);
    float delta_y = gamma_n(5) * (max_yt + max_zt);
    float delta_e = 2.0f * (gamma_n(2) * max_xt * max_yt + delta_y * max_xt + delta_x * max_yt);
    float max_e = fmax_(fmax_(fabsf(e0), fabsf(e1)), fabsf(e2));
    float delta_t = 3.0f * (gamma_n(3) * max_e * max_zt + delta_e * max_zt + delta_z * max_e) * fabsf(inv_det);
    if (t <= delta_t) return false;
    *t_out = t; *b0o = b0; *b1o = b1; *b2o = b2;
    return true;
}
/* Per-primitive attributes.  A scene with shading records (DScene::srec, the default) keeps everything below in the primitive's own
 * 128-byte record and does not carry prim_info / N / UV on the device at all (they are the same values gathered through the vertex indices:
 * 560 MB at 10 M triangles); without records (FTN_SREC=0) or for meshes with shading tangents the indexed arrays are resident. */
__device__ inline void prim_mat_light(const DScene& S, int prim, int* mat, int* light) {
    if (S.srec) { const float4* R = S.srec + 8 * (size_t)prim; *mat = (int)__float_as_uint(R[1].w); *light = (int)__float_as_uint(R[2].w); }
    else { const uint4 pi = S.prim_info[2 * prim]; *mat = (int)pi.x; *light = (int)pi.y; }
}
__device__ inline void prim_normals(const DScene& S, int prim, V3* n0, V3* n1, V3* n2) {
    if (S.srec) { const float4* R = S.srec + 8 * (size_t)prim; const float4 r3 = R[3], r4 = R[4], r5 = R[5]; *n0 = V3(r3.x, r3.y, r3.z); *n1 = V3(r4.x, r4.y, r4.z); *n2 = V3(r5.x, r5.y, r5.z); }
    else { const uint4 vi = S.prim_info[2 * prim + 1]; const float* N = S.N;
           *n0 = V3(N[3 * vi.x], N[3 * vi.x + 1], N[3 * vi.x + 2]); *n1 = V3(N[3 * vi.y], N[3 * vi.y + 1], N[3 * vi.y + 2]); *n2 = V3(N[3 * vi.z], N[3 * vi.z + 1], N[3 * vi.z + 2]); }
}
__device__ inline void prim_uvs(const DScene& S, int prim, float* u0x, float* u0y, float* u1x, float* u1y, float* u2x, float* u2y) {
    if (S.srec) { const float4* R = S.srec + 8 * (size_t)prim; const float4 r6 = R[6]; *u0x = R[3].w; *u0y = R[4].w; *u1x = R[5].w; *u1y = r6.x; *u2x = r6.y; *u2y = r6.z; }
    else { const uint4 vi = S.prim_info[2 * prim + 1]; const float* UV = S.UV;
           *u0x = UV[2 * vi.x]; *u0y = UV[2 * vi.x + 1]; *u1x = UV[2 * vi.y]; *u1y = UV[2 * vi.y + 1]; *u2x = UV[2 * vi.z]; *u2y = UV[2 * vi.z + 1]; }
}
/* the late `None` of triangle.rs:279-286: degenerate uvs AND a zero-area triangle */
__device__ inline bool tri_uv_degenerate_reject(const DScene& S, int prim, V3 p0, V3 p1, V3 p2) {
    float u0x, u0y, u1x, u1y, u2x, u2y;
    prim_uvs(S, prim, &u0x, &u0y, &u1x, &u1y, &u2x, &u2y);
    float determinant = (u0x - u2x) * (u1y - u2y) - (u0y - u2y) * (u1x - u2x);
    if (!(fabsf(determinant) < 1.0e-8f)) return false;
    V3 ng = cross(p2 - p0, p1 - p0);
    return len2(ng) == 0.0f;
}

/* ------------------------------------------------------------------ Sphere::intersect: sphere.rs:83-200.
 * Returns the hit distance; fills *si (world space) when si != nullptr. */
__device__ inline bool sphere_clipped(const DSphere& s, V3 p, float phi) {
    return (s.z_min > -s.radius && p.z < s.z_min) || (s.z_max < s.radius && p.z > s.z_max) || phi > s.phi_max;
}
/* FILL = false: hit / miss and t only (si is not looked at).  A template instead of a test of `si` against null: the address of a private
 * variable compared with null keeps the variable -- the caller's whole DSI -- in scratch memory (an alloca with an icmp user is not promoted) */
template <bool FILL = true>
FTN_DEV_NOINLINE bool sphere_intersect(const DSphere& s, const DRay& wr, float* t_out, DSI* si, DSIX* ex = nullptr) {
    V3 o_err, d_err;
    V3 ot = m4_point_exact_to_err(s.w2o, wr.o, &o_err);      /* Ray::tf_exact_to_err transform.rs:287-300 */
    V3 dt_ = m4_vector_exact_to_err(s.w2o, wr.d, &d_err);
    float tmax = wr.t_max;
    float lsq = len2(dt_);
    if (lsq > 0.0f) { float dt = dot(vabs(dt_), o_err) / lsq; ot = ot + dt_ * dt; tmax -= dt; }
    EF ox = ef_err(ot.x, o_err.x), oy = ef_err(ot.y, o_err.y), oz = ef_err(ot.z, o_err.z);
    EF dx = ef_err(dt_.x, d_err.x), dy = ef_err(dt_.y, d_err.y), dz = ef_err(dt_.z, d_err.z);
    EF a = dx * dx + dy * dy + dz * dz;
    EF b = EF(2.0f) * (dx * ox + dy * oy + dz * oz);
    EF c = ox * ox + oy * oy + oz * oz - EF(s.radius) * EF(s.radius);
    EF t0, t1;
    if (!quadratic(a, b, c, &t0, &t1)) return false;
    if (t0.hi > tmax || t1.lo <= 0.0f) return false;
    EF th = t0;
    if (th.lo <= 0.0f) { th = t1; if (th.hi > tmax) return false; }
    V3 p = ot + (dt_ * th.v);
    p = p * (s.radius / len(p - V3(0.0f, 0.0f, 0.0f)));
    if (p.x == 0.0f && p.y == 0.0f) p.x = 1.0e-5f * s.radius;
    float phi = ftn_det::atan2f_det(p.y, p.x);
    if (phi < 0.0f) phi += 2.0f * FTN_PI;
    if (sphere_clipped(s, p, phi)) {
        if (th.v == t1.v) return false;
        if (t1.hi > tmax) return false;
        th = t1;
        p = ot + (dt_ * th.v);
        p = p * (s.radius / len(p - V3(0.0f, 0.0f, 0.0f)));
        if (p.x == 0.0f && p.y == 0.0f) p.x = 1.0e-5f * s.radius;
        phi = ftn_det::atan2f_det(p.y, p.x);
        if (phi < 0.0f) phi += 2.0f * FTN_PI;
        if (sphere_clipped(s, p, phi)) return false;
    }
    *t_out = th.v;
    if (!FILL) return true;
    float theta = ftn_det::acosf_det(clampf(p.z / s.radius, -1.0f, 1.0f));
    float z_radius = sqrtf(p.x * p.x + p.y * p.y);
    float inv_zr = 1.0f / z_radius;
    float cos_phi = p.x * inv_zr, sin_phi = p.y * inv_zr;
    V3 dpdu(-s.phi_max * p.y, s.phi_max * p.x, 0.0f);
    V3 dpdv = (s.theta_max - s.theta_min) * V3(p.z * cos_phi, p.z * sin_phi, -s.radius * ftn_det::sinf_det(theta));
    V3 n = normalize(cross(dpdu, dpdv));
    V3 p_err = gamma_n(5) * vabs(p);
    if (s.reverse_orientation) n = n * -1.0f;
    /* SurfaceInteraction::transform(object_to_world): transform.rs:340-346, 369-385 */
    si->hit.p = m4_point_err_to_err(s.o2w, p, p_err, &si->hit.p_err);
    si->hit.n = normalize(m4_normal(s.o2w_inv, n));
    si->hit.time = wr.time;
    si->wo = normalize(m4_vector(s.o2w, -dt_));
    si->shading_n = normalize(m4_normal(s.o2w_inv, n));
    si->s_dpdu = m4_vector(s.o2w, dpdu);
    if (ex) {                                                   /* sphere.rs:131-160: uv, second derivatives -> dndu/dndv (Weingarten) */
        ex->uv = V2(phi / s.phi_max, (theta - s.theta_min) / (s.theta_max - s.theta_min));
        const V3 d2pduu = (-s.phi_max * s.phi_max) * V3(p.x, p.y, 0.0f);
        const V3 d2pduv = (s.theta_max - s.theta_min) * p.z * s.phi_max * V3(-sin_phi, cos_phi, 0.0f);
        const V3 d2pdvv = -(s.theta_max - s.theta_min) * (s.theta_max - s.theta_min) * V3(p.x, p.y, p.z);
        const float E = dot(dpdu, dpdu), F = dot(dpdu, dpdv), G = dot(dpdv, dpdv);
        const V3 N = normalize(cross(dpdu, dpdv));
        const float e = dot(N, d2pduu), f = dot(N, d2pduv), g = dot(N, d2pdvv);
        const float invEGF2 = 1.0f / (E * G - F * F);
        const V3 dndu = (f * F - e * G) * invEGF2 * dpdu + (e * F - f * E) * invEGF2 * dpdv;
        const V3 dndv = (g * F - f * G) * invEGF2 * dpdu + (f * F - g * E) * invEGF2 * dpdv;
        ex->dpdu = m4_vector(s.o2w, dpdu); ex->dpdv = m4_vector(s.o2w, dpdv);
        ex->dndu = m4_normal(s.o2w_inv, dndu); ex->dndv = m4_normal(s.o2w_inv, dndv);
    }
    return true;
}

/* ------------------------------------------------------------------ BVH traversal: bvh.rs:160-266
 * Stack: one uint32 per level per lane, in LDS, lane-interleaved: slot(sp) = base[sp * stride]. */
struct LdsStack {
    uint32_t* base; uint32_t stride;
    __device__ void push(int sp, uint32_t v) const { base[sp * stride] = v; }
    __device__ uint32_t pop(int sp) const { return base[sp * stride]; }
};
struct TravCount { uint32_t nodes, prims; };

template <bool ANY, bool COUNT>
__device__ inline bool traverse(const DScene& S, DRay& ray, const LdsStack& st, DHit* hit, TravCount* tc) {
    if (S.n_nodes == 0) return false;
    const V3 inv(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
    const uint32_t neg = (ray.d.x < 0.0f ? 1u : 0u) | (ray.d.y < 0.0f ? 2u : 0u) | (ray.d.z < 0.0f ? 4u : 0u);
    int sp = 0; uint32_t cur = 0; bool found = false;                    /* cur: byte offset of the node record */
    const uint32_t neg16 = neg << 16;
    for (;;) {
        const float4* rec = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.nodes) + cur);
        float4 na = rec[0], nb = rec[1];
        pin4(na); pin4(nb);
        if (COUNT) tc->nodes++;
        bool descend = false;
        if (slab_test_node(na, nb, ray.o, inv, ray.t_max)) {
            const uint32_t idx = __float_as_uint(nb.z), meta = __float_as_uint(nb.w);
            if (meta >> 24) {   /* leaf */
                const uint32_t n = meta & 0xffffu;
                for (uint32_t i = 0; i < n; i++) {
                    const uint32_t prim = idx + i;
                    float4 g0 = S.geom[S.geom_stride * prim], g1 = S.geom[S.geom_stride * prim + 1], g2 = S.geom[S.geom_stride * prim + 2];
                    pin4(g0); pin4(g1); pin4(g2);
                    if (COUNT) tc->prims++;
                    const uint32_t fl = __float_as_uint(g0.w);
                    float t, b0 = 0.0f, b1 = 0.0f, b2 = 0.0f; bool h;
                    if (fl & GF_KIND_SPHERE) h = sphere_intersect<false>(S.spheres[__float_as_uint(g1.w)], ray, &t, nullptr);
                    else {
                        V3 p0(g0.x, g0.y, g0.z), p1(g1.x, g1.y, g1.z), p2(g2.x, g2.y, g2.z);
                        h = tri_hit(ray.o, ray.d, ray.t_max, p0, p1, p2, &t, &b0, &b1, &b2);
                        if (h && (fl & GF_HAS_UVS) && tri_uv_degenerate_reject(S, (int)prim, p0, p1, p2)) h = false;
                    }
                    if (h) {
                        if (ANY) return true;
                        ray.t_max = t; found = true;
                        hit->t = t; hit->prim = (int)prim; hit->b0 = b0; hit->b1 = b1; hit->b2 = b2;
                    }
                }
            } else {
                if (meta & neg16) { st.push(sp++, cur + 32u); cur = idx; }       /* dir_is_neg[axis]: second child first */
                else { st.push(sp++, idx); cur = cur + 32u; }
                descend = true;
            }
        }
        if (!descend) {
            if (sp == 0) break;
            cur = st.pop(--sp);
        }
    }
    return found;
}

/* ------------------------------------------------------------------ shading geometry from a compact hit */
__device__ inline void load_tri(const DScene& S, int prim, V3* p0, V3* p1, V3* p2, uint32_t* flags) {
    const float4 g0 = S.geom[S.geom_stride * prim], g1 = S.geom[S.geom_stride * prim + 1], g2 = S.geom[S.geom_stride * prim + 2];
    *p0 = V3(g0.x, g0.y, g0.z); *p1 = V3(g1.x, g1.y, g1.z); *p2 = V3(g2.x, g2.y, g2.z); *flags = __float_as_uint(g0.w);
}
/* triangle.rs:270-393 */
/* the barycentric 1/det of the hit test (tri_hit above), which the reference's dndu/dndv use by accident (triangle.rs:361-363) */
__device__ inline float tri_inv_det(V3 o, V3 d, V3 p0, V3 p1, V3 p2) {
    V3 p0t = p0 - o, p1t = p1 - o, p2t = p2 - o;
    int kz = max_dimension(vabs(d));
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    V3 dir(d.get(kx), d.get(ky), d.get(kz));
    p0t = V3(p0t.get(kx), p0t.get(ky), p0t.get(kz)); p1t = V3(p1t.get(kx), p1t.get(ky), p1t.get(kz)); p2t = V3(p2t.get(kx), p2t.get(ky), p2t.get(kz));
    float sx = -dir.x / dir.z, sy = -dir.y / dir.z;
    p0t.x += sx * p0t.z; p0t.y += sy * p0t.z; p1t.x += sx * p1t.z; p1t.y += sy * p1t.z; p2t.x += sx * p2t.z; p2t.y += sy * p2t.z;
    float e0 = p1t.x * p2t.y - p1t.y * p2t.x, e1 = p2t.x * p0t.y - p2t.y * p0t.x, e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
        e0 = (float)((double)p1t.x * (double)p2t.y - (double)p1t.y * (double)p2t.x);
        e1 = (float)((double)p2t.x * (double)p0t.y - (double)p2t.y * (double)p0t.x);
        e2 = (float)((double)p0t.x * (double)p1t.y - (double)p0t.y * (double)p1t.x);
    }
    return 1.0f / (e0 + e1 + e2);
}
FTN_DEV_NOINLINE void tri_interaction(const DScene& S, const DHit& h, V3 ray_d, float time, DSI* si, DSIX* ex = nullptr, V3 ray_o = V3(0.0f, 0.0f, 0.0f)) {
    V3 p0, p1, p2, n0, n1, n2; uint32_t fl;
    const float b0 = h.b0, b1 = h.b1, b2 = h.b2;
    float u0x = 0.0f, u0y = 0.0f, u1x = 1.0f, u1y = 0.0f, u2x = 1.0f, u2y = 1.0f;
    if (S.srec) {                                                /* one 128-byte shading record (DScene::srec) */
        const float4* R = S.srec + 8 * (size_t)h.prim;
        const float4 r0 = R[0], r1 = R[1], r2 = R[2];
        p0 = V3(r0.x, r0.y, r0.z); p1 = V3(r1.x, r1.y, r1.z); p2 = V3(r2.x, r2.y, r2.z); fl = __float_as_uint(r0.w);
        si->mat = (int)__float_as_uint(r1.w); si->light = (int)__float_as_uint(r2.w);
        if (fl & (GF_HAS_NORMALS | GF_HAS_UVS)) {
            const float4 r3 = R[3], r4 = R[4], r5 = R[5];
            n0 = V3(r3.x, r3.y, r3.z); n1 = V3(r4.x, r4.y, r4.z); n2 = V3(r5.x, r5.y, r5.z);
            if (fl & GF_HAS_UVS) { const float4 r6 = R[6]; u0x = r3.w; u0y = r4.w; u1x = r5.w; u1y = r6.x; u2x = r6.y; u2y = r6.z; }
        }
    } else {
        load_tri(S, h.prim, &p0, &p1, &p2, &fl);
        const uint4 vi = S.prim_info[2 * h.prim + 1], pi = S.prim_info[2 * h.prim];
        si->mat = (int)pi.x; si->light = (int)pi.y;
        if (fl & GF_HAS_UVS) { const float* UV = S.UV; u0x = UV[2 * vi.x]; u0y = UV[2 * vi.x + 1]; u1x = UV[2 * vi.y]; u1y = UV[2 * vi.y + 1]; u2x = UV[2 * vi.z]; u2y = UV[2 * vi.z + 1]; }
        if (fl & GF_HAS_NORMALS) { const float* N = S.N; n0 = V3(N[3 * vi.x], N[3 * vi.x + 1], N[3 * vi.x + 2]); n1 = V3(N[3 * vi.y], N[3 * vi.y + 1], N[3 * vi.y + 2]); n2 = V3(N[3 * vi.z], N[3 * vi.z + 1], N[3 * vi.z + 2]); }
    }
    float d02x = u0x - u2x, d02y = u0y - u2y, d12x = u1x - u2x, d12y = u1y - u2y;
    V3 dp02 = p0 - p2, dp12 = p1 - p2;
    float determinant = d02x * d12y - d02y * d12x;
    V3 dpdu, dpdv;
    const bool degenerate_uv = fabsf(determinant) < 1.0e-8f;
    if (degenerate_uv) {
        V3 ng = cross(p2 - p0, p1 - p0);
        coordinate_system(normalize(ng), &dpdu, &dpdv);
    } else {
        float inv = 1.0f / determinant;
        dpdu = (d12y * dp02 - d02y * dp12) * inv;
        if (ex) dpdv = (-d12x * dp02 + d02x * dp12) * inv;
    }
    if (ex) {
        ex->uv = V2((b0 * u0x + b1 * u1x) + b2 * u2x, (b0 * u0y + b1 * u1y) + b2 * u2y);
        ex->dpdu = dpdu; ex->dpdv = dpdv; ex->dndu = V3(0.0f, 0.0f, 0.0f); ex->dndv = V3(0.0f, 0.0f, 0.0f);
    }
    float xs = fabsf(b0 * p0.x) + fabsf(b1 * p1.x) + fabsf(b2 * p2.x);
    float ys = fabsf(b0 * p0.y) + fabsf(b1 * p1.y) + fabsf(b2 * p2.y);
    float zs = fabsf(b0 * p0.z) + fabsf(b1 * p1.z) + fabsf(b2 * p2.z);
    si->hit.p_err = gamma_n(7) * V3(xs, ys, zs);
    si->hit.p = b0 * p0 + b1 * p1 + b2 * p2;
    si->hit.time = time;
    si->wo = -ray_d;
    V3 n = normalize(cross(dp02, dp12));
    V3 sn = n;
    if (fl & GF_FLIP) { n = n * -1.0f; sn = sn * -1.0f; }
    si->s_dpdu = dpdu;
    if (fl & (GF_HAS_NORMALS | GF_HAS_TANGENTS)) {               /* triangle.rs:332-391 */
        V3 ns = n;                                               /* no normals: the (already flipped) geometric normal, triangle.rs:337 */
        if (fl & GF_HAS_NORMALS) ns = normalize(b0 * n0 + b1 * n1 + b2 * n2);
        V3 ss;
        if (fl & GF_HAS_TANGENTS) {                              /* the mesh's interpolated tangent instead of dpdu, triangle.rs:341-342 */
            const uint4 vi = S.prim_info[2 * h.prim + 1]; const float* T = S.T;
            const V3 t0(T[3 * vi.x], T[3 * vi.x + 1], T[3 * vi.x + 2]), t1(T[3 * vi.y], T[3 * vi.y + 1], T[3 * vi.y + 2]), t2(T[3 * vi.z], T[3 * vi.z + 1], T[3 * vi.z + 2]);
            ss = normalize(b0 * t0 + b1 * t1 + b2 * t2);
        } else ss = normalize(dpdu);
        V3 ts = cross(ns, ss);
        if (len2(ts) > 0.0f) { ts = normalize(ts); ss = cross(ts, ns); }
        else coordinate_system(ns, &ts, &ss);      /* (v2, v3) bound as (ts, ss): triangle.rs:343-349 */
        si->s_dpdu = ss;
        sn = ns;
        n = faceforward(n, sn);
        if (ex && (fl & GF_HAS_NORMALS)) {                       /* triangle.rs:351-366 (a mesh with tangents only keeps dndu = dndv = 0, :367-369) */
            const V3 dn1 = n0 - n2, dn2 = n1 - n2;
            if (degenerate_uv) {
                const V3 dn = cross(n2 - n0, n1 - n0);
                if (len2(dn) == 0.0f) { ex->dndu = V3(0.0f, 0.0f, 0.0f); ex->dndv = V3(0.0f, 0.0f, 0.0f); }
                else coordinate_system(dn, &ex->dndu, &ex->dndv);
            } else {
                const float inv_det = tri_inv_det(ray_o, ray_d, p0, p1, p2);     /* sic: the barycentric determinant */
                ex->dndu = (d12y * dn1 - d02y * dn2) * inv_det;
                ex->dndv = (-d12x * dn1 + d02x * dn2) * inv_det;
            }
        }
    }
    si->hit.n = n; si->shading_n = sn; si->prim = h.prim;
}
__device__ inline bool make_interaction(const DScene& S, const DHit& h, const DRay& ray_before_hit, DSI* si, DSIX* ex = nullptr) {
    const float4 g0 = S.srec ? S.srec[8 * (size_t)h.prim] : S.geom[S.geom_stride * h.prim];
    if (__float_as_uint(g0.w) & GF_KIND_SPHERE) {
        DRay r = ray_before_hit; r.t_max = FTN_INF;   /* same root selection as at traversal time */
        float t; const float4 g1 = S.geom[S.geom_stride * h.prim + 1];
        bool ok = sphere_intersect(S.spheres[__float_as_uint(g1.w)], r, &t, si, ex);
        si->prim = h.prim; prim_mat_light(S, h.prim, &si->mat, &si->light);
        return ok;
    }
    tri_interaction(S, h, ray_before_hit.d, ray_before_hit.time, si, ex, ray_before_hit.o);
    return true;
}

/* ------------------------------------------------------------------ shapes as emitters: shapes/mod.rs:39-66 */
FTN_DEV_NOINLINE DSurfHit shape_sample(const DScene& S, int prim, V2 u) {
    const float4 g0 = S.geom[S.geom_stride * prim], g1 = S.geom[S.geom_stride * prim + 1], g2 = S.geom[S.geom_stride * prim + 2];
    const uint32_t fl = __float_as_uint(g0.w);
    DSurfHit h;
    if (fl & GF_KIND_SPHERE) {                                   /* sphere.rs:202-218 */
        const DSphere& s = S.spheres[__float_as_uint(g1.w)];
        V3 p_obj = V3(0.0f, 0.0f, 0.0f) + s.radius * uniform_sample_sphere(u);
        V3 n = normalize(m4_normal(s.o2w_inv, p_obj));
        if (s.reverse_orientation) n = n * -1.0f;
        p_obj = p_obj * (s.radius / len(p_obj - V3(0.0f, 0.0f, 0.0f)));
        V3 pe = gamma_n(5) * vabs(p_obj);
        h.p = m4_point_err_to_err(s.o2w, p_obj, pe, &h.p_err);
        h.time = 0.0f; h.n = n;
        return h;
    }
    V2 b = uniform_sample_triangle(u);                           /* triangle.rs:395-420 */
    V3 p0(g0.x, g0.y, g0.z), p1(g1.x, g1.y, g1.z), p2(g2.x, g2.y, g2.z);
    float bz = 1.0f - b.x - b.y;
    V3 sp = b.x * p0 + b.y * p1 + bz * p2;
    V3 n = normalize(cross(p1 - p0, p2 - p0));
    V3 sn;
    if (fl & GF_HAS_NORMALS) {
        V3 n0, n1, n2; prim_normals(S, prim, &n0, &n1, &n2);
        V3 ns = normalize(b.x * n0 + b.y * n1 + bz * n2);
        sn = faceforward(n, ns);
    } else if (fl & GF_FLIP) sn = n * -1.0f;
    else sn = n;
    V3 pas = vabs(b.x * p0) + vabs(b.y * p1) + vabs(bz * p2);
    h.p = V3(0.0f, 0.0f, 0.0f) + sp; h.p_err = gamma_n(6) * pas; h.time = 0.0f; h.n = sn;
    return h;
}
/* Shape::pdf_from_ref: intersects the light's own shape, bypassing the BVH (shapes/mod.rs:55-66) */
FTN_DEV_NOINLINE float shape_pdf_from_ref(const DScene& S, int prim, float area, const DSurfHit& ref, V3 wi) {
    DRay ray = spawn_ray(ref, wi);
    const float4 g0 = S.geom[S.geom_stride * prim];
    V3 hp, hn;
    if (__float_as_uint(g0.w) & GF_KIND_SPHERE) {
        DSI si; float t; const float4 g1 = S.geom[S.geom_stride * prim + 1];
        if (!sphere_intersect(S.spheres[__float_as_uint(g1.w)], ray, &t, &si)) return 0.0f;
        hp = si.hit.p; hn = si.hit.n;
    } else {
        V3 p0, p1, p2; uint32_t fl; load_tri(S, prim, &p0, &p1, &p2, &fl);
        DHit h; h.prim = prim;
        if (!tri_hit(ray.o, ray.d, ray.t_max, p0, p1, p2, &h.t, &h.b0, &h.b1, &h.b2)) return 0.0f;
        if ((fl & GF_HAS_UVS) && tri_uv_degenerate_reject(S, prim, p0, p1, p2)) return 0.0f;
        DSI si; tri_interaction(S, h, ray.d, ray.time, &si);
        hp = si.hit.p; hn = si.hit.n;
    }
    return len2(ref.p - hp) / (abs_dot(hn, -wi) * area);
}

/* ------------------------------------------------------------------ BSDF: reflection/{mod,bsdf,microfacet}.rs, fresnel.rs */
enum : uint32_t { T_REFL = 1, T_TRANS = 2, T_DIFFUSE = 4, T_GLOSSY = 8, T_SPECULAR = 16, T_ALL = 31 };
enum : uint32_t { BX_LAMBERT = 0, BX_OREN = 1, BX_SPEC_R = 2, BX_SPEC_T = 3, BX_MF_R = 4, BX_MF_T = 5 };
enum : uint32_t { FR_NOOP = 0, FR_DIEL = 1, FR_COND = 2 };

struct DLobe {
    uint32_t kind, fresnel; Rgb r; float p0, p1;   /* OrenNayar a,b | alpha_x, alpha_y */
    float ei, et;                                  /* dielectric eta_i, eta_t (= eta_a, eta_b for transmission) */
    const ftn_material* m;                         /* conductor eta (a) / k (b) */
};
struct DBsdf { V3 ns, ng, ss, ts; int n; DLobe lobe[2]; };

__device__ inline uint32_t lobe_type(uint32_t k) {
    switch (k) { case BX_LAMBERT: case BX_OREN: return T_REFL | T_DIFFUSE; case BX_SPEC_R: return T_REFL | T_SPECULAR;
                 case BX_SPEC_T: return T_TRANS | T_SPECULAR; case BX_MF_R: return T_REFL | T_GLOSSY; default: return T_TRANS | T_GLOSSY; }
}
__device__ inline bool lobe_matches(uint32_t k, uint32_t flags) { uint32_t t = lobe_type(k); return (flags & t) == t; }

__device__ inline float cos2_theta(V3 w) { return w.z * w.z; }
__device__ inline float sin2_theta(V3 w) { return fmax_(0.0f, 1.0f - cos2_theta(w)); }
__device__ inline float sin_theta(V3 w) { return sqrtf(sin2_theta(w)); }
__device__ inline float tan_theta(V3 w) { return sin_theta(w) / w.z; }
__device__ inline float tan2_theta(V3 w) { return sin2_theta(w) / cos2_theta(w); }
__device__ inline float cos_phi(V3 w) { float s = sin_theta(w); return s == 0.0f ? 1.0f : clampf(w.x / s, -1.0f, 1.0f); }
__device__ inline float sin_phi(V3 w) { float s = sin_theta(w); return s == 0.0f ? 0.0f : clampf(w.y / s, -1.0f, 1.0f); }
__device__ inline bool same_hemisphere(V3 a, V3 b) { return sign_pos(a.z) == sign_pos(b.z); }

__device__ inline float fresnel_dielectric(float ci, float eta_i, float eta_t) {   /* fresnel.rs:4-22 */
    ci = clampf(ci, -1.0f, 1.0f);
    if (!(ci > 0.0f)) { float s = eta_i; eta_i = eta_t; eta_t = s; ci = fabsf(ci); }
    float si = sqrtf(fmax_(1.0f - ci * ci, 0.0f));
    float st = eta_i / eta_t * si;
    if (st >= 1.0f) return 1.0f;
    float ct = sqrtf(fmax_(1.0f - st * st, 0.0f));
    float rpar = ((eta_t * ci) - (eta_i * ct)) / ((eta_t * ci) + (eta_i * ct));
    float rper = ((eta_i * ci) - (eta_t * ct)) / ((eta_i * ci) + (eta_t * ct));
    return (rpar * rpar + rper * rper) / 2.0f;
}
FTN_DEV_NOINLINE Rgb fresnel_conductor(float ci, Rgb eta_i, Rgb eta_t, Rgb k) {   /* fresnel.rs:25-48 */
    ci = clampf(ci, -1.0f, 1.0f);
    Rgb eta = eta_t / eta_i, eta_k = k / eta_i;
    float c2 = ci * ci, s2 = 1.0f - c2;
    Rgb eta2 = eta * eta, etak2 = eta_k * eta_k;
    Rgb t0 = eta2 - etak2 - s2;
    Rgb a2b2 = rgb_sqrt(t0 * t0 + 4.0f * eta2 * etak2);
    Rgb t1 = a2b2 + c2;
    Rgb a = rgb_sqrt(0.5f * (a2b2 + t0));
    Rgb t2 = 2.0f * ci * a;
    Rgb Rs = (t1 - t2) / (t1 + t2);
    Rgb t3 = c2 * a2b2 + s2 * s2;
    Rgb t4 = t2 * s2;
    Rgb Rp = Rs * (t3 - t4) / (t3 + t4);
    return 0.5f * (Rp + Rs);
}
__device__ inline Rgb lobe_fresnel(const DLobe& L, float cos_i) {
    if (L.fresnel == FR_DIEL) return Rgb(fresnel_dielectric(cos_i, L.ei, L.et));
    if (L.fresnel == FR_COND) return fresnel_conductor(fabsf(cos_i), Rgb(1.0f), Rgb(L.m->a[0], L.m->a[1], L.m->a[2]), Rgb(L.m->b[0], L.m->b[1], L.m->b[2]));
    return Rgb(1.0f);
}
/* TrowbridgeReitzDistribution: microfacet.rs:119-187 */
__device__ inline float tr_d(float ax, float ay, V3 wh) {
    float t2 = tan2_theta(wh);
    if (is_inf(t2)) return 0.0f;
    float c4 = cos2_theta(wh) * cos2_theta(wh);
    float e = ((cos_phi(wh) * cos_phi(wh)) / (ax * ax) + (sin_phi(wh) * sin_phi(wh)) / (ay * ay)) * t2;
    return 1.0f / (FTN_PI * ax * ay * c4 * (1.0f + e) * (1.0f + e));
}
__device__ inline float tr_lambda(float ax, float ay, V3 w) {
    float att = fabsf(tan_theta(w));
    if (is_inf(att)) return 0.0f;
    float alpha = sqrtf((cos_phi(w) * cos_phi(w)) * ax * ax + (sin_phi(w) * sin_phi(w)) * ay * ay);
    float a2t2 = (alpha * att) * (alpha * att);
    return (-1.0f + sqrtf(1.0f + a2t2)) / 2.0f;
}
__device__ inline float tr_g(float ax, float ay, V3 wo, V3 wi) { return 1.0f / (1.0f + tr_lambda(ax, ay, wo) + tr_lambda(ax, ay, wi)); }
__device__ inline float tr_pdf(float ax, float ay, V3 wh) { return tr_d(ax, ay, wh) * fabsf(wh.z); }
__device__ inline V3 tr_sample_wh(float ax, float ay, V3 wo, V2 u) {
    float ct, phi;
    if (ax == ay) {
        float tt2 = (ax * ax) * u.x / (1.0f - u.x);
        ct = 1.0f / sqrtf(1.0f + tt2);
        phi = 2.0f * FTN_PI * u.y;
    } else {
        phi = ftn_det::atanf_det(ay / ax * ftn_det::tanf_det(2.0f * FTN_PI * u.y + 0.5f * FTN_PI));
        if (u.y > 0.5f) phi += FTN_PI;
        float sp, cp; ftn_det::sincosf_det(phi, &sp, &cp);
        float alpha2 = 1.0f / ((cp * cp) / (ax * ax) + (sp * sp) / (ay * ay));
        float tt2 = alpha2 * u.x / (1.0f - u.x);
        ct = 1.0f / sqrtf(1.0f + tt2);
    }
    float st = sqrtf(fmax_(0.0f, 1.0f - (ct * ct)));
    V3 wh = spherical_direction(st, ct, phi);
    return same_hemisphere(wo, wh) ? wh : -wh;
}
__device__ inline bool refract(V3 wi, V3 n, float eta, V3* wt) {                  /* reflection/mod.rs:75-83 */
    float ci = dot(n, wi);
    float s2i = fmax_(0.0f, 1.0f - ci * ci);
    float s2t = eta * eta * s2i;
    if (s2t >= 1.0f) return false;
    float ct = sqrtf(1.0f - s2t);
    *wt = eta * -wi + (eta * ci - ct) * n;
    return true;
}
__device__ inline float mt_eta(const DLobe& L, V3 wo) { return wo.z > 0.0f ? L.et / L.ei : L.ei / L.et; }   /* get_eta :378-380 */

FTN_DEV_NOINLINE Rgb lobe_f(const DLobe& L, V3 wo, V3 wi) {
    switch (L.kind) {
        case BX_LAMBERT: return L.r * FTN_INV_PI;
        case BX_OREN: {
            float sti = sin_theta(wi), sto = sin_theta(wo), max_cos = 0.0f;
            if (sti > 1.0e-4f && sto > 1.0e-4f) { float dc = cos_phi(wi) * cos_phi(wo) + sin_phi(wi) * sin_phi(wo); max_cos = fmax_(0.0f, dc); }
            float sa, tb;
            if (fabsf(wi.z) > fabsf(wo.z)) { sa = sto; tb = sti / fabsf(wi.z); } else { sa = sti; tb = sto / fabsf(wo.z); }
            return L.r * FTN_INV_PI * (L.p0 + (L.p1 * max_cos * sa * tb));
        }
        case BX_SPEC_R: case BX_SPEC_T: return Rgb(0.0f);
        case BX_MF_R: {
            float co = fabsf(wo.z), ci = fabsf(wi.z);
            V3 wh = wi + wo;
            if (ci == 0.0f || co == 0.0f || veq(wh, V3(0.0f, 0.0f, 0.0f))) return Rgb(0.0f);
            wh = normalize(wh);
            Rgb fr = lobe_fresnel(L, dot(wi, faceforward(wh, V3(0.0f, 0.0f, 1.0f))));
            return L.r * tr_d(L.p0, L.p1, wh) * tr_g(L.p0, L.p1, wo, wi) * fr / (4.0f * ci * co);
        }
        default: {
            if (same_hemisphere(wo, wi)) return Rgb(0.0f);
            float co = wo.z, ci = wi.z;
            if (co == 0.0f || ci == 0.0f) return Rgb(0.0f);
            float eta = mt_eta(L, wo);
            V3 wh = normalize(wo + wi * eta);
            if (wh.z < 0.0f) wh = -wh;
            Rgb fr = Rgb(fresnel_dielectric(dot(wo, wh), L.ei, L.et));
            float sd = dot(wo, wh) + eta * dot(wi, wh);
            float factor = 1.0f / eta;
            return (Rgb(1.0f) - fr) * L.r *
                   fabsf(tr_d(L.p0, L.p1, wh) * tr_g(L.p0, L.p1, wo, wi) * (eta * eta) * abs_dot(wi, wh) * abs_dot(wo, wh) * (factor * factor) / (ci * co * (sd * sd)));
        }
    }
}
FTN_DEV_NOINLINE float lobe_pdf(const DLobe& L, V3 wo, V3 wi) {
    switch (L.kind) {
        case BX_LAMBERT: case BX_OREN: return same_hemisphere(wo, wi) ? fabsf(wi.z) * FTN_INV_PI : 0.0f;
        case BX_SPEC_R: case BX_SPEC_T: return 0.0f;
        case BX_MF_R: {
            if (!same_hemisphere(wo, wi)) return 0.0f;
            V3 wh = normalize(wo + wi);
            return tr_pdf(L.p0, L.p1, wh) / (4.0f * dot(wo, wh));
        }
        default: {
            if (same_hemisphere(wo, wi)) return 0.0f;
            float eta = mt_eta(L, wo);
            V3 wh = normalize(wo + wi * eta);
            float sd = dot(wo, wh) + eta * dot(wi, wh);
            float dwh = fabsf(((eta * eta) * dot(wi, wh)) / (sd * sd));
            return tr_pdf(L.p0, L.p1, wh) * dwh;
        }
    }
}
struct DScatter { Rgb f; V3 wi; float pdf; uint32_t type; };
FTN_DEV_NOINLINE bool lobe_sample(const DLobe& L, V3 wo, V2 u, DScatter* o) {
    o->type = lobe_type(L.kind);
    switch (L.kind) {
        case BX_LAMBERT: case BX_OREN: {
            V3 wi = cosine_sample_hemisphere(u);
            if (wo.z < 0.0f) wi.z *= -1.0f;
            o->pdf = lobe_pdf(L, wo, wi); o->f = lobe_f(L, wo, wi); o->wi = wi;
            return true;
        }
        case BX_SPEC_R: {
            V3 wi(-wo.x, -wo.y, wo.z);
            o->f = lobe_fresnel(L, wi.z) * L.r / fabsf(wi.z); o->wi = wi; o->pdf = 1.0f;
            return true;
        }
        case BX_SPEC_T: {
            bool entering = wo.z > 0.0f;
            float ei = entering ? L.ei : L.et, et = entering ? L.et : L.ei;
            V3 n(0.0f, 0.0f, 1.0f); if (dot(n, wo) < 0.0f) n = -n;
            V3 wi;
            if (!refract(wo, n, ei / et, &wi)) return false;
            Rgb ft = L.r * (Rgb(1.0f) - Rgb(fresnel_dielectric(wi.z, L.ei, L.et)));
            o->f = ft / fabsf(wi.z); o->wi = wi; o->pdf = 1.0f;
            return true;
        }
        case BX_MF_R: {
            V3 wh = tr_sample_wh(L.p0, L.p1, wo, u);
            V3 wi = -wo + 2.0f * dot(wo, wh) * wh;   /* reflect :85-87 */
            if (!same_hemisphere(wo, wi)) return false;
            o->pdf = tr_pdf(L.p0, L.p1, wh) / (4.0f * dot(wo, wh));
            o->f = lobe_f(L, wo, wi); o->wi = wi;
            return true;
        }
        default: {
            if (wo.z == 0.0f) return false;
            V3 wh = tr_sample_wh(L.p0, L.p1, wo, u);
            if (dot(wo, wh) < 0.0f) return false;
            float eta = mt_eta(L, -wo);
            V3 wi;
            if (!refract(wo, wh, eta, &wi)) return false;
            o->f = lobe_f(L, wo, wi); o->wi = wi; o->pdf = lobe_pdf(L, wo, wi);
            return true;
        }
    }
}

/* Bsdf::new + Material::compute_scattering_functions (src/material/ *.rs); roughness already mapped to alpha on the host.
 * Returns false for the configuration the reference panics on (specular glass with allow_multiple_lobes). */
__device__ inline void add_lobe(DBsdf* B, const DLobe& L) { if (B->n == 0) B->lobe[0] = L; else B->lobe[1] = L; B->n++; }   /* static indices: stays in VGPRs */
__device__ inline DLobe mk_lobe(uint32_t kind, Rgb r, uint32_t fresnel, float p0, float p1, float ei, float et, const ftn_material* m) {
    DLobe L; L.kind = kind; L.fresnel = fresnel; L.r = r; L.p0 = p0; L.p1 = p1; L.ei = ei; L.et = et; L.m = m; return L;
}
/* MT >= 0: the material type is known at compile time (material-specialised shade kernels): the other cases fold away */
template <int MT = -1>
__device__ inline bool make_bsdf(const ftn_material& m, const DSI& si, bool allow_multiple_lobes, DBsdf* B) {
    B->ns = si.shading_n; B->ng = si.hit.n;
    B->ss = normalize(si.s_dpdu);
    B->ts = normalize(cross(B->ns, B->ss));
    B->n = 0;
    B->lobe[0] = mk_lobe(BX_LAMBERT, Rgb(0.0f), FR_NOOP, 0.0f, 0.0f, 1.0f, 1.0f, &m); B->lobe[1] = B->lobe[0];
    const Rgb a(m.a[0], m.a[1], m.a[2]), b(m.b[0], m.b[1], m.b[2]);
    switch (MT >= 0 ? (uint32_t)MT : m.type) {
        case FTN_MAT_MATTE: {                                  /* matte.rs:35-52; OrenNayar A/B precomputed in s1/s2 */
            Rgb r = clamp_positive(a);
            if (!r.is_black()) add_lobe(B, mk_lobe(m.s0 == 0.0f ? BX_LAMBERT : BX_OREN, r, FR_NOOP, m.s1, m.s2, 1.0f, 1.0f, &m));
            return true;
        }
        case FTN_MAT_METAL:                                     /* metal.rs:37-65 */
            add_lobe(B, mk_lobe(BX_MF_R, Rgb(1.0f), FR_COND, m.s1, m.s2, 1.0f, 1.0f, &m));
            return true;
        case FTN_MAT_MIRROR: {                                  /* mirror.rs:21-30 */
            Rgb r = clamp_positive(a);
            if (!r.is_black()) add_lobe(B, mk_lobe(BX_SPEC_R, r, FR_NOOP, 0.0f, 0.0f, 1.0f, 1.0f, &m));
            return true;
        }
        case FTN_MAT_PLASTIC:                                   /* plastic.rs:24-48 */
            if (!a.is_black()) add_lobe(B, mk_lobe(BX_LAMBERT, a, FR_NOOP, 0.0f, 0.0f, 1.0f, 1.0f, &m));
            if (!b.is_black()) add_lobe(B, mk_lobe(BX_MF_R, b, FR_DIEL, m.s1, m.s1, 1.5f, 1.0f, &m));
            return true;
        default: {                                              /* glass.rs:51-93 */
            Rgb r = clamp_positive(a), t = clamp_positive(b);
            bool is_spec = m.s1 == 0.0f && m.s2 == 0.0f;
            if (is_spec && allow_multiple_lobes) return false;
            if (!r.is_black()) add_lobe(B, mk_lobe(is_spec ? BX_SPEC_R : BX_MF_R, r, FR_DIEL, m.s1, m.s2, 1.0f, m.s0, &m));
            if (!t.is_black()) add_lobe(B, mk_lobe(is_spec ? BX_SPEC_T : BX_MF_T, t, FR_DIEL, m.s1, m.s2, 1.0f, m.s0, &m));
            return true;
        }
    }
}
__device__ inline V3 to_local(const DBsdf& B, V3 v) { return V3(dot(v, B.ss), dot(v, B.ts), dot(v, B.ns)); }
__device__ inline V3 to_world(const DBsdf& B, V3 v) {
    return V3(B.ss.x * v.x + B.ts.x * v.y + B.ns.x * v.z, B.ss.y * v.x + B.ts.y * v.y + B.ns.y * v.z, B.ss.z * v.x + B.ts.z * v.y + B.ns.z * v.z);
}
__device__ inline int bsdf_num(const DBsdf& B, uint32_t flags) {
    int n = 0;
#pragma unroll
    for (int i = 0; i < 2; i++) if (i < B.n) n += lobe_matches(B.lobe[i].kind, flags) ? 1 : 0;
    return n;
}
__device__ inline Rgb bsdf_sum_f(const DBsdf& B, V3 wo, V3 wi, bool refl, uint32_t flags) {
    Rgb sum(0.0f);
#pragma unroll
    for (int i = 0; i < 2; i++) {
        if (i < B.n) {
            const uint32_t t = lobe_type(B.lobe[i].kind);
            if ((flags & t) == t && ((refl && (t & T_REFL)) || (!refl && (t & T_TRANS)))) sum = sum + lobe_f(B.lobe[i], wo, wi);
        }
    }
    return sum;
}
__device__ inline Rgb bsdf_f(const DBsdf& B, V3 wo_w, V3 wi_w, uint32_t flags) {             /* bsdf.rs:67-82 */
    V3 wi = to_local(B, wi_w), wo = to_local(B, wo_w);
    if (wo.z == 0.0f) return Rgb(0.0f);
    bool refl = dot(wi_w, B.ng) * dot(wo_w, B.ng) > 0.0f;
    return bsdf_sum_f(B, wo, wi, refl, flags);
}
__device__ inline float bsdf_pdf(const DBsdf& B, V3 wo_w, V3 wi_w, uint32_t flags) {         /* bsdf.rs:131-144 */
    V3 wo = to_local(B, wo_w), wi = to_local(B, wi_w);
    if (wo.z == 0.0f) return 0.0f;
    float nm = (float)bsdf_num(B, flags), pdf = 0.0f;
#pragma unroll
    for (int i = 0; i < 2; i++) if (i < B.n && lobe_matches(B.lobe[i].kind, flags)) pdf = pdf + lobe_pdf(B.lobe[i], wo, wi);
    return nm > 0.0f ? pdf / nm : 0.0f;
}
__device__ inline bool bsdf_sample(const DBsdf& B, V3 wo_w, V2 u, uint32_t flags, DScatter* out) {   /* bsdf.rs:85-129 */
    float mc = (float)bsdf_num(B, flags);
    if (mc == 0.0f) return false;
    int comp = (int)f2usize(fmin_(floorf(u.x * mc), mc - 1.0f));
    /* the comp-th matching lobe; with at most two lobes: lobe 0 if it matches and comp == 0, else lobe 1 */
    const bool m0 = B.n > 0 && lobe_matches(B.lobe[0].kind, flags);
    const int sel = (m0 && comp == 0) ? 0 : 1;
    V2 ur(u.x * mc - (float)comp, u.y);
    V3 wo = to_local(B, wo_w);
    DScatter s;
    const DLobe Ls = sel == 0 ? B.lobe[0] : B.lobe[1];
    if (!lobe_sample(Ls, wo, ur, &s)) return false;
    float pdf = s.pdf; Rgb f = s.f;
    if (pdf == 0.0f) return false;
    V3 wi_w = to_world(B, s.wi);
    const bool spec = (lobe_type(Ls.kind) & T_SPECULAR) != 0;
    if (!spec && mc > 1.0f) {
        float extra = 0.0f;
#pragma unroll
        for (int i = 0; i < 2; i++) if (i < B.n && i != sel && lobe_matches(B.lobe[i].kind, flags)) extra = extra + lobe_pdf(B.lobe[i], wo, s.wi);
        pdf += extra;
    }
    if (mc > 1.0f) pdf /= mc;
    if (!spec) {
        bool refl = dot(wi_w, B.ng) * dot(wo_w, B.ng) > 0.0f;
        f = bsdf_sum_f(B, wo, s.wi, refl, flags);
    }
    out->f = f; out->wi = wi_w; out->pdf = pdf; out->type = s.type;
    return true;
}

/* ------------------------------------------------------------------ lights: src/light/ *.rs, sampling.rs:59-180, mipmap.rs:258-312 */
__host__ __device__ inline Rgb env_texel(const DLight& L, int s, int t) {
    int w = (int)L.env_w, h = (int)L.env_h;
    s = ((s % w) + w) % w; t = ((t % h) + h) % h;
    const float4 p = L.texels[(size_t)t * w + s];
    return Rgb(p.x, p.y, p.z);
}
__host__ __device__ inline Rgb env_lookup(const DLight& L, V2 st) {      /* triangle(0, st): mipmap.rs:258-272 */
    float s = st.x * (float)L.env_w - 0.5f, t = st.y * (float)L.env_h - 0.5f;
    int s0 = f2i_sat(floorf(s)), t0 = f2i_sat(floorf(t));
    float ds = s - (float)s0, dt = t - (float)t0;
    return env_texel(L, s0, t0) * (1.0f - ds) * (1.0f - dt) + env_texel(L, s0, t0 + 1) * (1.0f - ds) * dt +
           env_texel(L, s0 + 1, t0) * ds * (1.0f - dt) + env_texel(L, s0 + 1, t0 + 1) * ds * dt;
}
/* env_lookup out of the cell record centred at (cx, cy): the same four texels, the same weights, the same expression.  false: the
 * lookup's 2 x 2 block is not inside this cell's neighbourhood (the caller then reads the plain texel table) */
__device__ inline bool env_cell_lookup(const DLight& L, V2 st, int cx, int cy, Rgb* out) {
    float s = st.x * (float)L.env_w - 0.5f, t = st.y * (float)L.env_h - 0.5f;
    int s0 = f2i_sat(floorf(s)), t0 = f2i_sat(floorf(t));
    float ds = s - (float)s0, dt = t - (float)t0;
    const int dx = s0 - cx, dy = t0 - cy;                       /* -1 or 0 for the cell that contains st */
    if (dx < -1 || dx > 0 || dy < -1 || dy > 0) return false;
    const float* c = reinterpret_cast<const float*>(L.cells + 8 * ((size_t)cy * L.env_w + (size_t)cx)) + 3 * ((dy + 1) * 3 + dx + 1);
    const Rgb t00(c[0], c[1], c[2]), t10(c[3], c[4], c[5]), t01(c[9], c[10], c[11]), t11(c[12], c[13], c[14]);
    *out = t00 * (1.0f - ds) * (1.0f - dt) + t01 * (1.0f - ds) * dt + t10 * ds * (1.0f - dt) + t11 * ds * dt;
    return true;
}
__device__ inline float env_cell_func(const DLight& L, uint32_t cx, uint32_t cy) {
    return reinterpret_cast<const float*>(L.cells + 8 * ((size_t)cy * L.env_w + (size_t)cx))[27];
}
/* the cell that contains st = (u, v) in [0, 1]^2 */
__device__ inline void env_cell_of(const DLight& L, V2 st, int* cx, int* cy) {
    int x = f2i_sat(floorf(st.x * (float)L.env_w)), y = f2i_sat(floorf(st.y * (float)L.env_h));
    *cx = x < 0 ? 0 : (x > (int)L.env_w - 1 ? (int)L.env_w - 1 : x); *cy = y < 0 ? 0 : (y > (int)L.env_h - 1 ? (int)L.env_h - 1 : y);
}
__device__ inline uint32_t search_cdf(const float* cdf, uint32_t size, float u) {   /* sampling.rs:66-81 */
    uint32_t first = 0, len = size;
    while (len > 0) {
        uint32_t half = len >> 1, mid = first + half;
        if (cdf[mid] <= u) { first = mid + 1; len -= half + 1; } else len = half;
    }
    int v = (int)first - 1, hi = (int)size - 2;
    return (uint32_t)(v < 0 ? 0 : (v > hi ? hi : v));
}
/* number of elements <= u in a non-decreasing array (what the loop of search_cdf computes) */
__device__ inline uint32_t upper_bound_f(const float* a, uint32_t size, float u) {
    uint32_t first = 0, len = size;
    while (len > 0) {
        uint32_t half = len >> 1, mid = first + half;
        if (a[mid] <= u) { first = mid + 1; len -= half + 1; } else len = half;
    }
    return first;
}
/* search_cdf through the table of every 32nd entry: same count of elements <= u, hence the same interval, for a monotone CDF */
__device__ inline uint32_t search_cdf_blocked(const float* cdf, const float* coarse, uint32_t size, float u) {
    const uint32_t n = size - 1u, nb = (n + 31u) >> 5;
    uint32_t b = upper_bound_f(coarse, nb + 1u, u);
    b = b == 0u ? 0u : b - 1u; if (b > nb - 1u) b = nb - 1u;
    const uint32_t lo = b << 5, hi = lo + 32u < n ? lo + 32u : n;
    const uint32_t first = (cdf[lo] <= u ? lo + 1u : lo) + (cdf[lo] <= u ? upper_bound_f(cdf + lo + 1u, hi - lo, u) : 0u);
    int v = (int)first - 1, top = (int)size - 2;
    return (uint32_t)(v < 0 ? 0 : (v > top ? top : v));
}
/* (func == NULL: *pdf is left to the caller, who has the function value from somewhere else -- the cell record) */
__device__ inline void dist1d_sample(const float* func, const float* cdf, float integral, uint32_t n, float u, float* x, float* pdf, uint32_t* idx, const float* coarse = nullptr) {
    uint32_t i = (coarse && n > 0u) ? search_cdf_blocked(cdf, coarse, n + 1, u) : search_cdf(cdf, n + 1, u);
    float du = u - cdf[i];
    if (cdf[i + 1] - cdf[i] > 0.0f) du /= cdf[i + 1] - cdf[i];
    if (func) *pdf = func[i] / integral;
    *x = ((float)i + du) / (float)n;
    *idx = i;
}
FTN_DEV_NOINLINE Rgb light_Le_env(const DLight& L, V3 dir) {    /* infinite.rs:156-164 */
    V3 w = normalize(m4_vector(L.w2l, dir));
    V2 st(spherical_phi(w) * (1.0f / (2.0f * FTN_PI)), spherical_theta(w) * FTN_INV_PI);
    /* the plain texel table (12 MB at 1024^2: mostly served by the caches), not the cell records (128 MB: a miss each time) -- those pay where
     * a light SAMPLE needs the neighbourhood and the function value of the cell its search ended in (light_sample_env) */
    return env_lookup(L, st);
}
__device__ inline Rgb scene_env_Le(const DScene& S, V3 dir) {    /* scene/mod.rs:59-64: sum over all lights (non-infinite give 0) */
    Rgb sum(0.0f);
    /* 0 + 0 + ... + Le_k + 0 ...: adding the zero spectra of the other lights is exact, so only infinite lights are visited */
    for (uint32_t i = 0; i < S.n_inf_lights; i++) sum = sum + light_Le_env(S.lights[S.inf_lights[i]], dir);
    return sum;
}
__device__ inline Rgb area_Le(const DLight& L, V3 n, V3 w) {     /* diffuse.rs:44-50 */
    return dot(n, w) > 0.0f ? Rgb(L.rgb[0], L.rgb[1], L.rgb[2]) : Rgb(0.0f);
}
struct DLiSample { Rgb radiance; V3 wi; float pdf; DSurfHit p1; };
__device__ inline DLiSample light_sample_env(const DLight& L, const DSurfHit& ref, V2 u) {      /* infinite.rs:99-140 */
    DLiSample s;
    float d1, pdf1, d0, pdf0; uint32_t vi, ui;
    dist1d_sample(L.marg_func, L.marg_cdf, L.marg_integral, L.nv, u.y, &d1, &pdf1, &vi, L.marg_coarse);
    const float cint = L.cond_integral[vi];
    dist1d_sample(L.cells ? nullptr : L.cond_func + (size_t)vi * L.nu, L.cond_cdf + (size_t)vi * (L.nu + 1), cint, L.nu, u.x, &d0, &pdf0, &ui,
                  L.cond_coarse ? L.cond_coarse + (size_t)vi * (((L.nu + 31u) >> 5) + 1u) : nullptr);
    if (L.cells) pdf0 = env_cell_func(L, ui, vi) / cint;
    float map_pdf = pdf0 * pdf1;
    float theta = d1 * FTN_PI, phi = d0 * 2.0f * FTN_PI;
    float sth, cth, sph, cph;
    ftn_det::sincosf_det(theta, &sth, &cth); ftn_det::sincosf_det(phi, &sph, &cph);
    s.wi = m4_vector(L.l2w, V3(sth * cph, sth * sph, cth));
    s.pdf = (sth == 0.0f) ? 0.0f : map_pdf / (2.0f * FTN_PI * FTN_PI * sth);
    if (map_pdf == 0.0f) s.pdf = 0.0f;                                   /* reference: unimplemented!() */
    s.p1.p = ref.p + s.wi * (2.0f * L.world_radius); s.p1.p_err = V3(); s.p1.time = ref.time; s.p1.n = V3();
    if (!(L.cells && env_cell_lookup(L, V2(d0, d1), (int)ui, (int)vi, &s.radiance))) s.radiance = env_lookup(L, V2(d0, d1));
    return s;
}
__device__ inline float light_pdf_env(const DLight& L, V3 wi) {                                  /* infinite.rs:142-154 */
    V3 w = m4_vector(L.w2l, wi);
    float theta = spherical_theta(w), phi = spherical_phi(w);
    float sth = ftn_det::sinf_det(theta);
    if (sth == 0.0f) return 0.0f;
    float px = phi * (1.0f / (2.0f * FTN_PI)), py = theta * FTN_INV_PI;
    long long iu = f2usize(px * (float)L.nu); if (iu > (long long)L.nu - 1) iu = (long long)L.nu - 1;
    long long iv = f2usize(py * (float)L.nv); if (iv > (long long)L.nv - 1) iv = (long long)L.nv - 1;
    const float fv = L.cond_func[(size_t)iv * L.nu + iu];      /* (the 4-byte table, not the cell record: 4 MB stay cached where 128 MB do not) */
    return (fv / L.marg_integral) / (2.0f * FTN_PI * FTN_PI * sth);
}
FTN_DEV_NOINLINE DLiSample light_sample(const DScene& S, const DLight& L, const DSurfHit& ref, V2 u) {
    DLiSample s;
    switch (L.kind) {
        case LK_POINT: {
            V3 wp(L.v[0], L.v[1], L.v[2]);
            s.wi = normalize(wp - ref.p); s.pdf = 1.0f;
            s.p1.p = wp; s.p1.p_err = V3(); s.p1.time = ref.time; s.p1.n = V3();
            s.radiance = Rgb(L.rgb[0], L.rgb[1], L.rgb[2]) / len2(wp - ref.p);
            return s;
        }
        case LK_DISTANT: {
            V3 d(L.v[0], L.v[1], L.v[2]);
            s.p1.p = ref.p + d * (2.0f * L.world_radius); s.p1.p_err = V3(); s.p1.time = ref.time; s.p1.n = V3();
            s.radiance = Rgb(L.rgb[0], L.rgb[1], L.rgb[2]); s.wi = d; s.pdf = 1.0f;
            return s;
        }
        case LK_INFINITE: return light_sample_env(L, ref, u);
        default: {                                                               /* diffuse.rs:75-89 */
            DSurfHit ps = shape_sample(S, L.prim, u);
            s.wi = normalize(ps.p - ref.p);
            s.pdf = shape_pdf_from_ref(S, L.prim, L.area, ref, s.wi);
            s.p1 = ps;
            s.radiance = area_Le(L, ps.n, -s.wi);
            return s;
        }
    }
}
FTN_DEV_NOINLINE float light_pdf(const DScene& S, const DLight& L, const DSurfHit& ref, V3 wi) {
    if (L.kind == LK_AREA) return shape_pdf_from_ref(S, L.prim, L.area, ref, wi);
    if (L.kind != LK_INFINITE) return 0.0f;
    return light_pdf_env(L, wi);
}

/* ------------------------------------------------------------------ RNG: rand_xoshiro 0.2.0 Xoshiro256Plus / SplitMix64, rand 0.6.5 Standard<f32> */
struct Rng {
    uint64_t s0, s1, s2, s3;
    __device__ void seed(uint64_t x) {
        uint64_t z;
        #define FTN_SM() (x += 0x9e3779b97f4a7c15ULL, z = x, z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL, z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL, z ^ (z >> 31))
        s0 = FTN_SM(); s1 = FTN_SM(); s2 = FTN_SM(); s3 = FTN_SM();
        #undef FTN_SM
    }
    __device__ float next() {
        uint64_t result = s0 + s3;
        uint64_t t = s1 << 17;
        s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= t;
        s3 = (s3 << 45) | (s3 >> 19);
        return (float)((uint32_t)(result >> 32) >> 8) * (1.0f / 16777216.0f);
    }
    __device__ V2 next2() { float a = next(); float b = next(); return V2(a, b); }
    /* the state after n further draws (the update of next() without its output) */
    __device__ void skip(uint32_t n) {
        for (uint32_t i = 0; i < n; i++) { const uint64_t t = s1 << 17; s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= t; s3 = (s3 << 45) | (s3 >> 19); }
    }
};
__device__ inline uint64_t indexed_key(uint64_t seed, int px, int py, uint32_t sample) {
    return (seed * 0x9E3779B97F4A7C15ULL) ^ ((uint64_t)(uint32_t)py << 40) ^ ((uint64_t)(uint32_t)px << 20) ^ (uint64_t)sample;
}

/* ------------------------------------------------------------------ PerspectiveCamera::generate_ray_differential (main ray): camera/mod.rs:145-205.
 * Ray differentials only feed image-texture filtering (interaction.rs:124-173); all textures are constant, so they are not carried. */
__device__ inline DRay camera_ray(const DCamera& C, V2 p_film, V2 p_lens_u, float time_u) {
    V3 pc = m4_point(C.r2c, V3(p_film.x, p_film.y, 0.0f));
    float time = (1.0f - time_u) * C.shutter_open + time_u * C.shutter_close;
    V3 o(0.0f, 0.0f, 0.0f);
    V3 d = normalize(pc - o);
    if (C.lens_radius > 0.0f) {
        V2 dl = concentric_sample_disk(p_lens_u);
        V2 pl(C.lens_radius * dl.x, C.lens_radius * dl.y);
        float ft = C.focal_dist / d.z;
        V3 pf = o + (d * ft);
        o = V3(pl.x, pl.y, 0.0f);
        d = normalize(pf - o);
    }
    /* Ray::transform(camera_to_world): transform.rs:307-322 */
    V3 oe; V3 ot = m4_point_exact_to_err(C.c2w, o, &oe);
    V3 dw = m4_vector(C.c2w, d);
    float t_max = FTN_INF;
    float lsq = len2(dw);
    if (lsq > 0.0f) { float dt = dot(vabs(dw), oe) / lsq; ot = ot + dw * dt; t_max -= dt; }
    DRay r; r.o = ot; r.d = dw; r.t_max = t_max; r.time = time;
    return r;
}

}  // namespace ftn
#endif
