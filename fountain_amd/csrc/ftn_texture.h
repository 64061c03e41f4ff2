/*
 * ftn_texture.h -- textures, ray differentials and the MIP pyramid lookup on the device:
 *   src/texture/{mapping,uv,checkerboard,image}.rs, src/mipmap.rs:273-341, src/interaction.rs:124-173 (compute_tex_differentials),
 *   src/camera/mod.rs:145-205 (generate_ray_differential), src/integrator/mod.rs:58-84 / :119-163 (differentials of specular bounces).
 * Only compiled into the TEXTURED variants of the render kernels: scenes whose material parameters are all constants run the
 * exact code they ran before.
 *
 * HBM layout: textures = the ftn_texture array as given; images = one DImage per ftn_image with its pyramid levels stored as
 * float4 texels (rgb + pad, so a texel is one aligned 16-byte load) back to back in `texels`.
 */
#ifndef FTN_TEXTURE_H
#define FTN_TEXTURE_H

#include "ftn_device.h"

namespace ftn {

struct DRayDiff { V3 rxo, ryo, rxd, ryd; bool has; };
struct DTexDiffs { V3 dpdx, dpdy; float dudx, dvdx, dudy, dvdy; };

/* ------------------------------------------------------------------ the constant-per-material part of compute_scattering_functions
 * (matte.rs:39-50 + OrenNayar::new reflection/mod.rs:260-267, roughness_to_alpha microfacet.rs:40-45): done once at upload for
 * constant parameters, per hit for textured ones. */
FTN_HD float roughness_to_alpha_(float roughness) {
    float x = ftn_det::logf_det(fmax_(roughness, 1.0e-3f));
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}
FTN_HD void material_finalize(ftn_material& m) {
    if (m.type == FTN_MAT_MATTE) {
        m.s0 = clampf(m.s0, 0.0f, 90.0f);
        if (m.s0 != 0.0f) { float sg = m.s0 * (float)(3.14159265358979323846 / 180.0); float s2 = sg * sg; m.s1 = 1.0f - (s2 / (2.0f * (s2 + 0.33f))); m.s2 = 0.45f * s2 / (s2 + 0.09f); }
    } else if (m.type == FTN_MAT_METAL || m.type == FTN_MAT_GLASS) { if (m.remap_roughness) { m.s1 = roughness_to_alpha_(m.s1); m.s2 = roughness_to_alpha_(m.s2); m.remap_roughness = 0; } }
    else if (m.type == FTN_MAT_PLASTIC) { if (m.remap_roughness) { m.s1 = roughness_to_alpha_(m.s1); m.remap_roughness = 0; } }
}

/* ------------------------------------------------------------------ MIPMap lookups: mipmap.rs:273-341 */
__device__ inline int rem_euclid_(int a, int n) { int r = a % n; return r < 0 ? r + n : r; }
__device__ inline Rgb mip_texel(const DScene& S, const DImage& im, uint32_t level, int s, int t) {       /* get_texel_from_level */
    const int w = (int)im.lw[level], h = (int)im.lh[level];
    if (im.wrap == FTN_WRAP_REPEAT) { s = rem_euclid_(s, w); t = rem_euclid_(t, h); }
    else if (im.wrap == FTN_WRAP_CLAMP) { s = min(max(s, 0), w - 1); t = min(max(t, 0), h - 1); }
    else if (s < 0 || s >= w || t < 0 || t >= h) return Rgb(0.0f);
    const float4 v = S.texels[(size_t)im.off[level] + (size_t)t * (size_t)w + (size_t)s];
    return Rgb(v.x, v.y, v.z);
}
__device__ inline Rgb mip_triangle(const DScene& S, const DImage& im, int level, V2 st) {                  /* :294-306 */
    level = min(max(level, 0), (int)im.n_levels - 1);
    const float s = st.x * (float)im.lw[level] - 0.5f, t = st.y * (float)im.lh[level] - 0.5f;
    const int s0 = f2i_sat(floorf(s)), t0 = f2i_sat(floorf(t));
    const float ds = s - (float)s0, dt = t - (float)t0;
    return mip_texel(S, im, (uint32_t)level, s0, t0) * (1.0f - ds) * (1.0f - dt) + mip_texel(S, im, (uint32_t)level, s0, t0 + 1) * (1.0f - ds) * dt +
           mip_texel(S, im, (uint32_t)level, s0 + 1, t0) * ds * (1.0f - dt) + mip_texel(S, im, (uint32_t)level, s0 + 1, t0 + 1) * ds * dt;
}
__device__ inline Rgb mip_lookup_trilinear_width(const DScene& S, const DImage& im, V2 st, float width) {  /* :273-286 */
    const float level = (float)im.n_levels - 1.0f + ftn_det::log2f_det(fmax_(width, 1.0e-8f));
    if (level < 0.0f) return mip_triangle(S, im, 0, st);
    if (level >= (float)(im.n_levels - 1)) return mip_texel(S, im, im.n_levels - 1, 0, 0);
    const int lf = (int)f2usize(floorf(level));
    const float delta = level - truncf(level);
    return (1.0f - delta) * mip_triangle(S, im, lf, st) + delta * mip_triangle(S, im, lf + 1, st);
}
__device__ inline Rgb mip_lookup_trilinear(const DScene& S, const DImage& im, V2 st, V2 dst0, V2 dst1) {   /* :288-291 (dst0.y without abs, as written) */
    const float width = fmax_(fmax_(fabsf(dst0.x), dst0.y), fmax_(fabsf(dst1.x), fabsf(dst1.y)));
    return mip_lookup_trilinear_width(S, im, st, 2.0f * width);
}

/* ------------------------------------------------------------------ Texture::evaluate over the flat texture array */
__device__ inline Rgb tex_eval(const DScene& S, int id, V2 uv, const DTexDiffs& td) {
    for (int guard = 0; guard < 64; guard++) {
        const ftn_texture t = S.textures[id];
        if (t.kind == FTN_TEX_CONSTANT) return t.is_float ? Rgb(t.value[0]) : Rgb(t.value[0], t.value[1], t.value[2]);
        const V2 st(t.su * uv.x + t.du, t.sv * uv.y + t.dv);                                           /* UVMapping::evaluate mapping.rs:41-53 */
        if (t.kind == FTN_TEX_UV) return Rgb(st.x - floorf(st.x), st.y - floorf(st.y), 0.0f);          /* uv.rs:17-23 */
        if (t.kind == FTN_TEX_CHECKERBOARD) {                                                         /* checkerboard.rs:49-64 */
            const int a = f2i_sat(floorf(st.x)), b = f2i_sat(floorf(st.y));
            id = ((int)((uint32_t)a + (uint32_t)b) % 2 == 0) ? t.tex1 : t.tex2;
            continue;
        }
        const V2 dx(t.su * td.dudx, t.sv * td.dvdx), dy(t.su * td.dudy, t.sv * td.dvdy);               /* image.rs:28-34 */
        return mip_lookup_trilinear(S, S.images[t.image], st, dx, dy);
    }
    return Rgb(0.0f);
}
/* the material record with its textured parameters evaluated at this hit, ready for make_bsdf */
__device__ inline ftn_material material_resolve(const DScene& S, int mat, V2 uv, const DTexDiffs& td) {
    ftn_material m = S.materials[mat];                 /* raw (not finalized) for textured materials */
    const ftn_material_textures mt = S.mtex[mat];
    if (mt.a >= 0) { Rgb v = tex_eval(S, mt.a, uv, td); m.a[0] = v.r; m.a[1] = v.g; m.a[2] = v.b; }
    if (mt.b >= 0) { Rgb v = tex_eval(S, mt.b, uv, td); m.b[0] = v.r; m.b[1] = v.g; m.b[2] = v.b; }
    if (mt.s0 >= 0) m.s0 = tex_eval(S, mt.s0, uv, td).r;
    if (mt.s1 >= 0) m.s1 = tex_eval(S, mt.s1, uv, td).r;
    if (mt.s2 >= 0) m.s2 = tex_eval(S, mt.s2, uv, td).r;
    material_finalize(m);
    return m;
}
__device__ inline bool material_is_textured(const DScene& S, int mat) {
    if (!S.mtex) return false;
    const ftn_material_textures mt = S.mtex[mat];
    return (mt.a & mt.b & mt.s0 & mt.s1 & mt.s2) >= 0;      /* the AND is negative only when every slot is -1 */
}

/* ------------------------------------------------------------------ SurfaceInteraction::compute_tex_differentials: interaction.rs:124-173 */
__device__ inline bool solve_2x2(float a00, float a01, float a10, float a11, float b0, float b1, float* x0, float* x1) {   /* math.rs:88-102, A = from_cols((a00,a01),(a10,a11)) */
    const float det = a00 * a11 - a10 * a01;
    if (fabsf(det) < 1.0e-10f) return false;
    *x0 = (a11 * b0 - a10 * b1) / det;
    *x1 = (a00 * b1 - a01 * b0) / det;
    if (isnan(*x0) || isnan(*x1)) return false;
    return true;
}
__device__ inline DTexDiffs compute_tex_diffs(V3 p, V3 n, V3 dpdu, V3 dpdv, const DRayDiff& rd) {
    DTexDiffs z; z.dpdx = V3(0.0f, 0.0f, 0.0f); z.dpdy = z.dpdx; z.dudx = z.dvdx = z.dudy = z.dvdy = 0.0f;
    if (!rd.has) return z;
    const float d = dot(n, p);
    const float tx = -(dot(n, rd.rxo) - d) / dot(n, rd.rxd);
    const V3 px = rd.rxo + tx * rd.rxd;
    const float ty = -(dot(n, rd.ryo) - d) / dot(n, rd.ryd);
    const V3 py = rd.ryo + ty * rd.ryd;
    const V3 dpdx = px - p, dpdy = py - p;
    int d0, d1;
    if (fabsf(n.x) > fabsf(n.y) && fabsf(n.x) > fabsf(n.z)) { d0 = 1; d1 = 2; }
    else if (fabsf(n.y) > fabsf(n.z)) { d0 = 0; d1 = 2; }
    else { d0 = 0; d1 = 1; }
    float dudx, dvdx, dudy, dvdy;
    if (!solve_2x2(dpdu.get(d0), dpdu.get(d1), dpdv.get(d0), dpdv.get(d1), dpdx.get(d0), dpdx.get(d1), &dudx, &dvdx)) return z;
    if (!solve_2x2(dpdu.get(d0), dpdu.get(d1), dpdv.get(d0), dpdv.get(d1), dpdy.get(d0), dpdy.get(d1), &dudy, &dvdy)) return z;
    DTexDiffs o; o.dpdx = dpdx; o.dpdy = dpdy; o.dudx = dudx; o.dvdx = dvdx; o.dudy = dudy; o.dvdy = dvdy;
    return o;
}

/* ------------------------------------------------------------------ the differential part of generate_ray_differential (camera/mod.rs:145-205),
 * RayDifferential::transform (transform.rs:325-338) and scale_differentials(1/sqrt(spp)) (geometry/mod.rs:125-133, integrator/mod.rs:254) */
__device__ inline DRayDiff camera_ray_diff(const DCamera& C, V2 p_film, V2 p_lens_u, const DRay& world_ray, float spp_scale) {
    const V3 pc = m4_point(C.r2c, V3(p_film.x, p_film.y, 0.0f));
    const V3 dxc(C.dx_camera[0], C.dx_camera[1], C.dx_camera[2]), dyc(C.dy_camera[0], C.dy_camera[1], C.dy_camera[2]);
    V3 rxo(0.0f, 0.0f, 0.0f), ryo(0.0f, 0.0f, 0.0f), rxd, ryd;
    if (C.lens_radius > 0.0f) {
        const V2 dl = concentric_sample_disk(p_lens_u);
        const V2 pl(C.lens_radius * dl.x, C.lens_radius * dl.y);
        const V3 dx = normalize(pc + dxc);
        const float ftx = C.focal_dist / dx.z;
        const V3 pfx = V3(0.0f, 0.0f, 0.0f) + (ftx * dx);
        rxo = V3(pl.x, pl.y, 0.0f); rxd = normalize(pfx - rxo);
        const V3 dy = normalize(pc + dxc);                               /* sic: dx_camera (camera/mod.rs:173) */
        const float fty = C.focal_dist / dy.z;
        const V3 pfy = V3(0.0f, 0.0f, 0.0f) + (fty * dy);
        ryo = V3(pl.x, pl.y, 0.0f); ryd = normalize(pfy - ryo);
    } else { rxd = normalize(pc + dxc); ryd = normalize(pc + dyc); }
    DRayDiff r; r.has = true;
    r.rxo = m4_point(C.c2w, rxo); r.ryo = m4_point(C.c2w, ryo); r.rxd = m4_vector(C.c2w, rxd); r.ryd = m4_vector(C.c2w, ryd);
    r.rxo = world_ray.o + (r.rxo - world_ray.o) * spp_scale; r.ryo = world_ray.o + (r.ryo - world_ray.o) * spp_scale;
    r.rxd = world_ray.d + (r.rxd - world_ray.d) * spp_scale; r.ryd = world_ray.d + (r.ryd - world_ray.d) * spp_scale;
    return r;
}

/* ------------------------------------------------------------------ differentials of a specular bounce: integrator/mod.rs:58-84 (reflect), :119-163 (transmit) */
__device__ inline DRayDiff specular_diff(bool reflect, const DRayDiff& in, V3 p, V3 wo, V3 wi, V3 ns, V3 dndu, V3 dndv, const DTexDiffs& td, float bsdf_eta) {
    DRayDiff o; o.has = in.has;
    if (!in.has) return o;
    o.rxo = p + td.dpdx; o.ryo = p + td.dpdy;
    V3 dndx = dndu * td.dudx + dndv * td.dvdx, dndy = dndu * td.dudy + dndv * td.dvdy;
    if (reflect) {
        const V3 dwo_dx = -in.rxd - wo, dwo_dy = -in.ryd - wo;
        const float dDN_dx = dot(dwo_dx, ns) + dot(wo, dndx), dDN_dy = dot(dwo_dy, ns) + dot(wo, dndy);
        o.rxd = (wi - dwo_dx) + (2.0f * dot(wo, ns)) * dndx + dDN_dx * ns;
        o.ryd = (wi - dwo_dy) + (2.0f * dot(wo, ns)) * dndy + dDN_dy * ns;
    } else {
        V3 sn = ns; float eta = 1.0f / bsdf_eta;
        if (dot(wo, ns) < 0.0f) { eta = bsdf_eta; sn = -sn; dndx = -dndx; dndy = -dndy; }
        const V3 dwo_dx = -in.rxd - wo, dwo_dy = -in.ryd - wo;
        const float dDN_dx = dot(dwo_dx, ns) + dot(wo, dndx), dDN_dy = dot(dwo_dy, ns) + dot(wo, dndy);
        const float mu = eta * dot(wo, sn) - abs_dot(wi, sn);
        const float dmu_dx = (eta - (eta * eta * dot(wo, sn)) / dot(wi, sn)) * dDN_dx;
        const float dmu_dy = (eta - (eta * eta * dot(wo, sn)) / dot(wi, sn)) * dDN_dy;
        o.rxd = wi - (eta * dwo_dx) + (mu * dndx + dmu_dx * sn);
        o.ryd = wi - (eta * dwo_dy) + (mu * dndy + dmu_dy * sn);
    }
    return o;
}

}  // namespace ftn
#endif
