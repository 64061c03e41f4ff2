// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.hpp).
// orc_render.hpp: src/sampler/{mod,random}.rs, src/camera/mod.rs, src/film.rs, src/filter/mod.rs,
// src/integrator/{mod,path,direct_lighting}.rs.  RNG: rand_xoshiro 0.2.0 Xoshiro256Plus + SplitMix64,
// rand 0.6.5 Standard<f32> (restated from the published algorithms; parity unpinned).
#pragma once
#include "orc_scene.hpp"

namespace orc {

// ---- RNG
struct SplitMix64 {
    uint64_t x;
    uint64_t next_u64() {
        x += 0x9e3779b97f4a7c15ULL;
        uint64_t z = x;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        return z ^ (z >> 31);
    }
};
struct Xoshiro256Plus {
    uint64_t s[4];
    static Xoshiro256Plus seed_from_u64(uint64_t seed) {   // SplitMix64 fills the 32 seed bytes, little endian
        SplitMix64 sm{seed}; Xoshiro256Plus r;
        for (int i = 0; i < 4; i++) r.s[i] = sm.next_u64();
        return r;
    }
    uint64_t next_u64() {
        uint64_t result = s[0] + s[3];
        uint64_t t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t;
        s[3] = (s[3] << 45) | (s[3] >> 19);
        return result;
    }
    uint32_t next_u32() { return (uint32_t)(next_u64() >> 32); }   // upper bits (rand_xoshiro 0.2)
    Float gen_f32() { return (Float)(next_u32() >> 8) * (1.0f / 16777216.0f); }   // rand 0.6 Standard: 24 bits in [0,1)
};

inline uint64_t indexed_sample_key(uint64_t seed, int32_t px, int32_t py, uint32_t sample) {
    return (seed * 0x9E3779B97F4A7C15ULL) ^ ((uint64_t)(uint32_t)py << 40) ^ ((uint64_t)(uint32_t)px << 20) ^ (uint64_t)sample;
}

// ---- Sampler trait + RandomSampler: src/sampler/mod.rs:19-54, src/sampler/random.rs
struct CameraSample { Vec2 p_film, p_lens; Float time; };
struct Sampler {
    uint32_t kind; Xoshiro256Plus rng; uint64_t base_seed;
    uint32_t samples_per_pixel; int32_t px = 0, py = 0; uint32_t current_pixel_sample_num = 0;
    uint32_t first_sample = 0, last_sample = 0;   // indexed: render samples (first, last] in 1-based numbering
    Sampler clone_with_seed(uint64_t seed) const { Sampler s = *this; s.rng = Xoshiro256Plus::seed_from_u64(seed); return s; }  // random.rs:61-67
    void start_pixel(int32_t x, int32_t y) { px = x; py = y; current_pixel_sample_num = (kind == FTN_SAMPLER_INDEXED) ? first_sample : 0; }
    bool start_next_sample() {                                              // mod.rs:94-99
        current_pixel_sample_num += 1;
        bool more = current_pixel_sample_num <= ((kind == FTN_SAMPLER_INDEXED) ? last_sample : samples_per_pixel);
        if (more && kind == FTN_SAMPLER_INDEXED)
            rng = Xoshiro256Plus::seed_from_u64(indexed_sample_key(base_seed, px, py, current_pixel_sample_num - 1));
        return more;
    }
    Float get_1d() { return rng.gen_f32(); }
    Vec2 get_2d() { Float a = rng.gen_f32(); Float b = rng.gen_f32(); return Vec2(a, b); }
    CameraSample get_camera_sample(int32_t rx, int32_t ry) {                // mod.rs:43-51
        CameraSample cs; Vec2 j = get_2d();
        cs.p_film = Vec2((Float)rx + j.x, (Float)ry + j.y);
        cs.p_lens = get_2d(); cs.time = get_1d();
        return cs;
    }
};

// ---- PerspectiveCamera: src/camera/mod.rs:72-205
struct Camera {
    Transform camera_to_world, raster_to_camera;
    Float shutter_open, shutter_close, lens_radius, focal_dist; Vec3 dx_camera, dy_camera;
    static Float lerp(Float t, Float a, Float b) { return (1.0f - t) * a + t * b; }   // math.rs:14-16
    RayDifferential generate_ray_differential(const CameraSample& sample, Float* weight) const {   // :145-205
        Vec3 p_film(sample.p_film.x, sample.p_film.y, 0.0f);
        Vec3 p_camera = tf_point(raster_to_camera, p_film);
        Float time = lerp(sample.time, shutter_open, shutter_close);
        Vec3 origin(0, 0, 0);
        Vec3 dir = normalize(p_camera - origin);
        RayDifferential rd; rd.ray.origin = origin; rd.ray.dir = dir; rd.ray.time = time; rd.ray.t_max = INF; rd.has_diff = true;
        if (lens_radius > 0.0f) {
            Vec2 d = concentric_sample_disk(sample.p_lens);
            Vec2 p_lens(lens_radius * d.x, lens_radius * d.y);
            Float ft = focal_dist / rd.ray.dir.z;
            Vec3 p_focus = rd.ray.at(ft);
            rd.ray.origin = Vec3(p_lens.x, p_lens.y, 0.0f);
            rd.ray.dir = normalize(p_focus - rd.ray.origin);
            Vec3 dx = normalize(p_camera + dx_camera);
            Float ftx = focal_dist / dx.z;
            Vec3 pfx = Vec3(0, 0, 0) + (ftx * dx);
            rd.diff.rx_origin = Vec3(p_lens.x, p_lens.y, 0.0f);
            rd.diff.rx_dir = normalize(pfx - rd.diff.rx_origin);
            Vec3 dy = normalize(p_camera + dx_camera);                      // quirk: dx_camera, :173
            Float fty = focal_dist / dy.z;
            Vec3 pfy = Vec3(0, 0, 0) + (fty * dy);
            rd.diff.ry_origin = Vec3(p_lens.x, p_lens.y, 0.0f);
            rd.diff.ry_dir = normalize(pfy - rd.diff.ry_origin);
        } else {
            rd.diff.rx_origin = origin; rd.diff.ry_origin = origin;
            rd.diff.rx_dir = normalize(p_camera + dx_camera);
            rd.diff.ry_dir = normalize(p_camera + dy_camera);
        }
        *weight = 1.0f;
        return tf_ray_diff(camera_to_world, rd);
    }
};
// PerspectiveCamera::new + CameraProjection::new: :51-69, :85-114
inline bool make_perspective_camera(const Transform& c2w, int xres, int yres, const Float sw[4], Float sh0, Float sh1,
                                    Float lens_radius, Float focal_dist, Float fov, ftn_camera_desc* out) {
    Transform persp;
    if (!tf_perspective(fov, 1.0e-2f, 1000.0f, &persp)) return false;
    // screen_window = {min.x, min.y, max.x, max.y}
    Transform screen_to_raster = tf_scale((Float)xres, (Float)yres, 1.0f) *
                                 tf_scale(1.0f / (sw[2] - sw[0]), 1.0f / (sw[1] - sw[3]), 1.0f) *
                                 tf_translate(Vec3(-sw[0], -sw[3], 0.0f));
    Transform raster_to_screen = screen_to_raster.inverse();
    Transform raster_to_camera = persp.inverse() * raster_to_screen;
    Vec3 o = tf_point(raster_to_camera, Vec3(0, 0, 0));
    Vec3 dx = tf_point(raster_to_camera, Vec3(1, 0, 0)) - o;
    Vec3 dy = tf_point(raster_to_camera, Vec3(0, 1, 0)) - o;
    memcpy(out->camera_to_world.m, c2w.t.a, 64); memcpy(out->camera_to_world.inv, c2w.invt.a, 64);
    memcpy(out->raster_to_camera.m, raster_to_camera.t.a, 64); memcpy(out->raster_to_camera.inv, raster_to_camera.invt.a, 64);
    out->shutter_open = sh0; out->shutter_close = sh1; out->lens_radius = lens_radius; out->focal_dist = focal_dist;
    for (int i = 0; i < 3; i++) { out->dx_camera[i] = dx[i]; out->dy_camera[i] = dy[i]; }
    return true;
}

// ---- Film<BoxFilter>: src/film.rs
struct Bounds2i { int x0, y0, x1, y1; int area() const { return (x1 - x0) * (y1 - y0); } };
inline Bounds2i intersect2i(Bounds2i a, Bounds2i b) { return Bounds2i{std::max(a.x0, b.x0), std::max(a.y0, b.y0), std::min(a.x1, b.x1), std::min(a.y1, b.y1)}; }

struct FilmTilePixel { Spectrum contrib_sum; Float filter_weight_sum = 0.0f; };
struct FilmTile { Bounds2i pixel_bounds; Vec2 filter_radius, inv_filter_radius; std::vector<FilmTilePixel> pixels;
    size_t get_pixel_idx(int x, int y) const { int width = pixel_bounds.x1 - pixel_bounds.x0; return (size_t)((y - pixel_bounds.y0) * width + (x - pixel_bounds.x0)); } };

struct Film {
    int full_res[2]; Bounds2i crop; Vec2 radius, inv_radius;
    Float filter_table[16][16];
    void init(const ftn_film_desc* d) {
        full_res[0] = d->full_resolution[0]; full_res[1] = d->full_resolution[1];
        crop = Bounds2i{d->crop[0], d->crop[1], d->crop[2], d->crop[3]};
        radius = Vec2(d->filter_radius[0], d->filter_radius[1]);
        inv_radius = Vec2(1.0f / radius.x, 1.0f / radius.y);                // BoxFilter::default: (2.0, 2.0)
        for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++) filter_table[y][x] = 1.0f;   // BoxFilter::evaluate
    }
    Bounds2i sample_bounds() const {                                        // :86-93
        return Bounds2i{f2i32(floorf((Float)crop.x0 + 0.5f - radius.x)), f2i32(floorf((Float)crop.y0 + 0.5f - radius.y)),
                        f2i32(ceilf((Float)crop.x1 - 0.5f + radius.x)), f2i32(ceilf((Float)crop.y1 - 0.5f + radius.y))};
    }
    FilmTile get_film_tile(Bounds2i sb) const {                             // :95-113
        int p0x = f2i32(ceilf((Float)sb.x0 - 0.5f - radius.x)), p0y = f2i32(ceilf((Float)sb.y0 - 0.5f - radius.y));
        int p1x = f2i32(ceilf((Float)sb.x1 - 0.5f + radius.x + 1.0f)), p1y = f2i32(ceilf((Float)sb.y1 - 0.5f - radius.y + 1.0f));
        FilmTile t; t.pixel_bounds = intersect2i(Bounds2i{p0x, p0y, p1x, p1y}, crop);
        t.filter_radius = radius; t.inv_filter_radius = inv_radius;
        t.pixels.assign((size_t)std::max(t.pixel_bounds.area(), 0), FilmTilePixel());
        return t;
    }
    size_t get_pixel_idx(int x, int y) const { int width = crop.x1 - crop.x0; return (size_t)((x - crop.x0) + (y - crop.y0) * width); }
    // add_sample_to_tile: :136-172.  Returns the number of pixels touched.
    int add_sample_to_tile(FilmTile& tile, Vec2 p_film, Spectrum radiance, Float sample_weight) const {
        Vec2 pd(p_film.x - 0.5f, p_film.y - 0.5f);
        int p0x = f2i32(ceilf(pd.x - tile.filter_radius.x)), p0y = f2i32(ceilf(pd.y - tile.filter_radius.y));
        int p1x = f2i32(floorf(pd.x + tile.filter_radius.x)) + 1, p1y = f2i32(floorf(pd.y + tile.filter_radius.y)) + 1;
        p0x = std::max(p0x, tile.pixel_bounds.x0); p0y = std::max(p0y, tile.pixel_bounds.y0);
        p1x = std::min(p1x, tile.pixel_bounds.x1); p1y = std::min(p1y, tile.pixel_bounds.y1);
        int touched = 0;
        for (int y = p0y; y < p1y; y++) {
            Float filt_y = fabsf(((Float)y - pd.y) * tile.inv_filter_radius.y * 16.0f);
            int y_idx = (int)std::min<int64_t>(f2usize(floorf(filt_y)), 15);
            for (int x = p0x; x < p1x; x++) {
                Float filt_x = fabsf(((Float)x - pd.x) * tile.inv_filter_radius.x * 16.0f);
                int x_idx = (int)std::min<int64_t>(f2usize(floorf(filt_x)), 15);
                Float filter_weight = filter_table[y_idx][x_idx];
                FilmTilePixel& px = tile.pixels[tile.get_pixel_idx(x, y)];
                px.contrib_sum += radiance * sample_weight * filter_weight;
                px.filter_weight_sum += filter_weight;
                touched++;
            }
        }
        return touched;
    }
    // merge_film_tile: :121-132 (caller serialises)
    void merge_film_tile(const FilmTile& tile, ftn_pixel* pixels) const {
        for (int y = tile.pixel_bounds.y0; y < tile.pixel_bounds.y1; y++)
            for (int x = tile.pixel_bounds.x0; x < tile.pixel_bounds.x1; x++) {
                const FilmTilePixel& tp = tile.pixels[tile.get_pixel_idx(x, y)];
                ftn_pixel& mp = pixels[get_pixel_idx(x, y)];
                Float xyz[3]; rgb_to_xyz(tp.contrib_sum.c, xyz);
                for (int i = 0; i < 3; i++) mp.xyz[i] += xyz[i];
                mp.filter_weight_sum += tp.filter_weight_sum;
            }
    }
};
// Film::new crop computation: :43-58
inline void make_film(const int32_t res[2], const Float cw[4], ftn_film_desc* out) {
    out->full_resolution[0] = res[0]; out->full_resolution[1] = res[1];
    out->crop[0] = f2i32(ceilf((Float)res[0] * cw[0])); out->crop[1] = f2i32(ceilf((Float)res[1] * cw[1]));
    out->crop[2] = f2i32(ceilf((Float)res[0] * cw[2])); out->crop[3] = f2i32(ceilf((Float)res[1] * cw[3]));
    out->filter_radius[0] = 0.5f; out->filter_radius[1] = 0.5f;
}
// into_spectrum_buffer: :195-210
inline void film_resolve(const ftn_pixel* p, size_t n, Float* rgb_out) {
    for (size_t i = 0; i < n; i++) {
        Float rgb[3]; xyz_to_rgb(p[i].xyz, rgb);
        if (p[i].filter_weight_sum != 0.0f) {
            Float inv_wt = 1.0f / p[i].filter_weight_sum;
            for (int c = 0; c < 3; c++) rgb[c] = fmax_(0.0f, rgb[c] * inv_wt);
        }
        rgb_out[3 * i] = rgb[0]; rgb_out[3 * i + 1] = rgb[1]; rgb_out[3 * i + 2] = rgb[2];
    }
}

// ---- integrators
struct Integrator {
    uint32_t kind; uint32_t max_depth; Float rr_threshold;
};

// estimate_direct: src/integrator/mod.rs:307-395
inline Spectrum estimate_direct(const Bsdf& bsdf, const SurfaceInteraction& isect, Vec2 u_scattering, const Light& light, int light_index,
                                Vec2 u_light, const SceneData& scene) {
    const uint8_t bsdf_flags = BSDF_ALL & ~BSDF_SPECULAR;
    Spectrum radiance(0.0f);
    LiSample ls = light.sample_incident_radiance(isect.hit, u_light);
    if (ls.pdf > 0.0f && !ls.radiance.is_black()) {
        Spectrum f = bsdf.f(isect.wo, ls.wi, bsdf_flags) * abs_dot(ls.wi, isect.shading_n);
        Float scattering_pdf = bsdf.pdf(isect.wo, ls.wi, bsdf_flags);
        if (!f.is_black() && scene.unoccluded(ls.p0, ls.p1)) {
            if (light.is_delta()) radiance += f * ls.radiance / ls.pdf;
            else {
                Float weight = power_heuristic(1, ls.pdf, 1, scattering_pdf);
                radiance += f * ls.radiance * weight / ls.pdf;
            }
        }
    }
    if (!light.is_delta()) {
        ScatterSample sc;
        if (bsdf.sample_f(isect.wo, u_scattering, bsdf_flags, &sc)) {
            Spectrum f = sc.f * abs_dot(sc.wi, isect.shading_n);
            bool sampled_specular = (sc.sampled_type & BSDF_SPECULAR) != 0;
            if (f.is_black()) return radiance;
            Float weight;
            if (sampled_specular) weight = 1.0f;
            else {
                Float light_pdf = light.pdf_incident_radiance(isect.hit, sc.wi);
                if (light_pdf == 0.0f) return radiance;
                weight = power_heuristic(1, sc.pdf, 1, light_pdf);
            }
            Ray ray = isect.hit.spawn_ray(sc.wi);
            SurfaceInteraction si2;
            Spectrum incident(0.0f);
            if (scene.intersect(ray, &si2)) {
                // area light of the hit primitive must be *this* light (pointer identity, :370-381)
                if (scene.bvh.prims[si2.prim].light >= 0 && scene.bvh.prims[si2.prim].light == light_index)
                    incident = scene.emitted_radiance(si2, -sc.wi);
            } else {
                incident = light.environment_emitted_radiance(ray);
            }
            if (!incident.is_black()) radiance += f * incident * weight / sc.pdf;
        }
    }
    return radiance;
}
// uniform_sample_one_light: :289-305
inline Spectrum uniform_sample_one_light(const SurfaceInteraction& isect, const Bsdf& bsdf, const SceneData& scene, Sampler& sampler) {
    size_t n_lights = scene.lights.size();
    if (n_lights == 0) return Spectrum(0.0f);
    size_t light_num = (size_t)f2usize(fmin_(sampler.get_1d() * (Float)n_lights, (Float)(n_lights - 1)));
    const Light& light = scene.lights[light_num];
    Vec2 u_light = sampler.get_2d();
    Vec2 u_scattering = sampler.get_2d();
    return (Float)n_lights * estimate_direct(bsdf, isect, u_scattering, light, (int)light_num, u_light, scene);
}

// compute_scattering_functions: interaction.rs:111-121; returns 0 = None (no material), 1 = Some, <0 = error
inline int scattering_functions(SurfaceInteraction& si, const RayDifferential& ray, const SceneData& scene, bool allow_multiple_lobes, Bsdf* bsdf) {
    si.compute_tex_differentials(ray);
    int mat = scene.bvh.prims[si.prim].material;
    if (mat < 0) return 0;
    const ftn_material resolved = scene.tex.resolve(scene.materials[mat], scene.mtex.empty() ? nullptr : &scene.mtex[mat], si);   // Texture::evaluate(si)
    if (compute_scattering_functions(resolved, si, allow_multiple_lobes, bsdf) != MAT_OK) { scene.error.store(FTN_ERR_UNSUPPORTED); return -1; }
    return 1;
}

// PathIntegrator::incident_radiance: src/integrator/path.rs:25-95
inline Spectrum path_li(const Integrator& it, RayDifferential& ray, const SceneData& scene, Sampler& sampler) {
    Spectrum path_radiance(0.0f), throughput(1.0f);
    uint32_t bounces = 0; bool specular_bounce = false;
    for (;;) {
        SurfaceInteraction si; bool hit = scene.intersect(ray.ray, &si);
        if (bounces == 0 || specular_bounce) {
            if (hit) path_radiance += throughput * scene.emitted_radiance(si, -ray.ray.dir);
            else path_radiance += throughput * scene.environment_emitted_radiance(ray.ray);
        }
        if (!hit || bounces >= it.max_depth) break;
        Bsdf bsdf;
        int r = scattering_functions(si, ray, scene, true, &bsdf);
        if (r < 0) break;
        if (r == 1) {
            if (bsdf.num_components(BSDF_ALL & ~BSDF_SPECULAR) > 0) {
                Spectrum direct = throughput * uniform_sample_one_light(si, bsdf, scene, sampler);
                path_radiance += direct;
            }
            Vec3 wo = -ray.ray.dir;
            ScatterSample bs;
            bool ok = bsdf.sample_f(wo, sampler.get_2d(), BSDF_ALL, &bs);
            if (ok && !bs.f.is_black()) {
                throughput *= bs.f * abs_dot(bs.wi, si.shading_n) / bs.pdf;
                specular_bounce = (bs.sampled_type & BSDF_SPECULAR) != 0;
                ray.ray = si.hit.spawn_ray(bs.wi);                          // differentials carried over unchanged
            } else break;
        } else {
            ray.ray = si.hit.spawn_ray(ray.ray.dir);                        // null bsdf: no bounce++
            continue;
        }
        if (throughput.max_component_value() < it.rr_threshold && bounces > 3) {
            Float q = fmax_(0.05f, 1.0f - throughput.max_component_value());
            if (sampler.get_1d() < q) break;
            else throughput /= 1.0f - q;
        }
        bounces += 1;
    }
    return path_radiance;
}

// DirectLightingIntegrator (UniformSampleOne): src/integrator/direct_lighting.rs:50-110 + specular_reflect/transmit mod.rs:39-178
inline Spectrum direct_li(const Integrator& it, RayDifferential& ray, const SceneData& scene, Sampler& sampler, uint32_t depth);
inline Spectrum specular_bounce_li(const Integrator& it, const RayDifferential& ray, const SurfaceInteraction& isect, const Bsdf& bsdf,
                                   const SceneData& scene, Sampler& sampler, uint32_t depth, uint8_t flags) {
    Vec3 wo = isect.wo;
    ScatterSample sc;
    Vec2 u = sampler.get_2d();                                              // evaluated before the match, mod.rs:52 / :113
    if (!bsdf.sample_f(wo, u, flags, &sc)) return Spectrum(0.0f);
    if (abs_dot(sc.wi, isect.shading_n) == 0.0f) return Spectrum(0.0f);
    RayDifferential child; child.ray = isect.hit.spawn_ray(sc.wi); child.has_diff = ray.has_diff;
    if (ray.has_diff) {                                                     // ray.diff.map(..): mod.rs:58-84 (reflect) / :119-163 (transmit)
        const TextureDifferentials& td = isect.tex_diffs; const DiffGeom& sh = isect.shading_geom; const Differential& diff = ray.diff;
        Differential o;
        o.rx_origin = isect.hit.p + td.dpdx; o.ry_origin = isect.hit.p + td.dpdy;
        Vec3 dndx = sh.dndu * td.dudx + sh.dndv * td.dvdx, dndy = sh.dndu * td.dudy + sh.dndv * td.dvdy;
        Vec3 ns = isect.shading_n;
        if (flags & BSDF_REFLECTION) {
            Vec3 dwo_dx = -diff.rx_dir - wo, dwo_dy = -diff.ry_dir - wo;
            Float dDN_dx = dot(dwo_dx, ns) + dot(wo, dndx), dDN_dy = dot(dwo_dy, ns) + dot(wo, dndy);
            o.rx_dir = (sc.wi - dwo_dx) + (2.0f * dot(wo, ns)) * dndx + dDN_dx * ns;
            o.ry_dir = (sc.wi - dwo_dy) + (2.0f * dot(wo, ns)) * dndy + dDN_dy * ns;
        } else {
            Vec3 sn = ns; Float eta = 1.0f / bsdf.eta;
            if (dot(wo, ns) < 0.0f) { eta = bsdf.eta; sn = -sn; dndx = -dndx; dndy = -dndy; }
            Vec3 dwo_dx = -diff.rx_dir - wo, dwo_dy = -diff.ry_dir - wo;
            Float dDN_dx = dot(dwo_dx, ns) + dot(wo, dndx), dDN_dy = dot(dwo_dy, ns) + dot(wo, dndy);   // unflipped normal, flipped dndx (:142-143)
            Float mu = eta * dot(wo, sn) - abs_dot(sc.wi, sn);
            Float dmu_dx = (eta - (eta * eta * dot(wo, sn)) / dot(sc.wi, sn)) * dDN_dx;
            Float dmu_dy = (eta - (eta * eta * dot(wo, sn)) / dot(sc.wi, sn)) * dDN_dy;
            o.rx_dir = sc.wi - (eta * dwo_dx) + (mu * dndx + dmu_dx * sn);
            o.ry_dir = sc.wi - (eta * dwo_dy) + (mu * dndy + dmu_dy * sn);
        }
        child.diff = o;
    }
    Spectrum li = direct_li(it, child, scene, sampler, depth + 1);
    return sc.f * li * fabsf(dot(sc.wi, isect.shading_n)) / sc.pdf;
}
inline Spectrum direct_li(const Integrator& it, RayDifferential& ray, const SceneData& scene, Sampler& sampler, uint32_t depth) {
    Spectrum radiance(0.0f);
    SurfaceInteraction isect;
    if (!scene.intersect(ray.ray, &isect)) return scene.environment_emitted_radiance(ray.ray);
    Bsdf bsdf;
    int r = scattering_functions(isect, ray, scene, false, &bsdf);
    if (r <= 0) { if (r == 0) scene.error.store(FTN_ERR_UNSUPPORTED); return radiance; }   // unimplemented!() :103
    if (it.kind == FTN_INTEGRATOR_WHITTED) {               // whitted.rs:42-58: every light, one 2D sample each, no emission term
        for (size_t li = 0; li < scene.lights.size(); li++) {
            const Light& light = scene.lights[li];
            LiSample ls = light.sample_incident_radiance(isect.hit, sampler.get_2d());
            if (ls.radiance.is_black() || ls.pdf == 0.0f) continue;
            Spectrum f = bsdf.f(isect.wo, ls.wi, BSDF_ALL);
            if (!f.is_black() && scene.unoccluded(ls.p0, ls.p1)) radiance += f * ls.radiance * abs_dot(ls.wi, isect.shading_n) / ls.pdf;
        }
    } else {
        radiance += scene.emitted_radiance(isect, isect.wo);
        radiance += uniform_sample_one_light(isect, bsdf, scene, sampler);
    }
    if (depth + 1 < it.max_depth) {
        radiance += specular_bounce_li(it, ray, isect, bsdf, scene, sampler, depth, BSDF_REFLECTION | BSDF_SPECULAR);
        radiance += specular_bounce_li(it, ray, isect, bsdf, scene, sampler, depth, BSDF_TRANSMISSION | BSDF_SPECULAR);
    }
    return radiance;
}

// ---- SamplerIntegrator::render_tile: src/integrator/mod.rs:229-281
struct TileStats { uint64_t camera_samples = 0, spill_samples = 0; };
inline void render_tile(const SceneData& scene, const Camera& camera, const Film& film, const Integrator& it,
                        Sampler tile_sampler, Bounds2i tile, FilmTile* film_tile_out, TileStats* ts) {
    FilmTile film_tile = film.get_film_tile(tile);
    for (int y = tile.y0; y < tile.y1; y++)
        for (int x = tile.x0; x < tile.x1; x++) {
            tile_sampler.start_pixel(x, y);
            while (tile_sampler.start_next_sample()) {
                CameraSample cs = tile_sampler.get_camera_sample(x, y);
                Float ray_weight;
                RayDifferential rd = camera.generate_ray_differential(cs, &ray_weight);
                rd.scale_differentials(1.0f / sqrtf((Float)tile_sampler.samples_per_pixel));
                Spectrum radiance(0.0f);
                if (ray_weight > 0.0f) {
                    radiance = (it.kind == FTN_INTEGRATOR_DIRECT_LIGHTING || it.kind == FTN_INTEGRATOR_WHITTED) ? direct_li(it, rd, scene, tile_sampler, 0)
                                                                           : path_li(it, rd, scene, tile_sampler);
                    if (radiance.has_nans()) scene.error.store(FTN_ERR_NAN_RADIANCE);   // check_radiance :285-287
                }
                int touched = film.add_sample_to_tile(film_tile, cs.p_film, radiance, ray_weight);
                ts->camera_samples++;
                if (touched != 1) ts->spill_samples++;
            }
        }
    *film_tile_out = std::move(film_tile);
}

}  // namespace orc
