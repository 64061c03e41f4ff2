// ORACLE -- TEST INFRASTRUCTURE ONLY.  C entry points (orc_*) of the CPU restatement; twins of the ftn_*
// functions declared in include/fountain_hip.h, plus orc_kat_* hooks for the reference's known-answer tests.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
#include "orc_render.hpp"
#include <thread>
#include <mutex>
#include <chrono>
#include <string>
#include <functional>

using namespace orc;

struct orc_scene { SceneData data; };

static thread_local std::string g_err;
static int fail(int code, const char* msg) { g_err = msg; return code; }

static Transform from_abi(const ftn_transform* t) { return Transform::make(Mat4::from_flat(t->m), Mat4::from_flat(t->inv)); }
static void to_abi(const Transform& t, ftn_transform* o) { memcpy(o->m, t.t.a, 64); memcpy(o->inv, t.invt.a, 64); }

extern "C" {

const char* orc_last_error(void) { return g_err.c_str(); }
int orc_uses_detmath(void) {
#ifdef ORC_DETMATH
    return 1;
#else
    return 0;
#endif
}

// ---------------------------------------------------------------- host constructors
int orc_transform_identity(ftn_transform* out) { to_abi(Transform::identity(), out); return 0; }
int orc_transform_translate(const float d[3], ftn_transform* out) { to_abi(tf_translate(Vec3(d[0], d[1], d[2])), out); return 0; }
int orc_transform_scale(float sx, float sy, float sz, ftn_transform* out) { to_abi(tf_scale(sx, sy, sz), out); return 0; }
int orc_transform_rotate(float deg, const float a[3], ftn_transform* out) {
    Transform t; if (!tf_rotate(deg, Vec3(a[0], a[1], a[2]), &t)) return fail(FTN_ERR_INVALID_ARGUMENT, "singular"); to_abi(t, out); return 0; }
int orc_transform_look_at(const float p[3], const float l[3], const float u[3], ftn_transform* out) {
    Transform t; if (!tf_look_at(Vec3(p[0], p[1], p[2]), Vec3(l[0], l[1], l[2]), Vec3(u[0], u[1], u[2]), &t)) return fail(FTN_ERR_INVALID_ARGUMENT, "singular"); to_abi(t, out); return 0; }
int orc_transform_from_flat(const float m[16], ftn_transform* out) {
    Transform t; if (!Transform::from_mat(Mat4::from_flat(m), &t)) return fail(FTN_ERR_INVALID_ARGUMENT, "singular"); to_abi(t, out); return 0; }
int orc_transform_mul(const ftn_transform* a, const ftn_transform* b, ftn_transform* out) { to_abi(from_abi(a) * from_abi(b), out); return 0; }
int orc_transform_inverse(const ftn_transform* a, ftn_transform* out) { to_abi(from_abi(a).inverse(), out); return 0; }
int orc_transform_perspective(float fov, float n, float f, ftn_transform* out) {
    Transform t; if (!tf_perspective(fov, n, f, &t)) return fail(FTN_ERR_INVALID_ARGUMENT, "singular"); to_abi(t, out); return 0; }
int orc_transform_point(const ftn_transform* t, const float p[3], float o[3]) { Vec3 r = tf_point(from_abi(t), Vec3(p[0], p[1], p[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; return 0; }
int orc_transform_vector(const ftn_transform* t, const float p[3], float o[3]) { Vec3 r = tf_vector(from_abi(t), Vec3(p[0], p[1], p[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; return 0; }
int orc_transform_normal(const ftn_transform* t, const float p[3], float o[3]) { Vec3 r = tf_normal(from_abi(t), Vec3(p[0], p[1], p[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; return 0; }
int orc_transform_points(const ftn_transform* t, size_t n, const float* in, float* out) { Transform T = from_abi(t); for (size_t i = 0; i < n; i++) { Vec3 r = tf_point(T, Vec3(in[3*i], in[3*i+1], in[3*i+2])); out[3*i] = r.x; out[3*i+1] = r.y; out[3*i+2] = r.z; } return 0; }
int orc_transform_normals(const ftn_transform* t, size_t n, const float* in, float* out) { Transform T = from_abi(t); for (size_t i = 0; i < n; i++) { Vec3 r = tf_normal(T, Vec3(in[3*i], in[3*i+1], in[3*i+2])); out[3*i] = r.x; out[3*i+1] = r.y; out[3*i+2] = r.z; } return 0; }
int orc_transform_swaps_handedness(const ftn_transform* t) { return from_abi(t).swaps_handedness() ? 1 : 0; }

int orc_sphere_init(const ftn_transform* o2w, const ftn_transform* w2o, int rev, float radius, float z_min, float z_max, float phi_max_deg, ftn_sphere* out) {
    Sphere s = Sphere::make(from_abi(o2w), from_abi(w2o), rev != 0, radius, z_min, z_max, phi_max_deg);
    memset(out, 0, sizeof(*out));
    out->object_to_world = *o2w; out->world_to_object = *w2o;
    out->radius = s.radius; out->z_min = s.z_min; out->z_max = s.z_max; out->theta_min = s.theta_min; out->theta_max = s.theta_max; out->phi_max = s.phi_max;
    out->reverse_orientation = rev ? 1 : 0;
    return 0;
}
int orc_camera_perspective(const ftn_transform* c2w, const int32_t res[2], const float sw[4], const float sh[2], float lens_radius, float focal_dist, float fov, ftn_camera_desc* out) {
    memset(out, 0, sizeof(*out));
    if (!make_perspective_camera(from_abi(c2w), res[0], res[1], sw, sh[0], sh[1], lens_radius, focal_dist, fov, out)) return fail(FTN_ERR_INVALID_ARGUMENT, "singular");
    return 0;
}
int orc_film_init(const int32_t res[2], const float cw[4], ftn_film_desc* out) { make_film(res, cw, out); return 0; }
int orc_film_sample_bounds(const ftn_film_desc* f, int32_t out[4]) { Film film; film.init(f); Bounds2i b = film.sample_bounds(); out[0] = b.x0; out[1] = b.y0; out[2] = b.x1; out[3] = b.y1; return 0; }
static void iter_tiles(Bounds2i b, int tile_size, std::vector<Bounds2i>* tiles) {            // Bounds2i::iter_tiles, bounds.rs:85-97
    for (int y = b.y0; y < b.y1; y += tile_size) for (int x = b.x0; x < b.x1; x += tile_size)
        tiles->push_back(Bounds2i{x, y, std::min(x + tile_size, b.x1), std::min(y + tile_size, b.y1)});
}
static void list_tiles(const Film& film, std::vector<Bounds2i>* tiles, Bounds2i* sb_out) {   // render_parallel: sample_bounds.iter_tiles(16), integrator/mod.rs:218-227
    Bounds2i sb = film.sample_bounds(); *sb_out = sb;
    iter_tiles(sb, 16, tiles);
}
int orc_film_tile_count(const ftn_film_desc* f, uint32_t* out) { Film film; film.init(f); std::vector<Bounds2i> t; Bounds2i sb; list_tiles(film, &t, &sb); *out = (uint32_t)t.size(); return 0; }
int orc_film_resolve(const ftn_pixel* p, size_t n, float* rgb) { film_resolve(p, n, rgb); return 0; }

// ---------------------------------------------------------------- scene
int orc_scene_create(const ftn_scene_desc* d, orc_scene** out) {
    orc_scene* s = new orc_scene();
    int rc = build_scene(d, &s->data);
    if (rc != 0) { delete s; return fail(rc, "invalid scene description"); }
    *out = s; return 0;
}
void orc_scene_destroy(orc_scene* s) { delete s; }
int orc_scene_info(const orc_scene* s, uint32_t* n_nodes, uint32_t* n_prims, uint32_t* n_lights, uint32_t* max_depth, float wb[6]) {
    const BVH& b = s->data.bvh;
    if (n_nodes) *n_nodes = (uint32_t)b.nodes.size();
    if (n_prims) *n_prims = (uint32_t)b.prims.size();
    if (n_lights) *n_lights = (uint32_t)s->data.lights.size();
    if (max_depth) *max_depth = b.max_depth;
    if (wb) for (int i = 0; i < 3; i++) { wb[i] = b.bounds.min[i]; wb[3 + i] = b.bounds.max[i]; }
    return 0;
}
int orc_scene_get_nodes(const orc_scene* s, ftn_bvh_node* nodes, uint32_t* order) {
    const BVH& b = s->data.bvh;
    if (nodes) memcpy(nodes, b.nodes.data(), b.nodes.size() * sizeof(ftn_bvh_node));
    if (order) for (size_t i = 0; i < b.prims.size(); i++) order[i] = b.prims[i].original_index;
    return 0;
}
// area-light table in Scene::lights order: per light {kind, prim (BVH order) or -1}
int orc_scene_get_lights(const orc_scene* s, int32_t* kind, int32_t* prim) {
    const SceneData& d = s->data;
    for (size_t i = 0; i < d.lights.size(); i++) { kind[i] = (int)d.lights[i].kind; prim[i] = -1; }
    for (size_t i = 0; i < d.bvh.prims.size(); i++) if (d.bvh.prims[i].light >= 0) prim[d.bvh.prims[i].light] = (int)i;
    return 0;
}

static Ray ray_from(const float* r) { Ray ray; ray.origin = Vec3(r[0], r[1], r[2]); ray.dir = Vec3(r[3], r[4], r[5]); ray.t_max = r[6]; ray.time = r[7]; return ray; }

static void parallel_for(size_t n, int n_threads, const std::function<void(size_t, size_t)>& fn) {
    if (n_threads <= 1 || n < 1024) { fn(0, n); return; }
    std::vector<std::thread> th; size_t chunk = (n + n_threads - 1) / n_threads;
    for (int t = 0; t < n_threads; t++) { size_t a = t * chunk, b = std::min(n, a + chunk); if (a < b) th.emplace_back(fn, a, b); }
    for (auto& t : th) t.join();
}

int orc_intersect(const orc_scene* s, const float* rays, size_t n, float* t_hit, int32_t* prim, float* bary, ftn_stats* stats, int n_threads) {
    const SceneData& d = s->data;
    std::atomic<uint64_t> nodes{0}, prims{0};
    parallel_for(n, n_threads, [&](size_t a, size_t b) {
        TraversalCounters c;
        for (size_t i = a; i < b; i++) {
            Ray ray = ray_from(rays + 8 * i); SurfaceInteraction si; bool deep = false;
            bool hit = d.bvh.intersect(ray, &si, &c, &deep);
            if (t_hit) t_hit[i] = hit ? ray.t_max : INF;
            if (prim) prim[i] = hit ? si.prim : -1;
            (void)bary;
        }
        nodes += c.nodes_visited; prims += c.prims_tested;
    });
    if (stats) { memset(stats, 0, sizeof(*stats)); stats->rays_closest = n; stats->nodes_visited = nodes; stats->prims_tested = prims; }
    return 0;
}
int orc_intersect_test(const orc_scene* s, const float* rays, size_t n, uint8_t* occluded, ftn_stats* stats, int n_threads) {
    const SceneData& d = s->data;
    std::atomic<uint64_t> nodes{0}, prims{0};
    parallel_for(n, n_threads, [&](size_t a, size_t b) {
        TraversalCounters c;
        for (size_t i = a; i < b; i++) { Ray ray = ray_from(rays + 8 * i); bool deep = false; occluded[i] = d.bvh.intersect_test(ray, &c, &deep) ? 1 : 0; }
        nodes += c.nodes_visited; prims += c.prims_tested;
    });
    if (stats) { memset(stats, 0, sizeof(*stats)); stats->rays_any = n; stats->nodes_visited = nodes; stats->prims_tested = prims; }
    return 0;
}
// p[3] p_err[3] n[3] uv[2] wo[3] dpdu[3] dpdv[3] shading_n[3] t
int orc_intersect_full(const orc_scene* s, const float* rays, size_t n, float* out24, int n_threads) {
    const SceneData& d = s->data;
    parallel_for(n, n_threads, [&](size_t a, size_t b) {
        for (size_t i = a; i < b; i++) {
            Ray ray = ray_from(rays + 8 * i); SurfaceInteraction si; bool deep = false; float* o = out24 + 24 * i;
            if (!d.bvh.intersect(ray, &si, nullptr, &deep)) { for (int k = 0; k < 24; k++) o[k] = 0.0f; o[23] = -1.0f; continue; }
            Vec3 v[] = {si.hit.p, si.hit.p_err, si.hit.n};
            for (int k = 0; k < 3; k++) for (int c = 0; c < 3; c++) o[3 * k + c] = v[k][c];
            o[9] = si.uv.x; o[10] = si.uv.y;
            Vec3 w[] = {si.wo, si.shading_geom.dpdu /* what Bsdf::new builds its frame from (bsdf.rs:24-26) */, si.geom.dpdv, si.shading_n};
            for (int k = 0; k < 4; k++) for (int c = 0; c < 3; c++) o[11 + 3 * k + c] = w[k][c];
            o[23] = ray.t_max;
        }
    });
    return 0;
}

// ---------------------------------------------------------------- render
int orc_render(const orc_scene* s, const ftn_camera_desc* cam, const ftn_film_desc* fd, const ftn_sampler_desc* sd,
               const ftn_integrator_desc* id, const ftn_tile_range* tr, int n_threads, int count_traffic,
               ftn_pixel* out_pixels, ftn_stats* stats) {
    SceneData& scene = const_cast<SceneData&>(s->data);
    scene.count_traffic = count_traffic != 0;
    scene.rays_closest = 0; scene.rays_any = 0; scene.nodes_visited = 0; scene.prims_tested = 0; scene.nodes_any = 0; scene.prims_any = 0; scene.error = 0;
    Camera camera;
    camera.camera_to_world = from_abi(&cam->camera_to_world); camera.raster_to_camera = from_abi(&cam->raster_to_camera);
    camera.shutter_open = cam->shutter_open; camera.shutter_close = cam->shutter_close;
    camera.lens_radius = cam->lens_radius; camera.focal_dist = cam->focal_dist;
    camera.dx_camera = Vec3(cam->dx_camera[0], cam->dx_camera[1], cam->dx_camera[2]);
    camera.dy_camera = Vec3(cam->dy_camera[0], cam->dy_camera[1], cam->dy_camera[2]);
    Film film; film.init(fd);
    Integrator it{id->kind, id->max_depth, id->rr_threshold};
    Sampler base; base.kind = sd->kind; base.base_seed = sd->seed; base.samples_per_pixel = sd->samples_per_pixel;
    base.rng = Xoshiro256Plus::seed_from_u64(sd->seed);
    if (sd->kind == FTN_SAMPLER_INDEXED && ((uint64_t)sd->first_sample > (uint64_t)sd->samples_per_pixel || (uint64_t)sd->first_sample + (uint64_t)sd->sample_count > (uint64_t)sd->samples_per_pixel))
        return fail(FTN_ERR_INVALID_ARGUMENT, "sample range outside [0, samples_per_pixel]");     /* same contract as ftn_render_device */
    base.first_sample = sd->first_sample;
    base.last_sample = sd->first_sample + (sd->sample_count ? sd->sample_count : (sd->samples_per_pixel - sd->first_sample));
    if (sd->kind == FTN_SAMPLER_TILE_SERIAL && (sd->first_sample != 0 || (sd->sample_count != 0 && sd->sample_count != sd->samples_per_pixel)))
        return fail(FTN_ERR_INVALID_ARGUMENT, "sample ranges need FTN_SAMPLER_INDEXED");

    std::vector<Bounds2i> all_tiles; Bounds2i sb; list_tiles(film, &all_tiles, &sb);
    std::vector<size_t> sel;
    uint32_t stride = tr && tr->stride ? tr->stride : 1, first = tr ? tr->first : 0, count = tr ? tr->count : 0;
    for (size_t i = first, k = 0; i < all_tiles.size() && (count == 0 || k < count); i += stride, k++) sel.push_back(i);

    std::vector<FilmTile> done(sel.size());
    std::vector<TileStats> tstats(sel.size());
    std::atomic<size_t> next{0};
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&]() {
        for (;;) {
            size_t k = next.fetch_add(1); if (k >= sel.size()) break;
            Bounds2i tile = all_tiles[sel[k]];
            uint64_t tile_id = (uint64_t)(int64_t)(tile.y0 * sb.x1 + tile.x0);   // tile_id: integrator/mod.rs:182-185
            render_tile(scene, camera, film, it, base.clone_with_seed(tile_id), tile, &done[k], &tstats[k]);
        }
    };
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads <= 1) worker();
    else { std::vector<std::thread> th; for (int t = 0; t < n_threads; t++) th.emplace_back(worker); for (auto& t : th) t.join(); }
    auto t1 = std::chrono::steady_clock::now();
    // merge in tile order (Film::render's order; render_parallel's order is scheduling dependent)
    uint64_t cs = 0, spill = 0;
    for (size_t k = 0; k < sel.size(); k++) { film.merge_film_tile(done[k], out_pixels); cs += tstats[k].camera_samples; spill += tstats[k].spill_samples; }
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->rays_closest = scene.rays_closest; stats->rays_any = scene.rays_any;
        stats->nodes_visited = scene.nodes_visited; stats->prims_tested = scene.prims_tested;
        stats->nodes_visited_any = scene.nodes_any; stats->prims_tested_any = scene.prims_any;
        stats->camera_samples = cs; stats->spill_samples = spill;
        stats->kernel_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    }
    int err = scene.error.load();
    if (err) return fail(err, err == FTN_ERR_NAN_RADIANCE ? "NaN radiance" : (err == FTN_ERR_BVH_TOO_DEEP ? "BVH deeper than 64" : "unsupported material configuration"));
    return 0;
}

// ---------------------------------------------------------------- known-answer hooks (reference unit tests)
float orc_kat_fresnel_dielectric(float cos_i, float eta_i, float eta_t) { return fresnel_dielectric(cos_i, eta_i, eta_t); }   // fresnel.rs:110-116
void orc_kat_fresnel_conductor(float cos_i, const float eta[3], const float k[3], float out[3]) {
    Spectrum r = fresnel_conductor(fabsf(cos_i), Spectrum(1.0f), Spectrum(eta[0], eta[1], eta[2]), Spectrum(k[0], k[1], k[2]));
    out[0] = r[0]; out[1] = r[1]; out[2] = r[2]; }
int orc_kat_bounds_intersect(const float bmin[3], const float bmax[3], const float ray8[8], float out[2]) {   // bounds.rs:292-323
    Bounds3 b; b.min = Vec3(bmin[0], bmin[1], bmin[2]); b.max = Vec3(bmax[0], bmax[1], bmax[2]);
    return b.intersect_test(ray_from(ray8), &out[0], &out[1]) ? 1 : 0; }
int orc_kat_solve_2x2(const float a[4], const float b[2], float x[2]) { return solve_linear_system_2x2(a[0], a[1], a[2], a[3], b[0], b[1], &x[0], &x[1]) ? 1 : 0; }   // math.rs:88-102 (Matrix2::new(c0r0,c0r1,c1r0,c1r1))
int orc_kat_sign_differs(float a, float b, float c) { return sign_differs(a, b, c) ? 1 : 0; }                     // triangle.rs:441-450
void orc_kat_distribution1d(const float* f, size_t n, float u, float out[3]) {                                    // sampling.rs:188-198
    Distribution1D d; d.init(f, n); Float x, pdf; size_t idx; d.sample_continuous(u, &x, &pdf, &idx); out[0] = x; out[1] = pdf; out[2] = (float)idx; }
void orc_kat_concentric_disk(float u0, float u1, float out[2]) { Vec2 d = concentric_sample_disk(Vec2(u0, u1)); out[0] = d.x; out[1] = d.y; }   // sampling.rs:200-208
void orc_kat_cosine_hemisphere(float u0, float u1, float out[3]) { Vec3 d = cosine_sample_hemisphere(Vec2(u0, u1)); out[0] = d.x; out[1] = d.y; out[2] = d.z; }
void orc_kat_xoshiro(const uint64_t s[4], uint64_t* out, size_t n) { Xoshiro256Plus r; for (int i = 0; i < 4; i++) r.s[i] = s[i]; for (size_t i = 0; i < n; i++) out[i] = r.next_u64(); }
void orc_kat_splitmix(uint64_t seed, uint64_t* out, size_t n) { SplitMix64 r{seed}; for (size_t i = 0; i < n; i++) out[i] = r.next_u64(); }
void orc_kat_sampler_f32(uint64_t seed, float* out, size_t n) { Xoshiro256Plus r = Xoshiro256Plus::seed_from_u64(seed); for (size_t i = 0; i < n; i++) out[i] = r.gen_f32(); }
void orc_kat_indexed_f32(uint64_t seed, int32_t px, int32_t py, uint32_t sample, float* out, size_t n) {
    Xoshiro256Plus r = Xoshiro256Plus::seed_from_u64(indexed_sample_key(seed, px, py, sample)); for (size_t i = 0; i < n; i++) out[i] = r.gen_f32(); }
void orc_kat_permutation(int64_t* items, const int64_t* perm, size_t n) { std::vector<int64_t> o(n); for (size_t i = 0; i < n; i++) o[i] = items[perm[i]]; memcpy(items, o.data(), n * 8); }   // bvh.rs:391-398
float orc_kat_next_float_up(float v) { return next_float_up(v); }
float orc_kat_next_float_down(float v) { return next_float_down(v); }
float orc_kat_gamma(int n) { return gamma(n); }
// camera ray for the camera tests (camera/mod.rs:220-366): out = o[3] d[3]
void orc_kat_camera_ray(const ftn_camera_desc* cam, const float sample5[5], float out[6]) {
    Camera c; c.camera_to_world = from_abi(&cam->camera_to_world); c.raster_to_camera = from_abi(&cam->raster_to_camera);
    c.shutter_open = cam->shutter_open; c.shutter_close = cam->shutter_close; c.lens_radius = cam->lens_radius; c.focal_dist = cam->focal_dist;
    c.dx_camera = Vec3(cam->dx_camera[0], cam->dx_camera[1], cam->dx_camera[2]); c.dy_camera = Vec3(cam->dy_camera[0], cam->dy_camera[1], cam->dy_camera[2]);
    CameraSample cs; cs.p_film = Vec2(sample5[0], sample5[1]); cs.p_lens = Vec2(sample5[2], sample5[3]); cs.time = sample5[4];
    Float w; RayDifferential rd = c.generate_ray_differential(cs, &w);
    out[0] = rd.ray.origin.x; out[1] = rd.ray.origin.y; out[2] = rd.ray.origin.z; out[3] = rd.ray.dir.x; out[4] = rd.ray.dir.y; out[5] = rd.ray.dir.z;
}
// Transform::tf_err_to_err for points / vectors (transform.rs:411-437): out = t[3] err[3]
void orc_kat_tf_err(const ftn_transform* t, const float p[3], const float e[3], int is_point, float out[6]) {
    Vec3 err; Vec3 r = is_point ? tf_point_err_to_err(from_abi(t), Vec3(p[0], p[1], p[2]), Vec3(e[0], e[1], e[2]), &err)
                                : tf_vector_err_to_err(from_abi(t), Vec3(p[0], p[1], p[2]), Vec3(e[0], e[1], e[2]), &err);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = err.x; out[4] = err.y; out[5] = err.z;
}
// Ray::transform (transform.rs:307-322): in/out 8 floats
void orc_kat_tf_ray(const ftn_transform* t, const float in8[8], float out8[8]) {
    Ray r = tf_ray(from_abi(t), ray_from(in8));
    out8[0] = r.origin.x; out8[1] = r.origin.y; out8[2] = r.origin.z; out8[3] = r.dir.x; out8[4] = r.dir.y; out8[5] = r.dir.z; out8[6] = r.t_max; out8[7] = r.time;
}
// single-shape sphere intersect (sphere.rs:241-273): returns hit; out = t, p_err[3], p[3]
int orc_kat_sphere_intersect(const ftn_sphere* sp, const float ray8[8], float out[7]) {
    Sphere s; s.object_to_world = from_abi(&sp->object_to_world); s.world_to_object = from_abi(&sp->world_to_object);
    s.reverse_orientation = sp->reverse_orientation; s.radius = sp->radius; s.z_min = sp->z_min; s.z_max = sp->z_max;
    s.theta_min = sp->theta_min; s.theta_max = sp->theta_max; s.phi_max = sp->phi_max;
    Float t; SurfaceInteraction si;
    if (!s.intersect(ray_from(ray8), &t, &si)) return 0;
    out[0] = t; out[1] = si.hit.p_err.x; out[2] = si.hit.p_err.y; out[3] = si.hit.p_err.z; out[4] = si.hit.p.x; out[5] = si.hit.p.y; out[6] = si.hit.p.z;
    return 1;
}
float orc_kat_math(int which, float x, float y) {
    switch (which) { case 0: return m_sin(x); case 1: return m_cos(x); case 2: return m_tan(x); case 3: return m_acos(x);
                     case 4: return m_atan(x); case 5: return m_atan2(x, y); case 6: return m_ln(x); case 7: return m_log2(x); case 13: return m_pow(x, y); default: return 0.0f; }
}
float orc_kat_roughness_to_alpha(float r) { return roughness_to_alpha(r); }
// imageio/mod.rs:169-175 inverse_gamma_correct, applied by load_mipmap (imageio/mod.rs:101-107) to every channel of a gamma-encoded map
static inline Float inverse_gamma_correct(Float v) { return v <= 0.04045f ? v * 1.0f / 12.92f : m_pow((v + 0.055f) * 1.0f / 1.055f, 2.4f); }
int orc_image_inverse_gamma(float* texels, size_t n) { if (!texels && n) return FTN_ERR_INVALID_ARGUMENT; for (size_t i = 0; i < n; i++) texels[i] = inverse_gamma_correct(texels[i]); return FTN_OK; }
// Bounds2i::iter_points / iter_tiles (bounds.rs:76-97): out holds 2 ints per point / 4 per tile; returns the count
size_t orc_kat_iter_points(const int32_t b[4], int32_t* out, size_t cap) {
    size_t n = 0;
    for (int y = b[1]; y < b[3]; y++) for (int x = b[0]; x < b[2]; x++) { if (n < cap) { out[2 * n] = x; out[2 * n + 1] = y; } n++; }
    return n;
}
size_t orc_kat_iter_tiles(const int32_t b[4], int32_t tile_size, int32_t* out, size_t cap) {
    std::vector<Bounds2i> t; iter_tiles(Bounds2i{b[0], b[1], b[2], b[3]}, tile_size, &t);
    for (size_t i = 0; i < t.size() && i < cap; i++) { out[4 * i] = t[i].x0; out[4 * i + 1] = t[i].y0; out[4 * i + 2] = t[i].x1; out[4 * i + 3] = t[i].y1; }
    return t.size();
}
// ---- textures (orc_texture.hpp)
// a level of the pyramid MIPMap::new builds (mipmap.rs:78-145)
int orc_test_mipmap_level(uint32_t w, uint32_t h, const float* texels, uint32_t level, uint32_t* lw, uint32_t* lh, float* rgb_out) {
    MIPMap m = MIPMap::make((int)w, (int)h, texels, FTN_WRAP_REPEAT);
    if ((int)level >= m.levels()) return FTN_ERR_INVALID_ARGUMENT;
    const MipLevel& L = m.pyramid[level];
    if (lw) *lw = (uint32_t)L.w;
    if (lh) *lh = (uint32_t)L.h;
    if (rgb_out) for (size_t i = 0; i < L.data.size(); i++) for (int c = 0; c < 3; c++) rgb_out[3 * i + c] = L.data[i][c];
    return FTN_OK;
}
// Texture::evaluate for rows of {u, v, dudx, dvdx, dudy, dvdy}
int orc_test_texture_eval(const orc_scene* s, int32_t texture, const float* in6, size_t n, float* out3) {
    if (texture < 0 || (size_t)texture >= s->data.tex.textures.size()) return FTN_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < n; i++) {
        SurfaceInteraction si; si.uv = Vec2(in6[6 * i], in6[6 * i + 1]);
        si.tex_diffs.dudx = in6[6 * i + 2]; si.tex_diffs.dvdx = in6[6 * i + 3]; si.tex_diffs.dudy = in6[6 * i + 4]; si.tex_diffs.dvdy = in6[6 * i + 5];
        Spectrum c = s->data.tex.evaluate(texture, si);
        out3[3 * i] = c[0]; out3[3 * i + 1] = c[1]; out3[3 * i + 2] = c[2];
    }
    return FTN_OK;
}
// MIPMap::lookup_trilinear_width on an image built by MIPMap::new (custom = 0) or new_custom (custom = 1): rows of {s, t, width}
void orc_kat_mipmap_lookup(uint32_t w, uint32_t h, const float* texels, int wrap, int custom, const float* st_width3, size_t n, float* out3) {
    MIPMap m;
    if (custom) { std::vector<Spectrum> img((size_t)w * h); for (size_t i = 0; i < img.size(); i++) img[i] = Spectrum(texels[3 * i], texels[3 * i + 1], texels[3 * i + 2]); m = MIPMap::make_custom((int)w, (int)h, img, wrap); }
    else m = MIPMap::make((int)w, (int)h, texels, wrap);
    for (size_t i = 0; i < n; i++) { Spectrum c = m.lookup_trilinear_width(Vec2(st_width3[3 * i], st_width3[3 * i + 1]), st_width3[3 * i + 2]); out3[3 * i] = c[0]; out3[3 * i + 1] = c[1]; out3[3 * i + 2] = c[2]; }
}

}  // extern "C"
