/*
 * ftn_kernels.hip -- gfx950 kernels, first pipeline: one lane walks a whole path ("megakernel", BASELINE config 2),
 * plus batch Scene::intersect / intersect_test kernels and the film resolve.
 *
 *   k_render_mega   SamplerIntegrator::render_tile (src/integrator/mod.rs:229-281) with PathIntegrator::incident_radiance
 *                   (src/integrator/path.rs:25-95) or DirectLightingIntegrator (src/integrator/direct_lighting.rs:50-110)
 *                   inlined.  One 256-thread workgroup = one 16x16 film tile (4 wave64).
 *                     FTN_SAMPLER_INDEXED    : one lane per pixel, samples in order, film sums kept in registers;
 *   k_render_serial FTN_SAMPLER_TILE_SERIAL: the reference's per-tile RandomSampler stream is serial by construction (one
 *                   Xoshiro256+ state per 16x16 tile, a path-dependent number of draws per sample), so ONE LANE walks a whole
 *                   tile exactly like render_tile does and the tiles of the film are the parallel dimension (a 4096^2 film has
 *                   65 536 of them).
 *   k_trace_batch   Scene::intersect / intersect_test for arrays of rays (src/scene/mod.rs:51-57).
 *   k_film_resolve  Film::merge_film_tile (src/film.rs:121-132) from the three accumulators into Pixel{xyz,w}.
 *
 * The traversal stack lives in LDS, lane-interleaved ([level][lane], conflict-free for ds_write/read_b32); its depth is
 * the BVH's depth, so LDS per workgroup = depth * 256 * 4 bytes.
 */
#include "ftn_kernels.h"
#include "ftn_texture.h"

namespace ftn {

/* ------------------------------------------------------------------ wave-level stat flush */
__device__ inline unsigned long long wave_sum(unsigned int v) {
    unsigned long long s = v;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    return s;
}
struct LaneCounters { uint32_t closest = 0, any = 0, cam = 0, spill = 0, bc = 0; TravCount tc{0, 0}; };
__device__ inline void flush_counters(DevStats* st, const LaneCounters& c, bool count) {
    unsigned long long a = wave_sum(c.closest), b = wave_sum(c.any), d = wave_sum(c.cam), e = wave_sum(c.spill), f = wave_sum(c.bc);
    unsigned long long n = 0, p = 0;
    if (count) { n = wave_sum(c.tc.nodes); p = wave_sum(c.tc.prims); }
    if ((threadIdx.x & 63) == 0) {
        if (a) atomicAdd(&st->rays_closest, a);
        if (b) atomicAdd(&st->rays_any, b);
        if (d) atomicAdd(&st->camera_samples, d);
        if (e) atomicAdd(&st->spill_samples, e);
        if (f) atomicAdd(&st->bc_writes, f);
        if (n) atomicAdd(&st->nodes_visited, n);
        if (p) atomicAdd(&st->prims_tested, p);
    }
}

/* ------------------------------------------------------------------ Scene::intersect / intersect_test wrappers */
template <bool COUNT>
struct Tracer {
    const DScene& S; LdsStack st; LaneCounters& lc;
    __device__ bool closest(DRay& ray, DHit* h) { lc.closest++; return traverse<false, COUNT>(S, ray, st, h, &lc.tc); }
    __device__ bool any(DRay& ray) { lc.any++; DHit h; return traverse<true, COUNT>(S, ray, st, &h, &lc.tc); }
};

/* ------------------------------------------------------------------ estimate_direct + uniform_sample_one_light: integrator/mod.rs:289-395 */
template <bool COUNT>
__device__ inline Rgb estimate_direct(Tracer<COUNT>& T, const DBsdf& B, const DSI& si, V2 u_scatter, const DLight& L, int light_index, V2 u_light) {
    const DScene& S = T.S;
    const uint32_t flags = T_ALL & ~T_SPECULAR;
    Rgb radiance(0.0f);
    DLiSample ls = light_sample(S, L, si.hit, u_light);
    if (ls.pdf > 0.0f && !ls.radiance.is_black()) {
        Rgb f = bsdf_f(B, si.wo, ls.wi, flags) * abs_dot(ls.wi, si.shading_n);
        float sp = bsdf_pdf(B, si.wo, ls.wi, flags);
        if (!f.is_black()) {
            DRay sr = spawn_ray_to_hit(si.hit, ls.p1);
            if (!T.any(sr)) {
                if (L.kind == LK_POINT || L.kind == LK_DISTANT) radiance = radiance + f * ls.radiance / ls.pdf;
                else radiance = radiance + f * ls.radiance * power_heuristic(ls.pdf, sp) / ls.pdf;
            }
        }
    }
    if (!(L.kind == LK_POINT || L.kind == LK_DISTANT)) {
        DScatter sc;
        if (bsdf_sample(B, si.wo, u_scatter, flags, &sc)) {
            Rgb f = sc.f * abs_dot(sc.wi, si.shading_n);
            if (f.is_black()) return radiance;
            float weight;
            if (sc.type & T_SPECULAR) weight = 1.0f;
            else {
                float lp = light_pdf(S, L, si.hit, sc.wi);
                if (lp == 0.0f) return radiance;
                weight = power_heuristic(sc.pdf, lp);
            }
            DRay ray = spawn_ray(si.hit, sc.wi);
            const DRay ray0 = ray;
            DHit h; Rgb inc(0.0f);
            if (T.closest(ray, &h)) {
                int h_mat, h_light; prim_mat_light(S, h.prim, &h_mat, &h_light);
                if (h_light >= 0 && h_light == light_index) {          /* the hit primitive's area light is THIS light (:370-381) */
                    DSI s2; make_interaction(S, h, ray0, &s2);
                    inc = area_Le(L, s2.hit.n, -sc.wi);              /* si.emitted_radiance(-scatter.wi) */
                }
            } else if (L.kind == LK_INFINITE) inc = light_Le_env(L, ray.d);
            if (!inc.is_black()) radiance = radiance + f * inc * weight / sc.pdf;
        }
    }
    return radiance;
}
template <bool COUNT>
__device__ inline Rgb uniform_sample_one_light(Tracer<COUNT>& T, const DBsdf& B, const DSI& si, Rng& rng) {
    const uint32_t nl = T.S.n_lights;
    if (nl == 0) return Rgb(0.0f);
    uint32_t ln = (uint32_t)f2usize(fmin_(rng.next() * (float)nl, (float)(nl - 1)));
    V2 ul = rng.next2();
    V2 us = rng.next2();
    return (float)nl * estimate_direct(T, B, si, us, T.S.lights[ln], (int)ln, ul);
}
__device__ inline Rgb emitted(const DScene& S, const DSI& si, V3 w) {          /* interaction.rs:175-180 */
    if (si.light < 0) return Rgb(0.0f);
    return area_Le(S.lights[si.light], si.hit.n, w);
}

/* ------------------------------------------------------------------ PathIntegrator::incident_radiance: path.rs:25-95 */
/* TEX: some material parameter is a texture.  `rd` is the camera ray's differential; the path integrator hands it on unchanged at
 * every bounce (path.rs:73, :79), so texture footprints after the first hit come from the CAMERA offsets rays, as in the reference. */
template <bool COUNT, bool TEX>
__device__ inline Rgb path_li(Tracer<COUNT>& T, DRay ray, const DRayDiff& rd, Rng& rng, uint32_t max_depth, float rr_threshold, int* err) {
    const DScene& S = T.S;
    Rgb L(0.0f), beta(1.0f);
    uint32_t bounces = 0; bool specular_bounce = false;
    for (;;) {
        const DRay ray0 = ray;
        DHit h; DSI si;
        bool hit = T.closest(ray, &h);
        if (hit) make_interaction(S, h, ray0, &si);
        if (bounces == 0 || specular_bounce) {
            if (hit) L = L + beta * emitted(S, si, -ray.d);
            else L = L + beta * scene_env_Le(S, ray.d);
        }
        if (!hit || bounces >= max_depth) break;
        const int mat = si.mat;
        if (mat >= 0) {
            DBsdf B;
            ftn_material mloc; const ftn_material* mp = &S.materials[mat];
            if (TEX && material_is_textured(S, mat)) {            /* Texture::evaluate(si) for every textured parameter */
                DSIX ex; DSI s2; make_interaction(S, h, ray0, &s2, &ex);
                const DTexDiffs td = compute_tex_diffs(si.hit.p, si.hit.n, ex.dpdu, ex.dpdv, rd);
                mloc = material_resolve(S, mat, ex.uv, td); mp = &mloc;
            }
            if (!make_bsdf(*mp, si, true, &B)) { *err = FTN_ERR_UNSUPPORTED; break; }
            if (bsdf_num(B, T_ALL & ~T_SPECULAR) > 0) {
                Rgb direct = beta * uniform_sample_one_light(T, B, si, rng);
                L = L + direct;
            }
            V3 wo = -ray.d;
            DScatter bs;
            V2 u = rng.next2();
            bool ok = bsdf_sample(B, wo, u, T_ALL, &bs);
            if (ok && !bs.f.is_black()) {
                beta = beta * (bs.f * abs_dot(bs.wi, si.shading_n) / bs.pdf);
                specular_bounce = (bs.type & T_SPECULAR) != 0;
                ray = spawn_ray(si.hit, bs.wi);
            } else break;
        } else {
            ray = spawn_ray(si.hit, ray.d);   /* null bsdf: skip without bounce++ (:77-81) */
            continue;
        }
        if (beta.max_component() < rr_threshold && bounces > 3) {
            float q = fmax_(0.05f, 1.0f - beta.max_component());
            if (rng.next() < q) break;
            beta = beta / (1.0f - q);
        }
        bounces += 1;
    }
    return L;
}

/* ------------------------------------------------------------------ DirectLightingIntegrator (UniformSampleOne): direct_lighting.rs:50-110 with
 * specular_reflect / specular_transmit (integrator/mod.rs:39-178).  The recursion is unrolled for chains with at most one specular
 * branch per vertex (mirror, or no specular lobe at all); a BSDF with both specular lobes (specular glass) reports FTN_ERR_UNSUPPORTED. */
#define FTN_DL_MAX 8
#define FTN_OWN_SERIAL (-2147483647 - 1)   /* film_add: single-writer tile walk */
#define FTN_OWN_NONE (-2147483647)         /* film_add: lane owns no crop pixel */
template <bool COUNT, bool TEX>
__device__ inline Rgb direct_li(Tracer<COUNT>& T, DRay ray, DRayDiff rd, Rng& rng, uint32_t max_depth, bool whitted, int* err) {
    const DScene& S = T.S;
    Rgb local[FTN_DL_MAX]; Rgb wf[FTN_DL_MAX]; float wc[FTN_DL_MAX], wp[FTN_DL_MAX]; bool owes_t[FTN_DL_MAX];
    int depth = 0; Rgb tail(0.0f); bool have_tail = false;
    for (;;) {
        if (depth >= FTN_DL_MAX) { *err = FTN_ERR_UNSUPPORTED; break; }
        const DRay ray0 = ray; DHit h;
        owes_t[depth] = false;
        if (!T.closest(ray, &h)) { tail = scene_env_Le(S, ray.d); have_tail = true; break; }
        DSI si; make_interaction(S, h, ray0, &si);
        const int mat = si.mat;
        if (mat < 0) { *err = FTN_ERR_UNSUPPORTED; tail = Rgb(0.0f); have_tail = true; break; }   /* unimplemented!() :103 */
        DBsdf B;
        ftn_material mloc; const ftn_material* mp = &S.materials[mat];
        DSIX ex; DTexDiffs td; float bsdf_eta = 1.0f;
        if (TEX) {                                             /* differentials follow the specular chain (mod.rs:58-84, :119-163) */
            DSI s2; make_interaction(S, h, ray0, &s2, &ex);
            td = compute_tex_diffs(si.hit.p, si.hit.n, ex.dpdu, ex.dpdv, rd);
            if (material_is_textured(S, mat)) { mloc = material_resolve(S, mat, ex.uv, td); mp = &mloc; }
            if (mp->type == FTN_MAT_GLASS) bsdf_eta = mp->s0;  /* Bsdf::new(si, eta): glass.rs:60 */
        }
        if (!make_bsdf(*mp, si, false, &B)) { *err = FTN_ERR_UNSUPPORTED; tail = Rgb(0.0f); have_tail = true; break; }
        Rgb rad(0.0f);
        if (whitted) {                                           /* whitted.rs:42-58: every light, one 2D sample each, no emission term */
            for (uint32_t li = 0; li < S.n_lights; li++) {
                const DLight& L = S.lights[li];
                DLiSample ls = light_sample(S, L, si.hit, rng.next2());
                if (ls.radiance.is_black() || ls.pdf == 0.0f) continue;
                Rgb f = bsdf_f(B, si.wo, ls.wi, T_ALL);
                if (!f.is_black()) {
                    DRay sr = spawn_ray_to_hit(si.hit, ls.p1);
                    if (!T.any(sr)) rad = rad + f * ls.radiance * abs_dot(ls.wi, si.shading_n) / ls.pdf;
                }
            }
        } else {
            rad = rad + emitted(S, si, si.wo);
            rad = rad + uniform_sample_one_light(T, B, si, rng);
        }
        local[depth] = rad;
        if (!((uint32_t)depth + 1 < max_depth)) { tail = Rgb(0.0f); have_tail = false; depth++; break; }
        /* specular_reflect: the 2D sample is drawn before the match (mod.rs:52) */
        owes_t[depth] = true;
        V2 ur = rng.next2();
        DScatter sr;
        bool has_r = bsdf_sample(B, si.wo, ur, T_REFL | T_SPECULAR, &sr) && !(abs_dot(sr.wi, si.shading_n) == 0.0f);
        bool has_t = bsdf_num(B, T_TRANS | T_SPECULAR) > 0;
        if (has_r && has_t) { *err = FTN_ERR_UNSUPPORTED; depth++; break; }
        if (has_r) {
            wf[depth] = sr.f; wc[depth] = fabsf(dot(sr.wi, si.shading_n)); wp[depth] = sr.pdf;
            if (TEX) rd = specular_diff(true, rd, si.hit.p, si.wo, sr.wi, si.shading_n, ex.dndu, ex.dndv, td, bsdf_eta);
            ray = spawn_ray(si.hit, sr.wi);
            depth++;
            continue;
        }
        if (has_t) {
            /* reflect returned 0; transmit recursion */
            V2 ut = rng.next2(); owes_t[depth] = false;
            DScatter stt;
            if (bsdf_sample(B, si.wo, ut, T_TRANS | T_SPECULAR, &stt) && !(abs_dot(stt.wi, si.shading_n) == 0.0f)) {
                wf[depth] = stt.f; wc[depth] = fabsf(dot(stt.wi, si.shading_n)); wp[depth] = stt.pdf;
                local[depth] = local[depth] + Rgb(0.0f);      /* radiance += specular_reflect (= 0) */
                if (TEX) rd = specular_diff(false, rd, si.hit.p, si.wo, stt.wi, si.shading_n, ex.dndu, ex.dndv, td, bsdf_eta);
                ray = spawn_ray(si.hit, stt.wi);
                depth++;
                continue;
            }
        }
        depth++;
        break;
    }
    /* unwind: radiance = local + reflect_term (+ transmit_term); pending transmit draws are consumed on the way up */
    Rgb li = have_tail ? tail : Rgb(0.0f);
    bool child = have_tail;
    for (int d = depth - 1; d >= 0; d--) {
        Rgb r = local[d];
        if (child) r = r + wf[d] * li * wc[d] / wp[d];
        else if ((uint32_t)d + 1 < max_depth) r = r + Rgb(0.0f);
        if (owes_t[d]) { (void)rng.next2(); r = r + Rgb(0.0f); }
        li = r; child = true;
    }
    return li;
}

/* ------------------------------------------------------------------ Film::add_sample_to_tile: film.rs:136-172 (box filter: every table entry is 1.0) */
struct FilmCtx {
    int crop[4]; int tpb[4];     /* FilmTile::pixel_bounds of this tile (get_film_tile, film.rs:95-113) */
    int sb[4];                   /* the tile's sample bounds */
    float radius[2];
    float4 *A, *B, *C;
};
__device__ inline size_t film_idx(const FilmCtx& F, int x, int y) { return (size_t)(y - F.crop[1]) * (size_t)(F.crop[2] - F.crop[0]) + (size_t)(x - F.crop[0]); }
__device__ inline void atomic_add4(float4* p, Rgb c, float w) {
    float* f = reinterpret_cast<float*>(p);
    atomicAdd(f + 0, c.r); atomicAdd(f + 1, c.g); atomicAdd(f + 2, c.b); atomicAdd(f + 3, w);
}
/* Returns the number of pixels touched. own_(x,y): the pixel whose register accumulator `acc` belongs to the caller
 * (indexed mode); serial mode passes own_x = INT_MIN and writes in-tile pixels straight to A (single writer). */
__device__ inline int film_add(const FilmCtx& F, V2 p_film, Rgb L, float sample_weight, int own_x, int own_y, float4* acc, uint32_t* bc_writes) {
    float pdx = p_film.x - 0.5f, pdy = p_film.y - 0.5f;
    int p0x = f2i_sat(ceilf(pdx - F.radius[0])), p0y = f2i_sat(ceilf(pdy - F.radius[1]));
    int p1x = f2i_sat(floorf(pdx + F.radius[0])) + 1, p1y = f2i_sat(floorf(pdy + F.radius[1])) + 1;
    p0x = max(p0x, F.tpb[0]); p0y = max(p0y, F.tpb[1]); p1x = min(p1x, F.tpb[2]); p1y = min(p1y, F.tpb[3]);
    const Rgb contrib = L * sample_weight * 1.0f;
    int touched = 0;
    for (int y = p0y; y < p1y; y++)
        for (int x = p0x; x < p1x; x++) {
            touched++;
            if (x == own_x && y == own_y) { acc->x += contrib.r; acc->y += contrib.g; acc->z += contrib.b; acc->w += 1.0f; continue; }
            const bool in_tile = x >= F.sb[0] && x < F.sb[2] && y >= F.sb[1] && y < F.sb[3];
            const size_t i = film_idx(F, x, y);
            if (in_tile && own_x == FTN_OWN_SERIAL) { float4 v = F.A[i]; v.x += contrib.r; v.y += contrib.g; v.z += contrib.b; v.w += 1.0f; F.A[i] = v; }
            else { atomic_add4(in_tile ? &F.B[i] : &F.C[i], contrib, 1.0f); (*bc_writes)++; }
        }
    return touched;
}
__device__ inline void make_film_ctx(const RenderParams& P, const DTile& t, FilmCtx* F) {
    for (int i = 0; i < 4; i++) F->crop[i] = P.crop[i];
    F->sb[0] = t.x0; F->sb[1] = t.y0; F->sb[2] = t.x1; F->sb[3] = t.y1;
    F->radius[0] = P.radius[0]; F->radius[1] = P.radius[1];
    int p0x = f2i_sat(ceilf((float)t.x0 - 0.5f - P.radius[0])), p0y = f2i_sat(ceilf((float)t.y0 - 0.5f - P.radius[1]));
    int p1x = f2i_sat(ceilf((float)t.x1 - 0.5f + P.radius[0] + 1.0f)), p1y = f2i_sat(ceilf((float)t.y1 - 0.5f - P.radius[1] + 1.0f));   /* sic: -radius, film.rs:100 */
    F->tpb[0] = max(p0x, P.crop[0]); F->tpb[1] = max(p0y, P.crop[1]); F->tpb[2] = min(p1x, P.crop[2]); F->tpb[3] = min(p1y, P.crop[3]);
    F->A = P.accA; F->B = P.accB; F->C = P.accC;
}

/* ------------------------------------------------------------------ one camera sample: render_tile's inner loop body (mod.rs:244-274) */
template <bool COUNT, bool TEX>
__device__ inline void render_sample(const RenderParams& P, Tracer<COUNT>& T, const FilmCtx& F, Rng& rng, int px, int py, int own_x, int own_y,
                                     float4* acc, int* err) {
    V2 j = rng.next2();
    V2 p_film((float)px + j.x, (float)py + j.y);
    V2 p_lens = rng.next2();
    float time_u = rng.next();
    DRay ray = camera_ray(P.C, p_film, p_lens, time_u);
    DRayDiff rd; rd.has = false;
    if (TEX) rd = camera_ray_diff(P.C, p_film, p_lens, ray, 1.0f / sqrtf((float)P.spp));     /* generate_ray_differential + scale_differentials (mod.rs:252-254) */
    Rgb L = (P.integrator_kind != FTN_INTEGRATOR_PATH) ? direct_li<COUNT, TEX>(T, ray, rd, rng, P.max_depth, P.integrator_kind == FTN_INTEGRATOR_WHITTED, err)
                                                                  : path_li<COUNT, TEX>(T, ray, rd, rng, P.max_depth, P.rr_threshold, err);
    if (L.has_nans()) *err = FTN_ERR_NAN_RADIANCE;       /* check_radiance :285-287 */
    int touched = film_add(F, p_film, L, 1.0f, own_x, own_y, acc, &T.lc.bc);
    T.lc.cam++;
    if (touched != 1) T.lc.spill++;
}

template <bool COUNT, bool TEX>
__global__ void __launch_bounds__(256) k_render_mega(RenderParams P) {
    extern __shared__ uint32_t lds_stack[];
    const DTile tile = P.tiles[blockIdx.x];
    LaneCounters lc; int err = 0;
    Tracer<COUNT> T{P.S, LdsStack{lds_stack + threadIdx.x, 256u}, lc};
    FilmCtx F; make_film_ctx(P, tile, &F);
    if (P.sampler_kind == FTN_SAMPLER_INDEXED) {
        const int px = tile.x0 + (int)(threadIdx.x & 15u), py = tile.y0 + (int)(threadIdx.x >> 4);
        if (px < tile.x1 && py < tile.y1) {
            const bool in_crop = px >= P.crop[0] && px < P.crop[2] && py >= P.crop[1] && py < P.crop[3];
            float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            size_t ai = 0;
            if (in_crop) { ai = film_idx(F, px, py); acc = P.accA[ai]; }
            for (uint32_t s = P.first_sample; s < P.last_sample; s++) {
                Rng rng; rng.seed(indexed_key(P.seed, px, py, s));
                render_sample<COUNT, TEX>(P, T, F, rng, px, py, in_crop ? px : FTN_OWN_NONE, py, &acc, &err);
            }
            if (in_crop) P.accA[ai] = acc;
        }
    }
    flush_counters(P.stats, lc, COUNT);
    if (err) atomicCAS(&P.stats->error, 0, err);
}

template <bool COUNT, bool TEX>
__global__ void __launch_bounds__(256) k_render_serial(RenderParams P) {
    extern __shared__ uint32_t lds_stack[];
    const uint32_t ti = blockIdx.x * 256u + threadIdx.x;
    LaneCounters lc; int err = 0;
    Tracer<COUNT> T{P.S, LdsStack{lds_stack + threadIdx.x, 256u}, lc};
    if (ti < P.n_tiles) {
        const DTile tile = P.tiles[ti];
        FilmCtx F; make_film_ctx(P, tile, &F);
        Rng rng; rng.seed(tile.tile_id);                 /* clone_with_seed(tile_id): random.rs:61-67 */
        float4 dummy = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (int py = tile.y0; py < tile.y1; py++)
            for (int px = tile.x0; px < tile.x1; px++)
                for (uint32_t s = 0; s < P.spp; s++)
                    render_sample<COUNT, TEX>(P, T, F, rng, px, py, FTN_OWN_SERIAL, 0, &dummy, &err);
    }
    flush_counters(P.stats, lc, COUNT);
    if (err) atomicCAS(&P.stats->error, 0, err);
}

void launch_render_mega(const RenderParams& p, bool count, hipStream_t stream) {
    if (p.n_tiles == 0) return;
    if (p.sampler_kind != FTN_SAMPLER_INDEXED) {
        const size_t lds_s = (size_t)p.stack_entries * 256 * sizeof(uint32_t);
        const dim3 g((p.n_tiles + 255u) / 256u);
        const bool tex_s = p.S.n_textures != 0;
        if (tex_s) { if (count) hipLaunchKernelGGL((k_render_serial<true, true>), g, dim3(256), lds_s, stream, p); else hipLaunchKernelGGL((k_render_serial<false, true>), g, dim3(256), lds_s, stream, p); }
        else { if (count) hipLaunchKernelGGL((k_render_serial<true, false>), g, dim3(256), lds_s, stream, p); else hipLaunchKernelGGL((k_render_serial<false, false>), g, dim3(256), lds_s, stream, p); }
        return;
    }
    size_t lds = (size_t)p.stack_entries * 256 * sizeof(uint32_t);
    const bool tex = p.S.n_textures != 0;
    if (tex) { if (count) hipLaunchKernelGGL((k_render_mega<true, true>), dim3(p.n_tiles), dim3(256), lds, stream, p); else hipLaunchKernelGGL((k_render_mega<false, true>), dim3(p.n_tiles), dim3(256), lds, stream, p); }
    else { if (count) hipLaunchKernelGGL((k_render_mega<true, false>), dim3(p.n_tiles), dim3(256), lds, stream, p); else hipLaunchKernelGGL((k_render_mega<false, false>), dim3(p.n_tiles), dim3(256), lds, stream, p); }
}

/* ------------------------------------------------------------------ Film::merge_film_tile: film.rs:121-132.  pixel.xyz += to_xyz(tile sum) per contributing tile */
__global__ void __launch_bounds__(256) k_film_resolve(const float4* __restrict__ A, const float4* __restrict__ B, const float4* __restrict__ C,
                                                      ftn_pixel* __restrict__ out, size_t n, const DevStats* __restrict__ stats) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    /* no sample landed outside its own pixel (the rule with the box filter of radius 0.5): B and C are all zero, adding them changes
     * nothing (x + 0 == x), and half of this kernel's reads can stay away */
    const bool spilled = stats->bc_writes != 0;
    for (; i < n; i += stride) {
        const float4 a = A[i], zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f), b = spilled ? B[i] : zero, c = spilled ? C[i] : zero;
        float4 o = *reinterpret_cast<const float4*>(&out[i]);
        float xyz[3];
        rgb_to_xyz(Rgb(a.x + b.x, a.y + b.y, a.z + b.z), xyz);          /* the home tile's FilmTilePixel */
        o.x += xyz[0]; o.y += xyz[1]; o.z += xyz[2]; o.w += a.w + b.w;
        if (c.w != 0.0f) {                                                /* neighbouring tiles' border pixels */
            rgb_to_xyz(Rgb(c.x, c.y, c.z), xyz);
            o.x += xyz[0]; o.y += xyz[1]; o.z += xyz[2]; o.w += c.w;
        }
        *reinterpret_cast<float4*>(&out[i]) = o;
    }
}
void launch_film_resolve(const RenderParams& p, ftn_pixel* device_pixels, hipStream_t stream) {
    size_t n = (size_t)(p.crop[2] - p.crop[0]) * (size_t)(p.crop[3] - p.crop[1]);
    if (n == 0) return;
    unsigned grid = (unsigned)((n + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_film_resolve, dim3(grid), dim3(256), 0, stream, p.accA, p.accB, p.accC, device_pixels, n, p.stats);
}

/* ------------------------------------------------------------------ Film::into_spectrum_buffer: film.rs:195-210.  16 B in, 12 B out per pixel */
__global__ void __launch_bounds__(256) k_spectrum_buffer(const float4* __restrict__ px, float* __restrict__ rgb_out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float4 p = px[i];
        const float xyz[3] = {p.x, p.y, p.z};
        float rgb[3]; xyz_to_rgb(xyz, rgb);
        if (p.w != 0.0f) { const float inv = 1.0f / p.w; rgb[0] = fmax_(0.0f, rgb[0] * inv); rgb[1] = fmax_(0.0f, rgb[1] * inv); rgb[2] = fmax_(0.0f, rgb[2] * inv); }
        rgb_out[3 * i] = rgb[0]; rgb_out[3 * i + 1] = rgb[1]; rgb_out[3 * i + 2] = rgb[2];
    }
}
void launch_spectrum_buffer(const ftn_pixel* device_pixels, size_t n, float* device_rgb, hipStream_t stream) {
    if (n == 0) return;
    unsigned grid = (unsigned)((n + 255) / 256); if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(k_spectrum_buffer, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const float4*>(device_pixels), device_rgb, n);
}

/* ------------------------------------------------------------------ test hook: Texture::evaluate for arrays of (u, v, dudx, dvdx, dudy, dvdy) */
__global__ void __launch_bounds__(256) k_test_texture_eval(DScene S, int texture, const float* __restrict__ in6, size_t n, float* __restrict__ out3) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    DTexDiffs td; td.dpdx = V3(0.0f, 0.0f, 0.0f); td.dpdy = td.dpdx;
    td.dudx = in6[6 * i + 2]; td.dvdx = in6[6 * i + 3]; td.dudy = in6[6 * i + 4]; td.dvdy = in6[6 * i + 5];
    const Rgb c = tex_eval(S, texture, V2(in6[6 * i], in6[6 * i + 1]), td);
    out3[3 * i] = c.r; out3[3 * i + 1] = c.g; out3[3 * i + 2] = c.b;
}
void launch_test_texture_eval(const DScene& S, int texture, const float* uv_diffs6, size_t n, float* rgb_out, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_test_texture_eval, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, S, texture, uv_diffs6, n, rgb_out);
}

/* ------------------------------------------------------------------ batch Scene::intersect / intersect_test / full interaction */
template <int MODE, bool COUNT>
__global__ void __launch_bounds__(256) k_trace_batch(DScene S, const float* __restrict__ rays, size_t n, float* t_hit, int* prim, float* bary,
                                                     unsigned char* occluded, float* out24, DevStats* stats) {
    extern __shared__ uint32_t lds_stack[];
    LaneCounters lc;
    Tracer<COUNT> T{S, LdsStack{lds_stack + threadIdx.x, 256u}, lc};
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const float* r = rays + 8 * i;
        DRay ray; ray.o = V3(r[0], r[1], r[2]); ray.d = V3(r[3], r[4], r[5]); ray.t_max = r[6]; ray.time = r[7];
        if (MODE == 1) occluded[i] = T.any(ray) ? 1 : 0;
        else {
            const DRay ray0 = ray; DHit h;
            bool hit = T.closest(ray, &h);
            if (MODE == 0) {
                if (t_hit) t_hit[i] = hit ? h.t : FTN_INF;
                if (prim) prim[i] = hit ? h.prim : -1;
                if (bary) { bary[3 * i] = hit ? h.b0 : 0.0f; bary[3 * i + 1] = hit ? h.b1 : 0.0f; bary[3 * i + 2] = hit ? h.b2 : 0.0f; }
            } else {
                float* o = out24 + 24 * i;
                if (!hit) { for (int k = 0; k < 24; k++) o[k] = 0.0f; o[23] = -1.0f; }
                else {
                    DSI si; make_interaction(S, h, ray0, &si);
                    o[0] = si.hit.p.x; o[1] = si.hit.p.y; o[2] = si.hit.p.z; o[3] = si.hit.p_err.x; o[4] = si.hit.p_err.y; o[5] = si.hit.p_err.z;
                    o[6] = si.hit.n.x; o[7] = si.hit.n.y; o[8] = si.hit.n.z; o[9] = 0.0f; o[10] = 0.0f;
                    o[11] = si.wo.x; o[12] = si.wo.y; o[13] = si.wo.z; o[14] = si.s_dpdu.x; o[15] = si.s_dpdu.y; o[16] = si.s_dpdu.z;
                    o[17] = 0.0f; o[18] = 0.0f; o[19] = 0.0f; o[20] = si.shading_n.x; o[21] = si.shading_n.y; o[22] = si.shading_n.z; o[23] = h.t;
                }
            }
        }
    }
    flush_counters(stats, lc, COUNT);
}
void launch_trace_batch(const DScene& S, const float* rays, size_t n, int mode, float* t_hit, int* prim, float* bary, unsigned char* occluded,
                        float* out24, DevStats* stats, uint32_t stack_entries, bool count, hipStream_t stream) {
    if (n == 0) return;
    size_t lds = (size_t)stack_entries * 256 * sizeof(uint32_t);
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
#define FTN_LAUNCH(M, Cn) hipLaunchKernelGGL((k_trace_batch<M, Cn>), grid, block, lds, stream, S, rays, n, t_hit, prim, bary, occluded, out24, stats)
    if (mode == 0) { if (count) FTN_LAUNCH(0, true); else FTN_LAUNCH(0, false); }
    else if (mode == 1) { if (count) FTN_LAUNCH(1, true); else FTN_LAUNCH(1, false); }
    else { FTN_LAUNCH(2, false); }
#undef FTN_LAUNCH
}

/* ------------------------------------------------------------------ test hook: device evaluation of the scalar math */
FTN_HD float eval_math(int which, float x, float y) {
    switch (which) {
        case 0: return ftn_det::sinf_det(x); case 1: return ftn_det::cosf_det(x); case 2: return ftn_det::tanf_det(x);
        case 3: return ftn_det::acosf_det(x); case 4: return ftn_det::atanf_det(x); case 5: return ftn_det::atan2f_det(x, y);
        case 6: return ftn_det::logf_det(x); case 7: return ftn_det::log2f_det(x); case 8: return sqrtf(x); case 9: return x / y;
        case 10: return (float)sqrt((double)x * (double)y); case 11: return next_up(x); case 12: return next_down(x);
        default: return 0.0f;
    }
}
__global__ void k_test_math(int which, const float* x, const float* y, size_t n, float* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = eval_math(which, x[i], y[i]);
}
void launch_test_math(int which, const float* x, const float* y, size_t n, float* out, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(k_test_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, which, x, y, n, out);
}

}  // namespace ftn
