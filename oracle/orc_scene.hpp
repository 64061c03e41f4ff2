// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.hpp).
// orc_scene.hpp: src/bvh.rs, src/primitive.rs, src/scene/mod.rs, src/light/*.rs, src/sampling.rs
// (Distribution1D/2D), level-0 bilinear lookup of src/mipmap.rs.
#pragma once
#include "orc_reflection.hpp"
#include "orc_texture.hpp"
#include <memory>
#include <deque>
#include <atomic>

namespace orc {

// ---- Distribution1D / 2D: src/sampling.rs:59-180
inline size_t search_sorted_cdf(const std::vector<Float>& cdf, Float u) {   // :66-81 with key = cdf[i] <= u
    size_t size = cdf.size(), first = 0, len = size;
    while (len > 0) {
        size_t half = len >> 1, middle = first + half;
        if (cdf[middle] <= u) { first = middle + 1; len -= half + 1; } else len = half;
    }
    int64_t v = (int64_t)first - 1;   // usize subtraction; never underflows because cdf[0] = 0 <= u
    int64_t hi = (int64_t)size - 2;
    return (size_t)(v < 0 ? 0 : (v > hi ? hi : v));
}
struct Distribution1D {
    std::vector<Float> func, cdf; Float func_integral;
    void init(const Float* f, size_t n) {                                  // :84-107
        func.assign(f, f + n); cdf.assign(n + 1, 0.0f);
        for (size_t i = 1; i < n + 1; i++) cdf[i] = cdf[i - 1] + (func[i - 1] / (Float)n);
        func_integral = cdf[n];
        if (func_integral == 0.0f) { for (size_t i = 1; i < n + 1; i++) cdf[i] = (Float)i / (Float)n; }
        else { for (size_t i = 1; i < n + 1; i++) cdf[i] /= func_integral; }
    }
    void sample_continuous(Float u, Float* x, Float* pdf, size_t* idx_out) const {   // :121-134
        size_t idx = search_sorted_cdf(cdf, u);
        Float du = u - cdf[idx];
        if (cdf[idx + 1] - cdf[idx] > 0.0f) du /= cdf[idx + 1] - cdf[idx];
        *pdf = func[idx] / func_integral;
        *x = ((Float)idx + du) / (Float)func.size();
        *idx_out = idx;
    }
};
struct Distribution2D {
    std::vector<Distribution1D> p_conditional_v; Distribution1D p_marginal;
    void init(const Float* func, size_t nu, size_t nv) {                    // :145-161
        p_conditional_v.resize(nv);
        std::vector<Float> marginal(nv);
        for (size_t v = 0; v < nv; v++) { p_conditional_v[v].init(func + v * nu, nu); marginal[v] = p_conditional_v[v].func_integral; }
        p_marginal.init(marginal.data(), nv);
    }
    Vec2 sample_continuous(Vec2 u, Float* pdf) const {                      // :163-169
        Float d1, pdf1, d0, pdf0; size_t v_idx, dummy;
        p_marginal.sample_continuous(u.y, &d1, &pdf1, &v_idx);
        p_conditional_v[v_idx].sample_continuous(u.x, &d0, &pdf0, &dummy);
        *pdf = pdf0 * pdf1;
        return Vec2(d0, d1);
    }
    Float pdf(Vec2 p) const {                                               // :171-179
        int64_t u_len = (int64_t)p_conditional_v[0].func.size();
        int64_t iu = std::min(std::max(f2usize(p.x * (Float)u_len), (int64_t)0), u_len - 1);
        int64_t v_len = (int64_t)p_marginal.func.size();
        int64_t iv = std::min(std::max(f2usize(p.y * (Float)v_len), (int64_t)0), v_len - 1);
        return p_conditional_v[iv].func[iu] / p_marginal.func_integral;
    }
};

// ---- MIPMap level 0, ImageWrap::Repeat: src/mipmap.rs:258-312
struct EnvMap {
    int w = 0, h = 0; std::vector<Float> texels;   // texels[(t*w + s)*3 + c]
    Spectrum texel(int s, int t) const {
        s = ((s % w) + w) % w; t = ((t % h) + h) % h;   // rem_euclid
        const Float* p = &texels[((size_t)t * w + s) * 3];
        return Spectrum(p[0], p[1], p[2]);
    }
    Spectrum triangle(Vec2 st) const {                                      // :258-272 (level 0)
        Float s = st.x * (Float)w - 0.5f, t = st.y * (Float)h - 0.5f;
        Float sf = floorf(s), tf = floorf(t);
        int s0 = f2i32(sf), t0 = f2i32(tf);
        Float ds = s - (Float)s0, dt = t - (Float)t0;
        return texel(s0, t0) * (1.0f - ds) * (1.0f - dt) + texel(s0, t0 + 1) * (1.0f - ds) * dt +
               texel(s0 + 1, t0) * ds * (1.0f - dt) + texel(s0 + 1, t0 + 1) * ds * dt;
    }
    // lookup_trilinear_width(st, 0.0): level = levels-1+log2(1e-8) < 0 -> triangle(0, st)   (:245-256)
    Spectrum lookup0(Vec2 st) const { if (w == 1 && h == 1) return lookup_1x1(st); return triangle(st); }
    // 1x1 map (new_uniform): levels = 1; width 0 -> level = 0 + log2(1e-8) < 0 -> triangle(0, st) as well
    Spectrum lookup_1x1(Vec2 st) const { return triangle(st); }
};

// ---- Lights: src/light/*.rs
struct SceneData;
struct LiSample { Spectrum radiance; Vec3 wi; Float pdf; SurfaceHit p0, p1; };

struct Light {
    enum Kind { POINT, DISTANT, INFINITE, AREA } kind;
    // point / distant
    Spectrum rgb; Vec3 v;
    // distant / infinite: preprocess
    Vec3 world_center; Float world_radius = 0.0f;
    // infinite
    Transform light_to_world, world_to_light; const EnvMap* l_map = nullptr; Distribution2D distribution;
    // area (DiffuseAreaLight, light/diffuse.rs:24-94)
    const Shape* shape = nullptr; Spectrum emit; Float area = 0.0f;

    bool is_delta() const { return kind == POINT || kind == DISTANT; }      // light/mod.rs:64-71

    Spectrum area_emitted_radiance(const SurfaceHit& hit, Vec3 w) const {   // diffuse.rs:44-50
        return dot(hit.n, w) > 0.0f ? emit : Spectrum(0.0f);
    }
    // compute_distribution: infinite.rs:63-78  (NB (height,width) = resolution() name swap: a w x h map gives a distribution with h columns, w rows)
    void compute_distribution() {
        int height = l_map->w, width = l_map->h;
        std::vector<Float> img((size_t)width * height);
        for (int j = 0; j < height; j++) {
            Float vv = (Float)j / (Float)height;
            Float sin_theta = m_sin(PI * ((Float)j + 0.5f) / (Float)height);
            for (int i = 0; i < width; i++) {
                Float uu = (Float)i / (Float)width;
                // filter = 1/max(w,h) -> level = floor(log2 max) - log2 max: exactly 0 for a power of two -> lerp(0, tri(0), tri(1)) = tri(0)*1 + tri(1)*0,
                // negative otherwise -> triangle(0) alone (mipmap.rs:247-249): pyramid level 1 never contributes;
                // a 1x1 map takes the `level >= levels-1` branch = texel(0,0,0)
                Spectrum tex = (l_map->w == 1 && l_map->h == 1) ? l_map->texel(0, 0) : l_map->triangle(Vec2(uu, vv));
                img[i + (size_t)j * width] = tex.luminance() * sin_theta;
            }
        }
        distribution.init(img.data(), width, height);
    }
    LiSample sample_incident_radiance(const SurfaceHit& reference, Vec2 u) const {
        LiSample s;
        switch (kind) {
            case POINT: {                                                   // point.rs:43-62
                s.wi = normalize(v - reference.p); s.pdf = 1.0f;
                s.p0 = reference; s.p1.p = v; s.p1.p_err = Vec3(); s.p1.time = reference.time; s.p1.n = Vec3();
                s.radiance = rgb / magnitude2(v - reference.p);
                return s;
            }
            case DISTANT: {                                                 // distant.rs:49-71
                Vec3 p_outside = reference.p + v * (2.0f * world_radius);
                s.p0 = reference; s.p1.p = p_outside; s.p1.p_err = Vec3(); s.p1.time = reference.time; s.p1.n = Vec3();
                s.radiance = rgb; s.wi = v; s.pdf = 1.0f;
                return s;
            }
            case INFINITE: {                                                // infinite.rs:99-140
                Float map_pdf; Vec2 uv = distribution.sample_continuous(u, &map_pdf);
                // map_pdf == 0 -> unimplemented!() in the reference; callers treat pdf 0 as "no sample"
                Float theta = uv.y * PI, phi = uv.x * 2.0f * PI;
                s.wi = tf_vector(light_to_world, Vec3(m_sin(theta) * m_cos(phi), m_sin(theta) * m_sin(phi), m_cos(theta)));
                s.pdf = (m_sin(theta) == 0.0f) ? 0.0f : map_pdf / (2.0f * PI * PI * m_sin(theta));
                if (map_pdf == 0.0f) s.pdf = 0.0f;
                s.p0 = reference; s.p1.p = reference.p + s.wi * (2.0f * world_radius); s.p1.p_err = Vec3(); s.p1.time = reference.time; s.p1.n = Vec3();
                s.radiance = l_map->lookup0(uv);
                return s;
            }
            default: {                                                      // diffuse.rs:75-89
                SurfaceHit p_shape = shape->sample(u);                      // sample_from_ref == sample (shapes/mod.rs:51-53)
                s.wi = normalize(p_shape.p - reference.p);
                s.pdf = shape->pdf_from_ref(reference, s.wi);
                s.p0 = reference; s.p1 = p_shape;
                s.radiance = area_emitted_radiance(p_shape, -s.wi);
                return s;
            }
        }
    }
    Float pdf_incident_radiance(const SurfaceHit& reference, Vec3 wi) const {
        switch (kind) {
            case POINT: case DISTANT: return 0.0f;
            case INFINITE: {                                                // infinite.rs:142-154
                Vec3 w = tf_vector(world_to_light, wi);
                Float theta = spherical_theta(w), phi = spherical_phi(w);
                if (m_sin(theta) == 0.0f) return 0.0f;
                return distribution.pdf(Vec2(phi * (1.0f / (2.0f * PI)), theta * FRAC_1_PI)) / (2.0f * PI * PI * m_sin(theta));
            }
            default: return shape->pdf_from_ref(reference, wi);            // diffuse.rs:91-93
        }
    }
    Spectrum environment_emitted_radiance(const Ray& ray) const {           // infinite.rs:156-164; default light/mod.rs:33
        if (kind != INFINITE) return Spectrum(0.0f);
        Vec3 w = normalize(tf_vector(world_to_light, ray.dir));
        Vec2 st(spherical_phi(w) * (1.0f / (2.0f * PI)), spherical_theta(w) * FRAC_1_PI);
        return l_map->lookup0(st);
    }
};

// ---- Primitive: src/primitive.rs:25-70
struct Primitive {
    const Shape* shape = nullptr;
    int material = -1;        // index into SceneData::materials
    int light = -1;           // index into SceneData::lights (area light), -1 = None
    int area_emit = -1;       // descriptor index, used to create the light after the BVH build
    uint32_t original_index = 0;
};

// ---- BVH: src/bvh.rs
struct TraversalCounters { uint64_t nodes_visited = 0, prims_tested = 0; };

struct BVH {
    std::vector<Primitive> prims;   // BVH order after apply_permutation
    Bounds3 bounds;
    std::vector<ftn_bvh_node> nodes;
    uint32_t max_depth = 0;

    struct PrimInfo { size_t prim_id; Bounds3 bounds; Vec3 centroid; };
    struct BuildNode { Bounds3 bounds; bool leaf; uint32_t first_prim_idx; uint16_t n_prims; BuildNode* children[2]; uint8_t split_axis; };

    // `partition` crate 0.1.x: Hoare-style in-place partition, returns the split point.
    // (Intra-side order does not influence the tree: every later step uses order-independent min/max folds.)
    static size_t partition_pred(PrimInfo* data, size_t len, int ax, Float midpoint) {
        if (len == 0) return 0;
        size_t l = 0, r = len - 1;
        for (;;) {
            while (l < len && data[l].centroid[ax] < midpoint) l++;
            while (r > 0 && !(data[r].centroid[ax] < midpoint)) r--;
            if (l >= r) return l;
            std::swap(data[l], data[r]);
        }
    }
    static BuildNode* recursive_build(std::deque<BuildNode>& arena, PrimInfo* info, size_t n,
                                      std::vector<int64_t>& ordering, uint32_t depth, uint32_t* max_depth) {   // :66-120
        if (depth > *max_depth) *max_depth = depth;
        Bounds3 node_bounds = Bounds3::empty(), centroid_bounds = Bounds3::empty();
        for (size_t i = 0; i < n; i++) { node_bounds = node_bounds.join(info[i].bounds); centroid_bounds = centroid_bounds.join_point(info[i].centroid); }
        arena.emplace_back();
        BuildNode* node = &arena.back();
        if (n == 1 || centroid_bounds.is_point()) {
            node->leaf = true; node->first_prim_idx = (uint32_t)ordering.size(); node->n_prims = (uint16_t)n; node->bounds = node_bounds;
            for (size_t i = 0; i < n; i++) ordering.push_back((int64_t)info[i].prim_id);
            return node;
        }
        int ax = centroid_bounds.maximum_extent();
        Float midpoint = (centroid_bounds.min[ax] + centroid_bounds.max[ax]) / 2.0f;
        size_t mid = partition_pred(info, n, ax, midpoint);
        if (mid == 0 || mid == n) {                                          // partition_equal_counts :122-131
            mid = n / 2;
            std::nth_element(info, info + mid, info + n, [ax](const PrimInfo& a, const PrimInfo& b) { return a.centroid[ax] < b.centroid[ax]; });
        }
        BuildNode* c0 = recursive_build(arena, info, mid, ordering, depth + 1, max_depth);
        BuildNode* c1 = recursive_build(arena, info + mid, n - mid, ordering, depth + 1, max_depth);
        node->leaf = false; node->children[0] = c0; node->children[1] = c1; node->split_axis = (uint8_t)ax;
        node->bounds = c0->bounds.join(c1->bounds);
        return node;
    }
    size_t flatten_tree(const BuildNode* node) {                            // :133-158
        ftn_bvh_node ln; memset(&ln, 0, sizeof(ln));
        for (int i = 0; i < 3; i++) { ln.bmin[i] = node->bounds.min[i]; ln.bmax[i] = node->bounds.max[i]; }
        if (node->leaf) {
            ln.is_leaf = 1; ln.idx = node->first_prim_idx; ln.n_prims = node->n_prims;
            nodes.push_back(ln);
            return 1;
        }
        ln.is_leaf = 0; ln.axis = node->split_axis; ln.idx = 0;
        nodes.push_back(ln);
        size_t my_idx = nodes.size() - 1;
        size_t first_len = flatten_tree(node->children[0]);
        nodes[my_idx].idx = (uint32_t)(my_idx + first_len + 1);
        size_t second_len = flatten_tree(node->children[1]);
        return first_len + second_len + 1;
    }
    void build(std::vector<Primitive> in_prims) {                           // :27-64
        prims = std::move(in_prims);
        nodes.clear(); max_depth = 0;
        if (prims.empty()) { bounds = Bounds3::empty(); return; }
        std::vector<PrimInfo> info(prims.size());
        for (size_t i = 0; i < prims.size(); i++) { info[i].prim_id = i; info[i].bounds = prims[i].shape->world_bound(); info[i].centroid = info[i].bounds.centroid(); }
        std::deque<BuildNode> arena;   // bumpalo arena in the reference
        std::vector<int64_t> ordering; ordering.reserve(prims.size());
        BuildNode* root = recursive_build(arena, info.data(), info.size(), ordering, 0, &max_depth);
        bounds = root->bounds;
        // apply_permutation (:355-374): items[i] <- items[ordering[i]]
        std::vector<Primitive> permuted(prims.size());
        for (size_t i = 0; i < prims.size(); i++) permuted[i] = prims[(size_t)ordering[i]];
        prims.swap(permuted);
        nodes.reserve(2 * prims.size());
        flatten_tree(root);
    }

    static Bounds3 node_bounds(const ftn_bvh_node& n) { Bounds3 b; b.min = Vec3(n.bmin[0], n.bmin[1], n.bmin[2]); b.max = Vec3(n.bmax[0], n.bmax[1], n.bmax[2]); return b; }

    // intersect: :160-215.  Returns false on miss. `too_deep` mirrors ArrayVec<[usize;64]>::push panicking.
    bool intersect(Ray& ray, SurfaceInteraction* out, TraversalCounters* ctr, bool* too_deep) const {
        if (nodes.empty()) return false;
        bool dir_is_neg[3] = {ray.dir.x < 0.0f, ray.dir.y < 0.0f, ray.dir.z < 0.0f};
        size_t stack[64]; int sp = 0; size_t cur = 0; bool found = false;
        for (;;) {
            const ftn_bvh_node& node = nodes[cur];
            if (ctr) ctr->nodes_visited++;
            if (node_bounds(node).intersect_test(ray)) {
                if (node.is_leaf) {
                    for (uint32_t i = 0; i < node.n_prims; i++) {
                        const Primitive& prim = prims[node.idx + i];
                        if (ctr) ctr->prims_tested++;
                        Float t; SurfaceInteraction si;
                        if (prim.shape->intersect(ray, &t, &si)) {        // GeometricPrimitive::intersect primitive.rs:48-54
                            ray.t_max = t; si.prim = (int)(node.idx + i); *out = si; found = true;
                        }
                    }
                    if (sp > 0) cur = stack[--sp]; else break;
                } else {
                    if (sp >= 64) { if (too_deep) *too_deep = true; return found; }
                    if (dir_is_neg[node.axis]) { stack[sp++] = cur + 1; cur = node.idx; }
                    else { stack[sp++] = node.idx; cur += 1; }
                }
            } else {
                if (sp > 0) cur = stack[--sp]; else break;
            }
        }
        return found;
    }
    // intersect_test: :217-266
    bool intersect_test(const Ray& ray, TraversalCounters* ctr, bool* too_deep) const {
        if (nodes.empty()) return false;
        bool dir_is_neg[3] = {ray.dir.x < 0.0f, ray.dir.y < 0.0f, ray.dir.z < 0.0f};
        size_t stack[64]; int sp = 0; size_t cur = 0;
        for (;;) {
            const ftn_bvh_node& node = nodes[cur];
            if (ctr) ctr->nodes_visited++;
            if (node_bounds(node).intersect_test(ray)) {
                if (node.is_leaf) {
                    for (uint32_t i = 0; i < node.n_prims; i++) {
                        if (ctr) ctr->prims_tested++;
                        if (prims[node.idx + i].shape->intersect_test(ray)) return true;
                    }
                    if (sp > 0) cur = stack[--sp]; else break;
                } else {
                    if (sp >= 64) { if (too_deep) *too_deep = true; return false; }
                    if (dir_is_neg[node.axis]) { stack[sp++] = cur + 1; cur = node.idx; }
                    else { stack[sp++] = node.idx; cur += 1; }
                }
            } else {
                if (sp > 0) cur = stack[--sp]; else break;
            }
        }
        return false;
    }
};

// ---- Scene: src/scene/mod.rs (flat, built from ftn_scene_desc)
struct SceneData {
    std::vector<Float> P, N, UV, S;
    std::vector<TriangleMesh> meshes;
    std::vector<Triangle> triangles;
    std::vector<Sphere> spheres;
    std::vector<ftn_material> materials;
    TextureSet tex; std::vector<ftn_material_textures> mtex;    // empty mtex: every parameter constant
    std::vector<EnvMap> envmaps;
    std::vector<Light> lights;
    BVH bvh;
    // statistics (atomics: the renderer is tile-parallel)
    mutable std::atomic<uint64_t> rays_closest{0}, rays_any{0}, nodes_visited{0}, prims_tested{0}, nodes_any{0}, prims_any{0};
    mutable std::atomic<int> error{0};   // 0 ok; FTN_ERR_* otherwise
    bool count_traffic = false;

    bool intersect(Ray& ray, SurfaceInteraction* si) const {                // :51-53
        TraversalCounters c; bool deep = false;
        bool hit = bvh.intersect(ray, si, count_traffic ? &c : nullptr, &deep);
        rays_closest.fetch_add(1, std::memory_order_relaxed);
        if (count_traffic) { nodes_visited.fetch_add(c.nodes_visited, std::memory_order_relaxed); prims_tested.fetch_add(c.prims_tested, std::memory_order_relaxed); }
        if (deep) error.store(FTN_ERR_BVH_TOO_DEEP);
        return hit;
    }
    bool intersect_test(const Ray& ray) const {                             // :55-57
        TraversalCounters c; bool deep = false;
        bool hit = bvh.intersect_test(ray, count_traffic ? &c : nullptr, &deep);
        rays_any.fetch_add(1, std::memory_order_relaxed);
        if (count_traffic) { nodes_visited.fetch_add(c.nodes_visited, std::memory_order_relaxed); prims_tested.fetch_add(c.prims_tested, std::memory_order_relaxed);
                             nodes_any.fetch_add(c.nodes_visited, std::memory_order_relaxed); prims_any.fetch_add(c.prims_tested, std::memory_order_relaxed); }
        if (deep) error.store(FTN_ERR_BVH_TOO_DEEP);
        return hit;
    }
    Spectrum environment_emitted_radiance(const Ray& ray) const {           // :59-64
        Spectrum sum(0.0f);
        for (const Light& l : lights) sum = sum + l.environment_emitted_radiance(ray);
        return sum;
    }
    // SurfaceInteraction::emitted_radiance: interaction.rs:175-180
    Spectrum emitted_radiance(const SurfaceInteraction& si, Vec3 w) const {
        const Primitive& prim = bvh.prims[si.prim];
        if (prim.light < 0) return Spectrum(0.0f);
        return lights[prim.light].area_emitted_radiance(si.hit, w);
    }
    // VisibilityTester::unoccluded: light/mod.rs:81-85
    bool unoccluded(const SurfaceHit& p0, const SurfaceHit& p1) const { return !intersect_test(p0.spawn_ray_to_hit(p1)); }
};

inline int build_scene(const ftn_scene_desc* d, SceneData* s) {
    s->P.assign(d->P, d->P + 3 * (size_t)d->n_vertices);
    if (d->N) s->N.assign(d->N, d->N + 3 * (size_t)d->n_vertices);
    if (d->UV) s->UV.assign(d->UV, d->UV + 2 * (size_t)d->n_vertices);
    if (d->S) s->S.assign(d->S, d->S + 3 * (size_t)d->n_vertices);
    s->meshes.resize(d->n_meshes);
    for (uint32_t i = 0; i < d->n_meshes; i++) {
        TriangleMesh& m = s->meshes[i];
        m.P = s->P.data(); m.N = s->N.empty() ? nullptr : s->N.data(); m.UV = s->UV.empty() ? nullptr : s->UV.data();
        m.has_normals = d->meshes[i].has_normals && m.N; m.has_uvs = d->meshes[i].has_uvs && m.UV;
        m.S = s->S.empty() ? nullptr : s->S.data(); m.has_tangents = d->meshes[i].has_tangents && m.S;
        m.flip_normals = d->meshes[i].flip_normals; m.reverse_orientation = d->meshes[i].reverse_orientation;
    }
    s->triangles.resize(d->n_triangles);
    for (uint32_t i = 0; i < d->n_triangles; i++) {
        if (d->tri_mesh[i] >= d->n_meshes) return FTN_ERR_INVALID_ARGUMENT;
        Triangle& t = s->triangles[i]; t.mesh = &s->meshes[d->tri_mesh[i]];
        for (int k = 0; k < 3; k++) { t.v[k] = d->tri_indices[3 * (size_t)i + k]; if (t.v[k] >= d->n_vertices) return FTN_ERR_INVALID_ARGUMENT; }
    }
    s->spheres.resize(d->n_spheres);
    for (uint32_t i = 0; i < d->n_spheres; i++) {
        const ftn_sphere& q = d->spheres[i]; Sphere& sp = s->spheres[i];
        sp.object_to_world = Transform::make(Mat4::from_flat(q.object_to_world.m), Mat4::from_flat(q.object_to_world.inv));
        sp.world_to_object = Transform::make(Mat4::from_flat(q.world_to_object.m), Mat4::from_flat(q.world_to_object.inv));
        sp.reverse_orientation = q.reverse_orientation != 0;
        sp.radius = q.radius; sp.z_min = q.z_min; sp.z_max = q.z_max; sp.theta_min = q.theta_min; sp.theta_max = q.theta_max; sp.phi_max = q.phi_max;
    }
    s->materials.assign(d->materials, d->materials + d->n_materials);
    { int trc = build_textures(d, &s->tex, &s->mtex); if (trc) return trc; }
    s->envmaps.resize(d->n_envmaps);
    for (uint32_t i = 0; i < d->n_envmaps; i++) {
        EnvMap& e = s->envmaps[i]; e.w = (int)d->envmaps[i].width; e.h = (int)d->envmaps[i].height;
        if (e.w <= 0 || e.h <= 0) return FTN_ERR_INVALID_ARGUMENT;
        e.texels.assign(d->envmaps[i].texels, d->envmaps[i].texels + (size_t)e.w * e.h * 3);
    }
    std::vector<Primitive> prims(d->n_prims);
    for (uint32_t i = 0; i < d->n_prims; i++) {
        const ftn_prim& p = d->prims[i];
        if (p.shape_kind == FTN_SHAPE_TRIANGLE) { if (p.shape_index >= d->n_triangles) return FTN_ERR_INVALID_ARGUMENT; prims[i].shape = &s->triangles[p.shape_index]; }
        else if (p.shape_kind == FTN_SHAPE_SPHERE) { if (p.shape_index >= d->n_spheres) return FTN_ERR_INVALID_ARGUMENT; prims[i].shape = &s->spheres[p.shape_index]; }
        else return FTN_ERR_INVALID_ARGUMENT;
        if (p.material >= (int)d->n_materials || p.area_emit >= (int)d->n_area_emit) return FTN_ERR_INVALID_ARGUMENT;
        prims[i].material = p.material; prims[i].area_emit = p.area_emit; prims[i].original_index = i;
    }
    s->bvh.build(std::move(prims));
    // Scene::new: preprocess explicit lights, then append area lights in BVH primitive order (scene/mod.rs:32-49)
    s->lights.clear();
    for (uint32_t i = 0; i < d->n_lights; i++) {
        const ftn_light& l = d->lights[i]; Light L;
        L.rgb = Spectrum(l.rgb[0], l.rgb[1], l.rgb[2]); L.v = Vec3(l.v[0], l.v[1], l.v[2]);
        if (l.type == FTN_LIGHT_POINT) L.kind = Light::POINT;
        else if (l.type == FTN_LIGHT_DISTANT) { L.kind = Light::DISTANT; s->bvh.bounds.bounding_sphere(&L.world_center, &L.world_radius); }
        else if (l.type == FTN_LIGHT_INFINITE) {
            if (l.envmap < 0 || l.envmap >= (int)d->n_envmaps) return FTN_ERR_INVALID_ARGUMENT;
            L.kind = Light::INFINITE; L.l_map = &s->envmaps[l.envmap];
            L.light_to_world = Transform::make(Mat4::from_flat(l.light_to_world.m), Mat4::from_flat(l.light_to_world.inv));
            L.world_to_light = L.light_to_world.inverse();
            L.compute_distribution();
            s->bvh.bounds.bounding_sphere(&L.world_center, &L.world_radius);
        } else return FTN_ERR_INVALID_ARGUMENT;
        s->lights.push_back(std::move(L));
    }
    for (size_t i = 0; i < s->bvh.prims.size(); i++) {
        Primitive& p = s->bvh.prims[i];
        if (p.area_emit >= 0) {
            Light L; L.kind = Light::AREA; L.shape = p.shape; L.area = p.shape->area();
            L.emit = Spectrum(d->area_emit[3 * p.area_emit], d->area_emit[3 * p.area_emit + 1], d->area_emit[3 * p.area_emit + 2]);
            p.light = (int)s->lights.size();
            s->lights.push_back(std::move(L));
        }
    }
    return FTN_OK;
}

}  // namespace orc
