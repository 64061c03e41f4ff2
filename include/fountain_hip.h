/*
 * fountain_hip.h -- C ABI of the MI355X path-tracing core that drops in behind
 * fountain's SamplerIntegrator::render*() (reference: src/integrator/mod.rs:193-227).
 *
 * The reference has no FFI; its seam is the Rust generic struct
 *   SamplerIntegrator<R: IntegratorRadiance>{camera, radiance}.render(&Scene, &Film<BoxFilter>, impl Sampler)
 * (src/integrator/mod.rs:22-25, :206, :218).  A Rust shim flattens
 * Scene / Camera / Film / Sampler / PathIntegrator into the POD descriptors below and
 * calls ftn_render(); the result is added into Film::pixels exactly as
 * Film::merge_film_tile does (src/film.rs:121-132).
 *
 * Conventions
 *   - plain C, plain pointers and sizes, no torch / HIP types in any signature;
 *   - every function returns 0 (FTN_OK) or a negative ftn_status; nothing unwinds
 *     across the boundary; ftn_last_error() gives a thread-local message;
 *   - descriptor memory is caller-owned and only borrowed for the duration of a call;
 *   - ftn_scene is an opaque handle that owns device (HBM) memory;
 *   - all floating point is IEEE binary32 ("Float = f32", src/math.rs:10);
 *   - 4x4 matrices are column-major, m[col*4+row], as cgmath::Matrix4 stores them
 *     (src/geometry/transform.rs:7-10).
 *
 * The same descriptors are consumed by the CPU oracle (oracle/, test infrastructure only),
 * which exports orc_* twins of the ftn_* entry points.
 */
#ifndef FOUNTAIN_HIP_H
#define FOUNTAIN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status codes */
typedef enum ftn_status {
    FTN_OK = 0,
    FTN_ERR_INVALID_ARGUMENT = -1,
    FTN_ERR_NO_DEVICE = -2,        /* no HIP device / runtime failure                      */
    FTN_ERR_OUT_OF_MEMORY = -3,
    FTN_ERR_NAN_RADIANCE = -4,     /* mirrors assert!(!l.has_nans()) integrator/mod.rs:285  */
    FTN_ERR_UNSUPPORTED = -5,      /* e.g. specular glass: todo!() in material/glass.rs:66  */
    FTN_ERR_BVH_TOO_DEEP = -6,     /* ArrayVec<[usize;64]> overflow, bvh.rs:168             */
    FTN_ERR_INTERNAL = -7
} ftn_status;

/* ------------------------------------------------------------------ math PODs */

/* Transform{t, invt}: src/geometry/transform.rs:7-10 */
typedef struct ftn_transform {
    float m[16];
    float inv[16];
} ftn_transform;

/* Pixel{xyz, filter_weight_sum}: src/film.rs:13-16 (16 bytes) */
typedef struct ftn_pixel {
    float xyz[3];
    float filter_weight_sum;
} ftn_pixel;

/* LinearBVHNode: src/bvh.rs:269-302 (32 bytes, DFS order, first child at idx+1).
 * is_leaf: idx = first_prim_idx, n_prims; interior: idx = second_child_idx, axis = split_axis. */
typedef struct ftn_bvh_node {
    float bmin[3];
    float bmax[3];
    uint32_t idx;
    uint16_t n_prims;
    uint8_t axis;
    uint8_t is_leaf;
} ftn_bvh_node;

/* ------------------------------------------------------------------ scene description
 * The flat equivalent of Scene{primitives_aggregate, lights, meshes} (src/scene/mod.rs:14-18)
 * BEFORE BVH::build: primitives are listed in insertion order; ftn_scene_create runs the
 * reference's build (src/bvh.rs:27-64), permutes the primitives and appends one area light per
 * emissive primitive in BVH order (src/scene/mod.rs:38-41).                                   */

enum { FTN_SHAPE_TRIANGLE = 0, FTN_SHAPE_SPHERE = 1 };

/* GeometricPrimitive{shape, material, light}: src/primitive.rs:25-29 */
typedef struct ftn_prim {
    uint32_t shape_kind;   /* FTN_SHAPE_*                                             */
    uint32_t shape_index;  /* triangle index (into tri_indices/3) or sphere index     */
    int32_t material;      /* index into materials, -1 = None                         */
    int32_t area_emit;     /* index into area_emit (DiffuseAreaLight L), -1 = None    */
} ftn_prim;

/* TriangleMesh flags: src/shapes/triangle.rs:10-27. Vertices / normals / tangents are already in world
 * space (TriangleMesh::new, :42-58).                                                           */
typedef struct ftn_mesh {
    uint32_t has_normals;
    uint32_t has_uvs;
    uint32_t flip_normals;        /* reverse_orientation ^ transform_swaps_handedness, shapes/mod.rs:27-29 */
    uint32_t reverse_orientation;
    uint32_t has_tangents;        /* "S" (constructors.rs:63): per-vertex shading tangents, triangle.rs:341-347 */
} ftn_mesh;

/* Sphere: src/shapes/sphere.rs:15-27 (fields after Sphere::new's clamping, :40-49) */
typedef struct ftn_sphere {
    ftn_transform object_to_world;
    ftn_transform world_to_object;
    float radius, z_min, z_max, theta_min, theta_max, phi_max;
    uint32_t reverse_orientation;
    uint32_t _pad;
} ftn_sphere;

enum {
    FTN_MAT_MATTE = 0,   /* src/material/matte.rs   : a = Kd, s0 = sigma (degrees)                       */
    FTN_MAT_METAL = 1,   /* src/material/metal.rs   : a = eta, b = k, s1/s2 = u/v roughness              */
    FTN_MAT_MIRROR = 2,  /* src/material/mirror.rs  : a = Kr                                             */
    FTN_MAT_PLASTIC = 3, /* src/material/plastic.rs : a = Kd, b = Ks, s1 = roughness                     */
    FTN_MAT_GLASS = 4    /* src/material/glass.rs   : a = Kr, b = Kt, s0 = eta, s1/s2 = u/v roughness    */
};

/* All textures are ConstantTexture (src/texture/mod.rs:34-42); 48 bytes. */
typedef struct ftn_material {
    uint32_t type;
    uint32_t remap_roughness;
    float a[3];
    float b[3];
    float s0, s1, s2;
    float _pad;
} ftn_material;

enum {
    FTN_LIGHT_POINT = 0,    /* src/light/point.rs   : rgb = I, v = world_point                   */
    FTN_LIGHT_DISTANT = 1,  /* src/light/distant.rs : rgb = L, v = dir_to_light (normalised)     */
    FTN_LIGHT_INFINITE = 2  /* src/light/infinite.rs: envmap index, light_to_world              */
};

/* Lights given explicitly in the scene (Scene::lights before area lights are appended). */
typedef struct ftn_light {
    uint32_t type;
    int32_t envmap;           /* FTN_LIGHT_INFINITE: index into envmaps                      */
    float rgb[3];
    float v[3];
    ftn_transform light_to_world;
} ftn_light;

/* Level 0 of the MIPMap<Spectrum> behind InfiniteAreaLight (src/mipmap.rs:245-312, ImageWrap::Repeat):
 * texels[(t*width + s)*3 + c]. Any size: the light only ever reads level 0 (infinite.rs:63-77). */
typedef struct ftn_envmap {
    uint32_t width, height;
    const float* texels;
} ftn_envmap;

/* Textures (the files under src/texture/) as a flat array; a material parameter is either the constant stored in ftn_material or, through
 * ftn_material_textures, the index of a texture evaluated at every hit (Texture::evaluate(si), texture/mod.rs:12-16).
 * CONSTANT  ConstantTexture                    value
 * UV        UVTexture            uv.rs:17-23   (s - floor s, t - floor t, 0) of the mapped coordinates
 * CHECKER   Checkerboard2DTexture checkerboard.rs:49-64 (AAMethod::None): tex1 if (floor s + floor t) % 2 == 0 else tex2
 * IMAGE     ImageTexture         image.rs:28-34 -> MIPMap::lookup_trilinear(st, dst/dx, dst/dy), mipmap.rs:273-306
 * Every non-constant texture carries its UVMapping (mapping.rs:12-53): st = (su*u + du, sv*v + dv).                          */
enum { FTN_TEX_CONSTANT = 0, FTN_TEX_UV = 1, FTN_TEX_CHECKERBOARD = 2, FTN_TEX_IMAGE = 3 };
enum { FTN_WRAP_REPEAT = 0, FTN_WRAP_BLACK = 1, FTN_WRAP_CLAMP = 2 };     /* ImageWrap, mipmap.rs:14-17 */
typedef struct ftn_texture {
    uint32_t kind;        /* FTN_TEX_*                                                  */
    uint32_t is_float;    /* 1: Texture<Output = Float> (value[0]); 0: Spectrum          */
    float value[3];       /* CONSTANT                                                   */
    int32_t tex1, tex2;   /* CHECKERBOARD: indices into textures (same output type)      */
    int32_t image;        /* IMAGE: index into images                                    */
    float su, sv, du, dv; /* UVMapping{scale_u, scale_v, offset_u, offset_v}             */
} ftn_texture;
/* The texels handed to MIPMap::new by load_mipmap (imageio/mod.rs:81-124: after scale, inverse gamma and the y flip), RGB,
 * texels[(t*width + s)*3 + c]; ftn_scene_create builds the pyramid (mipmap.rs:78-145).  Any size (no power-of-two rule here). */
typedef struct ftn_image {
    uint32_t width, height;
    uint32_t wrap;        /* FTN_WRAP_*                                                  */
    uint32_t _pad;
    const float* texels;
} ftn_image;
/* Per material (same index as materials[]): texture index for each parameter slot of ftn_material, -1 = use the constant. */
typedef struct ftn_material_textures {
    int32_t a, b, s0, s1, s2;
    int32_t _pad[3];
} ftn_material_textures;

typedef struct ftn_scene_desc {
    uint32_t n_prims;      const ftn_prim* prims;
    uint32_t n_triangles;  const uint32_t* tri_indices;  /* 3 per triangle, into the vertex pool */
                           const uint32_t* tri_mesh;     /* mesh id per triangle                  */
    uint32_t n_vertices;   const float* P;               /* 3 per vertex, world space             */
                           const float* N;               /* 3 per vertex or NULL                  */
                           const float* UV;              /* 2 per vertex or NULL                  */
    uint32_t n_meshes;     const ftn_mesh* meshes;
    uint32_t n_spheres;    const ftn_sphere* spheres;
    uint32_t n_materials;  const ftn_material* materials;
    uint32_t n_area_emit;  const float* area_emit;       /* 3 per entry (DiffuseAreaLight emit)   */
    uint32_t n_lights;     const ftn_light* lights;
    uint32_t n_envmaps;    const ftn_envmap* envmaps;
    /* optional (zero / NULL when every material parameter is constant) */
    uint32_t n_textures;   const ftn_texture* textures;
                           const ftn_material_textures* material_textures;   /* n_materials entries or NULL */
    uint32_t n_images;     const ftn_image* images;
    const float* S;        /* per-vertex shading tangents, 3 per vertex (world space), or NULL: no mesh has tangents */
} ftn_scene_desc;

/* ------------------------------------------------------------------ camera / film / sampler / integrator */

/* PerspectiveCamera after its constructor ran (src/camera/mod.rs:72-115). */
typedef struct ftn_camera_desc {
    ftn_transform camera_to_world;
    ftn_transform raster_to_camera;
    float shutter_open, shutter_close;
    float lens_radius, focal_dist;
    float dx_camera[3];
    float dy_camera[3];
} ftn_camera_desc;

/* Film<BoxFilter> (src/film.rs:18-26, 43-83); bounds are [x0,x1) x [y0,y1). */
typedef struct ftn_film_desc {
    int32_t full_resolution[2];
    int32_t crop[4];            /* cropped_pixel_bounds: x0, y0, x1, y1 */
    float filter_radius[2];     /* BoxFilter::default(): 0.5, 0.5 (src/filter/mod.rs:24-32) */
} ftn_film_desc;

enum {
    /* RandomSampler exactly as the reference: ONE Xoshiro256+ stream per 16x16 tile, seeded with
     * tile_id (src/integrator/mod.rs:197-204, src/sampler/random.rs:61-67). Serial per tile.     */
    FTN_SAMPLER_TILE_SERIAL = 0,
    /* Same generator behind the same Sampler surface, but re-seeded at every start_next_sample
     * from (tile_id, pixel, sample index) so that samples are independent units of work.        */
    FTN_SAMPLER_INDEXED = 1
};

typedef struct ftn_sampler_desc {
    uint32_t kind;
    uint32_t samples_per_pixel;
    uint64_t seed;              /* RandomSampler::new_with_seed(spp, 0): unused by tiles (random.rs:61-67);
                                   mixed into the FTN_SAMPLER_INDEXED key                              */
    uint32_t first_sample;      /* FTN_SAMPLER_INDEXED only: render samples [first, first+count)        */
    uint32_t sample_count;      /* 0 = all of samples_per_pixel                                         */
} ftn_sampler_desc;

enum {
    FTN_INTEGRATOR_PATH = 0,           /* src/integrator/path.rs:10-96                  */
    FTN_INTEGRATOR_DIRECT_LIGHTING = 1,/* src/integrator/direct_lighting.rs (UniformSampleOne) */
    FTN_INTEGRATOR_WHITTED = 2         /* src/integrator/whitted.rs:15-70: every light once, no emission at the hit, specular recursion */
};

typedef struct ftn_integrator_desc {
    uint32_t kind;
    uint32_t max_depth;     /* u16 in the reference */
    float rr_threshold;
    uint32_t _pad;
} ftn_integrator_desc;

/* Which 16x16 tiles of Film::sample_bounds() this call renders: tile indices
 * first, first+stride, ... (count of them; count==0 means "to the end"). Tiles are numbered
 * row-major as Bounds2i::iter_tiles yields them (src/geometry/bounds.rs:85-97).              */
typedef struct ftn_tile_range {
    uint32_t first;
    uint32_t stride;
    uint32_t count;
    uint32_t _pad;
} ftn_tile_range;

enum {
    FTN_PIPELINE_AUTO = 0,         /* the wavefront pipeline whenever it takes the request, else the megakernel                                   */
    FTN_PIPELINE_MEGAKERNEL = 1,   /* one lane walks a whole path (any sampler, any integrator)                                                   */
    FTN_PIPELINE_WAVEFRONT = 2     /* SoA queues, trace / shade kernels: FTN_SAMPLER_INDEXED; the path integrator always, direct lighting and
                                      Whitted for scenes without textures (Whitted: at most 4 lights); FTN_ERR_UNSUPPORTED otherwise              */
};

typedef struct ftn_render_options {
    uint32_t pipeline;        /* FTN_PIPELINE_*                                                 */
    int32_t device;           /* HIP device ordinal; -1 = current                               */
    uint32_t count_traffic;   /* 1: also tally nodes visited / prims tested with the reference's traversal for every ray (slower; equals the
                                    oracle's tally); 2: tally what the production configuration really walks: four-box records
                                    (ftn_stats.quad_records) and triangle tests, MIS rays toward an infinite light through the any-hit
                                    kernel (ftn_stats.mis_rays_any_hit)                                                  */
    uint32_t _pad;
} ftn_render_options;

typedef struct ftn_stats {
    uint64_t rays_closest;      /* Scene::intersect calls       (path.rs:42, integrator/mod.rs:367) */
    uint64_t rays_any;          /* Scene::intersect_test calls  (light/mod.rs:83)                    */
    uint64_t nodes_visited;     /* LinearBVHNode fetches (only with count_traffic)                   */
    uint64_t prims_tested;      /* primitive intersection tests (only with count_traffic)            */
    uint64_t camera_samples;
    uint64_t spill_samples;     /* film samples that touched more than one pixel (film.rs:138-139)   */
    double kernel_ms;           /* device time of the render kernels (HIP events)                    */
    double trace_ms;            /* device time of the closest-hit traversal kernel launches (wavefront)     */
    uint64_t trace_launches;    /* closest-hit traversal launches timed by trace_ms (wavefront)            */
    uint64_t nodes_visited_any; /* the share of nodes_visited / prims_tested spent in intersect_test rays   */
    uint64_t prims_tested_any;
    uint64_t mis_rays_any_hit;  /* Scene::intersect calls of estimate_direct (integrator/mod.rs:367) answered by the any-hit kernel: toward an
                                   infinite light only hit / miss matters.  Included in rays_closest (the reference's accounting).      */
    uint64_t quad_records;      /* count_traffic = 2: 128-byte four-box records fetched by the production traversal kernels (both) */
    uint64_t quad_records_any;  /*                    ... the any-hit kernel's share                                               */
    double any_ms;              /* device time of the any-hit traversal launches (wavefront); on small wavefronts they run on a second
                                 * stream beside the closest-hit launch of the same bounce, so trace_ms and any_ms can overlap in time      */
    uint64_t any_launches;
    double shade_ms;            /* device time of the classify and the shade launches (wavefront): two spans per bounce, the second one
                                 * starts behind the wait for the any-hit results                                                          */
    uint64_t shade_launches;    /* spans summed into shade_ms (two per bounce)                                                      */
    double sort_ms;             /* device time of the ray-queue coherence sorts (wavefront)                                        */
} ftn_stats;

/* ------------------------------------------------------------------ host-side constructors
 * (exact f32 restatements of the reference's host code; used by the Python/Rust shims so that
 *  no arithmetic happens outside this library)                                                  */

/* Transform algebra: src/geometry/transform.rs:20-140 */
int ftn_transform_identity(ftn_transform* out);
int ftn_transform_translate(const float delta[3], ftn_transform* out);
int ftn_transform_scale(float sx, float sy, float sz, ftn_transform* out);
int ftn_transform_rotate(float angle_deg, const float axis[3], ftn_transform* out);
int ftn_transform_look_at(const float pos[3], const float look[3], const float up[3], ftn_transform* out);
int ftn_transform_from_flat(const float m[16], ftn_transform* out);     /* from_flat + invert, :31-39 */
int ftn_transform_mul(const ftn_transform* a, const ftn_transform* b, ftn_transform* out); /* a * b, :143-149 */
int ftn_transform_inverse(const ftn_transform* a, ftn_transform* out);
int ftn_transform_perspective(float fov_deg, float near_z, float far_z, ftn_transform* out); /* :105-115 */
int ftn_transform_point(const ftn_transform* t, const float p[3], float out[3]);   /* :224 */
int ftn_transform_vector(const ftn_transform* t, const float v[3], float out[3]);  /* :177 */
int ftn_transform_normal(const ftn_transform* t, const float n[3], float out[3]);  /* :125-131 */
int ftn_transform_swaps_handedness(const ftn_transform* t);                       /* :121-123; returns 0/1 */
/* TriangleMesh::new's vertex / normal loops (src/shapes/triangle.rs:42-51): n points or normals, 3 floats each */
int ftn_transform_points(const ftn_transform* t, size_t n, const float* in, float* out);
int ftn_transform_normals(const ftn_transform* t, size_t n, const float* in, float* out);

/* Sphere::new clamping: src/shapes/sphere.rs:30-50 */
int ftn_sphere_init(const ftn_transform* o2w, const ftn_transform* w2o, int reverse_orientation,
                    float radius, float z_min, float z_max, float phi_max_deg, ftn_sphere* out);

/* PerspectiveCamera::new: src/camera/mod.rs:85-114; screen_window = {min.x, min.y, max.x, max.y} */
int ftn_camera_perspective(const ftn_transform* camera_to_world, const int32_t full_resolution[2],
                           const float screen_window[4], const float shutter[2],
                           float lens_radius, float focal_dist, float fov_deg, ftn_camera_desc* out);

/* Film::new (crop window -> cropped_pixel_bounds): src/film.rs:43-83; crop_window = {min.x, min.y, max.x, max.y} */
int ftn_film_init(const int32_t full_resolution[2], const float crop_window[4], ftn_film_desc* out);
/* Film::sample_bounds: src/film.rs:86-93; out = x0, y0, x1, y1 */
int ftn_film_sample_bounds(const ftn_film_desc* film, int32_t out[4]);
/* number of 16x16 tiles of sample_bounds (Bounds2i::iter_tiles, bounds.rs:85-97) */
int ftn_film_tile_count(const ftn_film_desc* film, uint32_t* out);
/* Film::into_spectrum_buffer: src/film.rs:195-210; rgb_out has 3 floats per pixel */
int ftn_film_resolve(const ftn_pixel* pixels, size_t n_pixels, float* rgb_out);

/* ------------------------------------------------------------------ scene */
typedef struct ftn_scene ftn_scene;

/* BVH::build + Scene::new (src/bvh.rs:27-64, src/scene/mod.rs:32-49), then upload to HBM. */
int ftn_scene_create(const ftn_scene_desc* desc, int device, ftn_scene** out);
void ftn_scene_destroy(ftn_scene* scene);

/* Host-side BVH build only (no device needed): fills caller arrays.
 * nodes_out: capacity 2*n_prims-1; prim_order_out: n_prims (ordered_prims[i] = original index).
 * Returns the node count in *n_nodes_out.                                                       */
int ftn_bvh_build(const ftn_scene_desc* desc, ftn_bvh_node* nodes_out, uint32_t* prim_order_out,
                  uint32_t* n_nodes_out, uint32_t* max_depth_out);

/* The four-box view of a flattened BVH that the traversal kernels walk (host only, no device needed): one
 * record of 32 floats (128 bytes) per two levels of the tree `nodes` (as returned by ftn_bvh_build; reference layout bvh.rs:269-302).
 * records_out: capacity 32 * (number of interior nodes) floats, or NULL to query the counts. */
int ftn_bvh_quads(const ftn_bvh_node* nodes, uint32_t n_nodes, float* records_out, uint32_t* n_records_out, uint32_t* stack_bound_out);

/* The eight-box occlusion records the any-hit kernel walks in triangle-only scenes (fountain_amd/csrc/ftn_host.cpp build_octs: a second
 * tree over the reference's leaves with outward-rounded 8-bit boxes; exact because intersect_test only depends on which LEAF boxes and
 * primitives a ray passes).  Host-only, for tests: 32 words per record, 8 floats {min, first primitive, max, 0} per explicit leaf box. */
int ftn_bvh_octs(const ftn_bvh_node* nodes, uint32_t n_nodes, uint32_t* records_out, uint32_t* n_records_out, uint32_t* stack_bound_out,
                 float* xbox_out, uint32_t* n_xbox_out);

/* Introspection for parity tests. */
int ftn_scene_info(const ftn_scene* scene, uint32_t* n_nodes, uint32_t* n_prims, uint32_t* n_lights,
                   uint32_t* max_depth, float world_bound[6]);
int ftn_scene_get_nodes(const ftn_scene* scene, ftn_bvh_node* nodes_out, uint32_t* prim_order_out);
/* Device bytes the scene holds, per array.  Arrays only a legacy knob reads are not allocated unless that knob is
 * set when the scene is created: the two-box records (FTN_TRACE4=0 / FTN_QUAD=0) and the indexed attribute arrays (FTN_SREC=0; meshes
 * with shading tangents keep them). */
typedef struct ftn_scene_memory {
    uint64_t nodes;             /* LinearBVHNode records, 32 B each (reference-order kernels: counting builds, exception rays, megakernel) */
    uint64_t quad;              /* four-box records, 128 B each (production traversal) */
    uint64_t oct;               /* eight-box occlusion records, 128 B each + explicit leaf boxes (any-hit traversal of triangle scenes) */
    uint64_t fat;               /* two-box records, 64 B each (legacy any-hit kernel; 0 unless asked for) */
    uint64_t geom;              /* triangle vertices, 48 B per primitive (leaf tests) */
    uint64_t srec;              /* shading records, 128 B per primitive */
    uint64_t indexed_attributes;/* prim_info + per-vertex normals / uvs / tangents (0 when the shading records carry everything) */
    uint64_t prim_class;        /* one byte per primitive */
    uint64_t lights;            /* light records, environment texels and distributions */
    uint64_t textures;          /* image pyramids */
    uint64_t other;             /* spheres, materials, statistics block */
    uint64_t total;
} ftn_scene_memory;
int ftn_scene_memory_info(const ftn_scene* scene, ftn_scene_memory* out);
/* Scene::lights after Scene::new: kind[i] (0 point, 1 distant, 2 infinite, 3 area), prim[i] = BVH-ordered primitive
 * index of an area light's shape or -1 */
int ftn_scene_get_lights(const ftn_scene* scene, int32_t* kind, int32_t* prim);

/* ------------------------------------------------------------------ the hot path */

/* Scene::intersect for a batch of rays (src/scene/mod.rs:51-53 -> src/bvh.rs:160-215).
 * rays: n x {o[3], d[3], t_max, time} (8 floats, HOST memory).  Outputs (HOST, may be NULL):
 * t_hit[n] (inf = miss), prim[n] (BVH-ordered primitive index, -1 = miss), bary[3n] (b0,b1,b2; triangles). */
int ftn_intersect(const ftn_scene* scene, const float* rays, size_t n,
                  float* t_hit, int32_t* prim, float* bary, ftn_stats* stats);
/* Scene::intersect_test (src/scene/mod.rs:55-57 -> src/bvh.rs:217-266): occluded[n] = 0/1 */
int ftn_intersect_test(const ftn_scene* scene, const float* rays, size_t n,
                       uint8_t* occluded, ftn_stats* stats);
/* Full SurfaceInteraction of Scene::intersect, for parity of the shading geometry:
 * per ray 24 floats: p[3] p_err[3] n[3] uv[2] wo[3] shading dpdu[3] dpdv[3] shading_n[3] t  (t<0 = miss);
 * uv and dpdv are filled by the oracle only (the device recomputes them where a texture needs them) */
int ftn_intersect_full(const ftn_scene* scene, const float* rays, size_t n, float* out24);

/* SamplerIntegrator::render_parallel (src/integrator/mod.rs:218-227): renders the selected tiles and
 * ADDS the result into out_pixels (crop_w*crop_h ftn_pixel, HOST memory, row-major) as
 * Film::merge_film_tile does.                                                                    */
int ftn_render(const ftn_scene* scene, const ftn_camera_desc* camera, const ftn_film_desc* film,
               const ftn_sampler_desc* sampler, const ftn_integrator_desc* integrator,
               const ftn_tile_range* tiles, const ftn_render_options* options,
               ftn_pixel* out_pixels, ftn_stats* stats);

/* Same, but the film stays resident in HBM: device_pixels is a DEVICE pointer to crop_w*crop_h
 * ftn_pixel (e.g. a torch tensor's data_ptr) on `stream` (a hipStream_t cast to void*, NULL = default
 * stream). Used by the multi-GPU driver: each rank renders its tiles into a zeroed buffer which is then
 * summed with one RCCL reduce.                                                                   */
int ftn_render_device(const ftn_scene* scene, const ftn_camera_desc* camera, const ftn_film_desc* film,
                      const ftn_sampler_desc* sampler, const ftn_integrator_desc* integrator,
                      const ftn_tile_range* tiles, const ftn_render_options* options,
                      void* device_pixels, void* stream, ftn_stats* stats);

/* ------------------------------------------------------------------ scene ingestion (host only)
 * The subset of the PBRT v3 format the reference evaluates: PbrtHeader::exec_stmt + make_camera / make_sampler / make_film
 * (src/loaders/pbrt.rs:426-532) and PbrtSceneBuilder::exec_stmt (pbrt.rs:178-330) with the defaults of
 * src/loaders/constructors.rs:38-359.  Include files and `Shape "plymesh"` resolve relative to the scene file.  Statements the
 * reference leaves unimplemented!() and non-constant textures return FTN_ERR_UNSUPPORTED; Integrator / PixelFilter / Accelerator
 * are ignored as in the reference (pbrt.rs:528-530).  The descriptors stay owned by the handle.                           */
typedef struct ftn_pbrt ftn_pbrt;
int ftn_pbrt_load(const char* path, ftn_pbrt** out);
void ftn_pbrt_destroy(ftn_pbrt* p);
const ftn_scene_desc* ftn_pbrt_scene(const ftn_pbrt* p);      /* pass to ftn_scene_create                      */
const ftn_camera_desc* ftn_pbrt_camera(const ftn_pbrt* p);    /* PbrtHeader::make_camera                       */
const ftn_film_desc* ftn_pbrt_film(const ftn_pbrt* p);        /* PbrtHeader::make_film                         */
int ftn_pbrt_samples_per_pixel(const ftn_pbrt* p);            /* Sampler "pixelsamples", default 16 (pbrt.rs:470-472) */
const char* ftn_pbrt_film_name(const ftn_pbrt* p);            /* Film "filename"                               */
const char* ftn_pbrt_last_error(void);                        /* message of the last failed ftn_pbrt_load / ftn_ply_load on this thread */
/* make_triangle_mesh_from_ply's reader (constructors.rs:94-190): ASCII or binary_little_endian, float x y z [nx ny nz] [u v],
 * triangle faces only.  Call once with NULL arrays for the counts, then with arrays of 3*nv / 3*nv / 2*nv floats and 3*nt indices. */
int ftn_ply_load(const char* path, uint32_t* n_vertices, uint32_t* n_triangles, float* P, float* N, float* UV,
                 uint32_t* indices, int* has_normals, int* has_uvs);

/* ------------------------------------------------------------------ output stage
 * Film::into_spectrum_buffer (src/film.rs:195-210) on the device: device_pixels -> device_rgb (3 floats per pixel), both DEVICE
 * pointers, on `stream`.  Bit-identical to ftn_film_resolve on the host.                                                  */
/* load_mipmap's gamma step (imageio/mod.rs:101-107, 169-175): every channel value v of a gamma-encoded image map becomes
 * v / 12.92 (v <= 0.04045) or ((v + 0.055) / 1.055)^2.4 -- BEFORE the texture's `scale` is applied.  In place, n floats, host memory.
 * The reference applies it to every non-EXR map and to any map with `"bool gamma" "true"`. */
int ftn_image_inverse_gamma(float* texels, size_t n);
int ftn_film_resolve_device(const void* device_pixels, size_t n_pixels, void* device_rgb_out, void* stream);
/* write_exr / read_exr (src/imageio/exr.rs:11-87): scanline OpenEXR, FLOAT channels R G B, layer "image"; rgb is row-major,
 * 3 floats per pixel.  The writer emits NO_COMPRESSION blocks (the reference's `exr` crate writes RLE: same samples, same layer);
 * the reader accepts NO_COMPRESSION, RLE, ZIPS and ZIP scanline files with FLOAT or HALF channels.  ftn_exr_read with rgb_out == NULL only
 * reports the size.                                                                                                        */
int ftn_exr_write(const char* path, const float* rgb, uint32_t width, uint32_t height);
int ftn_exr_read(const char* path, uint32_t* width, uint32_t* height, float* rgb_out);
const char* ftn_imageio_last_error(void);

/* ------------------------------------------------------------------ test hook
 * Evaluates the deterministic math of the kernels ON THE DEVICE for arrays of HOST floats (parity tests compare the bits
 * with a CPU evaluation): which = 0 sin, 1 cos, 2 tan, 3 acos, 4 atan, 5 atan2(x,y), 6 ln, 7 log2, 8 sqrt, 9 x/y,
 * 10 (float)sqrt((double)x * y)  [the f64 path of math.rs:37-42], 11 next_float_up, 12 next_float_down.                 */
int ftn_test_math(int which, const float* x, const float* y, size_t n, float* out);
/* Texture path hooks.  ftn_test_mipmap_level (host only): level `level` of the pyramid MIPMap::new builds for an image (mipmap.rs:78-145);
 * pass rgb_out = NULL for the size.  ftn_test_texture_eval: Texture::evaluate ON THE DEVICE for n rows of
 * {u, v, dudx, dvdx, dudy, dvdy} (HOST memory) -> rgb_out[3n] (a float texture repeats its value).                              */
int ftn_test_mipmap_level(uint32_t width, uint32_t height, const float* texels, uint32_t level, uint32_t* level_w, uint32_t* level_h, float* rgb_out);
int ftn_test_texture_eval(const ftn_scene* scene, int32_t texture, const float* uv_diffs6, size_t n, float* rgb_out);

/* ------------------------------------------------------------------ misc */
const char* ftn_last_error(void);
int ftn_device_count(void);
const char* ftn_version(void);
/* Layout version of the structs in this header (ftn_stats, ftn_mesh, ftn_scene_desc, ... are written to / read from caller memory):
 * a caller built against another header must not call anything else.  The shim compares it with the FTN_ABI_VERSION it was compiled
 * against when it loads the library.  History: 1 = round 1; 2 = round 2 (ftn_stats 96 -> 152 bytes, ftn_mesh 16 -> 20,
 * ftn_scene_desc gained `S`); 3 = round 3 (ftn_scene_memory added, no struct changed size). */
#define FTN_ABI_VERSION 3
int ftn_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FOUNTAIN_HIP_H */
