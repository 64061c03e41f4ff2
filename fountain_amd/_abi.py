"""ctypes mirror of include/fountain_hip.h (the C ABI of the MI355X path-tracing core).

Only layout lives here: every struct below must match the header field for field
(tests/test_abi.py checks sizes and that the shared library exports each declared symbol).
"""
import ctypes as C

c_f = C.c_float
c_u32 = C.c_uint32
c_i32 = C.c_int32
c_u64 = C.c_uint64

FTN_OK = 0
FTN_ERR_INVALID_ARGUMENT = -1
FTN_ERR_NO_DEVICE = -2
FTN_ERR_OUT_OF_MEMORY = -3
FTN_ERR_NAN_RADIANCE = -4
FTN_ERR_UNSUPPORTED = -5
FTN_ERR_BVH_TOO_DEEP = -6
FTN_ERR_INTERNAL = -7

FTN_SHAPE_TRIANGLE, FTN_SHAPE_SPHERE = 0, 1
FTN_MAT_MATTE, FTN_MAT_METAL, FTN_MAT_MIRROR, FTN_MAT_PLASTIC, FTN_MAT_GLASS = range(5)
FTN_LIGHT_POINT, FTN_LIGHT_DISTANT, FTN_LIGHT_INFINITE = range(3)
FTN_SAMPLER_TILE_SERIAL, FTN_SAMPLER_INDEXED = 0, 1
FTN_INTEGRATOR_PATH, FTN_INTEGRATOR_DIRECT_LIGHTING, FTN_INTEGRATOR_WHITTED = 0, 1, 2
FTN_PIPELINE_AUTO, FTN_PIPELINE_MEGAKERNEL, FTN_PIPELINE_WAVEFRONT = 0, 1, 2


class ftn_transform(C.Structure):
    _fields_ = [("m", c_f * 16), ("inv", c_f * 16)]


class ftn_pixel(C.Structure):
    _fields_ = [("xyz", c_f * 3), ("filter_weight_sum", c_f)]


class ftn_bvh_node(C.Structure):
    _fields_ = [("bmin", c_f * 3), ("bmax", c_f * 3), ("idx", c_u32), ("n_prims", C.c_uint16),
                ("axis", C.c_uint8), ("is_leaf", C.c_uint8)]


class ftn_prim(C.Structure):
    _fields_ = [("shape_kind", c_u32), ("shape_index", c_u32), ("material", c_i32), ("area_emit", c_i32)]


class ftn_mesh(C.Structure):
    _fields_ = [("has_normals", c_u32), ("has_uvs", c_u32), ("flip_normals", c_u32), ("reverse_orientation", c_u32), ("has_tangents", c_u32)]


class ftn_sphere(C.Structure):
    _fields_ = [("object_to_world", ftn_transform), ("world_to_object", ftn_transform),
                ("radius", c_f), ("z_min", c_f), ("z_max", c_f), ("theta_min", c_f), ("theta_max", c_f),
                ("phi_max", c_f), ("reverse_orientation", c_u32), ("_pad", c_u32)]


class ftn_material(C.Structure):
    _fields_ = [("type", c_u32), ("remap_roughness", c_u32), ("a", c_f * 3), ("b", c_f * 3),
                ("s0", c_f), ("s1", c_f), ("s2", c_f), ("_pad", c_f)]


class ftn_light(C.Structure):
    _fields_ = [("type", c_u32), ("envmap", c_i32), ("rgb", c_f * 3), ("v", c_f * 3),
                ("light_to_world", ftn_transform)]


class ftn_envmap(C.Structure):
    _fields_ = [("width", c_u32), ("height", c_u32), ("texels", C.POINTER(c_f))]


FTN_TEX_CONSTANT, FTN_TEX_UV, FTN_TEX_CHECKERBOARD, FTN_TEX_IMAGE = range(4)
FTN_WRAP_REPEAT, FTN_WRAP_BLACK, FTN_WRAP_CLAMP = range(3)


class ftn_texture(C.Structure):
    _fields_ = [("kind", c_u32), ("is_float", c_u32), ("value", c_f * 3), ("tex1", c_i32), ("tex2", c_i32), ("image", c_i32),
                ("su", c_f), ("sv", c_f), ("du", c_f), ("dv", c_f)]


class ftn_image(C.Structure):
    _fields_ = [("width", c_u32), ("height", c_u32), ("wrap", c_u32), ("_pad", c_u32), ("texels", C.POINTER(c_f))]


class ftn_material_textures(C.Structure):
    _fields_ = [("a", c_i32), ("b", c_i32), ("s0", c_i32), ("s1", c_i32), ("s2", c_i32), ("_pad", c_i32 * 3)]


class ftn_scene_desc(C.Structure):
    _fields_ = [
        ("n_prims", c_u32), ("prims", C.POINTER(ftn_prim)),
        ("n_triangles", c_u32), ("tri_indices", C.POINTER(c_u32)), ("tri_mesh", C.POINTER(c_u32)),
        ("n_vertices", c_u32), ("P", C.POINTER(c_f)), ("N", C.POINTER(c_f)), ("UV", C.POINTER(c_f)),
        ("n_meshes", c_u32), ("meshes", C.POINTER(ftn_mesh)),
        ("n_spheres", c_u32), ("spheres", C.POINTER(ftn_sphere)),
        ("n_materials", c_u32), ("materials", C.POINTER(ftn_material)),
        ("n_area_emit", c_u32), ("area_emit", C.POINTER(c_f)),
        ("n_lights", c_u32), ("lights", C.POINTER(ftn_light)),
        ("n_envmaps", c_u32), ("envmaps", C.POINTER(ftn_envmap)),
        ("n_textures", c_u32), ("textures", C.POINTER(ftn_texture)), ("material_textures", C.POINTER(ftn_material_textures)),
        ("n_images", c_u32), ("images", C.POINTER(ftn_image)),
        ("S", C.POINTER(c_f)),
    ]


class ftn_camera_desc(C.Structure):
    _fields_ = [("camera_to_world", ftn_transform), ("raster_to_camera", ftn_transform),
                ("shutter_open", c_f), ("shutter_close", c_f), ("lens_radius", c_f), ("focal_dist", c_f),
                ("dx_camera", c_f * 3), ("dy_camera", c_f * 3)]


class ftn_film_desc(C.Structure):
    _fields_ = [("full_resolution", c_i32 * 2), ("crop", c_i32 * 4), ("filter_radius", c_f * 2)]


class ftn_sampler_desc(C.Structure):
    _fields_ = [("kind", c_u32), ("samples_per_pixel", c_u32), ("seed", c_u64),
                ("first_sample", c_u32), ("sample_count", c_u32)]


class ftn_integrator_desc(C.Structure):
    _fields_ = [("kind", c_u32), ("max_depth", c_u32), ("rr_threshold", c_f), ("_pad", c_u32)]


class ftn_tile_range(C.Structure):
    _fields_ = [("first", c_u32), ("stride", c_u32), ("count", c_u32), ("_pad", c_u32)]


class ftn_render_options(C.Structure):
    _fields_ = [("pipeline", c_u32), ("device", c_i32), ("count_traffic", c_u32), ("_pad", c_u32)]


class ftn_stats(C.Structure):
    _fields_ = [("rays_closest", c_u64), ("rays_any", c_u64), ("nodes_visited", c_u64), ("prims_tested", c_u64),
                ("camera_samples", c_u64), ("spill_samples", c_u64), ("kernel_ms", C.c_double),
                ("trace_ms", C.c_double), ("trace_launches", c_u64), ("nodes_visited_any", c_u64),
                ("prims_tested_any", c_u64), ("mis_rays_any_hit", c_u64), ("quad_records", c_u64), ("quad_records_any", c_u64),
                ("any_ms", C.c_double), ("any_launches", c_u64), ("shade_ms", C.c_double), ("shade_launches", c_u64), ("sort_ms", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class ftn_scene_memory(C.Structure):
    _fields_ = [(k, c_u64) for k in ("nodes", "quad", "oct", "fat", "geom", "srec", "indexed_attributes", "prim_class", "lights", "textures", "other", "total")]


FTN_ABI_VERSION = 3          # include/fountain_hip.h; Backend() refuses a product library that reports another one

# Expected sizes (bytes) -- asserted against the header by the C side's static_asserts and tests/test_abi.py
SIZES = {
    "ftn_transform": 128, "ftn_pixel": 16, "ftn_bvh_node": 32, "ftn_prim": 16, "ftn_mesh": 20,
    "ftn_sphere": 288, "ftn_material": 48, "ftn_light": 160, "ftn_envmap": 16, "ftn_camera_desc": 296,
    "ftn_film_desc": 32, "ftn_sampler_desc": 24, "ftn_integrator_desc": 16, "ftn_tile_range": 16,
    "ftn_render_options": 16, "ftn_stats": 152, "ftn_scene_memory": 96, "ftn_texture": 48, "ftn_image": 24, "ftn_material_textures": 32,
}

# Every function the header declares (name -> None); used by the symbol-export test.
DECLARED_FUNCTIONS = [
    "ftn_transform_identity", "ftn_transform_translate", "ftn_transform_scale", "ftn_transform_rotate",
    "ftn_transform_look_at", "ftn_transform_from_flat", "ftn_transform_mul", "ftn_transform_inverse",
    "ftn_transform_perspective", "ftn_transform_point", "ftn_transform_vector", "ftn_transform_normal",
    "ftn_transform_swaps_handedness", "ftn_transform_points", "ftn_transform_normals",
    "ftn_sphere_init", "ftn_camera_perspective", "ftn_film_init",
    "ftn_film_sample_bounds", "ftn_film_tile_count", "ftn_film_resolve", "ftn_scene_create",
    "ftn_scene_destroy", "ftn_bvh_build", "ftn_bvh_quads", "ftn_bvh_octs", "ftn_scene_info", "ftn_scene_get_nodes", "ftn_scene_get_lights",
    "ftn_intersect", "ftn_intersect_test", "ftn_intersect_full", "ftn_render", "ftn_render_device",
    "ftn_last_error", "ftn_device_count", "ftn_version", "ftn_abi_version", "ftn_scene_memory_info", "ftn_test_math",
    "ftn_pbrt_load", "ftn_pbrt_destroy", "ftn_pbrt_scene", "ftn_pbrt_camera", "ftn_pbrt_film",
    "ftn_pbrt_samples_per_pixel", "ftn_pbrt_film_name", "ftn_pbrt_last_error", "ftn_ply_load",
    "ftn_test_mipmap_level", "ftn_test_texture_eval", "ftn_film_resolve_device", "ftn_exr_write", "ftn_exr_read", "ftn_imageio_last_error", "ftn_image_inverse_gamma",
]
