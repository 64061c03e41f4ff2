"""Malformed scene descriptions are refused with FTN_ERR_INVALID_ARGUMENT / FTN_ERR_UNSUPPORTED by the host-side validation (run through
ftn_bvh_build, which needs no GPU) and by the oracle -- never a crash."""
import ctypes as C

import numpy as np
import pytest

from fountain_amd import SceneBuilder, _abi as A


def base(be):
    b = SceneBuilder(be)
    b.texture("chk", "spectrum", "checkerboard", tex1=(0.1, 0.1, 0.1), tex2=(0.9, 0.9, 0.9))
    b.texture("img", "spectrum", "imagemap", texels=np.ones((4, 4, 3), np.float32))
    b.material("matte", Kd="chk")
    b.shape("trianglemesh", P=[(0, 0, 0), (1, 0, 0), (0, 1, 0)], N=[(0, 0, 1)] * 3, uv=[0, 0, 1, 0, 0, 1], indices=[0, 1, 2])
    b.material("matte", Kd="img")
    b.shape("sphere")
    b.light_source("infinite", texels=np.ones((4, 4, 3), np.float32))
    return b.build_desc()


def bvh_build(ftn, d):
    fn = ftn.lib.ftn_bvh_build
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    n = C.c_uint32()
    return fn(C.byref(d), None, None, C.cast(C.byref(n), C.c_void_p), None)


def orc_create(orc, d):
    h = C.c_void_p()
    rc = orc.lib.orc_scene_create(C.byref(d), C.byref(h))
    if rc == 0:
        orc.lib.orc_scene_destroy(h)
    return rc


def test_the_base_description_is_accepted(ftn, orc):
    d, keep = base(ftn)
    assert bvh_build(ftn, d) == 0
    d, keep = base(orc)
    assert orc_create(orc, d) == 0


MUTATIONS = {
    "prim shape index": lambda d: setattr(d.prims[0], "shape_index", 7),
    "prim shape kind": lambda d: setattr(d.prims[0], "shape_kind", 9),
    "prim material": lambda d: setattr(d.prims[0], "material", 99),
    "prim area light": lambda d: setattr(d.prims[0], "area_emit", 3),
    "vertex index": lambda d: d.tri_indices.__setitem__(1, 1000),
    "mesh id": lambda d: d.tri_mesh.__setitem__(0, 5),
    "light type": lambda d: setattr(d.lights[0], "type", 8),
    "envmap index": lambda d: setattr(d.lights[0], "envmap", 4),
    "envmap without texels": lambda d: setattr(d.envmaps[0], "width", 0),
    "checkerboard child": lambda d: setattr(d.textures[2], "tex1", 40),
    "texture kind": lambda d: setattr(d.textures[0], "kind", 17),
    "image index": lambda d: setattr(d.textures[3], "image", 2),
    "image wrap": lambda d: setattr(d.images[0], "wrap", 5),
    "material texture index": lambda d: setattr(d.material_textures[1], "a", 1000),
}


@pytest.mark.parametrize("name", sorted(MUTATIONS))
def test_malformed_description_is_refused(ftn, orc, name):
    for be in (ftn, orc):
        d, keep = base(be)
        MUTATIONS[name](d)
        rc = bvh_build(ftn, d) if be is ftn else orc_create(orc, d)
        assert rc in (A.FTN_ERR_INVALID_ARGUMENT, A.FTN_ERR_UNSUPPORTED), (name, be.prefix, rc)


def test_null_arrays_are_refused(ftn):
    d, keep = base(ftn)
    for field in ("prims", "tri_indices", "P", "meshes", "spheres", "materials", "lights", "envmaps"):
        d2, keep2 = base(ftn)
        setattr(d2, field, None)
        assert bvh_build(ftn, d2) == A.FTN_ERR_INVALID_ARGUMENT, field
    d2, keep2 = base(ftn)
    d2.N = None
    assert bvh_build(ftn, d2) == A.FTN_ERR_INVALID_ARGUMENT
    d2, keep2 = base(ftn)
    d2.materials[0].type = 11
    assert bvh_build(ftn, d2) == A.FTN_ERR_INVALID_ARGUMENT


# ------------------------------------------------------------------ render arguments (sample ranges are validated at the ABI)
BAD_RANGES = [(17, 0), (4, 13), (0, 17), (0xFFFFFFFF, 2), (16, 1)]          # (first_sample, sample_count) at 16 spp
GOOD_RANGES = [(0, 0), (0, 16), (15, 1), (16, 0), (3, 5)]


def _render_range(be, first, count, **kw):
    from fountain_amd import Film, PathIntegrator, RandomSampler, SamplerIntegrator, scenes
    b, cam, res = scenes.furnace(be, res=16)
    scene = b.create_scene()
    si = SamplerIntegrator(cam, PathIntegrator(3, 1.0))
    film = Film(be, res)
    return si.render_parallel(scene, film, RandomSampler(16, 0, indexed=True, first_sample=first, sample_count=count), tiles=(0, 1, 1), **kw), film


@pytest.mark.parametrize("first,count", BAD_RANGES)
def test_oracle_refuses_sample_ranges_outside_the_pixel_samples(orc, first, count):
    from fountain_amd.api import FountainError
    with pytest.raises(FountainError) as e:
        _render_range(orc, first, count)
    assert e.value.code == A.FTN_ERR_INVALID_ARGUMENT


@pytest.mark.parametrize("first,count", GOOD_RANGES)
def test_oracle_accepts_sample_ranges_inside_the_pixel_samples(orc, first, count):
    st, film = _render_range(orc, first, count)
    want = (16 - first) if count == 0 else count
    assert st["camera_samples"] == 256 * want


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", [A.FTN_PIPELINE_MEGAKERNEL, A.FTN_PIPELINE_WAVEFRONT])
def test_gpu_refuses_sample_ranges_outside_the_pixel_samples(gpu, pipeline):
    from fountain_amd.api import FountainError
    for first, count in BAD_RANGES:
        with pytest.raises(FountainError) as e:
            _render_range(gpu, first, count, pipeline=pipeline)
        assert e.value.code == A.FTN_ERR_INVALID_ARGUMENT, (first, count)
    for first, count in GOOD_RANGES:
        st, film = _render_range(gpu, first, count, pipeline=pipeline)
        assert st["camera_samples"] == 256 * ((16 - first) if count == 0 else count)


@pytest.mark.gpu
def test_gpu_refuses_a_device_other_than_the_scenes(gpu):
    from fountain_amd.api import FountainError
    with pytest.raises(FountainError) as e:
        _render_range(gpu, 0, 0, device=5)
    assert e.value.code == A.FTN_ERR_INVALID_ARGUMENT
