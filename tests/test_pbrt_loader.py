"""Scene ingestion: the library's C++ PBRT-subset reader and PLY reader against the call-by-call SceneBuilder
(the path every other parity test uses) -- descriptors must be byte-identical.  Host only: no GPU needed.
tests/golden/furnace_empty.pbrt is the reference's own test scene (testscenes/furnace_empty.pbrt, data, used by tests/furnace.rs)."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from fountain_amd import FountainError, PbrtScene, SceneBuilder, Transform, PerspectiveCamera, Film, load_ply, load_ply_ascii, scenes
from fountain_amd import _abi as A

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype)
    nbytes = n * np.dtype(dtype).itemsize
    return np.frombuffer(C.string_at(C.cast(ptr, C.c_void_p), nbytes), dtype=dtype).copy()


def desc_arrays(d):
    out = {
        "prims": _arr(d.prims, d.n_prims * 4, np.int32),
        "tri_indices": _arr(d.tri_indices, d.n_triangles * 3, np.uint32),
        "tri_mesh": _arr(d.tri_mesh, d.n_triangles, np.uint32),
        "P": _arr(d.P, d.n_vertices * 3, np.uint32),
        "N": _arr(d.N, d.n_vertices * 3, np.uint32) if d.N else None,
        "UV": _arr(d.UV, d.n_vertices * 2, np.uint32) if d.UV else None,
        "meshes": _arr(d.meshes, d.n_meshes * 5, np.uint32),
        "S": _arr(d.S, d.n_vertices * 3, np.uint32) if d.S else None,
        "spheres": _arr(d.spheres, d.n_spheres * 72, np.uint32),
        "materials": _arr(d.materials, d.n_materials * 11, np.uint32).reshape(-1, 11) if d.n_materials else np.zeros((0, 11), np.uint32),
        "area_emit": _arr(d.area_emit, d.n_area_emit * 3, np.uint32),
        "lights": _arr(d.lights, d.n_lights * 40, np.uint32),
        "textures": _arr(d.textures, d.n_textures * 12, np.uint32),
        "material_textures": _arr(d.material_textures, d.n_materials * 8, np.int32).reshape(-1, 8)[:, :5] if d.material_textures else None,
    }
    out["images"] = [(im.width, im.height, im.wrap, _arr(im.texels, im.width * im.height * 3, np.uint32)) for im in (d.images[i] for i in range(d.n_images))]
    # material _pad is not meaningful: compare the 11 leading words only (done by the reshape above: 48 B = 12 words)
    out["materials"] = _arr(d.materials, d.n_materials * 12, np.uint32).reshape(-1, 12)[:, :11]
    envs = []
    for i in range(d.n_envmaps):
        e = d.envmaps[i]
        envs.append((e.width, e.height, _arr(e.texels, e.width * e.height * 3, np.uint32)))
    out["envmaps"] = envs
    return out


def assert_same_desc(a, b):
    da, db = desc_arrays(a), desc_arrays(b)
    for k in da:
        if k in ("envmaps", "images"):
            assert len(da[k]) == len(db[k])
            for x, y in zip(da[k], db[k]):
                assert x[:-1] == y[:-1] and np.array_equal(x[-1], y[-1])
        elif da[k] is None or db[k] is None:
            assert da[k] is None and db[k] is None, k
        else:
            assert da[k].shape == db[k].shape, k
            assert np.array_equal(da[k], db[k]), k


def same_struct(a, b):
    return C.string_at(C.byref(a), C.sizeof(a)) == C.string_at(C.byref(b), C.sizeof(b))


def test_reference_furnace_scene_file(ftn):
    """testscenes/furnace_empty.pbrt == scenes.furnace(): sphere light with ReverseOrientation, LookAt camera, 16x16 film, 128 spp."""
    ps = PbrtScene(os.path.join(GOLD, "furnace_empty.pbrt"), ftn)
    b, cam, res = scenes.furnace(ftn, 16)
    b.attribute_begin(); b.material("matte", Kd=(1, 1, 1)); b.attribute_end()      # the file's second, shapeless block
    d, keep = b.build_desc()
    assert_same_desc(ps.desc, d)
    assert same_struct(ps.camera.desc, cam.desc)
    assert same_struct(ps.film().desc, Film(ftn, res).desc)
    assert ps.samples_per_pixel == 128 and ps.film_name == "furnace.exr"


def test_cornell_scene_file(ftn):
    ps = PbrtScene(os.path.join(GOLD, "cornell.pbrt"), ftn)
    b, cam, res = scenes.cornell(ftn, 64)
    d, keep = b.build_desc()
    assert_same_desc(ps.desc, d)
    assert same_struct(ps.camera.desc, cam.desc)
    assert same_struct(ps.film().desc, Film(ftn, res).desc)
    assert ps.samples_per_pixel == 8


def _write_ply(path, P, N, UV, F, binary):
    props = ["x", "y", "z"] + (["nx", "ny", "nz"] if N is not None else []) + (["u", "v"] if UV is not None else [])
    cols = [P] + ([N] if N is not None else []) + ([UV] if UV is not None else [])
    V = np.concatenate(cols, axis=1).astype("<f4")
    hdr = "ply\nformat %s 1.0\ncomment test\nelement vertex %d\n" % ("binary_little_endian" if binary else "ascii", len(P))
    hdr += "".join("property float %s\n" % p for p in props)
    hdr += "element face %d\nproperty list uchar int vertex_indices\nend_header\n" % len(F)
    with open(path, "wb") as f:
        f.write(hdr.encode())
        if binary:
            f.write(V.tobytes())
            for tri in F:
                f.write(struct.pack("<B3i", 3, *[int(x) for x in tri]))
        else:
            for row in V:
                f.write((" ".join(repr(float(x)) for x in row) + "\n").encode())
            for tri in F:
                f.write(("3 %d %d %d\n" % tuple(int(x) for x in tri)).encode())


@pytest.mark.parametrize("binary", [False, True])
def test_ply_reader_round_trips_the_reference_mesh(ftn, tmp_path, binary):
    """data/rounded_cube.ply's content (tests/golden/rounded_cube.npz) written as ASCII and binary PLY reads back bit-exactly."""
    P, N, F = scenes.rounded_cube_mesh()
    UV = (P[:, :2] * np.float32(0.25) + np.float32(0.5)).astype(np.float32)
    path = str(tmp_path / "m.ply")
    _write_ply(path, P, N, UV, F, binary)
    P2, N2, UV2, F2 = load_ply(path, ftn)
    assert np.array_equal(P2.view(np.uint32), np.asarray(P, np.float32).view(np.uint32))
    assert np.array_equal(N2.view(np.uint32), np.asarray(N, np.float32).view(np.uint32))
    assert np.array_equal(UV2.view(np.uint32), UV.view(np.uint32))
    assert np.array_equal(F2, np.asarray(F, np.uint32))
    if not binary:   # the independent Python reader agrees
        P3, N3, F3 = load_ply_ascii(path)
        assert np.array_equal(P3, P2) and np.array_equal(N3, N2) and np.array_equal(F3, F2)


def test_ply_reader_rejects_quads_and_missing_files(ftn, tmp_path):
    p = tmp_path / "q.ply"
    p.write_text("ply\nformat ascii 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\n"
                 "element face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n1 1 0\n0 1 0\n4 0 1 2 3\n")
    with pytest.raises(FountainError) as e:
        load_ply(str(p), ftn)
    assert e.value.code == A.FTN_ERR_UNSUPPORTED          # "Face with unsupported vertex count found" (constructors.rs:162)
    with pytest.raises(FountainError):
        load_ply(str(tmp_path / "missing.ply"), ftn)


FEATURES = """
# every statement kind of pbrt.rs:178-330 that is implemented in the reference
Film "image" "integer xresolution" [ 96 ] "integer yresolution" [ 48 ] "float cropwindow" [ 0.25 0.75 0.1 0.9 ]
Sampler "halton" "integer pixelsamples" 4
Scale -1 1 1
Rotate 12 0 0 1
LookAt 1 -4 2  0 0 0.25  0 0 1
Camera "perspective" "float fov" 35 "float lensradius" 0.05 "float focaldistance" 4.5 "float shutteropen" 0.25 "float shutterclose" 0.75
Accelerator "bvh"
WorldBegin
MakeNamedMaterial "gold" "string type" "metal" "rgb eta" [0.143 0.375 1.442] "rgb k" [3.983 2.386 1.603] "float roughness" 0.05
MakeNamedMaterial "brushed" "string type" "metal" "rgb eta" [0.2 0.9 1.1] "rgb k" [3.9 2.4 2.2] "float uroughness" 0.02 "float vroughness" 0.2 "bool remaproughness" "false"
LightSource "point" "rgb I" [10 9 8] "rgb scale" [2 2 2] "point from" [0 0 3]
LightSource "distant" "rgb L" [3 3 3] "point from" [1 1 1] "point to" [0 0 0]
TransformBegin
  Rotate -90 1 0 0
  LightSource "infinite" "rgb L" [0.25 0.5 0.75]
TransformEnd
Include "inc.pbrt"
AttributeBegin
  NamedMaterial "gold"
  ConcatTransform [1 0 0 0  0 1 0 0  0 0 1 0  0.5 0.25 0 1]
  Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0] "normal N" [0 0 1 0 0 1 0 0 1] "float uv" [0 0 1 0 0 1]
  Transform [2 0 0 0  0 2 0 0  0 0 2 0  0 0 1 1]
  NamedMaterial "brushed"
  Shape "plymesh" "string filename" "cube.ply"
AttributeEnd
AttributeBegin
  Material "glass" "float eta" 1.33 "rgb Kt" [0.9 0.9 1] "bool remaproughness" "false"
  Scale 1 1 -1
  Shape "sphere" "float radius" 0.5 "float zmin" -0.25 "float zmax" 0.4 "float phimax" 270
  Material "matte" "rgb Kd" [0.1 0.2 0.3] "float sigma" 20
  ReverseOrientation
  Identity
  Translate 0 0 -1
  Shape "sphere"
AttributeEnd
Shape "sphere" "float radius" 0.1
WorldEnd
"""
INC = """AttributeBegin
  Material "mirror"
  Translate 3 0 0
  Shape "sphere" "float radius" 0.75
AttributeEnd
"""


def test_every_supported_statement(ftn, tmp_path):
    P, N, F = scenes.rounded_cube_mesh()
    _write_ply(str(tmp_path / "cube.ply"), P, N, None, F, True)
    (tmp_path / "inc.pbrt").write_text(INC)
    (tmp_path / "s.pbrt").write_text(FEATURES)
    ps = PbrtScene(str(tmp_path / "s.pbrt"), ftn)

    b = SceneBuilder(ftn)
    gold = b._add_material(A.FTN_MAT_METAL, a=(0.143, 0.375, 1.442), b=(3.983, 2.386, 1.603), s1=0.05, s2=0.05)
    brushed = b._add_material(A.FTN_MAT_METAL, a=(0.2, 0.9, 1.1), b=(3.9, 2.4, 2.2), s1=0.02, s2=0.2, remap=False)
    b.light_source("point", I=(10, 9, 8), scale=(2, 2, 2), from_=(0, 0, 3))
    b.light_source("distant", L=(3, 3, 3), from_=(1, 1, 1), to=(0, 0, 0))
    b.attribute_begin(); b.rotate(-90, (1, 0, 0)); b.light_source("infinite", L=(0.25, 0.5, 0.75)); b.attribute_end()
    b.attribute_begin(); b.material("mirror"); b.translate((3, 0, 0)); b.shape("sphere", radius=0.75); b.attribute_end()
    b.attribute_begin()
    b._state[-1]["material"] = gold
    b.concat_transform(Transform.from_flat(ftn, [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0.5, 0.25, 0, 1]))
    b.shape("trianglemesh", indices=[0, 1, 2], P=[0, 0, 0, 1, 0, 0, 0, 1, 0], N=[0, 0, 1, 0, 0, 1, 0, 0, 1], uv=[0, 0, 1, 0, 0, 1])
    b._tf[-1] = Transform.from_flat(ftn, [2, 0, 0, 0, 0, 2, 0, 0, 0, 0, 2, 0, 0, 0, 1, 1])
    b._state[-1]["material"] = brushed
    b.add_mesh(P, N, None, F, b._tf[-1], False, brushed, -1)
    b.attribute_end()
    b.attribute_begin()
    b.material("glass", eta=1.33, Kt=(0.9, 0.9, 1.0), remaproughness=False)
    b.scale(1, 1, -1)
    b.shape("sphere", radius=0.5, zmin=-0.25, zmax=0.4, phimax=270.0)
    b.material("matte", Kd=(0.1, 0.2, 0.3), sigma=20.0)
    b.reverse_orientation(); b.identity(); b.translate((0, 0, -1)); b.shape("sphere")
    b.attribute_end()
    b.shape("sphere", radius=0.1)
    d, keep = b.build_desc()
    assert_same_desc(ps.desc, d)

    w2c = Transform.scale(ftn, -1, 1, 1) * Transform.rotate(ftn, 12, (0, 0, 1)) * Transform.look_at(ftn, (1, -4, 2), (0, 0, 0.25), (0, 0, 1))
    cam = PerspectiveCamera(ftn, w2c.inverse(), (96, 48), shutter=(0.25, 0.75), lens_radius=0.05, focal_dist=4.5, fov=35.0)
    assert same_struct(ps.camera.desc, cam.desc)
    assert same_struct(ps.film().desc, Film(ftn, (96, 48), crop_window=(0.25, 0.1, 0.75, 0.9)).desc)
    assert ps.samples_per_pixel == 4 and ps.film_name == "render.exr"


@pytest.mark.parametrize("body,code", [
    ('ObjectBegin "a"', A.FTN_ERR_UNSUPPORTED),                                             # unimplemented!() pbrt.rs:196
    ('Texture "t" "spectrum" "checkerboard"', A.FTN_ERR_INVALID_ARGUMENT),                 # tex1 / tex2 are required (constructors.rs:265-266)
    ('Texture "t" "float" "uv"', A.FTN_ERR_INVALID_ARGUMENT),                              # UnknownName("float uv") pbrt.rs:381
    ('Texture "t" "spectrum" "uv" "string mapping" "spherical"', A.FTN_ERR_INVALID_ARGUMENT),
    ('Material "matte" "texture Kd" "t"', A.FTN_ERR_INVALID_ARGUMENT),                     # TextureError: lookup_texture pbrt.rs:142-147
    ('Texture "t" "spectrum" "imagemap" "string filename" "x.png"', A.FTN_ERR_UNSUPPORTED),
    ('Material "velvet"', A.FTN_ERR_INVALID_ARGUMENT),                                      # UnknownName
    ('NamedMaterial "nope"', A.FTN_ERR_INVALID_ARGUMENT),                                   # MaterialError
    ('Shape "cone"', A.FTN_ERR_INVALID_ARGUMENT),
    ('Material "metal" "float roughness" 0.1', A.FTN_ERR_INVALID_ARGUMENT),                 # eta / k are required (constructors.rs:215)
    ('AttributeEnd', A.FTN_ERR_INVALID_ARGUMENT),
    ('CoordinateSystem "x"', A.FTN_ERR_UNSUPPORTED),
])
def test_errors_follow_the_reference(ftn, tmp_path, body, code):
    p = tmp_path / "e.pbrt"
    p.write_text('Camera "perspective"\nWorldBegin\n%s\nWorldEnd\n' % body)
    with pytest.raises(FountainError) as e:
        PbrtScene(str(p), ftn)
    assert e.value.code == code, e.value


def test_plastic_reads_lowercase_ks(ftn, tmp_path):
    """constructors.rs:232-238 looks up "ks": a file's "Ks" is ignored and the 0.25 default applies."""
    p = tmp_path / "p.pbrt"
    p.write_text('Camera "perspective"\nWorldBegin\nMaterial "plastic" "rgb Ks" [1 1 1]\nShape "sphere"\nWorldEnd\n')
    ps = PbrtScene(str(p), ftn)
    m = ps.desc.materials[ps.desc.prims[0].material]
    assert m.type == A.FTN_MAT_PLASTIC and list(m.b) == [0.25, 0.25, 0.25]


TEXTURED = """
Camera "perspective"
WorldBegin
Texture "chk" "spectrum" "checkerboard" "float uscale" 8 "float vscale" 4 "rgb tex1" [0.1 0.2 0.3] "rgb tex2" [0.8 0.7 0.6]
Texture "fchk" "float" "checkerboard" "float tex1" 0 "float tex2" 30 "float udelta" 0.5
Texture "grid" "color" "uv" "float uscale" 2 "float vdelta" 0.25
Texture "img" "spectrum" "imagemap" "string filename" "t.exr" "string wrap" "clamp" "float scale" 0.5
Texture "imgg" "spectrum" "imagemap" "string filename" "t.exr" "bool gamma" "true" "float scale" 2
Texture "nest" "spectrum" "checkerboard" "texture tex1" "grid" "texture tex2" "img"
Material "matte" "texture Kd" "chk" "texture sigma" "fchk"
Shape "sphere"
Material "matte" "texture Kd" "fchk"
Shape "sphere" "float radius" 2
Material "plastic" "texture Kd" "nest" "texture ks" "grid" "texture roughness" "fchk"
Shape "sphere" "float radius" 3
Material "metal" "texture eta" "img" "rgb k" [3 2 1] "texture roughness" "fchk"
Shape "sphere" "float radius" 4
Material "mirror" "texture Kr" "chk"
Shape "sphere" "float radius" 5
Material "glass" "texture Kr" "grid" "texture Kt" "chk" "texture uroughness" "fchk" "float vroughness" 0.2
Shape "sphere" "float radius" 6
WorldEnd
"""


def test_texture_statements(ftn, tmp_path):
    """Texture statements and "texture" parameters (pbrt.rs:362-385, constructors.rs:247-318) against SceneBuilder.texture()."""
    from fountain_amd import write_exr
    rng = np.random.default_rng(4)
    img = rng.random((6, 10, 3)).astype(np.float32)              # not a power of two: MIPMap::new takes any size
    write_exr(str(tmp_path / "t.exr"), img, ftn)
    (tmp_path / "s.pbrt").write_text(TEXTURED)
    ps = PbrtScene(str(tmp_path / "s.pbrt"), ftn)
    b = SceneBuilder(ftn)
    b.texture("chk", "spectrum", "checkerboard", uscale=8.0, vscale=4.0, tex1=(0.1, 0.2, 0.3), tex2=(0.8, 0.7, 0.6))
    b.texture("fchk", "float", "checkerboard", tex1=0.0, tex2=30.0, udelta=0.5)
    b.texture("grid", "color", "uv", uscale=2.0, vdelta=0.25)
    b.texture("img", "spectrum", "imagemap", texels=img, wrap="clamp", scale=0.5)
    b.texture("imgg", "spectrum", "imagemap", texels=img, gamma=True, scale=2.0)         # "bool gamma" "true": inverse_gamma_correct, THEN the scale
    b.texture("nest", "spectrum", "checkerboard", tex1="grid", tex2="img")
    b.material("matte", Kd="chk", sigma="fchk"); b.shape("sphere")
    b.material("matte", Kd="fchk"); b.shape("sphere", radius=2.0)             # float texture in a spectrum slot -> the default 0.5
    b.material("plastic", Kd="nest", Ks="grid", roughness="fchk"); b.shape("sphere", radius=3.0)
    b.material("metal", eta="img", k=(3, 2, 1), roughness="fchk"); b.shape("sphere", radius=4.0)
    b.material("mirror", Kr="chk"); b.shape("sphere", radius=5.0)
    b.material("glass", Kr="grid", Kt="chk", uroughness="fchk", vroughness=0.2); b.shape("sphere", radius=6.0)
    d, keep = b.build_desc()
    assert_same_desc(ps.desc, d)
    assert d.n_textures == 10 and d.n_images == 2 and list(d.materials[2].a) == [0.5, 0.5, 0.5] and d.material_textures[2].a == -1
    # the stored image is the file flipped in y and scaled (load_mipmap, imageio/mod.rs:100-117)
    got = np.frombuffer(C.string_at(C.cast(ps.desc.images[0].texels, C.c_void_p), 6 * 10 * 3 * 4), np.float32).reshape(6, 10, 3)
    assert np.array_equal(got, (img * np.float32(0.5))[::-1])
    # the gamma-encoded map (imageio/mod.rs:101-107, 169-175): v / 12.92 below 0.04045, ((v + 0.055) / 1.055)^2.4 above, then * scale, then the flip
    gg = np.frombuffer(C.string_at(C.cast(ps.desc.images[1].texels, C.c_void_p), 6 * 10 * 3 * 4), np.float32).reshape(6, 10, 3)
    v = img.astype(np.float64)
    want = np.where(img <= np.float32(0.04045), v / 12.92, ((v + 0.055) / 1.055) ** 2.4) * 2.0
    assert np.allclose(gg, want[::-1], rtol=3e-7, atol=0) and not np.array_equal(gg, (img * np.float32(2.0))[::-1])


# ------------------------------------------------------------------ SURVEY 8(f).1 on the GPU: parsed scene files rendered by the HIP path vs the oracle
def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _render_parsed(gpu, ps, pipeline, spp):
    sc = ps.create_scene()
    film = ps.film()
    from fountain_amd import PathIntegrator, SamplerIntegrator
    st = SamplerIntegrator(ps.camera, PathIntegrator.new(5, 1.0)).render_parallel(sc, film, ps.sampler(spp, indexed=True), pipeline=pipeline)
    return film.pixels, st


def _render_oracle(orc, builder, cam, res, spp, film_desc=None):
    from fountain_amd import PathIntegrator, RandomSampler, SamplerIntegrator
    sc = builder.create_scene()
    film = Film(orc, res) if film_desc is None else Film.from_desc(orc, film_desc)
    st = SamplerIntegrator(cam, PathIntegrator.new(5, 1.0)).render_parallel(sc, film, RandomSampler(spp, 0, indexed=True))
    return film.pixels, st


def _same_film(px, ref, n_spill, what):
    diff = (_bits(px) != _bits(ref)).any(axis=-1)
    assert int(diff.sum()) <= 4 * n_spill, "%s: %d pixels differ, only %d spill samples" % (what, int(diff.sum()), n_spill)
    assert np.allclose(px, ref, rtol=2e-6, atol=1e-7), what


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", [A.FTN_PIPELINE_MEGAKERNEL, A.FTN_PIPELINE_WAVEFRONT])
@pytest.mark.parametrize("which", ["furnace", "cornell", "plymesh"])
def test_parsed_scene_files_render_like_the_oracle(gpu, orc_det, tmp_path, which, pipeline):
    """ftn_pbrt_load -> ftn_scene_create -> ftn_render on the GPU against the ORACLE rendering the same scene assembled call by call
    (SceneBuilder over the oracle's constructors): the reference's own testscenes/furnace_empty.pbrt, a Cornell file, and a file whose
    geometry comes from a PLY file (data/rounded_cube.ply's content) under an environment light.  loaders/pbrt.rs:178-330,
    constructors.rs:94-190 -> integrator/mod.rs:218."""
    spp = 4
    if which == "furnace":
        ps = PbrtScene(os.path.join(GOLD, "furnace_empty.pbrt"), gpu)
        b, cam, res = scenes.furnace(orc_det, 16)
    elif which == "cornell":
        ps = PbrtScene(os.path.join(GOLD, "cornell.pbrt"), gpu)
        b, cam, res = scenes.cornell(orc_det, 64)
    else:
        P, N, F = scenes.rounded_cube_mesh()
        _write_ply(str(tmp_path / "cube.ply"), P, N, None, F, True)
        (tmp_path / "s.pbrt").write_text(
            'Film "image" "integer xresolution" [ 72 ] "integer yresolution" [ 56 ]\nSampler "random" "integer pixelsamples" 4\n'
            'LookAt 28 -28 14  0 0 -2  0 0 1\nCamera "perspective" "float fov" 40\nWorldBegin\n'
            'LightSource "infinite" "rgb L" [0.9 1.0 1.2]\n'
            'AttributeBegin\n  Material "metal" "rgb eta" [0.2 0.92 1.1] "rgb k" [3.9 2.45 2.14] "float roughness" 0.1\n  Rotate 20 0 0 1\n  Shape "plymesh" "string filename" "cube.ply"\nAttributeEnd\n'
            'AttributeBegin\n  Material "matte" "rgb Kd" [0.5 0.5 0.5]\n  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-60 -60 -9.99  60 -60 -9.99  60 60 -9.99  -60 60 -9.99]\nAttributeEnd\n'
            'WorldEnd\n')
        ps = PbrtScene(str(tmp_path / "s.pbrt"), gpu)
        b = SceneBuilder(orc_det)
        b.light_source("infinite", L=(0.9, 1.0, 1.2))
        b.attribute_begin(); b.material("metal", eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14), roughness=0.1); b.rotate(20, (0, 0, 1)); b.shape("trianglemesh", P=P, N=N, indices=F); b.attribute_end()
        b.attribute_begin(); b.material("matte", Kd=(0.5, 0.5, 0.5)); scenes._quad(b, (-60, -60, -9.99), (60, -60, -9.99), (60, 60, -9.99), (-60, 60, -9.99)); b.attribute_end()
        res = (72, 56)
        cam = PerspectiveCamera.look_at(orc_det, (28, -28, 14), (0, 0, -2), (0, 0, 1), res, fov=40.0)
    px, st = _render_parsed(gpu, ps, pipeline, spp)
    ref, sto = _render_oracle(orc_det, b, cam, res, spp)
    assert st["rays_closest"] == sto["rays_closest"] and st["rays_any"] == sto["rays_any"] and st["camera_samples"] == res[0] * res[1] * spp
    _same_film(px, ref, st["spill_samples"], which)
    assert np.isfinite(px).all() and px[..., :3].max() > 0


# ------------------------------------------------------------------ vertex tangents ("S": constructors.rs:63, triangle.rs:53-58, :341-347)
TANGENT_SCENE = """
Film "image" "integer xresolution" [ 64 ] "integer yresolution" [ 48 ]
Sampler "random" "integer pixelsamples" 4
LookAt 0.5 -3 2.2  0.5 0.5 0  0 0 1
Camera "perspective" "float fov" 38
WorldBegin
LightSource "point" "rgb I" [30 30 30] "point from" [0.2 0.1 3]
LightSource "distant" "rgb L" [2 2 2] "point from" [1 -1 2] "point to" [0 0 0]
AttributeBegin
  Material "metal" "rgb eta" [0.2 0.92 1.1] "rgb k" [3.9 2.45 2.14] "float uroughness" 0.02 "float vroughness" 0.35
  Rotate 25 0 0 1
  Scale 1.5 1 1
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1 -1 0  1 -1 0  1 1 0  -1 1 0] "normal N" [0 0 1 0 0 1 0 0 1 0 0 1]
        "vector S" [1 1 0  1 1 0  0 1 0.2  0 1 0]
AttributeEnd
AttributeBegin
  Material "metal" "rgb eta" [1.5 1.0 0.5] "rgb k" [3 2.5 2] "float uroughness" 0.3 "float vroughness" 0.03
  Translate 1.2 1.6 0.4
  Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0  1 0 0.5  0 1 0.5] "vector S" [0 1 0  0 1 0  1 1 0]
AttributeEnd
WorldEnd
"""


def _tangent_scene(be):
    b = SceneBuilder(be)
    b.light_source("point", I=(30, 30, 30), from_=(0.2, 0.1, 3))
    b.light_source("distant", L=(2, 2, 2), from_=(1, -1, 2), to=(0, 0, 0))
    b.attribute_begin(); b.material("metal", eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14), uroughness=0.02, vroughness=0.35)
    b.rotate(25, (0, 0, 1)); b.scale(1.5, 1, 1)
    b.shape("trianglemesh", indices=[0, 1, 2, 0, 2, 3], P=[(-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0)], N=[(0, 0, 1)] * 4, S=[(1, 1, 0), (1, 1, 0), (0, 1, 0.2), (0, 1, 0)])
    b.attribute_end()
    b.attribute_begin(); b.material("metal", eta=(1.5, 1.0, 0.5), k=(3, 2.5, 2), uroughness=0.3, vroughness=0.03)
    b.translate((1.2, 1.6, 0.4))
    b.shape("trianglemesh", indices=[0, 1, 2], P=[(0, 0, 0), (1, 0, 0.5), (0, 1, 0.5)], S=[(0, 1, 0), (0, 1, 0), (1, 1, 0)])      # tangents without normals
    b.attribute_end()
    cam = PerspectiveCamera.look_at(be, (0.5, -3, 2.2), (0.5, 0.5, 0), (0, 0, 1), (64, 48), fov=38.0)
    return b, cam, (64, 48)


def test_vertex_tangents_are_parsed_like_the_builder(ftn, tmp_path):
    (tmp_path / "t.pbrt").write_text(TANGENT_SCENE)
    ps = PbrtScene(str(tmp_path / "t.pbrt"), ftn)
    b, cam, res = _tangent_scene(ftn)
    d, keep = b.build_desc()
    assert_same_desc(ps.desc, d)
    assert same_struct(ps.camera.desc, cam.desc) and ps.desc.meshes[0].has_tangents == 1 and ps.desc.meshes[1].has_normals == 0 and ps.desc.meshes[1].has_tangents == 1
    # per-vertex array length must match P (make_triangle_mesh's assert_eq!, triangle.rs:54)
    (tmp_path / "bad.pbrt").write_text(TANGENT_SCENE.replace('"vector S" [0 1 0  0 1 0  1 1 0]', '"vector S" [0 1 0  0 1 0]'))
    with pytest.raises(FountainError):
        PbrtScene(str(tmp_path / "bad.pbrt"), ftn)


def test_oracle_shading_frame_follows_the_vertex_tangent(orc):
    """triangle.rs:341-349 on the oracle: with a normal (0,0,1) and a tangent in the plane, the shading dpdu IS the normalised tangent
    (ts = ns x ss, ss = ts x ns); without `S` it is the normalised dpdu of the default uvs, a different direction; a mesh with tangents
    but no normals takes the geometric normal as ns."""
    from fountain_amd import make_rays
    P = [(0, 0, 0), (2, 0, 0), (0, 2, 0)]
    rays = make_rays([(0.5, 0.5, 1.0)], [(0.0, 0.0, -1.0)])
    frames = {}
    for name, kw in (("plain", dict(N=[(0, 0, 1)] * 3)), ("tangent", dict(N=[(0, 0, 1)] * 3, S=[(0, 3, 0)] * 3)), ("tangent_only", dict(S=[(0.6, 0.8, 5.0)] * 3))):
        b = SceneBuilder(orc)
        b.material("matte")
        b.shape("trianglemesh", indices=[0, 1, 2], P=P, **kw)
        frames[name] = b.create_scene().intersect_full(rays)[0]
    assert np.allclose(frames["tangent"][14:17], (0, 1, 0), atol=1e-7) and np.allclose(frames["tangent"][20:23], (0, 0, 1), atol=1e-7)
    assert not np.allclose(frames["plain"][14:17], frames["tangent"][14:17], atol=1e-3)
    # tangents only: ns = geometric normal (0,0,1); ss = the tangent's projection onto the plane, normalised
    assert np.allclose(frames["tangent_only"][20:23], (0, 0, 1), atol=1e-7) and np.allclose(frames["tangent_only"][14:17], (0.6, 0.8, 0), atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", [A.FTN_PIPELINE_MEGAKERNEL, A.FTN_PIPELINE_WAVEFRONT])
def test_vertex_tangents_render_like_the_oracle(gpu, orc_det, tmp_path, pipeline):
    """anisotropic metals whose shading frame comes from per-vertex tangents (one mesh with normals + tangents, one with tangents only):
    the parsed file on the GPU against the oracle's render of the builder scene, and the full interaction records of a ray batch"""
    from fountain_amd import make_rays
    (tmp_path / "t.pbrt").write_text(TANGENT_SCENE)
    ps = PbrtScene(str(tmp_path / "t.pbrt"), gpu)
    px, st = _render_parsed(gpu, ps, pipeline, 4)
    b, cam, res = _tangent_scene(orc_det)
    ref, sto = _render_oracle(orc_det, b, cam, res, 4)
    assert st["rays_closest"] == sto["rays_closest"] and st["rays_any"] == sto["rays_any"]
    _same_film(px, ref, st["spill_samples"], "vertex tangents")
    assert px[..., :3].max() > 0
    rng = np.random.default_rng(4)
    o = np.array([0.5, -3, 2.2], np.float32) + rng.normal(size=(4000, 3)).astype(np.float32) * 0.05
    tgt = rng.uniform(-1.5, 2.5, (4000, 3)).astype(np.float32) * np.array([1, 1, 0.2], np.float32)
    rays = make_rays(o, tgt - o)
    fg, fo = ps.create_scene().intersect_full(rays), b.create_scene().intersect_full(rays)
    hit = fo[:, 23] >= 0
    assert hit.sum() > 500 and np.array_equal(fg[:, 23] >= 0, hit)
    for sl in (slice(0, 9), slice(11, 17), slice(20, 24)):          # p, p_err, n | wo, shading dpdu | shading_n, t  (9-10 uv and 17-19 dpdv: oracle only)
        assert np.array_equal(_bits(fg[:, sl]), _bits(fo[:, sl])), sl
