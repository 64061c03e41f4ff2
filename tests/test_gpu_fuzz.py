"""Differential fuzzing (-m gpu): seeded random scenes -- every shape, material, texture and light kind the path supports, random
transforms (rotations, non-uniform and mirroring scales), partial spheres, meshes with and without normals / uvs, thin-lens or pinhole
cameras, odd film sizes and crop windows -- rendered by the HIP path and by the deterministic-math oracle from the same recipe.
Films must agree bit for bit (up to the last bit of spill pixels), ray counts exactly; an error must be the same error."""
import numpy as np
import pytest

from fountain_amd import (DirectLightingIntegrator, Film, FountainError, PathIntegrator, PerspectiveCamera, RandomSampler, SamplerIntegrator,
                          SceneBuilder, WhittedIntegrator, _abi as A)

pytestmark = pytest.mark.gpu
MEGA, WAVE = A.FTN_PIPELINE_MEGAKERNEL, A.FTN_PIPELINE_WAVEFRONT


def make_recipe(seed, env_only=False, tri_only=False):
    rng = np.random.default_rng((5000 if env_only else 1000) + seed + (900000 if tri_only else 0))
    U = lambda a, b: float(rng.uniform(a, b))
    col = lambda lo=0.05, hi=0.95: tuple(float(x) for x in rng.uniform(lo, hi, 3))
    ops = []
    tex_spec, tex_float = [], []
    if rng.random() < 0.7 and not env_only:      # (textured scenes take the generic shading kernel)
        ops.append(("texture", "chk", "spectrum", "checkerboard", dict(uscale=U(1, 9), vscale=U(1, 9), tex1=col(), tex2=col()))); tex_spec.append("chk")
        ops.append(("texture", "grid", "spectrum", "uv", dict(uscale=U(0.5, 4), vscale=U(0.5, 4), udelta=U(-1, 1)))); tex_spec.append("grid")
        img = rng.random((int(rng.integers(1, 14)), int(rng.integers(1, 14)), 3)).astype(np.float32)
        ops.append(("texture", "img", "spectrum", "imagemap", dict(texels=img, wrap=str(rng.choice(["repeat", "black", "clamp"])), scale=U(0.5, 1.5), uscale=U(0.5, 3)))); tex_spec.append("img")
        ops.append(("texture", "nest", "spectrum", "checkerboard", dict(uscale=U(1, 4), vscale=U(1, 4), tex1="img", tex2="grid"))); tex_spec.append("nest")
        ops.append(("texture", "frough", "float", "checkerboard", dict(uscale=U(1, 6), vscale=U(1, 6), tex1=U(0.01, 0.2), tex2=U(0.2, 0.8)))); tex_float.append("frough")
    S = lambda: (str(rng.choice(tex_spec)) if tex_spec and rng.random() < 0.5 else col())
    F = lambda lo, hi: (str(rng.choice(tex_float)) if tex_float and rng.random() < 0.4 else U(lo, hi))

    def material():
        k = int(rng.integers(0, 6))
        if k == 0: return ("material", "matte", dict(Kd=S(), sigma=(0.0 if rng.random() < 0.5 else U(1, 60))))
        if k == 1: return ("material", "mirror", dict(Kr=S()))
        if k == 2: return ("material", "plastic", dict(Kd=S(), Ks=S(), roughness=F(0.02, 0.5), remaproughness=bool(rng.random() < 0.7)))
        if k == 3: return ("material", "metal", dict(eta=S(), k=col(1.0, 4.0), uroughness=F(0.01, 0.4), vroughness=F(0.01, 0.4), remaproughness=bool(rng.random() < 0.7)))
        if k == 4: return ("material", "metal", dict(eta=col(0.1, 1.5), k=col(1.0, 4.0), roughness=F(0.01, 0.4)))
        return ("material", "glass", dict(Kr=S(), Kt=S(), uroughness=U(0.05, 0.5), vroughness=U(0.05, 0.5), eta=U(1.1, 1.8)))

    def xform():
        t = [("translate", tuple(float(x) for x in rng.uniform(-2.5, 2.5, 3)))]
        if rng.random() < 0.7: t.append(("rotate", U(-180, 180), tuple(float(x) for x in rng.normal(size=3))))
        if rng.random() < 0.5:
            s = [U(0.5, 1.6), U(0.5, 1.6), U(0.5, 1.6)]
            if rng.random() < 0.3: s[int(rng.integers(0, 3))] *= -1.0              # swaps handedness
            t.append(("scale",) + tuple(s))
        return t

    # lights
    n_expl = 0
    for _ in range(1 if env_only else int(rng.integers(1, 4))):
        k = 2 if env_only else int(rng.integers(0, 3))
        if k == 0: ops.append(("light", "point", dict(I=col(5, 40), from_=tuple(float(x) for x in rng.uniform(-4, 4, 3) + np.array([0, 0, 5])))))
        elif k == 1: ops.append(("light", "distant", dict(L=col(0.5, 3), from_=tuple(float(x) for x in rng.normal(size=3) + np.array([0, 0, 2])), to=(0.0, 0.0, 0.0))))
        else:
            n = int(rng.choice([1, 4, 16]))
            tex = (rng.random((n, n, 3)) ** 3 * 2).astype(np.float32)
            ops.append(("begin",)); ops.append(("rotate", U(0, 360), (0.3, 0.2, 1.0))); ops.append(("light", "infinite", dict(texels=tex))); ops.append(("end",))
        n_expl += 1
    # ground
    ops.append(material())
    ops.append(("shape", "trianglemesh", dict(P=[(-8, -8, -3), (8, -8, -3), (8, 8, -3), (-8, 8, -3)], uv=[0, 0, 1, 0, 1, 1, 0, 1], indices=[0, 1, 2, 0, 2, 3])))
    # objects
    for _ in range(int(rng.integers(4, 12))):
        ops.append(("begin",))
        ops.append(material())
        if rng.random() < 0.25 and not env_only: ops.append(("area", col(2, 12)))
        if rng.random() < 0.3: ops.append(("reverse",))
        ops.extend(xform())
        if rng.random() < 0.55 and not tri_only:
            r = U(0.3, 1.1)
            kw = dict(radius=r)
            if rng.random() < 0.4: kw.update(zmin=-r * U(0.2, 1.0), zmax=r * U(0.2, 1.0), phimax=U(90, 360))
            ops.append(("shape", "sphere", kw))
        else:
            nv, nt = int(rng.integers(4, 40)), int(rng.integers(2, 60))
            P = rng.normal(size=(nv, 3)).astype(np.float32) * 0.7
            kw = dict(P=P, indices=rng.integers(0, nv, size=(nt, 3)).astype(np.uint32))     # degenerate triangles included
            if rng.random() < 0.5: kw["N"] = (P / np.maximum(np.linalg.norm(P, axis=1, keepdims=True), 1e-6)).astype(np.float32)
            if rng.random() < 0.5: kw["uv"] = rng.random((nv, 2)).astype(np.float32)
            ops.append(("shape", "trianglemesh", kw))
        ops.append(("end",))
    res = (int(rng.integers(40, 130)), int(rng.integers(33, 100)))
    eye = tuple(float(x) for x in rng.normal(size=3) * np.array([1.5, 1.5, 0.6]) + np.array([0, -7, 1.5]))
    cam = dict(eye=eye, look=(U(-0.5, 0.5), 0.0, U(-1, 0.5)), fov=U(25, 70))
    if rng.random() < 0.5: cam.update(lens_radius=U(0.02, 0.3), focal_dist=U(4, 9))
    crop = (0.0, 0.0, 1.0, 1.0) if rng.random() < 0.5 else (U(0, 0.3), U(0, 0.3), U(0.6, 1.0), U(0.6, 1.0))
    return ops, res, cam, crop


def build(be, recipe):
    ops, res, cam, crop = recipe
    b = SceneBuilder(be)
    for op in ops:
        k = op[0]
        if k == "texture": b.texture(op[1], op[2], op[3], **op[4])
        elif k == "material": b.material(op[1], **op[2])
        elif k == "light": b.light_source(op[1], **op[2])
        elif k == "shape": b.shape(op[1], **op[2])
        elif k == "begin": b.attribute_begin()
        elif k == "end": b.attribute_end()
        elif k == "translate": b.translate(op[1])
        elif k == "rotate": b.rotate(op[1], op[2])
        elif k == "scale": b.scale(op[1], op[2], op[3])
        elif k == "area": b.area_light_source("diffuse", L=op[1])
        elif k == "reverse": b.reverse_orientation()
    kw = {k: v for k, v in cam.items() if k in ("fov", "lens_radius", "focal_dist")}
    camera = PerspectiveCamera.look_at(be, cam["eye"], cam["look"], (0, 0, 1), res, **kw)
    return b.create_scene(), camera, res, crop


def render(be, scene, camera, res, crop, integ, sampler, pipeline):
    film = Film(be, res, crop)
    si = SamplerIntegrator(camera, integ)
    kw = dict(pipeline=pipeline) if not be.is_oracle else {}
    try:
        st = si.render_parallel(scene, film, sampler, **kw)
        return film.pixels, st, None
    except FountainError as e:
        return film.pixels, None, e.code


@pytest.mark.parametrize("seed", range(24))
def test_random_scene(gpu, orc_det, seed):
    check_recipe(gpu, orc_det, make_recipe(seed), seed)


@pytest.mark.parametrize("seed", range(8))
def test_random_environment_lit_scene(gpu, orc_det, seed):
    """same, with ONE infinite light and no emissive shape: the scenes the environment-only shading kernels take"""
    check_recipe(gpu, orc_det, make_recipe(seed, env_only=True), seed)


@pytest.mark.parametrize("seed", range(12))
def test_random_triangle_only_scene(gpu, orc_det, seed):
    """scenes without spheres (random meshes with degenerate, repeated and sliver triangles; every light kind): their shadow and MIS rays
    walk the eight-box occlusion records (k_wf_trace8_any) -- multi-primitive leaves with explicit boxes, flat boxes, tiny next to large"""
    check_recipe(gpu, orc_det, make_recipe(seed, env_only=seed % 3 == 0, tri_only=True), seed)


def check_recipe(gpu, orc_det, recipe, seed):
    g = build(gpu, recipe)
    o = build(orc_det, recipe)
    assert g[0].info()["n_nodes"] == o[0].info()["n_nodes"]
    gn, go = g[0].nodes(), o[0].nodes()
    assert np.array_equal(gn[0], go[0]) and np.array_equal(gn[1], go[1])                     # same BVH, same primitive order
    cases = [(PathIntegrator.new(5, 1.0), RandomSampler(4, seed, indexed=True), WAVE), (PathIntegrator.new(5, 1.0), RandomSampler(4, seed, indexed=True), MEGA),
             (PathIntegrator.new(3, 0.5), RandomSampler(1, 0), MEGA), (DirectLightingIntegrator(3), RandomSampler(2, 0, indexed=True), MEGA),
             (WhittedIntegrator(3), RandomSampler(2, 0, indexed=True), MEGA),
             # AUTO: the wavefront stages of the two integrators (textured scenes included; Whitted up to 32 lights), the megakernel otherwise
             (DirectLightingIntegrator(4), RandomSampler(2, seed, indexed=True), A.FTN_PIPELINE_AUTO), (WhittedIntegrator(4), RandomSampler(2, seed, indexed=True), A.FTN_PIPELINE_AUTO)]
    for integ, smp, pl in cases:
        px, st, err = render(gpu, *g, integ, smp, pl)
        pxo, sto, erro = render(orc_det, *o, integ, smp, pl)
        what = "seed %d %s pipeline %d" % (seed, type(integ).__name__, pl)
        assert err == erro, what
        if err is not None:
            continue
        assert st["rays_closest"] == sto["rays_closest"] and st["rays_any"] == sto["rays_any"] and st["camera_samples"] == sto["camera_samples"], what
        diff = (px.view(np.uint32) != pxo.view(np.uint32)).any(axis=-1)
        assert int(diff.sum()) <= 4 * st["spill_samples"], "%s: %d pixels differ, %d spill samples" % (what, int(diff.sum()), st["spill_samples"])
        assert np.allclose(px, pxo, rtol=2e-6, atol=1e-7, equal_nan=True), what
