"""Synthetic scene descriptions for the BASELINE.json configs, built through the reference-named
SceneBuilder.  Everything is closed-form or a fixed-seed generator: no files besides the reference's own test mesh
(tests/golden/rounded_cube.npz, converted from data/rounded_cube.ply).
"""
import os

import numpy as np

from .api import Film, PathIntegrator, PerspectiveCamera, RandomSampler, SamplerIntegrator, SceneBuilder, Transform

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rounded_cube_mesh():
    d = np.load(os.path.join(_ROOT, "tests", "golden", "rounded_cube.npz"))
    return d["P"], d["N"], d["F"]


# ------------------------------------------------------------------ config 1: testscenes/furnace_empty.pbrt
def furnace(be, res=16):
    b = SceneBuilder(be)
    b.attribute_begin()
    b.material("matte", Kd=(0.5, 0.5, 0.5))
    b.area_light_source("diffuse", L=(1.0, 1.0, 1.0))
    b.reverse_orientation()
    b.shape("sphere", radius=100.0)
    b.attribute_end()
    cam = PerspectiveCamera.look_at(be, (0, -2, 0), (0, 0, 0), (0, 0, 1), (res, res), fov=60.0)
    return b, cam, (res, res)


# ------------------------------------------------------------------ config 2: Cornell-style box from reference features only
def _quad(b, p0, p1, p2, p3):
    b.shape("trianglemesh", P=[p0, p1, p2, p3], indices=[0, 1, 2, 0, 2, 3])


def cornell(be, res=512, with_plastic=True):
    """5 matte walls, an emissive ceiling quad (2 triangle lights, L = 15), a matte sphere, a mirror sphere and a plastic
    sphere.  Box spans [-1,1]^3, camera on -y looking at +y, z up."""
    b = SceneBuilder(be)
    white, red, green = (0.73, 0.73, 0.73), (0.65, 0.05, 0.05), (0.12, 0.45, 0.15)
    b.material("matte", Kd=white)
    _quad(b, (-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1))      # floor (normal +z)
    _quad(b, (-1, -1, 1), (-1, 1, 1), (1, 1, 1), (1, -1, 1))          # ceiling
    _quad(b, (-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1))          # back wall
    b.material("matte", Kd=red)
    _quad(b, (-1, -1, -1), (-1, 1, -1), (-1, 1, 1), (-1, -1, 1))      # left
    b.material("matte", Kd=green)
    _quad(b, (1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1))          # right
    b.attribute_begin()
    b.material("matte", Kd=(0.0, 0.0, 0.0))
    b.area_light_source("diffuse", L=(15.0, 15.0, 15.0))
    _quad(b, (-0.25, -0.25, 0.995), (-0.25, 0.25, 0.995), (0.25, 0.25, 0.995), (0.25, -0.25, 0.995))   # clockwise from above: faces -z
    b.attribute_end()
    b.attribute_begin()
    b.material("matte", Kd=(0.6, 0.6, 0.3))
    b.translate((-0.45, 0.3, -0.65))
    b.shape("sphere", radius=0.35)
    b.attribute_end()
    b.attribute_begin()
    b.material("mirror", Kr=(0.9, 0.9, 0.9))
    b.translate((0.45, 0.1, -0.6))
    b.shape("sphere", radius=0.4)
    b.attribute_end()
    if with_plastic:
        b.attribute_begin()
        b.material("plastic", Kd=(0.2, 0.3, 0.6), Ks=(0.3, 0.3, 0.3), roughness=0.1)
        b.translate((0.0, -0.45, -0.8))
        b.shape("sphere", radius=0.2)
        b.attribute_end()
    cam = PerspectiveCamera.look_at(be, (0, -3.4, 0), (0, 0, 0), (0, 0, 1), (res, res), fov=40.0)
    return b, cam, (res, res)


# ------------------------------------------------------------------ procedural lat-long environment map (configs 3-5)
def sky_envmap(n=512):
    """Closed-form n x n (power of two) RGB lat-long map: vertical sky gradient + one bright lobe. texels[t][s][c]."""
    t = (np.arange(n, dtype=np.float64) + 0.5) / n
    s = (np.arange(n, dtype=np.float64) + 0.5) / n
    theta = (t * np.pi)[:, None]
    phi = (s * 2.0 * np.pi)[None, :]
    up = np.cos(theta)                                  # +1 at zenith (theta = 0)
    horizon = np.exp(-(up * 3.0) ** 2)
    base = np.stack([0.25 + 0.35 * horizon + 0.0 * phi, 0.35 + 0.35 * horizon + 0.0 * phi, 0.55 + 0.30 * (up > 0) * up + 0.0 * phi], axis=-1)
    ground = (up < 0)[..., None] * np.ones_like(phi)[..., None]
    base = base * (1.0 - 0.75 * ground)
    sun_dir = np.array([0.45, -0.35, 0.82]); sun_dir /= np.linalg.norm(sun_dir)
    d = np.stack([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta) + 0.0 * phi], axis=-1)
    cosang = (d * sun_dir).sum(-1)
    lobe = 40.0 * np.exp((cosang - 1.0) * 180.0)
    tex = base + lobe[..., None] * np.array([1.0, 0.93, 0.8])
    return np.ascontiguousarray(tex.astype(np.float32))


# ------------------------------------------------------------------ config 3: rounded_cube.ply + env light
def rounded_cube_env(be, res=1024, env_n=512):
    P, N, F = rounded_cube_mesh()
    b = SceneBuilder(be)
    b.light_source("infinite", texels=sky_envmap(env_n))
    b.material("matte", Kd=(0.5, 0.5, 0.5))
    _quad(b, (-60, -60, -9.99), (60, -60, -9.99), (60, 60, -9.99), (-60, 60, -9.99))     # ground
    b.attribute_begin()
    b.material("matte", Kd=(0.7, 0.5, 0.3))
    b.shape("trianglemesh", P=P, N=N, indices=F)
    b.attribute_end()
    cam = PerspectiveCamera.look_at(be, (28, -28, 14), (0, 0, -2), (0, 0, 1), (res, res), fov=40.0)
    return b, cam, (res, res)


# ------------------------------------------------------------------ configs 4/5: baked copies of the mesh (no true instancing in the reference)
def _lcg(seed):
    state = np.uint64(seed)
    while True:
        state = (state * np.uint64(6364136223846793005) + np.uint64(1442695040888963407)) & np.uint64(0xFFFFFFFFFFFFFFFF)
        yield float(int(state >> np.uint64(40))) / float(1 << 24)


def instanced_cubes(be, n_copies=2309, res=(4096, 4096), env_n=1024, seed=7, spacing=26.0, lens_radius=0.0, metal_every=3, textured=False, metal_roughness=0.05):
    """n_copies transformed copies of rounded_cube (4332 triangles each) on a jittered 3-D grid, transforms baked into the
    vertices by TriangleMesh::new (triangle.rs:42-58).  2309 copies = 10,002,588 triangles (config 5)."""
    P, N, F = rounded_cube_mesh()
    b = SceneBuilder(be)
    b.light_source("infinite", texels=sky_envmap(env_n))
    side = int(np.ceil(n_copies ** (1.0 / 3.0)))
    rnd = _lcg(seed)
    mats = [("matte", dict(Kd=(0.55, 0.55, 0.55))), ("matte", dict(Kd=(0.7, 0.35, 0.25))),
            ("metal", dict(eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14), roughness=metal_roughness))]
    if textured:           # checkerboard albedo on the first matte (per-triangle default uvs: the mesh has none)
        b.texture("chk", "spectrum", "checkerboard", uscale=4.0, vscale=4.0, tex1=(0.7, 0.7, 0.7), tex2=(0.25, 0.3, 0.45))
        mats[0] = ("matte", dict(Kd="chk"))
    with np.errstate(over="ignore"):
        for c in range(n_copies):
            ix, iy, iz = c % side, (c // side) % side, c // (side * side)
            jitter = [(next(rnd) - 0.5) * 0.35 * spacing for _ in range(3)]
            pos = ((ix - 0.5 * (side - 1)) * spacing + jitter[0], (iy - 0.5 * (side - 1)) * spacing + jitter[1], (iz - 0.5 * (side - 1)) * spacing + jitter[2])
            axis = (next(rnd) - 0.5, next(rnd) - 0.5, next(rnd) - 0.5 + 1e-3)
            ang = 360.0 * next(rnd)
            sc = 0.6 + 0.7 * next(rnd)
            b.attribute_begin()
            name, kw = mats[2] if (metal_every and c % metal_every == 0) else mats[c % 2]
            b.material(name, **kw)
            b.translate(pos)
            b.rotate(ang, axis)
            b.scale(sc, sc, sc)
            b.shape("trianglemesh", P=P, N=N, indices=F)
            b.attribute_end()
    ext = 0.5 * side * spacing
    eye = (1.9 * ext, -2.3 * ext, 1.4 * ext)
    dist = float(np.linalg.norm(np.array(eye)))
    cam = PerspectiveCamera.look_at(be, eye, (0, 0, 0), (0, 0, 1), res, fov=38.0, lens_radius=lens_radius, focal_dist=dist)
    return b, cam, res


def render(be, builder, cam, res, integrator, sampler, backend_kwargs=None, scene=None, tiles=None, crop=(0.0, 0.0, 1.0, 1.0)):
    """Convenience: create_scene + Film + SamplerIntegrator.render_parallel -> (rgb, pixels, stats, scene)."""
    scene = scene or builder.create_scene()
    film = Film(be, res, crop)
    si = SamplerIntegrator(cam, integrator)
    stats = si.render_parallel(scene, film, sampler, tiles=tiles, **(backend_kwargs or {}))
    rgb, _ = film.into_spectrum_buffer()
    return rgb, film.pixels, stats, scene
