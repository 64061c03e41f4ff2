"""Golden films of the hot path: a few small renders by the CPU oracle (deterministic-math build), committed as
tests/golden/render_fixtures.npz.  They pin the oracle against drift (tests/test_golden_films.py, CPU) and are what the HIP path
has to reproduce bit for bit (same test, -m gpu).  Inputs are the synthetic scenes of fountain_amd/scenes.py; nothing of the
reference is needed to run this.

    python tests/golden/make_render_fixtures.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from fountain_amd import DirectLightingIntegrator, Film, PathIntegrator, RandomSampler, SamplerIntegrator, scenes  # noqa: E402

CASES = {
    # name: (scene factory, integrator, sampler)
    "furnace_tile_serial": (lambda be: scenes.furnace(be, res=16), lambda: PathIntegrator.new(10, 0.0), lambda: RandomSampler(8, 0)),
    "cornell_indexed": (lambda be: scenes.cornell(be, res=32), lambda: PathIntegrator.new(5, 1.0), lambda: RandomSampler(4, 0, indexed=True)),
    "cube_env_indexed": (lambda be: scenes.rounded_cube_env(be, res=24, env_n=32), lambda: PathIntegrator.new(5, 1.0), lambda: RandomSampler(4, 2, indexed=True)),
    "cubes_metal_dof": (lambda be: scenes.instanced_cubes(be, n_copies=8, res=(32, 24), env_n=16, lens_radius=0.05), lambda: PathIntegrator.new(5, 1.0),
                        lambda: RandomSampler(3, 1, indexed=True)),
    "cornell_direct": (lambda be: scenes.cornell(be, res=24), lambda: DirectLightingIntegrator(3), lambda: RandomSampler(2, 0, indexed=True)),
}


def render_case(be, name, **kw):
    make, integ, smp = CASES[name]
    b, cam, res = make(be)
    film = Film(be, res)
    st = SamplerIntegrator(cam, integ()).render_parallel(b.create_scene(), film, smp(), **kw)
    return film.pixels, st


if __name__ == "__main__":
    from oracle_loader import oracle_backend
    be = oracle_backend(det=True)
    out = {}
    for name in CASES:
        px, st = render_case(be, name)
        out[name] = px
        out[name + "__rays"] = np.array([st["rays_closest"], st["rays_any"], st["camera_samples"]], np.int64)
        print(name, px.shape, "mean", float(px[..., :3].mean()), "rays", out[name + "__rays"])
    path = os.path.join(HERE, "render_fixtures.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
