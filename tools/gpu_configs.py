"""Mrays/s of the wavefront pipeline on the BASELINE.json configs 2-5 (synthetic scenes of fountain_amd.scenes), a few spp each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fountain_amd import *
from fountain_amd import scenes, _abi as A
gpu = default_backend()
def run(name, make, spp):
    b, cam, res = make(gpu); sc = b.create_scene(); info = sc.info()
    si = SamplerIntegrator(cam, PathIntegrator(5, 1.0))
    best = None
    for rep in range(3):
        st = si.render_parallel(sc, Film(gpu, res), RandomSampler(4096, 0, indexed=True, first_sample=rep * spp, sample_count=spp), pipeline=A.FTN_PIPELINE_WAVEFRONT)
        if best is None or st["kernel_ms"] < best["kernel_ms"]: best = st
    rays = best["rays_closest"] + best["rays_any"]
    print("%-44s %9d tris %4dx%-4d %2d spp: %7.2f ms  %6.0f Mrays/s  (%.2f rays per camera sample)" % (name, info["n_prims"], res[0], res[1], spp, best["kernel_ms"], rays / best["kernel_ms"] / 1e3, rays / best["camera_samples"]), flush=True)
run("config 2: Cornell box", lambda be: scenes.cornell(be, res=512), 16)
run("config 3: rounded cube + env map", lambda be: scenes.rounded_cube_env(be, res=1024, env_n=512), 8)
run("config 4: 46 metal cubes, DoF, 1080p", lambda be: scenes.instanced_cubes(be, n_copies=46, res=(1920, 1080), env_n=1024, lens_radius=0.4), 4)
run("config 5: 2309 cubes (10M triangles)", lambda be: scenes.instanced_cubes(be, n_copies=2309, res=(4096, 4096)), 1)
