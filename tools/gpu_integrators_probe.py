"""Measurement for DESIGN.md section 8: the three integrators on the wavefront pipeline and in the per-pixel megakernel.
Prints Mrays/s (kernel time) per integrator and pipeline."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fountain_amd import DirectLightingIntegrator, Film, PathIntegrator, RandomSampler, SamplerIntegrator, WhittedIntegrator, default_backend, scenes, _abi as A

gpu = default_backend()
MEGA, WAVE = A.FTN_PIPELINE_MEGAKERNEL, A.FTN_PIPELINE_WAVEFRONT


def run(name, make, spp):
    b, cam, res = make(gpu)
    sc = b.create_scene()
    for label, integ, pl in (("path, wavefront", PathIntegrator(5, 1.0), WAVE), ("path, megakernel", PathIntegrator(5, 1.0), MEGA),
                             ("direct lighting (depth 5), wavefront", DirectLightingIntegrator(5), WAVE), ("direct lighting (depth 5), megakernel", DirectLightingIntegrator(5), MEGA),
                             ("whitted (depth 5), wavefront", WhittedIntegrator(5), WAVE), ("whitted (depth 5), megakernel", WhittedIntegrator(5), MEGA)):
        si = SamplerIntegrator(cam, integ)
        best = None
        for rep in range(3):
            st = si.render_parallel(sc, Film(gpu, res), RandomSampler(spp, 0, indexed=True), pipeline=pl)
            if best is None or st["kernel_ms"] < best["kernel_ms"]:
                best = st
        rays = best["rays_closest"] + best["rays_any"]
        print("%-28s %-40s %8.2f ms  %10d rays  %8.1f Mrays/s  (%.2f rays per camera sample)" % (name, label, best["kernel_ms"], rays, rays / best["kernel_ms"] / 1e3, rays / best["camera_samples"]), flush=True)


run("cornell 512^2 x 16", lambda be: scenes.cornell(be, res=512), 16)
run("rounded cube + env 1024^2 x 4", lambda be: scenes.rounded_cube_env(be, res=1024), 4)
run("343 cubes 1024^2 x 4", lambda be: scenes.instanced_cubes(be, n_copies=343, res=(1024, 1024), env_n=256), 4)
