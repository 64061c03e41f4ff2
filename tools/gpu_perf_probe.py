"""Quick timing probe of both pipelines on the BASELINE scenes (not a test)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fountain_amd import *
from fountain_amd import scenes, _abi as A
gpu = default_backend()
def run(name, make, spp, pipelines=(A.FTN_PIPELINE_MEGAKERNEL, A.FTN_PIPELINE_WAVEFRONT), reps=2):
    t = time.time(); b, cam, res = make(gpu); sc = b.create_scene(); info = sc.info()
    print("%s: scene built in %.1fs: %d prims %d nodes depth %d" % (name, time.time() - t, info["n_prims"], info["n_nodes"], info["max_depth"]), flush=True)
    si = SamplerIntegrator(cam, PathIntegrator(5, 1.0))
    for pl in pipelines:
        for r in range(reps):
            film = Film(gpu, res)
            st = si.render_parallel(sc, film, RandomSampler(spp, 0, indexed=True), pipeline=pl)
        rays = st["rays_closest"] + st["rays_any"]
        print("  pipeline %d: %.1f ms kernel, %d rays -> %.1f Mrays/s (trace %.1f ms over %d launches)" % (pl, st["kernel_ms"], rays, rays / st["kernel_ms"] / 1e3, st["trace_ms"], st["trace_launches"]), flush=True)
    st = si.render_parallel(sc, Film(gpu, res), RandomSampler(min(spp, 2), 0, indexed=True), pipeline=A.FTN_PIPELINE_WAVEFRONT, count_traffic=True)
    rays = st["rays_closest"] + st["rays_any"]
    print("  traffic: %.1f nodes/ray %.2f prims/ray" % (st["nodes_visited"] / rays, st["prims_tested"] / rays), flush=True)
which = sys.argv[1:] or ["cornell", "cube", "cubes"]
if "cornell" in which: run("cornell 512x512x16", lambda be: scenes.cornell(be, res=512), 16)
if "cube" in which: run("rounded cube 1024x1024x4", lambda be: scenes.rounded_cube_env(be, res=1024), 4)
if "cubes" in which: run("125 cubes 1024x1024x4", lambda be: scenes.instanced_cubes(be, n_copies=125, res=(1024, 1024), env_n=256), 4)
if "big" in which: run("2309 cubes (10M tris) 2048x2048x1", lambda be: scenes.instanced_cubes(be, n_copies=2309, res=(2048, 2048)), 1, pipelines=(A.FTN_PIPELINE_WAVEFRONT,))
