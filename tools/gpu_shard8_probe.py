import os, sys
sys.path.insert(0, "/root/repo")
from fountain_amd import *
from fountain_amd import scenes, _abi as A
gpu = default_backend()
b, cam, r = scenes.instanced_cubes(gpu, n_copies=2309, res=(4096, 4096)); sc = b.create_scene()
si = SamplerIntegrator(cam, PathIntegrator(5, 1.0))
for rep in range(3):
    st = si.render_parallel(sc, Film(gpu, r), RandomSampler(4096, 0, indexed=True, first_sample=rep * 128, sample_count=128), tiles=(0, 8, 0), pipeline=A.FTN_PIPELINE_WAVEFRONT)
    rays = st["rays_closest"] + st["rays_any"]
    print("rank-0-of-8 shard, 128 spp: %.1f ms, %.0f Mrays/s, camera samples %d" % (st["kernel_ms"], rays / st["kernel_ms"] / 1e3, st["camera_samples"]), flush=True)
