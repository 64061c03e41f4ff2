"""Runs a few wavefront steps on an instanced scene (for rocprofv3 --pmc passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fountain_amd import *
from fountain_amd import scenes, _abi as A
gpu = default_backend()
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 343
res = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
b, cam, r = scenes.instanced_cubes(gpu, n_copies=copies, res=(res, res)); sc = b.create_scene()
si = SamplerIntegrator(cam, PathIntegrator(5, 1.0))
for i in range(steps):
    st = si.render_parallel(sc, Film(gpu, r), RandomSampler(4096, 0, indexed=True, first_sample=i, sample_count=1), pipeline=A.FTN_PIPELINE_WAVEFRONT)
print(st)
