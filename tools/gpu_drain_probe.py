"""Drain of the any-hit launches (temporary instrumentation build: FTN_WF_DEBUG prints first-dry / last-dry / end per launch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["FTN_WF_DEBUG"] = "1"
os.environ["FTN_WF_OVERLAP"] = "0"
from fountain_amd import *
from fountain_amd import scenes, _abi as A
gpu = default_backend()
b, cam, r = scenes.instanced_cubes(gpu, n_copies=2309, res=(4096, 4096)); sc = b.create_scene()
si = SamplerIntegrator(cam, PathIntegrator(5, 1.0))
for steal in ("0", "1"):
    os.environ["FTN_ANY2_STEAL"] = steal
    print("steal", steal, file=sys.stderr)
    for rep in range(2):
        st = si.render_parallel(sc, Film(gpu, r), RandomSampler(4096, 0, indexed=True, first_sample=rep, sample_count=1), pipeline=A.FTN_PIPELINE_WAVEFRONT)
