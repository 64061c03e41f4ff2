"""Counting run of the wavefront pipeline on the config-5 scene: lane utilisation of the traversal kernel's node / leaf steps (FTN_WF_DEBUG=1).
argv: copies res max_depth"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["FTN_WF_DEBUG"] = "1"
from fountain_amd import *
from fountain_amd import scenes, _abi as A
gpu = default_backend()
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 2309
res = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 5
b, cam, r = scenes.instanced_cubes(gpu, n_copies=copies, res=(res, res)); sc = b.create_scene()
si = SamplerIntegrator(cam, PathIntegrator(depth, 1.0))
for count in (True, False):
    st = si.render_parallel(sc, Film(gpu, r), RandomSampler(4096, 0, indexed=True, first_sample=0, sample_count=1), pipeline=A.FTN_PIPELINE_WAVEFRONT, count_traffic=count)
    print("count" if count else "timed", st)
