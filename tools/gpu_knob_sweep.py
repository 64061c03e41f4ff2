"""Knob sweep of k_wf_trace on the config-5 scene (experiment)."""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fountain_amd import *
from fountain_amd import scenes, _abi as A
gpu = default_backend()
b, cam, r = scenes.instanced_cubes(gpu, n_copies=2309, res=(4096, 4096)); sc = b.create_scene()
si = SamplerIntegrator(cam, PathIntegrator(5, 1.0))
def run(env):
    for k, v in env.items(): os.environ[k] = str(v)
    best = None
    for rep in range(2):
        st = si.render_parallel(sc, Film(gpu, r), RandomSampler(4096, 0, indexed=True, first_sample=rep, sample_count=1), pipeline=A.FTN_PIPELINE_WAVEFRONT)
        if best is None or st["kernel_ms"] < best["kernel_ms"]: best = st
    print("%-70s total %.2f ms  trace(closest) %.2f ms" % (env, best["kernel_ms"], best["trace_ms"]), flush=True)
    for k in env: os.environ.pop(k)
run({})
for refill, lb, burst in itertools.product((4, 8, 16, 32), (1, 2, 4, 8), (4, 8, 16)):
    run({"FTN_TRACE_REFILL": refill, "FTN_TRACE_LEAF_BATCH": lb, "FTN_TRACE_BURST": burst})
