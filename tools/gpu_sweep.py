"""Knob sweep for k_wf_trace on the mid-size instanced scene (experiments only)."""
import os, sys, time, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fountain_amd import *
from fountain_amd import scenes, _abi as A
gpu = default_backend()
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 343
res = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
b, cam, r = scenes.instanced_cubes(gpu, n_copies=copies, res=(res, res)); sc = b.create_scene()
si = SamplerIntegrator(cam, PathIntegrator(5, 1.0))
def run(env):
    for k, v in env.items(): os.environ[k] = str(v)
    best = None
    for rep in range(3):
        st = si.render_parallel(sc, Film(gpu, r), RandomSampler(4096, 0, indexed=True, first_sample=rep, sample_count=1), pipeline=A.FTN_PIPELINE_WAVEFRONT)
        if best is None or st["kernel_ms"] < best["kernel_ms"]: best = st
    rays = best["rays_closest"] + best["rays_any"]
    print("%-60s total %.2f ms  trace(closest) %.2f ms  -> %.0f Mrays/s" % (env, best["kernel_ms"], best["trace_ms"], rays / best["kernel_ms"] / 1e3), flush=True)
run({"FTN_TRACE_FAT": 0})
run({"FTN_TRACE_FAT": 1})
for burst in (2, 4, 8):
    for lb in (2, 4, 8):
        run({"FTN_TRACE_FAT": 1, "FTN_TRACE_BURST": burst, "FTN_TRACE_LEAF_BATCH": lb, "FTN_TRACE_REFILL": 16})
run({"FTN_TRACE_FAT": 1, "FTN_TRACE_BURST": 4, "FTN_TRACE_LEAF_BATCH": 2, "FTN_TRACE_REFILL": 8})
run({"FTN_TRACE_FAT": 1, "FTN_TRACE_BURST": 4, "FTN_TRACE_LEAF_BATCH": 2, "FTN_TRACE_REFILL": 32})
