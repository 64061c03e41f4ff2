"""Occupancy-matched comparison of the node-per-visit and fat-node traversal kernels on the config-5 scene (experiment)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fountain_amd import *
from fountain_amd import scenes, _abi as A
gpu = default_backend()
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 2309
res = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
b, cam, r = scenes.instanced_cubes(gpu, n_copies=copies, res=(res, res)); sc = b.create_scene()
si = SamplerIntegrator(cam, PathIntegrator(5, 1.0))
SPP = int(os.environ.get("PROBE_SPP", "1"))
def run(env):
    for k, v in env.items(): os.environ[k] = str(v)
    best = None
    for rep in range(3):
        st = si.render_parallel(sc, Film(gpu, r), RandomSampler(4096, 0, indexed=True, first_sample=rep * SPP, sample_count=SPP), pipeline=A.FTN_PIPELINE_WAVEFRONT)
        if best is None or st["kernel_ms"] < best["kernel_ms"]: best = st
    rays = best["rays_closest"] + best["rays_any"]
    print("%-70s total %.2f ms (%d spp per call)  trace(closest) %.2f ms / %d  -> %.0f Mrays/s" % (env, best["kernel_ms"], SPP, best["trace_ms"], best["trace_launches"], rays / best["kernel_ms"] / 1e3), flush=True)
    for k in env: os.environ.pop(k)
for cfg in [x.split(",") for x in sys.argv[3:]] or [[]]:
    run(dict(kv.split("=") for kv in cfg if kv))
