"""Textured variant of the instanced-cubes scene: specialised shading kernels (one per material type, environment-lit variant) vs the
generic textured kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fountain_amd import *
from fountain_amd import scenes, _abi as A
gpu = default_backend()
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 343
res = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
b, cam, r = scenes.instanced_cubes(gpu, n_copies=copies, res=(res, res), textured=True); sc = b.create_scene()
si = SamplerIntegrator(cam, PathIntegrator(5, 1.0))
for spec in ("0", "1", "0", "1"):
    os.environ["FTN_SHADE_SPECIALISE"] = spec
    best = None
    for rep in range(2):
        st = si.render_parallel(sc, Film(gpu, r), RandomSampler(64, 0, indexed=True, first_sample=8 * rep, sample_count=8), pipeline=A.FTN_PIPELINE_WAVEFRONT)
        best = st if best is None or st["kernel_ms"] < best["kernel_ms"] else best
    print("specialised shading %s: %.2f ms, %.0f Mrays/s" % (spec, best["kernel_ms"], (best["rays_closest"] + best["rays_any"]) / best["kernel_ms"] / 1e3), flush=True)
