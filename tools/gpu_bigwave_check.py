"""One wavefront of 268 M paths (16 spp of the 4096^2 film) against the same call cut into passes of 64 Mi paths: identical film bits
and identical counters are expected (same accumulators, same per-pixel sample order)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fountain_amd import *
from fountain_amd import scenes, _abi as A
gpu = default_backend()
b, cam, r = scenes.instanced_cubes(gpu, n_copies=2309, res=(4096, 4096)); sc = b.create_scene()
si = SamplerIntegrator(cam, PathIntegrator(5, 1.0))
out = {}
for m in ("64", "256"):
    os.environ["FTN_WF_PATHS_M"] = m
    f = Film(gpu, r)
    st = si.render_parallel(sc, f, RandomSampler(4096, 0, indexed=True, first_sample=16, sample_count=16), pipeline=A.FTN_PIPELINE_WAVEFRONT)
    out[m] = (f.pixels.copy(), st)
    print(m, "Mi paths per pass: %.1f ms, rays %d + %d, spills %d" % (st["kernel_ms"], st["rays_closest"], st["rays_any"], st["spill_samples"]), flush=True)
a, b2 = out["64"], out["256"]
diff = (a[0].view(np.uint32) != b2[0].view(np.uint32)).any(axis=-1)
print("pixels that differ: %d (spill samples %d); counters equal: %s" % (int(diff.sum()), a[1]["spill_samples"], all(a[1][k] == b2[1][k] for k in ("rays_closest", "rays_any", "camera_samples", "spill_samples"))))
