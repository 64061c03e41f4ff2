"""The wavefront's path state (44 GB by default for 134 M paths) must shrink gracefully when the GPU has less free memory:
occupy most of the HBM with a torch tensor, render 8 spp, compare with the same render done with plenty of room."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fountain_amd import *
from fountain_amd import scenes, _abi as A
gpu = default_backend()
b, cam, r = scenes.instanced_cubes(gpu, n_copies=64, res=(2048, 2048), env_n=64)
si = SamplerIntegrator(cam, PathIntegrator(5, 1.0))
def render():
    sc = b.create_scene()                      # fresh scene handle: fresh work buffers
    f = Film(gpu, r)
    st = si.render_parallel(sc, f, RandomSampler(64, 0, indexed=True, first_sample=0, sample_count=32), pipeline=A.FTN_PIPELINE_WAVEFRONT)
    return f.pixels.copy(), st
ref, st0 = render()                            # 2048^2 x 32 spp = 134 M paths in one wavefront
free, total = torch.cuda.mem_get_info()
hog = torch.empty(int(free - (30 << 30)), dtype=torch.uint8, device="cuda")     # leave ~30 GB: two halvings needed (44 -> 22 -> 11 GB ... )
print("free before %.1f GB, after the hog %.1f GB" % (free / 2**30, torch.cuda.mem_get_info()[0] / 2**30), flush=True)
px, st1 = render()
print("kernel ms: roomy %.1f, squeezed %.1f; identical films: %s; rays equal: %s" % (st0["kernel_ms"], st1["kernel_ms"], np.array_equal(ref.view(np.uint32), px.view(np.uint32)), st0["rays_closest"] == st1["rays_closest"]))
