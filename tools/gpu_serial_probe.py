"""Throughput of the reference's exact per-tile RandomSampler stream (FTN_SAMPLER_TILE_SERIAL: one lane per tile) on the config-5 scene."""
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
import numpy as np
films = {}
for name, pl in (("queues (one path per tile in flight)", A.FTN_PIPELINE_WAVEFRONT), ("megakernel (one lane per tile)", A.FTN_PIPELINE_MEGAKERNEL)):
    for rep in range(2):
        film = Film(gpu, r)
        st = si.render_parallel(sc, film, RandomSampler(1, 0), pipeline=pl)
        rays = st["rays_closest"] + st["rays_any"]
        print("tile-serial sampler, %s, 1 spp: %.1f ms, %d rays -> %.0f Mrays/s (%d bounce rounds)" % (name, st["kernel_ms"], rays, rays / st["kernel_ms"] / 1e3, st["shade_launches"]), flush=True)
    films[pl] = film.pixels.copy()
a, b2 = films[A.FTN_PIPELINE_WAVEFRONT].view(np.uint32), films[A.FTN_PIPELINE_MEGAKERNEL].view(np.uint32)
print("films of the two pipelines: %d of %d pixels differ (spill pixels may, in the last bit)" % (int((a != b2).any(axis=-1).sum()), a.shape[0] * a.shape[1]), flush=True)
st = si.render_parallel(sc, Film(gpu, r), RandomSampler(4096, 0, indexed=True, sample_count=1), pipeline=A.FTN_PIPELINE_MEGAKERNEL)
rays = st["rays_closest"] + st["rays_any"]
print("indexed sampler, megakernel, 1 spp: %.1f ms -> %.0f Mrays/s" % (st["kernel_ms"], rays / st["kernel_ms"] / 1e3))
