"""Debug helper: the configurations of test_four_box_kernels_render_like_the_two_record_kernels one by one, flushing before each render."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fountain_amd import Film, PathIntegrator, RandomSampler, SamplerIntegrator, default_backend, scenes, _abi as A
gpu = default_backend()
b, cam, res = scenes.instanced_cubes(gpu, n_copies=27, res=(160, 160), env_n=32)
sc = b.create_scene()
si = SamplerIntegrator(cam, PathIntegrator.new(5, 1.0))
ref = None
for env in [dict(a.split("=") for a in s.split()) for s in sys.argv[1:]] or [{}]:
    for k in list(os.environ):
        if k.startswith("FTN_"):
            del os.environ[k]
    os.environ.update(env)
    for count in (0, 2):
        print("config", env, "count", count, flush=True)
        f = Film(gpu, res)
        st = si.render_parallel(sc, f, RandomSampler(2, 0, indexed=True), pipeline=A.FTN_PIPELINE_WAVEFRONT, count_traffic=count)
        if ref is None:
            ref = f.pixels.copy()
        print("   ok: rays %d+%d, quad records %d, same film: %s" % (st["rays_closest"], st["rays_any"], st["quad_records"], np.array_equal(ref.view(np.uint32), f.pixels.view(np.uint32))), flush=True)
