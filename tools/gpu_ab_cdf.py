import os, sys
sys.path.insert(0, "/root/repo")
from fountain_amd import *
from fountain_amd import scenes, _abi as A
gpu = default_backend()
b, cam, r = scenes.instanced_cubes(gpu, n_copies=2309, res=(4096, 4096))
os.environ["FTN_NO_COARSE_CDF"] = "1"; sc0 = b.create_scene()
del os.environ["FTN_NO_COARSE_CDF"]; sc1 = b.create_scene()
si = SamplerIntegrator(cam, PathIntegrator(5, 1.0))
film = Film(gpu, r)
for rep in range(4):
    for name, sc in (("plain", sc0), ("coarse", sc1)):
        st = si.render_parallel(sc, film, RandomSampler(4096, 0, indexed=True, first_sample=8 * rep, sample_count=8), pipeline=A.FTN_PIPELINE_WAVEFRONT)
        print(rep, name, "%.2f ms" % st["kernel_ms"], flush=True)
