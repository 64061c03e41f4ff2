"""Times library variants (FTN_LIB) on the 343-copy scene in separate processes."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
sys.path.insert(0, %r)
from fountain_amd import *
from fountain_amd import scenes, _abi as A
gpu = default_backend()
b, cam, r = scenes.instanced_cubes(gpu, n_copies=343, res=(2048, 2048)); sc = b.create_scene()
si = SamplerIntegrator(cam, PathIntegrator(5, 1.0))
best = None
for rep in range(4):
    st = si.render_parallel(sc, Film(gpu, r), RandomSampler(4096, 0, indexed=True, first_sample=rep, sample_count=1), pipeline=A.FTN_PIPELINE_WAVEFRONT)
    if best is None or st["kernel_ms"] < best["kernel_ms"]: best = st
rays = best["rays_closest"] + best["rays_any"]
print("%%-28s total %%.2f ms trace(closest) %%.2f ms -> %%.0f Mrays/s" %% (os.environ.get("FTN_LIB", "default"), best["kernel_ms"], best["trace_ms"], rays / best["kernel_ms"] / 1e3))
''' % ROOT
for lib in sys.argv[1:] or ["libfountain_hip.so"]:
    env = dict(os.environ, FTN_LIB=lib)
    subprocess.run([sys.executable, "-c", code], env=env)
