"""Drain of the any-hit launches.  Needs a library built with -DFTN_DRAIN_PROBE (ftn_wavefront.hip): each launch then reports on stderr
when its first / last wave found the queue dry and when it ended.  FTN_LIB selects the library, e.g.
    hipcc ... -DFTN_DRAIN_PROBE -c ftn_wavefront.hip -o build/probe.o && hipcc -shared ... -o ../libfountain_hip_probe.so
    FTN_LIB=libfountain_hip_probe.so python tools/gpu_drain_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["FTN_WF_OVERLAP"] = "0"
from fountain_amd import *
from fountain_amd import scenes, _abi as A
gpu = default_backend()
b, cam, r = scenes.instanced_cubes(gpu, n_copies=2309, res=(4096, 4096)); sc = b.create_scene()
si = SamplerIntegrator(cam, PathIntegrator(5, 1.0))
for rep in range(2):
    if True:
        st = si.render_parallel(sc, Film(gpu, r), RandomSampler(4096, 0, indexed=True, first_sample=rep, sample_count=1), pipeline=A.FTN_PIPELINE_WAVEFRONT)
