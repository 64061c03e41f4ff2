"""ftn_intersect / ftn_intersect_test on a large batch of incoherent rays (config-5 scene): the wavefront traversal kernel vs the plain
one-lane-per-ray loop (FTN_BATCH_SIMPLE=1); both must return the same bits."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fountain_amd import *
from fountain_amd import scenes
gpu = default_backend()
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 2309
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000000
b, cam, r = scenes.instanced_cubes(gpu, n_copies=copies, res=(64, 64)); sc = b.create_scene()
info = sc.info(); lo, hi = info["world_bound"][:3], info["world_bound"][3:]
rng = np.random.default_rng(1)
o = (lo + rng.random((n, 3)) * (hi - lo)).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32)
rays = make_rays(o, d)
res = {}
for mode in ("wavefront", "simple"):
    if mode == "simple": os.environ["FTN_BATCH_SIMPLE"] = "1"
    else: os.environ.pop("FTN_BATCH_SIMPLE", None)
    t, prim, bary, st = sc.intersect(rays)
    occ, st2 = sc.intersect_test(rays)
    res[mode] = (t, prim, bary, occ)
    print("%-10s closest %.2f ms (%.0f Mrays/s)  any %.2f ms (%.0f Mrays/s)  nodes/ray %.1f" % (mode, st["kernel_ms"], n / st["kernel_ms"] / 1e3, st2["kernel_ms"], n / st2["kernel_ms"] / 1e3, st["nodes_visited"] / n), flush=True)
a, bb = res["wavefront"], res["simple"]
print("identical:", all(np.array_equal(x.view(np.uint8), y.view(np.uint8)) for x, y in zip(a, bb)), " hits:", int((a[1] >= 0).sum()))
