"""First GPU shake-out: furnace, traversal parity, small render parity against the CPU oracle (det build)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle_loader import oracle_backend
from fountain_amd import *
from fountain_amd import scenes, _abi as A

gpu = default_backend()
orc = oracle_backend(det=True)
print("devices:", gpu.fn("device_count")())

def cmp(name, a, b):
    a = np.asarray(a); b = np.asarray(b)
    same = np.array_equal(a.view(np.uint32), b.view(np.uint32)) if a.dtype == np.float32 else np.array_equal(a, b)
    nd = int((a != b).sum())
    print("  %-28s %s  (differing elements: %d / %d, max abs diff %.3g)" % (name, "BIT-EXACT" if same else "DIFFERS", nd, a.size, float(np.nanmax(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0))
    return same

# ---- furnace
for name, integ, exp, eps in (("path rr", PathIntegrator(10, 1.0), 2.0, 0.1), ("path no rr", PathIntegrator(10, 0.0), 2.0, 1e-3), ("direct", DirectLightingIntegrator(3), 1.5, 1e-5)):
    for indexed in (False, True):
        out = {}
        for be, tag in ((gpu, "gpu"), (orc, "orc")):
            b, cam, res = scenes.furnace(be)
            t = time.time()
            rgb, px, st, _ = scenes.render(be, b, cam, res, integ, RandomSampler(128, 0, indexed=indexed))
            out[tag] = (rgb, px, st, time.time() - t)
        rgb = out["gpu"][0]
        ok = np.abs(rgb - exp).max() <= eps
        print("furnace %-10s indexed=%d: gpu min %.6f max %.6f %s  rays gpu %d/%d orc %d/%d  (%.2fs / %.2fs)" % (name, indexed, rgb.min(), rgb.max(), "ok" if ok else "FAIL",
              out["gpu"][2]["rays_closest"], out["gpu"][2]["rays_any"], out["orc"][2]["rays_closest"], out["orc"][2]["rays_any"], out["gpu"][3], out["orc"][3]))
        cmp("pixels vs oracle", out["gpu"][1], out["orc"][1])

# ---- traversal parity on the rounded cube
rng = np.random.default_rng(5)
for be in (gpu, orc):
    b = SceneBuilder(be); P, N, F = scenes.rounded_cube_mesh(); b.material("none"); b.shape("trianglemesh", P=P, N=N, indices=F)
    sc = b.create_scene()
    if be is gpu: sg = sc
    else: so = sc
d = rng.normal(size=(100000, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
rays = make_rays(np.zeros((1, 3), np.float32), d)
tg, pg, bg, stg = sg.intersect(rays); to, po, bo, sto = so.intersect(rays)
print("rounded cube: hits gpu %d orc %d of %d" % ((pg >= 0).sum(), (po >= 0).sum(), len(pg)))
cmp("closest t", tg, to); cmp("closest prim", pg, po)
og, _ = sg.intersect_test(rays); oo, _ = so.intersect_test(rays)
cmp("any-hit", og, oo)
print("  counters gpu nodes %d prims %d | orc nodes %d prims %d" % (stg["nodes_visited"], stg["prims_tested"], sto["nodes_visited"], sto["prims_tested"]))
fg = sg.intersect_full(rays[:20000]); fo = so.intersect_full(rays[:20000])
for nm, sl in (("p", slice(0, 3)), ("p_err", slice(3, 6)), ("n", slice(6, 9)), ("wo", slice(11, 14)), ("shading dpdu", slice(14, 17)), ("shading_n", slice(20, 23)), ("t", slice(23, 24))):
    a = fg[:, sl]; bb = fo[:, sl] if nm != "shading dpdu" else None
    if bb is not None: cmp("full." + nm, a, bb)
nodes_g, order_g = sg.nodes(); nodes_o, order_o = so.nodes()
cmp("bvh nodes", nodes_g.view(np.uint8), nodes_o.view(np.uint8)); cmp("bvh prim order", order_g, order_o)

# ---- small Cornell render, both pipelines' common path (megakernel), indexed sampler
out = {}
for be, tag in ((gpu, "gpu"), (orc, "orc")):
    b, cam, res = scenes.cornell(be, res=64)
    t = time.time()
    rgb, px, st, _ = scenes.render(be, b, cam, res, PathIntegrator(5, 1.0), RandomSampler(16, 0, indexed=True), backend_kwargs=dict(count_traffic=True))
    out[tag] = (rgb, px, st, time.time() - t)
print("cornell 64x64x16: gpu %.2fs orc %.2fs; rays gpu %d+%d orc %d+%d; nodes %d/%d prims %d/%d spill %d/%d" % (out["gpu"][3], out["orc"][3],
      out["gpu"][2]["rays_closest"], out["gpu"][2]["rays_any"], out["orc"][2]["rays_closest"], out["orc"][2]["rays_any"],
      out["gpu"][2]["nodes_visited"], out["orc"][2]["nodes_visited"], out["gpu"][2]["prims_tested"], out["orc"][2]["prims_tested"], out["gpu"][2]["spill_samples"], out["orc"][2]["spill_samples"]))
cmp("cornell pixels", out["gpu"][1], out["orc"][1])
d = out["gpu"][0].astype(np.float64) - out["orc"][0].astype(np.float64)
print("  RMSE %.3g  mean %.4f" % (np.sqrt((d ** 2).mean()), out["gpu"][0].mean()))
