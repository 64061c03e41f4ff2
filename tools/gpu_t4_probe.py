"""A/B probe of the traversal kernels on the bench scene (not a test): one scene build, then bench-sized steps under different knob
settings (the library re-reads its FTN_* knobs at every render call).  Prints per-group device times (HIP events inside the library).
  python tools/gpu_t4_probe.py [copies res spp] -- SET 'FTN_TRACE4=0' 'FTN_T4_WG=4 FTN_T4_BURST=4' ...
Every SET is a space-separated list of NAME=VALUE; the empty string '' is the default configuration."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from fountain_amd import Film, PathIntegrator, RandomSampler, SamplerIntegrator, default_backend, scenes, _abi as A  # noqa: E402

argv = sys.argv[1:]
sets = [""]
if "--" in argv:
    k = argv.index("--")
    argv, sets = argv[:k], argv[k + 1:]
copies = int(argv[0]) if len(argv) > 0 else 2309
res = int(argv[1]) if len(argv) > 1 else 4096
spp = int(argv[2]) if len(argv) > 2 else 16
reps = int(os.environ.get("PROBE_REPS", "2"))

gpu = default_backend()
t0 = time.time()
b, cam, r = scenes.instanced_cubes(gpu, n_copies=copies, res=(res, res), env_n=int(os.environ.get("PROBE_ENV_N", "1024")))
sc = b.create_scene()
info = sc.info()
print("scene: %d prims, %d nodes, depth %d, built in %.1f s" % (info["n_prims"], info["n_nodes"], info["max_depth"], time.time() - t0), flush=True)
si = SamplerIntegrator(cam, PathIntegrator(5, 1.0))
film = Film(gpu, r)
dev = torch.zeros((film.height, film.width, 4), dtype=torch.float32, device="cuda:0")
stream = torch.cuda.current_stream()


def step(i, count=0):
    smp = RandomSampler(4096, 0, indexed=True, first_sample=(i * spp) % 4096, sample_count=spp)
    return si.render_device(sc, film, smp, dev.data_ptr(), stream.cuda_stream, pipeline=A.FTN_PIPELINE_WAVEFRONT, count_traffic=count)


base_env = dict(os.environ)
ref = None
for s in sets:
    for k in list(os.environ):
        if k.startswith("FTN_") and k not in base_env:
            del os.environ[k]
    for kv in s.split():
        k, v = kv.split("=")
        os.environ[k] = v
    dev.zero_()
    step(0)                                   # warm-up (allocations, first use of a kernel variant)
    best = None
    for rep in range(reps):
        st = step(1 + rep)
        if best is None or st["kernel_ms"] < best["kernel_ms"]:
            best = st
    torch.cuda.synchronize()
    rays = best["rays_closest"] + best["rays_any"]
    out = {"set": s, "ms": round(best["kernel_ms"], 2), "mrays": round(rays / best["kernel_ms"] / 1e3, 1), "closest_ms": round(best["trace_ms"], 2),
           "closest_launches": best["trace_launches"], "any_ms": round(best["any_ms"], 2), "shade_ms": round(best["shade_ms"], 2), "sort_ms": round(best["sort_ms"], 2)}
    if os.environ.get("PROBE_COUNT"):
        c = step(1, count=2)
        rc = c["rays_closest"] - c["mis_rays_any_hit"]
        ra = c["rays_any"] + c["mis_rays_any_hit"]
        out.update({"rec_per_closest_ray": round((c["quad_records"] - c["quad_records_any"]) / max(rc, 1), 2), "rec_per_any_ray": round(c["quad_records_any"] / max(ra, 1), 2),
                    "prims_per_closest_ray": round((c["prims_tested"] - c["prims_tested_any"]) / max(rc, 1), 3), "prims_per_any_ray": round(c["prims_tested_any"] / max(ra, 1), 3),
                    "nodes_per_ray_twobox": round(c["nodes_visited"] / max(rc + ra, 1), 2)})
    h = dev.sum().item()
    out["film_sum"] = h
    print(json.dumps(out), flush=True)
