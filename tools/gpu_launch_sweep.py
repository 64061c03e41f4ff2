"""Per-launch times of the traversal kernels under knob settings (FTN_WF_DEBUG=1 makes the library print every span): is the camera
rays' launch -- coherent, a third of the closest-hit time -- best served by the same control parameters as the bounce launches?
  python tools/gpu_launch_sweep.py 'FTN_T4_BURST=1' 'FTN_T4_BURST=4' ...      ('' = defaults)"""
import os, re, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from fountain_amd import Film, PathIntegrator, RandomSampler, SamplerIntegrator, default_backend, scenes, _abi as A  # noqa: E402

sets = sys.argv[1:] or [""]
gpu = default_backend()
b, cam, r = scenes.instanced_cubes(gpu, n_copies=2309, res=(4096, 4096), env_n=1024)
sc = b.create_scene()
si = SamplerIntegrator(cam, PathIntegrator(5, 1.0))
film = Film(gpu, r)
dev = torch.zeros((film.height, film.width, 4), dtype=torch.float32, device="cuda:0")
stream = torch.cuda.current_stream()


SPP = int(os.environ.get("SWEEP_SPP", "16"))


def step(i):
    smp = RandomSampler(4096, 0, indexed=True, first_sample=(i * SPP) % 4096, sample_count=SPP)
    return si.render_device(sc, film, smp, dev.data_ptr(), stream.cuda_stream, pipeline=A.FTN_PIPELINE_WAVEFRONT)


base_env = dict(os.environ)
step(0)
for s in sets:
    for k in list(os.environ):
        if k.startswith("FTN_") and k not in base_env: del os.environ[k]
    for kv in s.split():
        k, v = kv.split("="); os.environ[k] = v
    step(1)
    os.environ["FTN_WF_DEBUG"] = "1"
    sys.stderr.flush()
    with tempfile.TemporaryFile(mode="w+") as tf:
        old = os.dup(2); os.dup2(tf.fileno(), 2)
        try:
            st = step(2); torch.cuda.synchronize()
        finally:
            os.dup2(old, 2); os.close(old)
        tf.seek(0); log = tf.read()
    del os.environ["FTN_WF_DEBUG"]
    cl = [float(x) for x in re.findall(r"closest-hit trace: ([0-9.]+) ms", log)]
    an = [float(x) for x in re.findall(r"any-hit trace: ([0-9.]+) ms", log)]
    print("%-40s closest %s = %.1f | any-hit %s = %.1f" % (s or "(defaults)", " ".join("%.1f" % x for x in cl), sum(cl), " ".join("%.1f" % x for x in an), sum(an)), flush=True)
