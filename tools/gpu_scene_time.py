import os, sys, time
sys.path.insert(0, os.getcwd())
from fountain_amd import default_backend, scenes
gpu = default_backend()
t0 = time.time()
b, cam, r = scenes.instanced_cubes(gpu, n_copies=2309, res=(4096, 4096), env_n=1024)
t1 = time.time()
sc = b.create_scene()
t2 = time.time()
print("python scene description %.2f s, create_scene %.2f s" % (t1 - t0, t2 - t1), flush=True)
