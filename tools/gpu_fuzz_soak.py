"""One-off soak: the differential fuzz of tests/test_gpu_fuzz.py over many more seeds (argv: first last [env | tri])."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle_loader import oracle_backend
from fountain_amd import default_backend
import test_gpu_fuzz as F
gpu, orc = default_backend(), oracle_backend(det=True)
first, last = int(sys.argv[1]), int(sys.argv[2])
env_only = len(sys.argv) > 3 and sys.argv[3] == "env"
tri_only = len(sys.argv) > 3 and sys.argv[3] == "tri"
bad = 0
for seed in range(first, last):
    try:
        F.check_recipe(gpu, orc, F.make_recipe(seed, env_only=env_only or (tri_only and seed % 3 == 0), tri_only=tri_only), seed)
    except AssertionError as e:
        bad += 1
        print("seed %d FAILED: %s" % (seed, str(e)[:300]), flush=True)
    if seed % 20 == 0: print("seed", seed, "done", flush=True)
print("seeds %d..%d: %d failures" % (first, last, bad))
