"""Converts the reference's test mesh data/rounded_cube.ply (used by tests/tri_watertight.rs) into a compact
binary fixture.  Run once in the build container (needs /root/reference); the output is committed.

    python tests/golden/make_rounded_cube_fixture.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fountain_amd.api import load_ply_ascii  # noqa: E402

src = "/root/reference/data/rounded_cube.ply"
P, N, F = load_ply_ascii(src)
assert P.shape == (8664, 3) and N.shape == (8664, 3) and F.shape == (4332, 3)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rounded_cube.npz")
np.savez_compressed(out, P=P.astype(np.float32), N=N.astype(np.float32), F=F.astype(np.uint32))
print("wrote", out, os.path.getsize(out), "bytes; bounds", P.min(0), P.max(0))
