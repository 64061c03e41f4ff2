"""Loads the CPU oracle (oracle/liboracle*.so) for tests.  The product package never does this."""
import os
import subprocess

from fountain_amd.api import Backend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_cache = {}


def oracle_backend(det=False):
    name = "liboracle_det.so" if det else "liboracle.so"
    if name not in _cache:
        path = os.path.join(ROOT, "oracle", name)
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), name])
        _cache[name] = Backend(path, "orc_", True)
    return _cache[name]
