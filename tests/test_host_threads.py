"""ftn_scene_create's host passes run on several threads for large scenes (parallel_for / ParallelBvh, ftn_host.cpp).  Every pass
writes an element from that element's inputs alone, so the arrays must not depend on the thread count: the BVH (nodes, primitive
order), the four-box records and the eight-box records of a 300 k-triangle scene with FTN_BVH_THREADS=1 and =7, byte for byte.  CPU only
(the host-only entry points of the C ABI)."""
import ctypes as C
import os

import numpy as np

from fountain_amd import _abi as A, scenes


def _arrays(ftn, desc):
    fn = ftn.lib.ftn_bvh_build
    fn.argtypes = [C.c_void_p] * 5
    nodes = (A.ftn_bvh_node * max(1, 2 * desc.n_prims))()
    order = (C.c_uint32 * max(1, desc.n_prims))()
    n, depth = C.c_uint32(), C.c_uint32()
    assert fn(C.byref(desc), nodes, order, C.byref(n), C.byref(depth)) == 0
    nb = np.frombuffer(nodes, dtype=np.uint8)[: n.value * C.sizeof(A.ftn_bvh_node)].copy()
    q = ftn.lib.ftn_bvh_quads
    q.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 3
    nrec, bound = C.c_uint32(), C.c_uint32()
    assert q(nodes, n.value, None, C.byref(nrec), C.byref(bound)) == 0
    quad = np.zeros((nrec.value, 32), np.float32)
    assert q(nodes, n.value, quad.ctypes.data_as(C.c_void_p), C.byref(nrec), C.byref(bound)) == 0
    o = ftn.lib.ftn_bvh_octs
    o.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 5
    norec, obound, nx = C.c_uint32(), C.c_uint32(), C.c_uint32()
    assert o(nodes, n.value, None, C.byref(norec), C.byref(obound), None, C.byref(nx)) == 0
    oct_ = np.zeros((norec.value, 32), np.uint32)
    xbox = np.zeros((max(nx.value, 1), 8), np.float32)
    assert o(nodes, n.value, oct_.ctypes.data_as(C.c_void_p), C.byref(norec), C.byref(obound), xbox.ctypes.data_as(C.c_void_p), C.byref(nx)) == 0
    return dict(nodes=nb, order=np.frombuffer(order, dtype=np.uint32).copy(), depth=depth.value, quad=quad.view(np.uint32), quad_bound=bound.value,
                oct=oct_, oct_bound=obound.value, xbox=xbox[: nx.value].view(np.uint32))


def test_host_arrays_do_not_depend_on_the_thread_count(ftn):
    b, _, _ = scenes.instanced_cubes(ftn, n_copies=70, res=(64, 64))           # 70 x 4332 triangles: above every pass's threshold for threads
    desc, keep = b.build_desc()
    assert desc.n_prims > 4 * 65536
    got = {}
    old = os.environ.get("FTN_BVH_THREADS")
    try:
        for nt in (1, 7):
            os.environ["FTN_BVH_THREADS"] = str(nt)
            got[nt] = _arrays(ftn, desc)
    finally:
        if old is None: os.environ.pop("FTN_BVH_THREADS", None)
        else: os.environ["FTN_BVH_THREADS"] = old
    for k, v in got[1].items():
        w = got[7][k]
        assert np.array_equal(v, w) if isinstance(v, np.ndarray) else v == w, k
