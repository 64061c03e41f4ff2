"""The eight-box occlusion records (ftn_bvh_octs / build_octs): a second tree over the reference's leaves that only Scene::intersect_test
walks.  intersect_test (bvh.rs:217-266) never shrinks t_max and returns a boolean, so a ray is occluded iff some LEAF's own box passes the
slab test and one of its primitives passes its test; the interior boxes only have to let every such leaf be reached.  CPU only:
  * structure: every leaf of the reference tree hangs under exactly one record slot, every record is reachable, the stack bound holds;
  * containment: a slot's decoded box (origin + q * step, 8 bits per plane) contains the reference node's box;
  * soundness of the kernel's conservative test (one fused multiply-add per plane, margin 2^-20, far side scaled by 1 + 4 gamma(3)):
    on random rays, whenever the reference's Bounds3f::intersect_test passes for a leaf box, the conservative test passes for every
    record slot on the way down to that leaf -- so the walk of k_wf_trace8_any reaches the leaf.
The kernel itself is compared with the oracle on the GPU (tests/test_gpu_parity.py, tests/test_gpu_fuzz.py, -m gpu)."""
import ctypes as C

import numpy as np
import pytest

from fountain_amd import scenes
from test_quad_bvh import K_FAR, slab

F = np.float32
GAMMA3 = (F(3.0) * F(2.0 ** -24)) / (F(1.0) - F(3.0) * F(2.0 ** -24))
K2 = F(1.0) + F(4.0) * GAMMA3


def build(ftn, desc):
    fn = ftn.lib.ftn_bvh_build
    fn.argtypes = [C.c_void_p] * 5
    from fountain_amd import _abi as A
    nodes = (A.ftn_bvh_node * max(1, 2 * desc.n_prims))()
    order = (C.c_uint32 * max(1, desc.n_prims))()
    n, depth = C.c_uint32(), C.c_uint32()
    assert fn(C.byref(desc), nodes, order, C.byref(n), C.byref(depth)) == 0
    q = ftn.lib.ftn_bvh_octs
    q.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 5
    nrec, bound, nx = C.c_uint32(), C.c_uint32(), C.c_uint32()
    assert q(nodes, n.value, None, C.byref(nrec), C.byref(bound), None, C.byref(nx)) == 0
    rec = np.zeros((nrec.value, 32), np.uint32)
    xbox = np.zeros((max(nx.value, 1), 8), np.float32)
    assert q(nodes, n.value, rec.ctypes.data_as(C.c_void_p), C.byref(nrec), C.byref(bound), xbox.ctypes.data_as(C.c_void_p), C.byref(nx)) == 0
    return [nodes[i] for i in range(n.value)], rec, bound.value, xbox[: nx.value]


def decode(rec):
    """-> origin[3], step[3], links[8], lo[8][3], hi[8][3] (float64: origin + q * step is exact there)"""
    org = rec[:3].view(np.float32).astype(np.float64)
    step = np.array([2.0 ** (int((rec[3] >> (8 * a)) & 0xff) - 127) for a in range(3)])
    links = rec[4:12]
    q = np.zeros((6, 8), np.uint32)
    for j in range(6):
        for k in range(8):
            q[j, k] = (rec[12 + 2 * j + (k >> 2)] >> (8 * (k & 3))) & 0xff
    lo = np.stack([org[a] + q[2 * a] * step[a] for a in range(3)], axis=1)
    hi = np.stack([org[a] + q[2 * a + 1] * step[a] for a in range(3)], axis=1)
    return org, step, links, q, lo, hi


def slab_many(lo, hi, o, inv, t_max):
    """Bounds3f::intersect_test (bounds.rs:214-233) for many boxes at once, float32 operation by operation (== test_quad_bvh.slab)"""
    t0 = np.zeros(len(lo), np.float32); t1 = np.full(len(lo), t_max, np.float32); ok = np.ones(len(lo), bool)
    with np.errstate(all="ignore"):
        for i in range(3):
            tn = ((lo[:, i] - o[i]).astype(np.float32) * inv[i]).astype(np.float32); tf = ((hi[:, i] - o[i]).astype(np.float32) * inv[i]).astype(np.float32)
            sw = tn > tf
            tn, tf = np.where(sw, tf, tn), np.where(sw, tn, tf)
            tf = (tf * K_FAR).astype(np.float32)
            t0 = np.fmax(t0, tn); t1 = np.fmin(t1, tf)
            ok &= ~(t0 > t1)
    return ok


def conservative(rec, o, inv, t_max):
    """t8_any_step's test of the eight slots, in float32 with the kernel's expressions (the fused multiply-add through float64: the
    product of two float32 is exact there)"""
    org = rec[:3].view(np.float32)
    links = rec[4:12]
    t0 = np.zeros(8, np.float32); t1 = np.full(8, t_max, np.float32); far = np.full(8, np.inf, np.float32)
    with np.errstate(all="ignore"):
        for a in range(3):
            step = F(2.0 ** (int((rec[3] >> (8 * a)) & 0xff) - 127))
            aa = F(step * inv[a]); b = F(F(org[a] - o[a]) * inv[a])
            m = F(F(np.abs(b) + F(F(255.0) * np.abs(aa))) * F(2.0 ** -20))
            bn, bf = F(b - m), F(b + m)
            qlo = np.array([(rec[12 + 4 * a + (k >> 2)] >> (8 * (k & 3))) & 0xff for k in range(8)], np.float64)
            qhi = np.array([(rec[12 + 4 * a + 2 + (k >> 2)] >> (8 * (k & 3))) & 0xff for k in range(8)], np.float64)
            qn, qf = (qhi, qlo) if inv[a] < 0 else (qlo, qhi)
            tn = (qn * np.float64(aa) + np.float64(bn)).astype(np.float32)
            tf = (qf * np.float64(aa) + np.float64(bf)).astype(np.float32)
            t0 = np.fmax(t0, tn); far = np.fmin(far, tf)
        t1 = np.fmin(t1, (far * K2).astype(np.float32))
    return (~(t0 > t1)) & (links != 0xffffffff)


@pytest.mark.parametrize("name", ["cubes3", "cube", "loose_triangles"])
def test_eight_box_records(ftn, name):
    if name == "cubes3":
        b, cam, res = scenes.instanced_cubes(ftn, n_copies=3, res=(16, 16), env_n=4)
    elif name == "cube":
        b, cam, res = scenes.rounded_cube_env(ftn, res=16, env_n=4)
    else:
        from fountain_amd import SceneBuilder
        b = SceneBuilder(ftn); b.material("matte")
        rng = np.random.default_rng(3)
        for _ in range(40):                                              # loose triangles of very different sizes, some degenerate boxes
            c = rng.uniform(-50, 50, 3); s = 10.0 ** rng.uniform(-3, 1.5)
            P = c + rng.normal(size=(3, 3)) * s
            if rng.random() < 0.2: P[:, 2] = c[2]                        # flat in z
            b.shape("trianglemesh", P=P.astype(np.float32), indices=[0, 1, 2])
    desc, keep = b.build_desc()
    nodes, rec, bound, xbox = build(ftn, desc)
    n_leaves = sum(1 for n in nodes if n.is_leaf)
    # ---- structure + containment
    leaf_of_first_prim = {n.idx: i for i, n in enumerate(nodes) if n.is_leaf}
    seen_leaf, seen_rec, parent_slot = set(), {0}, {}
    path_to_leaf = {}                                                    # leaf node index -> [(record index, slot), ...] from the root
    todo, max_pending = [(0, [])], 0
    while todo:
        max_pending = max(max_pending, len(todo) - 1)
        q, path = todo.pop()
        org, step, links, qq, lo, hi = decode(rec[q])
        n_valid = int((links != 0xffffffff).sum())
        assert 2 <= n_valid <= 8
        for k in range(8):
            if links[k] == 0xffffffff:
                continue
            if links[k] >> 31:
                assert (links[k] >> 30) & 1                              # (no vertex data in this entry point: explicit boxes)
                xb = xbox[int(links[k] & 0x3fffffff)]
                node = leaf_of_first_prim[int(xb[3:4].view(np.uint32)[0])]
                assert node not in seen_leaf
                seen_leaf.add(node)
                path_to_leaf[node] = path + [(q, k)]
                assert np.array_equal(xb[:3], np.array(nodes[node].bmin[:], np.float32)) and np.array_equal(xb[4:7], np.array(nodes[node].bmax[:], np.float32))
                cb = nodes[node]
            else:
                child = int(links[k]) // 128
                assert int(links[k]) % 128 == 0 and child not in seen_rec and child > q
                seen_rec.add(child)
                todo.append((child, path + [(q, k)]))
                cb = None
            if cb is not None:
                assert (lo[k] <= np.array(cb.bmin[:], np.float64)).all() and (hi[k] >= np.array(cb.bmax[:], np.float64)).all()
    assert len(seen_leaf) == n_leaves and len(seen_rec) == len(rec)
    assert len(rec) < n_leaves / 3 + 2                                   # ~ leaves / (fan-out - 1), fan-out well above four
    # an interior slot's decoded box contains every leaf box below it (the records below re-quantise on their own grids and may stick out)
    dec = [decode(r_) for r_ in rec]
    for li, path in path_to_leaf.items():
        for (q, k) in path:
            lo, hi = dec[q][4], dec[q][5]
            assert (lo[k] <= np.array(nodes[li].bmin[:], np.float64)).all() and (hi[k] >= np.array(nodes[li].bmax[:], np.float64)).all()
    # ---- soundness of the conservative test on rays
    rng = np.random.default_rng(11)
    wb_lo = np.array(nodes[0].bmin[:], np.float32); wb_hi = np.array(nodes[0].bmax[:], np.float32)
    leaves = [i for i, n in enumerate(nodes) if n.is_leaf]
    leaf_lo = np.array([nodes[i].bmin[:] for i in leaves], np.float32); leaf_hi = np.array([nodes[i].bmax[:] for i in leaves], np.float32)
    one = slab(leaf_lo[0], leaf_hi[0], wb_lo - F(1.0), F(1.0) / (leaf_lo[0] - wb_lo + F(1.5)), F(np.inf))[0]
    assert one == slab_many(leaf_lo[:1], leaf_hi[:1], wb_lo - F(1.0), F(1.0) / (leaf_lo[0] - wb_lo + F(1.5)), F(np.inf))[0]
    checked = 0
    for r in range(150 if len(leaves) > 2000 else 900):
        o = rng.uniform(wb_lo - (wb_hi - wb_lo), wb_hi + (wb_hi - wb_lo)).astype(np.float32)
        tgt = nodes[leaves[rng.integers(len(leaves))]]
        p = rng.uniform(np.array(tgt.bmin[:]), np.array(tgt.bmax[:]))   # aim at a leaf: grazing and through-going rays alike
        d = (p - o + rng.normal(size=3) * (0.02 if r % 3 else 0.0) * np.linalg.norm(p - o)).astype(np.float32)
        if r % 7 == 0: d = d * F(1e-6)
        if (d == 0).any(): continue
        with np.errstate(all="ignore"):
            inv = (F(1.0) / d).astype(np.float32)
        t_max = F(np.inf) if r % 2 else F(rng.uniform(0.3, 1.5))
        cache = {}
        for li in np.array(leaves)[slab_many(leaf_lo, leaf_hi, o, inv, t_max)]:
            for (q, k) in path_to_leaf[int(li)]:
                if q not in cache:
                    cache[q] = conservative(rec[q], o, inv, t_max)
                assert cache[q][k], ("a leaf the reference enters is cut off", name, r, li, q, k)
                checked += 1
    assert checked > 300
    assert bound >= max_pending
