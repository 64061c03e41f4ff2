"""The four-box records the production traversal kernels walk (ftn_bvh_quads: the reference's tree, two levels per 128-byte record)
visit exactly the leaves the reference walk visits, in the reference's order, under the same evolution of t_max.

CPU only.  Two walkers in numpy float32, fed the same rays and the same (synthetic, deterministic) leaf-hit function:
  * the reference walk, restated from src/bvh.rs:160-215 over the LinearBVHNode array ftn_bvh_build returns;
  * the walk of ftn_trace4.hip's closest-hit kernel over the records ftn_bvh_quads returns: boxes of the two intermediate levels never
    tested, grandchildren ordered by the three split axes, deferred children re-tested at pop time by `t0 > t_max`.
The sequences of (leaf, t_max at that moment) must be identical -- that is what makes closest hits bit-identical whatever the
triangle test does.  The synthetic hit function shrinks t_max aggressively so that the pop-time culling is exercised on every ray.
The real kernels are compared with the oracle on the GPU (tests/test_gpu_parity.py, -m gpu)."""
import ctypes as C

import numpy as np
import pytest

from fountain_amd import SceneBuilder, _abi as A, scenes

F = np.float32
EPS = F(2.0 ** -24)
SCALE = [F(1.0)]          # length scale of the scene under test (set by check())
K_FAR = F(1.0) + F(2.0) * ((F(3.0) * EPS) / (F(1.0) - F(3.0) * EPS))        # 1 + 2 * gamma(3), err_float.rs:7-10


def build(ftn, desc):
    fn = ftn.lib.ftn_bvh_build
    fn.argtypes = [C.c_void_p] * 5
    nodes = (A.ftn_bvh_node * max(1, 2 * desc.n_prims))()
    order = (C.c_uint32 * max(1, desc.n_prims))()
    n, depth = C.c_uint32(), C.c_uint32()
    assert fn(C.byref(desc), nodes, order, C.byref(n), C.byref(depth)) == 0
    q = ftn.lib.ftn_bvh_quads
    q.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    nrec, bound = C.c_uint32(), C.c_uint32()
    assert q(nodes, n.value, None, C.byref(nrec), C.byref(bound)) == 0
    rec = np.zeros((max(1, nrec.value), 32), np.float32)
    assert q(nodes, n.value, rec.ctypes.data_as(C.c_void_p), C.byref(nrec), C.byref(bound)) == 0
    arr = np.ctypeslib.as_array(nodes)[: n.value] if n.value else None
    return [nodes[i] for i in range(n.value)], rec[: nrec.value], bound.value, depth.value


def slab(lo, hi, o, inv, t_max):
    """Bounds3f::intersect_test (bounds.rs:214-233): (passes, t0)"""
    t0, t1 = F(0.0), F(t_max)
    with np.errstate(all="ignore"):
        for i in range(3):
            tn, tf = (lo[i] - o[i]) * inv[i], (hi[i] - o[i]) * inv[i]
            if tn > tf:
                tn, tf = tf, tn
            tf = tf * K_FAR
            t0, t1 = np.fmax(t0, tn), np.fmin(t1, tf)           # f32::max / f32::min ignore a NaN operand
            if t0 > t1:
                return False, t0
    return True, t0


def fake_hit(ray_id, prim, t0_leaf, t_max):
    """stands in for Primitive::intersect: deterministic in (ray, primitive, the leaf's entry distance), accepts t <= t_max"""
    h = (ray_id * 2654435761 + prim * 40503 + 12345) & 0xFFFFFFFF
    h ^= h >> 15
    h = (h * 2246822519) & 0xFFFFFFFF
    h ^= h >> 13
    if (h & 3) != 0:
        return None
    v = F((h >> 8) & 0xFFFF) / F(65536.0)
    with np.errstate(all="ignore"):
        t = F(t0_leaf * (F(1.0) + v * F(0.5)) + v * SCALE[0])
    if not np.isfinite(t) or t > t_max:
        return None
    return t


def reference_walk(nodes, ray_id, o, d, t_max):
    """bvh.rs:160-215"""
    with np.errstate(all="ignore"):
        inv = F(1.0) / d
    neg = [d[0] < 0, d[1] < 0, d[2] < 0]
    visits, stack, cur = [], [], 0
    n_tests = 0
    while True:
        n = nodes[cur]
        n_tests += 1
        ok, t0 = slab(np.array(n.bmin[:], F), np.array(n.bmax[:], F), o, inv, t_max)
        if ok and n.is_leaf:
            for i in range(n.n_prims):
                visits.append((n.idx + i, float(t_max)))
                t = fake_hit(ray_id, n.idx + i, t0, t_max)
                if t is not None:
                    t_max = t
            if not stack:
                break
            cur = stack.pop()
        elif ok:
            if neg[n.axis]:
                stack.append(cur + 1)
                cur = n.idx
            else:
                stack.append(n.idx)
                cur += 1
        else:
            if not stack:
                break
            cur = stack.pop()
    return visits, n_tests


def quad_walk(nodes, rec, bound, ray_id, o, d, t_max):
    """k_wf_trace4 (ftn_trace4.hip), one lane"""
    with np.errstate(all="ignore"):
        inv = F(1.0) / d
    neg24 = ((1 if d[0] < 0 else 0) | (2 if d[1] < 0 else 0) | (4 if d[2] < 0 else 0)) * 0x010101
    visits, n_rec = [], 0
    root = nodes[0]
    ok, t0 = slab(np.array(root.bmin[:], F), np.array(root.bmax[:], F), o, inv, t_max)
    if not ok:
        return visits, 0, 0
    bits = rec.view(np.uint32)
    stack, deepest = [], 0

    def leaf(first, t0_leaf, t_max):
        # the leaf's primitives in order; geom's GF_LEAF_END flag = last primitive of the BVH leaf holding `first`
        n = leaf_size[first]
        for i in range(n):
            visits.append((first + i, float(t_max)))
            t = fake_hit(ray_id, first + i, t0_leaf, t_max)
            if t is not None:
                t_max = t
        return t_max

    leaf_size = {n.idx: n.n_prims for n in nodes if n.is_leaf}
    if root.is_leaf:
        leaf(0, t0, t_max)
        return visits, 0, 0
    cur = 0
    while True:
        n_rec += 1
        r, rb = rec[cur // 128], bits[cur // 128]
        slots = []
        for k in range(4):
            s = r[8 * k: 8 * k + 8]
            meta = int(rb[8 * k + 7])
            ok, t0 = slab(np.array([s[0], s[2], s[4]], F), np.array([s[1], s[3], s[5]], F), o, inv, t_max)
            assert not (ok and meta & (1 << 25)), "an empty slot's box must fail the slab test by itself"
            e = int(rb[8 * k + 6])                     # as stored: record offset, or first primitive | bit 31
            slots.append((e, t0, ok, meta))
        axes = slots[0][3] & neg24                      # slot 0's meta: one-hot axes of A, R, B in bytes 0, 1, 2
        sA, sR, sB = (axes & 0xFF) != 0, (axes & 0xFF00) != 0, (axes & 0xFF0000) != 0
        pa = [slots[1], slots[0]] if sA else [slots[0], slots[1]]
        pb = [slots[3], slots[2]] if sB else [slots[2], slots[3]]
        seq = [x for x in (pb + pa if sR else pa + pb) if x[2]]
        nxt = None
        if seq:
            nxt = seq[0]
            for x in reversed(seq[1:]):
                stack.append((x[0], x[1]))
        deepest = max(deepest, len(stack))
        while True:
            if nxt is None:
                while stack:
                    e, t0 = stack.pop()
                    if not (t0 > t_max):
                        nxt = (e, t0)
                        break
                if nxt is None:
                    return visits, n_rec, deepest
            if nxt[0] >> 31:
                t_max = leaf(nxt[0] & 0x7FFFFFFF, nxt[1], t_max)
                nxt = None
                continue
            cur = nxt[0]
            break
    return visits, n_rec, deepest


def rays_for(lo, hi, n, seed):
    rng = np.random.default_rng(seed)
    ext = hi - lo
    o = (lo - 0.5 * ext + rng.random((n, 3)) * 2.0 * ext).astype(F)
    d = rng.normal(size=(n, 3)).astype(F)
    aim = (lo + rng.random((n, 3)) * ext).astype(F)                 # two thirds of the rays aim at a point inside the world box
    d[: 2 * n // 3] = (aim - o)[: 2 * n // 3] * rng.uniform(0.2, 3.0, (2 * n // 3, 1)).astype(F)
    # axis-parallel directions (zeros, negative zeros) and origins exactly on box planes: the NaN paths of the slab test
    k = n // 6
    d[-2 * k:-k, 0] = 0.0
    d[-2 * k + k // 2:-k, 1] = -0.0
    o[-2 * k: -2 * k + k // 3, 0] = lo[0]
    o[-2 * k + k // 3: -2 * k + k // 2, 2] = hi[2]
    o[-k:] = ((lo + hi) * 0.5 + rng.normal(size=(k, 3)) * 0.05 * ext).astype(F)       # from inside
    t_max = np.where(rng.random(n) < 0.6, np.inf, rng.random(n) * np.linalg.norm(ext) * 2).astype(F)
    return o, d, t_max


def check(ftn, desc, n_rays, seed):
    nodes, rec, bound, depth = build(ftn, desc)
    lo, hi = np.array(nodes[0].bmin[:], F), np.array(nodes[0].bmax[:], F)
    o, d, tm = rays_for(lo, hi, n_rays, seed)
    SCALE[0] = F(0.3 * float(np.linalg.norm(hi - lo)))
    tot_ref = tot_rec = n_vis = 0
    deepest = 0
    n_exc = 0
    for i in range(n_rays):
        va, na = reference_walk(nodes, i, o[i], d[i], tm[i])
        with np.errstate(all="ignore"):
            if not np.isfinite(F(1.0) / d[i]).all():
                n_exc += 1                      # ray_is_exceptional (ftn_trace4.hip): the kernels hand such rays to the reference-order kernel
                continue
        vb, nb, dp = quad_walk(nodes, rec, bound, i, o[i], d[i], tm[i])
        assert va == vb, "ray %d: the four-box walk visits other leaves (or another t_max) than the reference walk" % i
        tot_ref += na
        tot_rec += nb
        n_vis += len(va)
        deepest = max(deepest, dp)
    assert deepest <= bound and n_exc >= n_rays // 8
    print("%d nodes (depth %d), %d records, stack bound %d (deepest seen %d); %d rays: %d node tests in the reference walk, %d record fetches, %d leaf visits"
          % (len(nodes), depth, len(rec), bound, deepest, n_rays, tot_ref, tot_rec, n_vis))
    return tot_ref, tot_rec, n_vis, len(nodes), len(rec), bound, depth


def test_rounded_cube(ftn):
    P, N, Fc = scenes.rounded_cube_mesh()
    b = SceneBuilder(ftn)
    b.material("matte")
    b.shape("trianglemesh", P=P, N=N, indices=Fc)
    desc, keep = b.build_desc()
    tot_ref, tot_rec, n_vis, n_nodes, n_rec, bound, depth = check(ftn, desc, 400, 1)
    assert n_vis > 400 and n_nodes > 8000
    assert tot_rec * 2 < tot_ref                      # fewer than half as many dependent fetches as node visits
    assert bound <= 3 * ((depth + 1) // 2 + 1)


def test_cornell_with_spheres_and_many_copies(ftn):
    b, cam, res = scenes.cornell(ftn, res=32)
    desc, keep = b.build_desc()
    check(ftn, desc, 300, 2)
    b, cam, res = scenes.instanced_cubes(ftn, n_copies=5, res=(32, 32), env_n=8)
    desc, keep = b.build_desc()
    tot_ref, tot_rec, n_vis, *_ = check(ftn, desc, 250, 3)
    assert n_vis > 0 and tot_rec * 2 < tot_ref


def test_degenerate_trees(ftn):
    """one primitive (the root is a leaf), two primitives (one record with two leaf slots), identical centroids (a leaf holding several
    primitives), and a lopsided chain (the equal-counts fallback, bvh.rs:122-131)"""
    tri = [(0, 0, 0), (1, 0, 0), (0, 1, 0)]
    for n_copies, offs in ((1, [0.0]), (2, [0.0, 3.0]), (3, [0.0, 0.0, 0.0]), (7, [0.0, 0.0, 1.0, 1.0, 1.0, 5.0, 9.0])):
        b = SceneBuilder(ftn)
        b.material("matte")
        for dz in offs:
            b.shape("trianglemesh", P=[(x, y, z + dz) for x, y, z in tri], indices=[0, 1, 2])
        desc, keep = b.build_desc()
        check(ftn, desc, 150, 10 + n_copies)


def test_skipping_the_intermediate_box_is_exact():
    """pass(child box) implies pass(parent box) for any t_max' >= t_max whenever 1/d is finite in every component -- the property the
    four-box records rely on -- and NOT otherwise: with a zero direction component and the origin on a bounding plane, 0 * inf = NaN
    drops a constraint from the child's test that the parent's test keeps (why such rays take the reference-order kernel)."""
    rng = np.random.default_rng(5)
    grid = np.array([-2.0, -1.0, -0.5, 0.0, 0.5, 1.0, 2.0], F)
    bad = bad_exceptional = 0
    for it in range(8000):
        plo = rng.choice(grid[:4], 3).astype(F)
        phi = np.maximum(plo, rng.choice(grid[3:], 3).astype(F))
        clo = np.minimum(np.maximum(plo, rng.choice(grid, 3).astype(F)), phi)
        chi = np.minimum(np.maximum(clo, rng.choice(grid, 3).astype(F)), phi)
        o = rng.choice(grid, 3).astype(F)
        d = rng.choice(np.array([-1.0, -0.0, 0.0, 1.0, 0.37, -2.5], F), 3).astype(F)
        if not d.any():
            continue
        with np.errstate(all="ignore"):
            inv = F(1.0) / d
        t_c = F(rng.choice([0.5, 1.0, 3.0, np.inf]))
        t_p = F(t_c * F(rng.choice([1.0, 1.5]))) if np.isfinite(t_c) else t_c
        if it % 2:                                   # also off the grid: rounding in the subtraction and the product
            o = (o + rng.normal(size=3) * 1e-3).astype(F)
            d = np.where(d != 0, d * F(1.0 + rng.normal() * 1e-2), d).astype(F)
            with np.errstate(all="ignore"):
                inv = F(1.0) / d
        ok_c, _ = slab(clo, chi, o, inv, t_c)
        ok_p, _ = slab(plo, phi, o, inv, t_p)
        if ok_c and not ok_p:
            if np.isfinite(inv).all():
                bad += 1
            else:
                bad_exceptional += 1
    assert bad == 0 and bad_exceptional > 0
