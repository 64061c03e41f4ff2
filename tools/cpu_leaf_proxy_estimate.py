"""How many leaf tests could a plane-slab proxy of the leaf's triangle reject before the triangle is fetched?  (the
traversal kernels wait mostly for the triangle reads of their leaf tests -- 3.9 per closest-hit ray, of which one is the hit.)
CPU estimate, no GPU: uniformly random lines through the AABB of a rounded_cube triangle; counted: lines that hit the triangle, lines that
cross its plane inside the AABB, and lines that pass a 32-bit proxy (normal quantised to 3 x 6 bits, the vertices' extent along it quantised
outwards to 7 + 7 bits of the box's extent along it).  A line that passes neither can be skipped without reading the vertices."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fountain_amd import scenes  # noqa: E402

P, N, F = scenes.rounded_cube_mesh()
P = np.asarray(P, np.float64).reshape(-1, 3)
tri = P[np.asarray(F).reshape(-1, 3)]
rng = np.random.default_rng(0)
tot = hit_n = plane_n = proxy_n = viol = 0


def sph(n):
    v = rng.normal(size=(n, 3))
    return v / np.linalg.norm(v, axis=1, keepdims=True)


for it in range(40):
    tr = tri[rng.integers(0, len(tri), 20000)]
    lo, hi = tr.min(1), tr.max(1)
    c, R = (lo + hi) / 2, np.linalg.norm(hi - lo, axis=1, keepdims=True) / 2 + 1e-9
    a, b = c + sph(len(tr)) * R, c + sph(len(tr)) * R
    o, d = a - (b - a) * 3, b - a
    inv = 1 / np.where(d == 0, 1e-30, d)
    t1, t2 = (lo - o) * inv, (hi - o) * inv
    tn, tf = np.minimum(t1, t2).max(1), np.maximum(t1, t2).min(1)
    inb = (tn <= tf) & (tf > 0)
    e1, e2 = tr[:, 1] - tr[:, 0], tr[:, 2] - tr[:, 0]
    pv = np.cross(d, e2); det = (e1 * pv).sum(1); ok = np.abs(det) > 1e-30
    tv = o - tr[:, 0]; u = (tv * pv).sum(1) / np.where(ok, det, 1); qv = np.cross(tv, e1)
    v = (d * qv).sum(1) / np.where(ok, det, 1); t = (e2 * qv).sum(1) / np.where(ok, det, 1)
    hit = ok & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > 0)
    n = np.cross(e1, e2); dn = (n * d).sum(1)
    pp = o + (((n * (tr[:, 0] - o)).sum(1)) / np.where(dn == 0, 1e-30, dn))[:, None] * d
    ext = hi - lo
    pin = ((pp >= lo - 1e-9 * ext - 1e-12) & (pp <= hi + 1e-9 * ext + 1e-12)).all(1)
    nq = np.round(n / np.abs(n).max(1, keepdims=True) * 31)
    dv = (tr * nq[:, None, :]).sum(2)
    base = np.minimum(nq * lo, nq * hi).sum(1); step = (np.abs(nq) * ext).sum(1) / 127.0 + 1e-12
    dlo = np.floor((dv.min(1) - base) / step) * step + base - 1e-9; dhi = np.ceil((dv.max(1) - base) / step) * step + base + 1e-9
    so, sd = (nq * o).sum(1), (nq * d).sum(1)
    fn, ff = so + tn * sd, so + tf * sd
    qpass = (np.maximum(fn, ff) >= dlo - 1e-9) & (np.minimum(fn, ff) <= dhi + 1e-9)
    tot += inb.sum(); hit_n += (inb & hit).sum(); plane_n += (inb & pin).sum(); proxy_n += (inb & qpass).sum(); viol += (inb & hit & ~qpass).sum()
print("lines through a triangle's AABB: %d; hit the triangle %.3f; cross its plane inside the box %.3f; pass the 32-bit proxy %.3f (hits rejected: %d)"
      % (tot, hit_n / tot, plane_n / tot, proxy_n / tot, viol))
