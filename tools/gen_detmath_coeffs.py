"""Generates the polynomial coefficients used by fountain_amd/csrc/detmath.h (run once; output pasted into the header).
High-precision Chebyshev fits (mpmath), converted to the monomial basis and rounded to binary64.

  asin(s) = s + s*u*G(u),  u = s^2 in [0, 0.25]      (acos uses it with s = sqrt((1-|x|)/2) for |x| > 0.5)
  atan(t) = t*H(u),        u = t^2 in [0, 1]
"""
import mpmath as mp

mp.mp.dps = 60


def fit(f, a, b, tol):
    for n in range(6, 40):
        coeffs, err = mp.chebyfit(f, [a, b], n, error=True)
        if err < tol:
            return coeffs[::-1], err      # lowest degree first
    raise RuntimeError("no fit")


def G(u):
    if u == 0:
        return mp.mpf(1) / 6
    s = mp.sqrt(u)
    return (mp.asin(s) - s) / (s * u)


def H(u):
    if u == 0:
        return mp.mpf(1)
    t = mp.sqrt(u)
    return mp.atan(t) / t


for name, f, a, b, tol in (("ASIN_G", G, 0, mp.mpf("0.25"), mp.mpf("4e-14")), ("ATAN_H", H, 0, 1, mp.mpf("4e-14"))):
    c, err = fit(f, a, b, tol)
    print("/* %s: degree %d in u, max abs error %s */" % (name, len(c) - 1, mp.nstr(err, 3)))
    print("static const double %s[%d] = {" % (name, len(c)))
    for x in c:
        print("    %s," % mp.nstr(x, 20, min_fixed=-mp.inf, max_fixed=-mp.inf))
    print("};")
