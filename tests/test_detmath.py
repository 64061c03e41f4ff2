"""fountain_amd/csrc/detmath.h: accuracy against correctly rounded references (CPU, via the oracle's det build which compiles
the same header) and bit-identity of the device evaluation with the host evaluation (GPU)."""
import ctypes as C

import numpy as np
import pytest

f32 = np.float32
FUNCS = {0: ("sin", np.sin, (-20, 20)), 1: ("cos", np.cos, (-20, 20)), 2: ("tan", np.tan, (-1.55, 1.55)), 3: ("acos", np.arccos, (-1, 1)),
         4: ("atan", np.arctan, (-50, 50)), 6: ("ln", np.log, (1e-6, 1e4)), 7: ("log2", np.log2, (1e-6, 1e4))}


def host_eval(orc_det, which, x, y):
    fn = orc_det.lib.orc_kat_math
    fn.restype = C.c_float
    fn.argtypes = [C.c_int, C.c_float, C.c_float]
    return np.array([fn(which, float(a), float(b)) for a, b in zip(x, y)], dtype=f32)


@pytest.mark.parametrize("which", sorted(FUNCS))
def test_accuracy_vs_correct_rounding(orc_det, which):
    name, ref, (lo, hi) = FUNCS[which]
    rng = np.random.default_rng(which)
    x = rng.uniform(lo, hi, 20000).astype(f32)
    got = host_eval(orc_det, which, x, x)
    want = ref(x.astype(np.float64)).astype(f32)
    bad = got != want
    assert bad.mean() <= 1e-3, name                       # misrounds: ~1e-5 expected
    ulp = np.abs(got.astype(np.float64) - ref(x.astype(np.float64))) / np.spacing(np.abs(want)).astype(np.float64)
    assert ulp.max() <= 1.0, name


def test_powf_correctly_rounded(orc_det):
    """f32::powf for a positive base (imageio/mod.rs:173 is its only caller on this path): exp(y ln x) in binary64, one rounding"""
    rng = np.random.default_rng(13)
    x = np.concatenate([rng.uniform(0.09, 1.0, 15000), rng.uniform(1e-3, 60.0, 5000)]).astype(f32)
    y = np.concatenate([np.full(15000, 2.4), rng.uniform(-3.0, 3.0, 5000)]).astype(f32)
    got = host_eval(orc_det, 13, x, y)
    exact = np.power(x.astype(np.float64), y.astype(np.float64))
    assert (got != exact.astype(f32)).mean() <= 1e-3
    assert (np.abs(got.astype(np.float64) - exact) / np.spacing(np.abs(exact.astype(f32))).astype(np.float64)).max() <= 1.0
    assert list(host_eval(orc_det, 13, f32([1.0, 4.0, 0.0, 2.0]), f32([7.5, 0.5, 2.4, 0.0]))) == [1.0, 2.0, 0.0, 1.0]


def test_atan2_and_special_values(orc_det):
    rng = np.random.default_rng(5)
    y, x = rng.uniform(-5, 5, 20000).astype(f32), rng.uniform(-5, 5, 20000).astype(f32)
    got = host_eval(orc_det, 5, y, x)
    want = np.arctan2(y.astype(np.float64), x.astype(np.float64)).astype(f32)
    assert (got != want).mean() <= 1e-3
    one = host_eval(orc_det, 7, f32([1.0 / 512, 1024.0, 1.0]), f32([0, 0, 0]))
    assert list(one) == [-9.0, 10.0, 0.0]                  # exact on powers of two (MIP level selection)
    assert host_eval(orc_det, 3, f32([1.0, -1.0]), f32([0, 0]))[0] == 0.0
    assert host_eval(orc_det, 5, f32([0.0]), f32([-1.0]))[0] == f32(np.pi)
    assert np.signbit(host_eval(orc_det, 5, f32([-0.0]), f32([1.0]))[0])


@pytest.mark.gpu
def test_device_math_is_bit_identical_to_host(gpu, orc_det):
    """Every scalar building block the kernels rely on, evaluated on the MI355X and on the host: sin/cos/.../ln in binary64 with
    fixed operation order, IEEE f32 sqrt and divide, the f64 sqrt of the quadratic, next_float_up/down."""
    rng = np.random.default_rng(9)
    n = 200000
    fn = gpu.lib.ftn_test_math
    cases = {0: (-30, 30), 1: (-30, 30), 2: (-1.55, 1.55), 3: (-1, 1), 4: (-100, 100), 5: (-5, 5), 6: (1e-8, 1e6), 7: (1e-8, 1e6)}
    for which, (lo, hi) in cases.items():
        x = rng.uniform(lo, hi, n).astype(f32)
        y = rng.uniform(-5, 5, n).astype(f32)
        out = np.empty(n, f32)
        gpu.check(fn(which, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), C.c_size_t(n), out.ctypes.data_as(C.c_void_p)))
        ref = host_eval(orc_det, which, x[:4000], y[:4000])
        assert np.array_equal(out[:4000].view(np.uint32), ref.view(np.uint32)), which
        if which in FUNCS:                                   # and the full array against the correctly rounded value
            want = FUNCS[which][1](x.astype(np.float64)).astype(f32)
            assert (out != want).mean() <= 1e-3
    x = np.abs(rng.normal(size=n)).astype(f32) * f32(1e3)
    y = (rng.normal(size=n).astype(f32) + f32(1e-3)) * f32(7)
    for which, ref in ((8, np.sqrt(x)), (9, x / y), (10, np.sqrt(x.astype(np.float64) * np.abs(y).astype(np.float64)).astype(f32))):
        out = np.empty(n, f32)
        yy = np.abs(y) if which == 10 else y
        gpu.check(fn(which, x.ctypes.data_as(C.c_void_p), yy.ctypes.data_as(C.c_void_p), C.c_size_t(n), out.ctypes.data_as(C.c_void_p)))
        assert np.array_equal(out.view(np.uint32), ref.astype(f32).view(np.uint32)), which     # IEEE correctly rounded
    v = np.concatenate([rng.normal(size=1000).astype(f32), f32([0.0, -0.0, np.inf, -np.inf, 1e-45, -1e-45])])
    for which, name in ((11, "orc_kat_next_float_up"), (12, "orc_kat_next_float_down")):
        out = np.empty(len(v), f32)
        gpu.check(fn(which, v.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), C.c_size_t(len(v)), out.ctypes.data_as(C.c_void_p)))
        h = getattr(orc_det.lib, name)
        h.restype = C.c_float
        h.argtypes = [C.c_float]
        ref = np.array([h(float(a)) for a in v], dtype=f32)
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), name
