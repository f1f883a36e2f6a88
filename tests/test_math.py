"""include/mcs_math.h (the deterministic elementary functions shared by the oracle's
det mode and the HIP kernels) against glibc libm: accuracy in ulps and exact identities."""
import ctypes as ct

import numpy as np

from conftest import mcs, orc

dp = ct.POINTER(ct.c_double)


def ev(lib, fn, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = a if b is None else np.ascontiguousarray(b, dtype=np.float64)
    out = np.zeros_like(a)
    assert lib.orc_eval_fn(mcs.capi.FN[fn], len(a), a.ctypes.data_as(dp), b.ctypes.data_as(dp), out.ctypes.data_as(dp)) == 0
    return out


def ulps(x, y):
    return np.abs(x.view(np.int64) - y.view(np.int64))


def test_accuracy_vs_libm():
    det, libm = orc.load("det", mcs.capi), orc.load("libm", mcs.capi)
    assert det.orc_math_mode() == b"det" and libm.orc_math_mode() == b"libm"
    rng = np.random.default_rng(7)
    n = 400_000
    cases = {
        "sin": (rng.uniform(-12, 12, n), None, 2), "cos": (rng.uniform(-12, 12, n), None, 2),
        "asin": (rng.uniform(-1, 1, n), None, 2), "acos": (rng.uniform(-1, 1, n), None, 2),
        "atan2": (rng.normal(size=n), rng.normal(size=n), 3), "log10": (10 ** rng.uniform(-40, 40, n), None, 2),
        "hypot1": (10 ** rng.uniform(-8, 12, n), None, 1), "sqrt": (10 ** rng.uniform(-60, 60, n), None, 0),
        "div": (rng.normal(size=n), rng.normal(size=n), 0),
    }
    for fn, (a, b, tol) in cases.items():
        d, l = ev(det, fn, a, b), ev(libm, fn, a, b)
        assert ulps(d, l).max() <= tol, (fn, int(ulps(d, l).max()))


def test_exact_values():
    det = orc.load("det", mcs.capi)
    z = np.array([0.0])
    assert ev(det, "sin", z)[0] == 0.0 and ev(det, "cos", z)[0] == 1.0
    assert ev(det, "asin", np.array([1.0]))[0] == np.pi / 2 and ev(det, "asin", np.array([-1.0]))[0] == -np.pi / 2
    assert ev(det, "acos", np.array([1.0]))[0] == 0.0
    assert ev(det, "log10", np.array([1.0]))[0] == 0.0
    assert ev(det, "atan2", np.array([0.0]), np.array([1.0]))[0] == 0.0
    # prevfloat(1.0), the clamp of src/scattering.jl:3, is a legal asin argument
    assert np.isfinite(ev(det, "asin", np.array([np.nextafter(1.0, 0.0)]))[0])


def test_mod2pi_range_and_identity():
    det = orc.load("det", mcs.capi)
    rng = np.random.default_rng(3)
    x = rng.uniform(-50, 50, 200_000)
    r = ev(det, "mod2pi", x)
    assert np.all(r >= 0) and np.all(r < 2 * np.pi)
    inside = rng.uniform(0, 6.28, 1000)
    assert np.array_equal(ev(det, "mod2pi", inside), inside)       # Base.mod2pi returns x itself in [0, 2pi)
    k = np.rint((x - r) / (2 * np.pi))
    assert np.max(np.abs(x - k * 2 * np.pi - r)) < 1e-13
