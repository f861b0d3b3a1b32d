"""The floating-point contract (fpspec.h) against double-precision numpy."""
import numpy as np

import oracle_lib as ol


def ulp_err(got, exact):
    exact32 = exact.astype(np.float32)
    ulp = np.spacing(np.abs(exact32)).astype(np.float64)
    ulp = np.maximum(ulp, 2.0 ** -149)
    return np.abs(got.astype(np.float64) - exact) / ulp


def test_sin_cos_on_phi_range():
    rs = np.random.RandomState(1)
    x = (rs.rand(200000) * 7.0).astype(np.float32)
    for op, fn in (("sin", np.sin), ("cos", np.cos)):
        got = ol.fps(op, x)
        exact = fn(x.astype(np.float64))
        err = ulp_err(got, exact)
        absolute = np.abs(got - exact)
        # <= 3 ulp, or <= 1e-13 absolute next to a zero of the function
        assert np.all((err <= 3.0) | (absolute <= 1e-13)), op


def test_exp_log_acos_atan_tan_pow():
    rs = np.random.RandomState(2)
    x = (-rs.rand(200000) * 100.0).astype(np.float32)
    assert ulp_err(ol.fps("exp", x), np.exp(x.astype(np.float64))).max() <= 3.0
    x = np.concatenate([rs.rand(100000), rs.rand(100000) * 1e-6]).astype(np.float32)
    x = x[x > 0]
    assert ulp_err(ol.fps("log", x), np.log(x.astype(np.float64))).max() <= 3.0
    x = (rs.rand(200000) * 2 - 1).astype(np.float32)
    assert ulp_err(ol.fps("acos", x), np.arccos(x.astype(np.float64))).max() <= 3.0
    assert ulp_err(ol.fps("asin", x), np.arcsin(x.astype(np.float64))).max() <= 3.0
    x = ((rs.rand(200000) * 2 - 1) * 50).astype(np.float32)
    assert ulp_err(ol.fps("atan", x), np.arctan(x.astype(np.float64))).max() <= 3.0
    x = (rs.rand(200000) * 8).astype(np.float32)
    exact = np.tan(x.astype(np.float64))
    keep = np.abs(exact) < 1e4
    assert ulp_err(ol.fps("tan", x)[keep], exact[keep]).max() <= 3.0
    x = (rs.rand(100000) * 4 + 1e-5).astype(np.float32)
    y = np.full_like(x, 1 / 2.2)
    assert ulp_err(ol.fps("pow", x, y), np.power(x.astype(np.float64), y.astype(np.float64))).max() <= 6.0


def test_special_values():
    inf, nan = np.float32(np.inf), np.float32(np.nan)
    assert ol.fps("exp", [0.0])[0] == 1.0 and ol.fps("exp", [-inf])[0] == 0.0 and ol.fps("exp", [200.0])[0] == inf
    assert ol.fps("log", [1.0])[0] == 0.0 and ol.fps("log", [0.0])[0] == -inf and np.isnan(ol.fps("log", [-1.0])[0])
    assert ol.fps("acos", [1.0])[0] == 0.0 and np.isnan(ol.fps("acos", [1.0000001])[0])
    assert ol.fps("sin", [0.0])[0] == 0.0 and ol.fps("cos", [0.0])[0] == 1.0
    # NaN operands are dropped by fmin/fmax, as CUDA's fminf/fmaxf do
    assert ol.fps("fmin", [nan, 2.0], [1.0, nan]).tolist() == [1.0, 2.0]
    assert ol.fps("fmax", [nan, 2.0], [1.0, nan]).tolist() == [1.0, 2.0]
    # float -> int: NaN -> 0, saturating
    assert ol.fps("f2i", [nan, 3.9, -3.9, 1e20, -1e20]).tolist() == [0.0, 3.0, -3.0, 2147483648.0, -2147483648.0]
    assert ol.fps("rsqrt", [4.0])[0] == 0.5
