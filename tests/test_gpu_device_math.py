"""The DEVICE branches of the scalar helpers (csrc/clc_math.hpp, csrc/clc_rows.hpp): v_rsq_f64 / v_rcp_f64 seeds + Newton steps,
frexp + the polynomial logarithm.  The host unit shims (tests/test_lm_controller_host.py, tests/test_rows_math_host.py) compile the
plain expressions of the same functions, so these branches are pinned here, element-wise on the GPU, against numpy's correctly
rounded results: the error bounds the source comments state, in ulps, over the ranges the solver feeds them."""
import numpy as np
import pytest

import camlasercalibratool_amd as clc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sv():
    s = clc.Solver(0)
    yield s
    s.close()


def _ulps(got, want):
    return np.abs(got - want) / np.spacing(np.abs(want))


def _positive(rng, n, lo, hi):
    """log-uniform in [2^lo, 2^hi) with random mantissas + the edges"""
    x = np.exp2(rng.uniform(lo, hi, size=n))
    x[:4] = [np.exp2(lo), np.exp2(hi - 1), 1.0, 3.0]
    return x


def test_reciprocal_and_reciprocal_square_root_sequences(sv):
    rng = np.random.default_rng(5)
    x = _positive(rng, 200000, -300, 300)
    assert _ulps(sv.debug_math(0, x), 1.0 / np.sqrt(x)).max() <= 2.0      # rsqrt_pos: "error ~1 ulp" (against a doubly rounded reference)
    assert _ulps(sv.debug_math(1, x), 1.0 / x).max() <= 1.0                # rcp_pos: "<= 1 ulp"
    assert _ulps(sv.debug_math(2, x), 1.0 / x).max() <= 1.0                # rcp_pos_safe
    assert _ulps(sv.debug_math(3, x), np.sqrt(x)).max() <= 2.0             # sqrt_pos: "<= 2 ulp"
    assert sv.debug_math(3, np.zeros(3)).tolist() == [0.0, 0.0, 0.0]       # ... and 0 for 0 (0 * inf selected away)
    # subnormal / tiny arguments of rcp_pos_safe: the IEEE quotient (+inf where 1/x overflows), never NaN
    tiny = np.array([5e-324, 1e-320, 2.2250738585072014e-308, 1e-310, 4e-309])
    got = sv.debug_math(2, tiny)
    with np.errstate(over="ignore"):
        want = 1.0 / tiny
    assert not np.isnan(got).any() and np.array_equal(np.isinf(got), np.isinf(want))
    fin = np.isfinite(want)
    assert _ulps(got[fin], want[fin]).max() <= 1.0


def test_cauchy_weight_and_cost_logarithm(sv):
    rng = np.random.default_rng(6)
    x = 1.0 + _positive(rng, 200000, -60, 120)  # 1 + r0^2 / lf^2 >= 1
    assert _ulps(sv.debug_math(4, x), 1.0 / x).max() <= 1.0        # rcp_ge1: "~1 ulp"
    assert _ulps(sv.debug_math(5, x), 1.0 / x).max() <= 10.0       # rcp_ge1_weight: "<= 10 ulp" (one Newton step)
    y = _positive(rng, 200000, -1000, 1000)                        # the running product of the cost, any finite positive value
    got = sv.debug_math(6, y)
    want = np.log(y)
    rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-300)
    near1 = np.abs(y - 1.0) < 1e-3
    assert _ulps(got[~near1], want[~near1]).max() <= 2.0           # "< 1 ulp" of the polynomial + the final rounding
    z = 1.0 + rng.uniform(-1e-3, 1e-3, size=100000)                # around 1: the result is tiny, the error bound is relative to f = m - 1
    gz, wz = sv.debug_math(6, z), np.log(z)
    assert (np.abs(gz - wz) <= 4 * np.spacing(np.abs(wz)) + 1e-19).all() and rel.max() < 1e-12
