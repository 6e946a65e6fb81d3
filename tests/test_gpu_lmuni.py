"""GPU tests of the wave-uniform, register-resident LM controller (csrc/clc_lmuni.hpp) that the on-chip solves run between two passes.

It replaces ceres::Solve's trust-region loop (src/LaseCamCalCeres.cpp:301-307) for the problems that stay on chip.  Every
expression keeps the operand order and the fused multiply-adds of the serial controller (csrc/clc_lm.hpp), so for the same totals
the two must agree BIT FOR BIT: the single-workgroup kernel is instantiated with both (the hooks build's clc_debug_single_controller puts the cooperative
kernel's controller in place of the wavefront controller on the LDS state, itself bit-identical to the serial one; the evaluation
pass, the reduction and therefore the totals are the same code) and whole
solves — pose, summary, every field of every iteration record — are compared for equality, on noisy, ragged, far-start, outlier,
degenerate, option-limited and invalid-step problems."""
import numpy as np
import pytest

import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd

pytestmark = pytest.mark.gpu

X0 = sd.pose7_from_T(np.eye(4))


@pytest.fixture(scope="module")
def sv():
    s = clc.Solver(0)
    yield s
    s.close()


def _both(sv, rec, x0, opt=None):
    sv.set_launch(0, -1)
    sv.upload(rec)
    assert sv.debug_resident_single()[0]
    out = []
    for uni in (False, True):
        sv.debug_single_controller(uni)
        out.append(sv.solve(x0, opt) if opt is not None else sv.solve(x0))
    sv.debug_single_controller(False)
    return out


def _assert_identical(a, b, what):
    assert np.array_equal(a.pose, b.pose), what
    sa, sb = a.summary, b.summary
    for f in ("termination", "num_iterations", "num_successful_steps", "num_unsuccessful_steps", "num_evaluations", "initial_cost", "final_cost"):
        assert getattr(sa, f) == getattr(sb, f), (what, f, getattr(sa, f), getattr(sb, f))
    assert len(a.trace) == len(b.trace) == sa.num_iterations + 1, what
    for i, (p, q) in enumerate(zip(a.trace, b.trace)):
        for f in ("iteration", "step_is_valid", "step_is_successful", "cost", "cost_change", "gradient_max_norm", "step_norm",
                  "relative_decrease", "trust_region_radius"):
            assert getattr(p, f) == getattr(q, f), (what, i, f, getattr(p, f), getattr(q, f))


def _problems():
    yield "c1", clc.flatten_observations(sd.GenerateSimData(1, noise_sigma=0.01), False), X0
    yield "ragged noisy", clc.flatten_observations(sd.GenerateSimData(7, n_poses=90, noise_sigma=0.03), False), X0
    yield "20 x 500", clc.flatten_observations(sd.sim_fixed_count(3, 20, 500, noise_sigma=0.01), False), X0
    far = X0.copy()
    far[:3] = (30.0, -20.0, 5.0)
    yield "far start", clc.flatten_observations(sd.sim_fixed_count(5, 40, 120, noise_sigma=0.01), False), far
    rec = clc.flatten_observations(sd.sim_fixed_count(11, 30, 200, noise_sigma=0.005), False).copy()
    rng = np.random.default_rng(3)
    bad = rng.choice(rec.shape[0], rec.shape[0] // 10, replace=False)
    rec[bad, 4:6] += rng.normal(0, 0.5, (bad.size, 2))  # gross outliers: rejected steps on the way
    yield "outliers", rec, X0
    yield "noise free", clc.flatten_observations(sd.sim_fixed_count(13, 25, 100, noise_sigma=0.0), False), X0
    yield "board edges", clc.flatten_observations(sd.sim_board_edges(6, n_poses=40, pts_per_pose=60, noise_sigma=0.002), True, True), X0


def test_register_controller_is_bit_identical_to_the_lds_controller(sv):
    for name, rec, x0 in _problems():
        a, b = _both(sv, rec, x0)
        _assert_identical(a, b, name)


def test_register_controller_rejected_steps(sv):
    """Rejected steps — HandleUnsuccessfulStep shrinks the radius and the step is recomputed from the Gauss-Newton system kept at x (the
    totals buffer the state points to).  This cost is so benign that Ceres' default acceptance threshold never rejects (54 far starts x
    first radii tried: not one rejected step), so the threshold is raised: above 1 no step is ever good enough and the solve ends on
    the minimum trust-region radius after a run of rejections; just below 1 rejections and acceptances alternate."""
    rec = clc.flatten_observations(sd.GenerateSimData(5, noise_sigma=0.02), False)
    n = 0
    for angles, t, mrd, min_radius in (((0.0, 0.0, 0.0), (0.0, 0.0, 0.0), 1.5, 1e-3), ((2.0, -1.0, 2.5), (3.0, -2.0, 1.0), 0.9999, 1e-32),
                                       ((0.1, 0.2, -0.1), (0.1, 0.0, 0.2), 0.999, 1e-32), ((-2.5, 0.7, 1.0), (0.0, 1.0, -1.0), 1.01, 1e-6)):
        x = sd.pose7_from_T(sd.tlc_to_tcl(sd.rot_zyx(*angles)[0], np.array(t))) if any(angles) else X0.copy()
        o = clc.default_options()
        o.min_relative_decrease = mrd
        o.min_trust_region_radius = min_radius
        o.max_num_iterations = 60
        a, b = _both(sv, rec, x, o)
        _assert_identical(a, b, (angles, t, mrd))
        n += a.summary.num_unsuccessful_steps
    assert n >= 4  # (8 on MI355X)


def test_register_controller_options_and_limits(sv):
    rec = clc.flatten_observations(sd.sim_fixed_count(17, 30, 150, noise_sigma=0.02), False)
    for kw in (dict(use_loss=0), dict(max_num_iterations=3), dict(max_num_iterations=0), dict(function_tolerance=1e-14, parameter_tolerance=1e-3),
               dict(gradient_tolerance=1e-2), dict(jacobi_scaling=0), dict(initial_trust_region_radius=1e-3), dict(min_trust_region_radius=1e3),
               dict(max_trust_region_radius=2e4), dict(min_relative_decrease=0.9)):
        o = clc.default_options()
        for k, v in kw.items():
            setattr(o, k, v)
        a, b = _both(sv, rec, X0, o)
        _assert_identical(a, b, kw)


def test_register_controller_invalid_steps_take_the_serial_path(sv):
    """A trust-region radius so large that the damping vanishes next to a rank-deficient Gauss-Newton matrix (every scan on the same
    board pose: the solve cannot fix all six degrees of freedom) makes the Cholesky pivots fail: HandleInvalidStep — the in-register
    loop of lmu_post against the serial loop on the LDS state — and a non-finite start point fails the solve in iteration zero."""
    S = sd.sim_fixed_count(19, 1, 400, noise_sigma=0.0)
    rec = np.tile(clc.flatten_observations(S, False), (3, 1))
    o = clc.default_options()
    o.initial_trust_region_radius = 1e300
    o.max_trust_region_radius = 1e308
    o.min_lm_diagonal = 1e-300
    a, b = _both(sv, rec, X0, o)
    _assert_identical(a, b, "rank deficient")
    o = clc.default_options()
    o.max_num_consecutive_invalid_steps = 2
    o.initial_trust_region_radius = 1e300
    o.max_trust_region_radius = 1e308
    o.min_lm_diagonal = 1e-300
    a, b = _both(sv, rec, X0, o)
    _assert_identical(a, b, "rank deficient, 2 invalid steps allowed")
    bad = X0.copy()
    bad[0] = np.inf
    rec2 = clc.flatten_observations(sd.sim_fixed_count(23, 10, 100, noise_sigma=0.01), False)
    sv.upload(rec2)
    for uni in (False, True):
        sv.debug_single_controller(uni)
        try:
            r = sv.solve(bad)
            assert r.summary.termination == 6
        except clc.ClcError:
            pass
    sv.debug_single_controller(False)
