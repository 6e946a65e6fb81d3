"""GPU tests of the on-chip resident batched solver (csrc/clc_resident.hpp): one workgroup per problem, the
problem's scan points read from HBM once into registers + LDS, every LM pass of the solve run from there.
Replaces one ceres::Solve per problem (src/LaseCamCalCeres.cpp:301-307).

The resident kernel sums in a different order than the row-layout kernels (a lane holds points of ONE scan and
expands its moments once per pass), so it is compared with the lockstep form to rounding — same termination, same
iteration count, pose and cost inside the BASELINE gates — and with the oracle's DENSE_QR solve on samples."""
import numpy as np
import pytest

import camlasercalibratool_amd as clc
from camlasercalibratool_amd import _capi, simdata as sd

pytestmark = pytest.mark.gpu

T_TOL = 1e-6
COST_TOL = 1e-8
BASE = 2 | 16 | 32 | 128 | 256 | 512
LOCKSTEP_256 = BASE | 2048 | 1024   # lockstep launches, 256-thread workgroups, serial controller
NO_RESIDENT = BASE | 4096           # whatever the library ran before the resident kernel existed


@pytest.fixture(scope="module")
def sv():
    s = clc.Solver(0)
    yield s
    s.close()


def _dT(a, b):
    return np.abs(sd.T_from_pose7(a) - sd.T_from_pose7(b)).max()


def _key(s):
    return (s.termination, s.num_iterations, s.num_evaluations, s.num_successful_steps, s.num_unsuccessful_steps)


def _batch(seed, P, n_poses, K, noise=0.01):
    if n_poses <= 32:
        return sd.sim_shard_records(seed, 0, P, n_poses, K, noise)
    sh = sd.sim_shard(seed, 0, P, n_poses, K, noise, candidates=4 * n_poses)  # (more board poses than the default draws)
    rec, off = sh.records()
    return rec, off, sh.start_poses(), sh.gt_Tcl


def _cut(rec, off, keep):
    P = len(off) - 1
    idx = np.concatenate([np.arange(off[k], off[k] + keep[k]) for k in range(P)]) if P else np.zeros(0, dtype=np.int64)
    off2 = np.zeros(P + 1, dtype=np.int64)
    off2[1:] = np.cumsum(keep)
    return np.ascontiguousarray(rec[idx.astype(np.int64)]), off2


def test_resident_layout_is_built_and_is_the_default(sv, oracle_mod):
    """24 problems x 12 scans x 97 points, ragged (one cut inside a scan, one of a single observation, one empty): the
    lane layout is built, the default solve runs on it and agrees with the lockstep form and with the oracle."""
    P, n_poses, K = 24, 12, 97
    rec, off, x0, gt = _batch(21, P, n_poses, K)
    keep = np.full(P, n_poses * K)
    keep[3], keep[7], keep[11] = 130, 1, 0
    rec, off = _cut(rec, off, keep)
    sv.set_launch(0, -1)
    sv.upload_batched(rec, off)
    ok, lanes, ppl, rows = sv.debug_resident()
    assert ok and lanes in (256, 512) and 1 <= ppl <= 44 and rows >= P - 1
    pr, sr = sv.solve_batched(x0)
    sv.set_launch(0, LOCKSTEP_256)
    pl, sl = sv.solve_batched(x0)
    sv.set_launch(0, -1)
    for k in range(P):
        assert _key(sr[k]) == _key(sl[k]), k
        assert np.abs(pr[k] - pl[k]).max() <= 1e-9, k
        assert abs(sr[k].final_cost - sl[k].final_cost) <= 1e-11 * max(1.0, abs(sl[k].final_cost)), k
        assert abs(sr[k].initial_cost - sl[k].initial_cost) <= 1e-11 * max(1.0, abs(sl[k].initial_cost)), k
        if keep[k] == 0:
            continue
        ref = oracle_mod.solve(rec[off[k]:off[k + 1]], x0[k], linear_solver="qr")
        assert sr[k].termination == ref.summary.termination and sr[k].num_iterations == ref.summary.num_iterations, k
        assert _dT(pr[k], ref.pose) <= T_TOL and abs(sr[k].final_cost - ref.summary.final_cost) <= COST_TOL, k
    # solving twice is bitwise repeatable
    pr2, sr2 = sv.solve_batched(x0)
    assert np.array_equal(pr, pr2) and [s.final_cost for s in sr] == [s.final_cost for s in sr2]


@pytest.mark.parametrize("use_loss", [1, 0])
@pytest.mark.parametrize("n_poses,K", [(20, 500), (7, 1409), (40, 33), (300, 5), (1, 64), (500, 4)])
def test_resident_scan_shapes(sv, oracle_mod, n_poses, K, use_loss):
    """Scan shapes that exercise the lane split: C4's 20 x 500, long scans over many lanes, short scans of one lane each,
    500 scans of four points (every lane of the 512-lane form its own plane), a single scan — with and without the Cauchy loss."""
    P = 6
    rec, off, x0, gt = _batch(77 + n_poses, P, n_poses, K)
    sv.set_launch(0, -1)
    sv.upload_batched(rec, off)
    ok, lanes, ppl, rows = sv.debug_resident()
    assert ok, (n_poses, K)
    o = clc.default_options()
    o.use_loss = use_loss
    pr, sr = sv.solve_batched(x0, o)
    sv.set_launch(0, LOCKSTEP_256)
    pl, sl = sv.solve_batched(x0, o)
    sv.set_launch(0, -1)
    oo = oracle_mod.default_options()
    oo.use_loss = use_loss
    for k in range(P):
        assert _key(sr[k]) == _key(sl[k]), k
        assert np.abs(pr[k] - pl[k]).max() <= 1e-8, k
        assert abs(sr[k].final_cost - sl[k].final_cost) <= 1e-10 * max(1.0, abs(sl[k].final_cost)), k
    ref = oracle_mod.solve(rec[off[2]:off[3]], x0[2], options=oo, linear_solver="qr")
    assert sr[2].termination == ref.summary.termination and sr[2].num_iterations == ref.summary.num_iterations
    if n_poses >= 7:  # (fewer scans do not determine the pose: the answer is then whatever LM stops at, compared above)
        assert _dT(pr[2], ref.pose) <= T_TOL and abs(sr[2].final_cost - ref.summary.final_cost) <= COST_TOL


def test_resident_falls_back_when_a_problem_does_not_fit(sv, oracle_mod):
    """A problem beyond what a workgroup holds (or with more scans than lanes) leaves the batch to the streaming kernels; the results
    do not change.  (A batch with p.z != 0 stays on chip since round 5: the 512-lane form with 24-byte slots.)"""
    rec, off, x0, gt = _batch(5, 3, 30, 500)  # 15 000 observations per problem
    sv.set_launch(0, -1)
    sv.upload_batched(rec, off)
    assert not sv.debug_resident()[0]
    p1, s1 = sv.solve_batched(x0)
    sv.set_launch(0, LOCKSTEP_256)
    p2, s2 = sv.solve_batched(x0)  # (several workgroups per problem here: another summation order)
    sv.set_launch(0, -1)
    assert [_key(s) for s in s1] == [_key(s) for s in s2] and np.abs(p1 - p2).max() <= 1e-9
    rec, off, x0, gt = _batch(6, 2, 600, 2)  # 600 scans of two points: more scans than lanes
    sv.upload_batched(rec, off)
    assert not sv.debug_resident()[0]
    rec, off, x0, gt = _batch(7, 4, 12, 97)
    rec[5, 6] = 1e-3  # one point off the lidar plane: the z form of the on-chip kernel
    sv.upload_batched(rec, off)
    assert sv.debug_resident()[0] and sv.path_info().batched_points_carry_z == 1
    p3, s3 = sv.solve_batched(x0)
    ref = oracle_mod.solve(rec[off[0]:off[1]], x0[0], linear_solver="qr")
    assert _dT(p3[0], ref.pose) <= T_TOL and abs(s3[0].final_cost - ref.summary.final_cost) <= COST_TOL


def test_resident_flag_4096_is_the_previous_default(sv):
    """Flag 4096 at solve time: the round-2 kernels on the row layout (same decisions as the lockstep form, results to
    rounding — the workgroups per problem differ); at upload time: no lane layout."""
    rec, off, x0, gt = _batch(9, 64, 8, 200)
    sv.set_launch(0, -1)
    sv.upload_batched(rec, off)
    assert sv.debug_resident()[0]
    sv.set_launch(0, NO_RESIDENT)
    pa, sa = sv.solve_batched(x0)
    sv.set_launch(0, LOCKSTEP_256)
    pb, sb = sv.solve_batched(x0)
    sv.set_launch(0, NO_RESIDENT)
    assert [_key(s) for s in sa] == [_key(s) for s in sb] and np.abs(pa - pb).max() <= 1e-9
    sv.upload_batched(rec, off)  # flag 4096 at upload: the lane layout is not built at all
    assert not sv.debug_resident()[0]
    sv.set_launch(0, -1)


def test_resident_wide_ragged_batch(sv, oracle_mod):
    """8 192 ragged problems (6 scans x 80 points cut to random lengths, some of one observation, some empty): the
    resident kernel against the lockstep form on every problem, against the oracle on a sample."""
    P, n_poses, K = 8192, 6, 80
    rec, off, x0, gt = _batch(4242, P, n_poses, K)
    per = n_poses * K
    rng = np.random.default_rng(5)
    keep = rng.integers(per // 2, per + 1, size=P)
    keep[::97] = 1
    keep[5::211] = 0
    rec, off = _cut(rec, off, keep)
    sv.set_launch(0, -1)
    sv.upload_batched(rec, off)
    assert sv.debug_resident()[0]
    pr, sr = sv.solve_batched(x0)
    sv.set_launch(0, LOCKSTEP_256)
    pl, sl = sv.solve_batched(x0)
    sv.set_launch(0, -1)
    n_diff = 0
    for k in range(P):
        if _key(sr[k]) != _key(sl[k]):
            n_diff += 1  # a tolerance test may flip on a last-bit difference of the sums; rare
            continue
        if keep[k] >= per // 2:
            assert _dT(pr[k], pl[k]) <= T_TOL and abs(sr[k].final_cost - sl[k].final_cost) <= COST_TOL, k
    assert n_diff <= P // 500, n_diff
    for k in list(range(0, P, 512)) + [97, 5]:
        r1 = rec[off[k]:off[k + 1]]
        if r1.shape[0] == 0:
            continue
        ref = oracle_mod.solve(r1, x0[k], linear_solver="qr")
        assert sr[k].termination == ref.summary.termination and sr[k].num_iterations == ref.summary.num_iterations, k
        if keep[k] >= per // 2:
            assert _dT(pr[k], ref.pose) <= T_TOL and abs(sr[k].final_cost - ref.summary.final_cost) <= COST_TOL, k


def test_solve_batched_in_place_on_the_handles_buffers(sv):
    """clc_batched_host_buffers + clc_solve_batched on exactly those pointers: same results as the copying call, no staging."""
    rec, off, x0, gt = _batch(31, 40, 10, 120)
    sv.set_launch(0, -1)
    sv.upload_batched(rec, off)
    pa, sa = sv.solve_batched(x0)
    pb, sb = sv.batched_buffers()
    assert pb.shape == (40, 7)
    pb[:] = x0
    pc, sc = sv.solve_batched_inplace()
    assert np.array_equal(pc, pa) and pc.ctypes.data == pb.ctypes.data
    assert [_key(s) for s in sc] == [_key(s) for s in sa] and [s.final_cost for s in sc] == [s.final_cost for s in sa]
    L = _capi.lib()
    import ctypes as C
    bad = (clc.Summary * 40)()
    assert L.clc_solve_batched(sv._h, None, pb.ctypes.data_as(C.POINTER(C.c_double)), bad) == -1  # one of the two pointers only


X0 = sd.pose7_from_T(np.eye(4))
STEP_CHAIN = BASE  # explicit flags: the 256-workgroup step_kernel chain (what clc_solve runs for problems beyond one workgroup)


@pytest.mark.parametrize("seed,noise", [(1, 0.01), (7, 0.03), (3, 0.0)])
def test_single_problem_resident_solve(sv, oracle_mod, seed, noise):
    """The reference's own problem size (simulation_lasercamcal_node: 50 poses x ~114 points): clc_solve with the default
    flags runs the whole LM loop in ONE single-workgroup launch from registers + LDS.  Same decisions and iteration trace as
    the step-kernel chain (to rounding: another summation order) and as the oracle's DENSE_QR solve."""
    S = sd.GenerateSimData(seed, noise_sigma=noise)
    rec = clc.flatten_observations(S, False)
    sv.set_launch(0, -1)
    sv.upload(rec)
    ok, lanes, ppl = sv.debug_resident_single()
    assert ok and lanes == 512 and 1 <= ppl <= 22
    r = sv.solve(X0)
    sv.set_launch(0, STEP_CHAIN)
    c = sv.solve(X0)
    sv.set_launch(0, -1)
    ref = oracle_mod.solve(rec, X0, linear_solver="qr")
    assert _key(r.summary) == _key(c.summary) and r.summary.termination == ref.summary.termination
    assert r.summary.num_iterations == ref.summary.num_iterations and len(r.trace) == len(c.trace) == len(ref.trace)
    assert np.abs(r.pose - c.pose).max() <= 1e-9 and abs(r.summary.final_cost - c.summary.final_cost) <= 1e-12 * max(1.0, c.summary.final_cost)
    assert _dT(r.pose, ref.pose) <= T_TOL and abs(r.summary.final_cost - ref.summary.final_cost) <= COST_TOL
    for a, b in zip(r.trace, c.trace):
        assert (a.iteration, a.step_is_valid, a.step_is_successful) == (b.iteration, b.step_is_valid, b.step_is_successful)
        # (the two paths sum in different orders: quantities that are differences of nearly equal numbers — the gradient at
        # the minimum, the last cost changes and what is derived from them — agree to an absolute, not a relative, tolerance)
        tol = {"cost": (1e-12, 1e-16), "cost_change": (1e-6, 1e-14), "gradient_max_norm": (1e-7, 1e-12), "step_norm": (1e-6, 1e-14),
               "relative_decrease": (1e-5, 1e-7), "trust_region_radius": (1e-5, 0.0)}
        for f, (rt, at) in tol.items():
            x, y = getattr(a, f), getattr(b, f)
            assert abs(x - y) <= rt * abs(y) + at, (f, a.iteration, x, y)
    # bitwise repeatable
    r2 = sv.solve(X0)
    assert np.array_equal(r.pose, r2.pose) and r.summary.final_cost == r2.summary.final_cost


def test_single_problem_resident_options_and_limits(sv, oracle_mod):
    """Solver options reach the in-kernel controller (no loss, iteration cap); a problem beyond 512 x 22 points, or with
    p.z != 0, is not held by one workgroup (the cooperative kernel takes it); profile_events = 1 asks for per-pass events and gets the launch pair."""
    S = sd.sim_fixed_count(5, 20, 500, noise_sigma=0.01)  # 10 000 points: fits
    rec = clc.flatten_observations(S, False)
    sv.set_launch(0, -1)
    sv.upload(rec)
    assert sv.debug_resident_single()[0]
    for kw in (dict(use_loss=0), dict(max_num_iterations=3), dict(function_tolerance=1e-12)):
        o, oo = clc.default_options(), oracle_mod.default_options()
        for k, v in kw.items():
            setattr(o, k, v)
            setattr(oo, k, v)
        r = sv.solve(X0, o)
        ref = oracle_mod.solve(rec, X0, options=oo, linear_solver="qr")
        assert r.summary.termination == ref.summary.termination and r.summary.num_iterations == ref.summary.num_iterations, kw
        assert _dT(r.pose, ref.pose) <= T_TOL and abs(r.summary.final_cost - ref.summary.final_cost) <= COST_TOL, kw
    o = clc.default_options()
    o.profile_events = 1
    r = sv.solve(X0, o)
    assert r.summary.eval_kernel_launches == r.summary.num_evaluations and r.summary.eval_kernel_ms > 0
    S2 = sd.sim_fixed_count(5, 24, 500, noise_sigma=0.01)  # 12 000 points: does not fit one workgroup
    sv.upload(clc.flatten_observations(S2, False))
    assert not sv.debug_resident_single()[0]
    rec[7, 6] = 1e-3
    sv.upload(rec)
    assert not sv.debug_resident_single()[0]
    r = sv.solve(X0)
    ref = oracle_mod.solve(rec, X0, linear_solver="qr")
    assert _dT(r.pose, ref.pose) <= T_TOL and abs(r.summary.final_cost - ref.summary.final_cost) <= COST_TOL


def test_multistart_on_shared_observations_equals_the_batch_of_copies(oracle_mod):
    """clc_solve_multistart (north_star: multi-hypothesis calibration): S start poses on ONE uploaded problem — a workgroup per start, every
    workgroup on the same on-chip layout — against (a) clc_solve_batched of S uploaded COPIES of the problem, bit for bit (same kernel, same
    arithmetic, same lane layout), and (b) the oracle's solve from every start.  Starts spread 0-15 cm / 0-9 deg around the truth: different
    iteration counts, some rejected steps.  Both lane widths (256: the default of a batch; 512: a problem with more than 256 scans)."""
    rng = np.random.default_rng(11)
    S = 70
    for n_poses, pts in ((20, 500), (300, 20)):
        rec = clc.flatten_observations(sd.sim_fixed_count(5 + n_poses, n_poses, pts, noise_sigma=0.01), False)
        n = rec.shape[0]
        x_true = sd.pose7_from_T(sd.tlc_to_tcl(sd.GT_RLC, sd.GT_TLC))
        with clc.Solver(0) as s:
            starts = s.pose_plus(np.tile(x_true, (S, 1)), rng.normal(size=(S, 6)) * np.linspace(0.0, 0.05, S)[:, None])
            s.upload_batched(rec, np.array([0, n], dtype=np.int64))
            pi = s.path_info()
            assert pi.batched_resident == 1 and pi.batched_lanes == (256 if n_poses <= 256 else 512)
            poses, sms = s.solve_multistart(starts)
            again, _ = s.solve_multistart(starts)
            assert np.array_equal(poses, again)
            one, sm1 = s.solve_multistart(starts[3:4])   # a single start: the same answer as in the crowd
            assert np.array_equal(one[0], poses[3]) and sm1[0].num_iterations == sms[3].num_iterations
            # the same S problems as uploaded copies
            s.upload_batched(np.tile(rec, (S, 1)), np.arange(S + 1, dtype=np.int64) * n)
            cp, csm = s.solve_batched(starts)
            with pytest.raises(clc.ClcError):
                s.solve_multistart(starts)               # a batch of S problems is not ONE shared problem
        assert np.array_equal(poses, cp)
        its = set()
        for k in range(S):
            assert (sms[k].num_iterations, sms[k].termination, sms[k].final_cost) == (csm[k].num_iterations, csm[k].termination, csm[k].final_cost)
            ref = oracle_mod.solve(rec, starts[k], linear_solver="qr")
            assert sms[k].num_iterations == ref.summary.num_iterations and sms[k].termination == ref.summary.termination, k
            assert np.abs(sd.T_from_pose7(poses[k]) - sd.T_from_pose7(ref.pose)).max() <= 1e-6 and abs(sms[k].final_cost - ref.summary.final_cost) <= 1e-8, k
            its.add(sms[k].num_iterations)
        assert len(its) >= 2  # (the starts really differ)


def test_multistart_on_a_problem_beyond_one_workgroup_runs_the_starts_in_turn(oracle_mod):
    """A shared problem that no workgroup holds (60 poses x 500 points = 30 000 observations): clc_solve_multistart still answers every
    start — one after the other on the batch's streaming path — with ONE copy of the data on the device."""
    rec = clc.flatten_observations(sd.sim_fixed_count(9, 60, 500, noise_sigma=0.01), False)
    x_true = sd.pose7_from_T(sd.tlc_to_tcl(sd.GT_RLC, sd.GT_TLC))
    with clc.Solver(0) as s:
        starts = s.pose_plus(np.tile(x_true, (5, 1)), np.random.default_rng(2).normal(size=(5, 6)) * 0.03)
        s.upload_batched(rec, np.array([0, rec.shape[0]], dtype=np.int64))
        assert s.path_info().batched_resident == 0
        poses, sms = s.solve_multistart(starts)
    for k in range(5):
        ref = oracle_mod.solve(rec, starts[k], linear_solver="qr")
        assert sms[k].num_iterations == ref.summary.num_iterations
        assert np.abs(sd.T_from_pose7(poses[k]) - sd.T_from_pose7(ref.pose)).max() <= 1e-6 and abs(sms[k].final_cost - ref.summary.final_cost) <= 1e-8


def test_multistart_with_points_off_the_lidar_plane_and_timed(oracle_mod):
    """The shared problem's points carry z (the 512-lane, 24-byte-slot form of the resident kernel): clc_solve_multistart on it, against the
    oracle from every start; profile_events = 1 reports the one launch's duration in every summary."""
    rng = np.random.default_rng(4)
    rec = clc.flatten_observations(sd.sim_fixed_count(12, 18, 400, noise_sigma=0.01), False)
    rec[:, 6] = rng.normal(size=rec.shape[0]) * 0.01
    x_true = sd.pose7_from_T(sd.tlc_to_tcl(sd.GT_RLC, sd.GT_TLC))
    S = 9
    with clc.Solver(0) as s:
        starts = s.pose_plus(np.tile(x_true, (S, 1)), rng.normal(size=(S, 6)) * 0.03)
        s.upload_batched(rec, np.array([0, rec.shape[0]], dtype=np.int64))
        pi = s.path_info()
        assert pi.batched_resident == 1 and pi.batched_points_carry_z == 1 and pi.batched_lanes == 512
        o = clc.default_options()
        o.profile_events = 1
        poses, sms = s.solve_multistart(starts, o)
    for k in range(S):
        ref = oracle_mod.solve(rec, starts[k], linear_solver="qr")
        assert sms[k].num_iterations == ref.summary.num_iterations and sms[k].termination == ref.summary.termination
        assert np.abs(sd.T_from_pose7(poses[k]) - sd.T_from_pose7(ref.pose)).max() <= 1e-6 and abs(sms[k].final_cost - ref.summary.final_cost) <= 1e-8
        assert sms[k].eval_kernel_launches == 1 and sms[k].eval_kernel_ms > 0


def test_calibration_from_starts_mirror_picks_the_lowest_cost(oracle_mod):
    """clc.CamLaserCalibrationFromStarts (the Python mirror of clc_adapter::Session::CalibrationFromStarts): 24 initial guesses on the
    simulation node's observations, refined in place; the winner is the oracle's solve from the same start."""
    S = sd.GenerateSimData(3, noise_sigma=0.01)
    rng = np.random.default_rng(8)
    x_true = sd.pose7_from_T(sd.tlc_to_tcl(sd.GT_RLC, sd.GT_TLC))
    starts = np.stack([sd.T_from_pose7(oracle_mod.pose_plus(x_true, rng.normal(size=6) * 0.1)) for _ in range(24)])
    T = starts.copy()
    best, costs, sms = clc.CamLaserCalibrationFromStarts(S, T, False, False)
    assert costs[best] == costs.min() and not np.array_equal(T, starts)
    rec = clc.flatten_observations(S, False, False)
    ref = oracle_mod.solve(rec, sd.pose7_from_T(starts[best]), linear_solver="qr")
    assert np.abs(T[best] - sd.T_from_pose7(ref.pose)).max() <= 1e-6 and abs(costs[best] - ref.summary.final_cost) <= 1e-8
    assert sms[best].num_iterations == ref.summary.num_iterations
