"""GPU tests of the on-chip resident batched solver (csrc/clc_resident.hpp): one workgroup per problem, the
problem's scan points read from HBM once into registers + LDS, every LM pass of the solve run from there.
Replaces one ceres::Solve per problem (src/LaseCamCalCeres.cpp:301-307).

The resident kernel sums in a different order than the row-layout kernels (a lane holds points of ONE scan and
expands its moments once per pass), so it is compared with the lockstep form to rounding — same termination, same
iteration count, pose and cost inside the BASELINE gates — and with the oracle's DENSE_QR solve on samples."""
import numpy as np
import pytest

import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd

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
    """A problem beyond what a workgroup holds (or with more scans than lanes, or with p.z != 0) leaves the batch to the
    streaming kernels; the results do not change."""
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
    rec[5, 6] = 1e-3  # one point off the lidar plane
    sv.upload_batched(rec, off)
    assert not sv.debug_resident()[0]
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
