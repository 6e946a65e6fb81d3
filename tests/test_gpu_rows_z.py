"""GPU tests of scan points OFF the lidar plane (p.z != 0): the row layout that carries z (streaming paths) and the 24-byte-slot form of
the cooperative on-chip kernel (csrc/clc_coop.hpp, WITH_Z — the default for a single problem of up to 65 536 lanes x 26 points).

Oberserve::points is std::vector<Eigen::Vector3d> (include/LaseCamCalCeres.h:22); the reference's own scan conversion writes
z = 0 (src/utilities.cpp:198-215), a caller of the Vector3d interface need not.  Such arrays keep the row layout with a
third 8-byte value per point (rows of 192 doubles, 14 moments per scan: csrc/clc_rows.hpp rows3_*): evaluation, the step
chain, the lockstep batched solver and the closed-form normal equations (which never read z, src/LaseCamCalCeres.cpp:147)
all run on it.  Compared with the oracle's per-residual evaluation / DENSE_QR solve and with the per-point layouts."""
import numpy as np
import pytest

import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd

pytestmark = pytest.mark.gpu

X0 = sd.pose7_from_T(np.eye(4))
T_TOL = 1e-6
COST_TOL = 1e-8
ROWS = 256 | 32           # row layout forced (below 2e5 observations the default keeps the per-point layout)
BASE = 2 | 16 | 32 | 128 | 256 | 512


@pytest.fixture(scope="module")
def sv():
    s = clc.Solver(0)
    yield s
    s.close()


def _dT(a, b):
    return np.abs(sd.T_from_pose7(a) - sd.T_from_pose7(b)).max()


def _off_plane(rec, seed, sigma=0.05, every=1):
    """The same scans with the points lifted off the lidar plane (a tilted / warped scan plane)."""
    rng = np.random.default_rng(seed)
    out = rec.copy()
    out[::every, 6] = rng.normal(size=out[::every].shape[0]) * sigma
    return out


def test_rows_with_z_match_oracle_and_the_per_point_paths(sv, oracle_mod):
    """Ragged C1 scans, C5 board-edge terms, a truncated array; every launch shape of the row kernel, loss on and off."""
    rng = np.random.default_rng(5)
    gt = sd.pose7_from_T(sd.tlc_to_tcl(sd.GT_RLC, sd.GT_TLC))
    cases = [
        _off_plane(clc.flatten_observations(sd.GenerateSimData(3, n_poses=50, noise_sigma=0.02), False), 1),
        _off_plane(clc.flatten_observations(sd.sim_board_edges(5, n_poses=300, pts_per_pose=137, noise_sigma=0.002), True, True), 2, every=7),
        _off_plane(clc.flatten_observations(sd.sim_fixed_count(8, 120, 500, 0.01), False)[:57311], 3, sigma=0.3),
    ]
    for rec in cases:
        sv.set_launch(0, -1)
        sv.upload(rec)
        ok, n_rows, _, _ = sv.debug_rows()
        assert ok and sv.rows_carry_z()[0] and n_rows >= (rec.shape[0] + 63) // 64
        assert not sv.debug_resident_single()[0]  # the on-chip lane layout holds (x, y) only
        for pose in (X0, oracle_mod.pose_plus(gt, rng.normal(size=6) * 0.03)):
            for with_loss in (True, False):
                c0, g0, H0 = oracle_mod.evaluate_ne(rec, pose, with_loss=with_loss)
                sv.set_launch(0, 2 | 16 | 32)  # compact per-point layout
                c2, g2, H2 = sv.eval(pose, with_loss=with_loss)
                for grid, flags in ((0, ROWS), (0, ROWS | 512), (5, ROWS), (0, 256), (3, 256 | 4)):
                    sv.set_launch(grid, flags)
                    c, g, H = sv.eval(pose, with_loss=with_loss)
                    for b in (c0, c2):
                        assert abs(c - b) <= 1e-11 * abs(b)
                    for b in (H0, H2):
                        assert np.abs(H - b).max() <= 1e-11 * np.abs(H0).max()
                    for b in (g0, g2):
                        assert np.abs(g - b).max() <= 1e-10 * np.abs(g0).max() + 1e-16
                    assert sv.eval(pose, with_loss=with_loss)[0] == c  # fixed lane -> row map: bitwise repeatable
    sv.set_launch(0, -1)


def test_rows_with_z_differ_from_the_flat_answer(sv, oracle_mod):
    """Guard against a kernel that silently drops z: the same scans with and without the lift give different sums."""
    rec0 = clc.flatten_observations(sd.sim_fixed_count(4, 40, 300, 0.01), False)
    rec = _off_plane(rec0, 9, sigma=0.2)
    sv.set_launch(0, ROWS)
    sv.upload(rec0)
    assert sv.debug_rows()[0] and not sv.rows_carry_z()[0]
    cf, gf, Hf = sv.eval(X0)
    sv.upload(rec)
    assert sv.rows_carry_z()[0]
    c, g, H = sv.eval(X0)
    c0, g0, H0 = oracle_mod.evaluate_ne(rec, X0)
    assert abs(c - c0) <= 1e-11 * abs(c0) and abs(c - cf) > 1e-3 * abs(cf)
    assert np.abs(H - H0).max() <= 1e-11 * np.abs(H0).max() and np.abs(H - Hf).max() > 1e-3 * np.abs(Hf).max()
    sv.upload(rec0)  # and back: the flag follows the data
    assert not sv.rows_carry_z()[0] and sv.eval(X0)[0] == cf
    sv.set_launch(0, -1)


@pytest.mark.parametrize("use_loss", [1, 0])
def test_solve_on_rows_with_z(sv, oracle_mod, use_loss):
    """clc_solve as the step chain on rows that carry z (forced), and with the default flags (-1: the cooperative kernel's z form)."""
    S = sd.sim_fixed_count(6, 60, 500, noise_sigma=0.01)
    rec = _off_plane(clc.flatten_observations(S, False), 4, sigma=0.02)
    o, oo = clc.default_options(), oracle_mod.default_options()
    o.use_loss = oo.use_loss = use_loss
    ref = oracle_mod.solve(rec, X0, options=oo, linear_solver="qr")
    sv.upload(rec)
    results = []
    for flags in (ROWS, ROWS | 4, -1):
        sv.set_launch(0, flags)
        r = sv.solve(X0, o)
        assert r.summary.termination == ref.summary.termination and r.summary.num_iterations == ref.summary.num_iterations, flags
        assert _dT(r.pose, ref.pose) <= T_TOL and abs(r.summary.final_cost - ref.summary.final_cost) <= COST_TOL, flags
        results.append(r)
    sv.set_launch(0, ROWS)
    assert np.array_equal(results[0].pose, sv.solve(X0, o).pose)  # repeatable bit for bit
    sv.set_launch(0, -1)


def test_solve_default_path_beyond_2e5_observations_with_z(sv, oracle_mod):
    """2.5e5 observations with z: rows that carry z are built (the streaming paths' layout) and the default solve runs on chip."""
    S = sd.sim_fixed_count(7, 500, 500, noise_sigma=0.01)
    rec = _off_plane(clc.flatten_observations(S, False), 5, sigma=0.01)
    sv.set_launch(0, -1)
    sv.upload(rec)
    assert sv.debug_rows()[0] and sv.rows_carry_z()[0]
    r = sv.solve(X0)
    ref = oracle_mod.solve(rec, X0, linear_solver="qr")
    assert r.summary.termination == ref.summary.termination and r.summary.num_iterations == ref.summary.num_iterations
    assert _dT(r.pose, ref.pose) <= T_TOL and abs(r.summary.final_cost - ref.summary.final_cost) <= COST_TOL
    ms, passes = sv.time_steps(X0, 2, 4)
    assert ms > 0 and passes >= 5


def test_closed_form_ignores_z_like_the_reference(sv, oracle_mod):
    """bar_p = (pt.x, pt.y, 1) (src/LaseCamCalCeres.cpp:147): the 9x9 normal equations on rows that carry z equal those
    of the flat scans."""
    S = sd.sim_fixed_count(3, 80, 300, noise_sigma=0.005)
    rec0 = clc.flatten_observations(S, False)
    rec = _off_plane(rec0, 6, sigma=0.1)
    for flags in (ROWS, -1):
        sv.set_launch(0, flags)
        sv.upload(rec0)
        T0, u0, s0 = sv.closed_form()
        sv.upload(rec)
        T1, u1, s1 = sv.closed_form()
        To, uo, so = oracle_mod.closed_form(rec)
        assert u0 == u1 == uo
        assert np.abs(T1 - T0).max() <= 1e-12 and np.abs(T1 - To).max() <= 1e-9
    sv.set_launch(0, -1)


def test_batched_problems_with_z(sv, oracle_mod):
    """A batch with points off the lidar plane runs ON CHIP: the 512-lane form of resident_solve_kernel with 24-byte slots (10 points per
    lane in registers + 12 in LDS), one launch, the batch read from HBM once.  It agrees with the lockstep launches on the rows that carry
    z (one wave per problem and 256-thread workgroups), with the per-point layout, and with the oracle; ragged scans and a problem with
    one scan only included; loss on and off."""
    P = 12
    rec, off, x0, gt = sd.sim_shard_records(31, 0, P, 12, 97, 0.01)
    rec = _off_plane(rec, 8, sigma=0.02)
    sv.set_launch(0, -1)
    sv.upload_batched(rec, off)
    pi = sv.path_info()
    assert sv.debug_rows()[2] and sv.rows_carry_z()[1]
    assert pi.batched_resident == 1 and pi.batched_points_carry_z == 1 and pi.batched_lanes == 512 and pi.batched_points_per_lane <= 22
    for use_loss in (1, 0):
        o, oo = clc.default_options(), oracle_mod.default_options()
        o.use_loss = oo.use_loss = use_loss
        sv.set_launch(0, -1)
        p1, s1 = sv.solve_batched(x0, o)
        sv.set_launch(0, BASE | 2048 | 1024)  # lockstep, 256-thread workgroups, serial controller (rows that carry z)
        p2, s2 = sv.solve_batched(x0, o)
        sv.set_launch(0, BASE | 4096)         # not the on-chip kernel: lockstep, one wave per problem
        p4, s4 = sv.solve_batched(x0, o)
        sv.set_launch(0, 2 | 16 | 32)         # compact per-point layout
        p3, s3 = sv.solve_batched(x0, o)
        sv.set_launch(0, -1)
        for k in range(P):
            ref = oracle_mod.solve(rec[off[k]:off[k + 1]], x0[k], options=oo, linear_solver="qr")
            for p, s in ((p1, s1), (p2, s2), (p3, s3), (p4, s4)):
                assert s[k].termination == ref.summary.termination and s[k].num_iterations == ref.summary.num_iterations, (k, use_loss)
                assert _dT(p[k], ref.pose) <= T_TOL and abs(s[k].final_cost - ref.summary.final_cost) <= COST_TOL, (k, use_loss)
        pr, sr = sv.solve_batched(x0, o)
        assert np.array_equal(pr, p1) and all(sr[k].final_cost == s1[k].final_cost for k in range(P))  # bit-repeatable
    # a later flat batch on the same handle goes back to the (x, y) form, two problems per CU
    rec0, off0, x00, _ = sd.sim_shard_records(32, 0, P, 12, 97, 0.01)
    sv.upload_batched(rec0, off0)
    pi = sv.path_info()
    assert not sv.rows_carry_z()[1] and pi.batched_resident == 1 and pi.batched_points_carry_z == 0 and pi.batched_lanes == 256


def test_batched_z_ragged_scans_and_idle_lanes(sv, oracle_mod):
    """Problems of the reference's C1 shape (50 ragged scans of 0-154 points: lanes with few points, idle lanes, masked padding slots)
    with z in every 3rd point, and a problem whose scans leave one point per lane — after a batch that filled every LDS slot with
    large coordinates."""
    dirty, offd, xd, _ = sd.sim_shard_records(77, 0, 4, 20, 500, 0.01)
    dirty = _off_plane(dirty, 3, sigma=0.5)
    dirty[:, 4:7] *= 300.0
    sv.set_launch(0, -1)
    sv.upload_batched(dirty, offd)
    assert sv.path_info().batched_points_carry_z == 1
    sv.solve_batched(xd)
    recs, x0s = [], []
    for k in range(6):
        S = sd.GenerateSimData(60 + k, noise_sigma=0.01)
        recs.append(_off_plane(clc.flatten_observations(S, False), 20 + k, sigma=0.03, every=3))
    recs.append(_off_plane(clc.flatten_observations(sd.sim_fixed_count(9, 300, 1, noise_sigma=0.01), False), 30, sigma=0.02))  # 300 scans of ONE point
    off = np.zeros(len(recs) + 1, dtype=np.int64)
    off[1:] = np.cumsum([r.shape[0] for r in recs])
    rec = np.concatenate(recs)
    x0 = np.tile(X0, (len(recs), 1))
    sv.upload_batched(rec, off)
    pi = sv.path_info()
    assert pi.batched_resident == 1 and pi.batched_points_carry_z == 1
    p, s = sv.solve_batched(x0)
    for k in range(len(recs)):
        ref = oracle_mod.solve(recs[k], X0, linear_solver="qr")
        assert s[k].termination == ref.summary.termination and s[k].num_iterations == ref.summary.num_iterations, k
        assert _dT(p[k], ref.pose) <= T_TOL and abs(s[k].final_cost - ref.summary.final_cost) <= COST_TOL, k


def test_c3_size_batch_with_z_on_chip(sv, oracle_mod):
    """BASELINE.json configs[2] with points off the lidar plane: 1 024 independent problems x 10^4 observations, every point with z != 0
    (Oberserve::points is a vector of Eigen::Vector3d, include/LaseCamCalCeres.h:22): ONE launch of the 512-lane z form, parity at
    the gates against the oracle on 64 sampled problems, the lockstep launches on all of them."""
    P = 1024
    rec, off, x0, gt = sd.sim_shard_records(4242, 0, P, 20, 500, 0.01)
    rec = _off_plane(rec, 5, sigma=0.02)
    sv.set_launch(0, -1)
    sv.upload_batched(rec, off)
    pi = sv.path_info()
    assert pi.batched_resident == 1 and pi.batched_points_carry_z == 1 and pi.batched_lanes == 512 and pi.batched_points_per_lane == 20
    p, s = sv.solve_batched(x0)
    sv.set_launch(0, BASE | 4096)
    p2, s2 = sv.solve_batched(x0)
    sv.set_launch(0, -1)
    assert all(s[k].termination == s2[k].termination and s[k].num_iterations == s2[k].num_iterations for k in range(P))
    assert np.abs(p - p2).max() <= 1e-9
    assert all(s[k].termination in (1, 2, 3) for k in range(P))
    for k in range(0, P, 16):
        ref = oracle_mod.solve(rec[off[k]:off[k + 1]], x0[k], linear_solver="qr")
        assert s[k].termination == ref.summary.termination and s[k].num_iterations == ref.summary.num_iterations, k
        assert _dT(p[k], ref.pose) <= T_TOL and abs(s[k].final_cost - ref.summary.final_cost) <= COST_TOL, k


def _on_chip(sv):
    pi = sv.path_info()
    return pi.coop_resident == 1 and pi.coop_points_carry_z == 1 and pi.single_resident == 0


@pytest.mark.parametrize("use_loss", [1, 0])
def test_cooperative_kernel_holds_points_with_z(sv, oracle_mod, use_loss):
    """The default clc_solve of a single problem whose points carry z: ONE launch of coop_solve_kernel<loss, nt, WITH_Z> (10 points
    per lane in registers + 16 in LDS, 24 bytes per slot) — C1 size (which without z one workgroup would hold), ragged scans,
    chunks that cut scans, z in every point and in one point only.  Same termination and iteration count as the oracle's DENSE_QR solve
    and as the step chain on the rows that carry z, T_cl / cost within the gates, bit-repeatable."""
    cases = [("C1 size", _off_plane(clc.flatten_observations(sd.GenerateSimData(2, noise_sigma=0.01), False), 1, sigma=0.03)),
             ("ragged scans", _off_plane(clc.flatten_observations(sd.GenerateSimData(4, n_poses=400, noise_sigma=0.02), False), 2, sigma=0.02, every=3)),
             ("24 scans x 5000: chunks cut scans", _off_plane(clc.flatten_observations(sd.sim_fixed_count(5, 24, 5000, noise_sigma=0.01), False), 3)),
             ("one point off the plane", clc.flatten_observations(sd.sim_fixed_count(7, 300, 400, noise_sigma=0.01), False))]
    cases[3][1][12345, 6] = 0.2
    o, oo = clc.default_options(), oracle_mod.default_options()
    o.use_loss = oo.use_loss = use_loss
    sv.set_launch(0, -1)
    for label, rec in cases:
        sv.upload(rec)
        assert _on_chip(sv), label
        n0 = sv.path_info().coop_solves
        r = sv.solve(X0, o)
        assert sv.path_info().coop_solves == n0 + 1 and sv.path_info().coop_timeouts == 0, label
        ref = oracle_mod.solve(rec, X0, options=oo, linear_solver="qr")
        assert r.summary.termination == ref.summary.termination and r.summary.num_iterations == ref.summary.num_iterations, label
        assert _dT(r.pose, ref.pose) <= T_TOL and abs(r.summary.final_cost - ref.summary.final_cost) <= COST_TOL, label
        assert len(r.trace) == len(ref.trace)
        r2 = sv.solve(X0, o)
        assert np.array_equal(r.pose, r2.pose) and r.summary.final_cost == r2.summary.final_cost, label
        sv.set_launch(0, BASE)  # the step chain on the rows that carry z: another summation order, the same decisions
        c = sv.solve(X0, o)
        sv.set_launch(0, -1)
        assert (c.summary.termination, c.summary.num_iterations, c.summary.num_evaluations) == (r.summary.termination, r.summary.num_iterations, r.summary.num_evaluations), label
        assert np.abs(c.pose - r.pose).max() <= 1e-9, label


def test_cooperative_kernel_z_capacity_edge(sv, oracle_mod):
    """65 536 lanes x 26 points with z is the last problem the kernel holds (scans of 13 full lanes); one observation more keeps the
    step chain on the rows that carry z.  Both agree with the oracle."""
    ppl = 26
    K = 16 * ppl      # 416 points per scan = 16 full lanes; a workgroup's chunk = 16 scans = 256 lanes
    n_scans = 256 * 16
    S = sd.sim_fixed_count(21, n_scans + 1, K, noise_sigma=0.01)
    full = _off_plane(clc.flatten_observations(S, False), 9, sigma=0.02)
    rec = np.ascontiguousarray(full[: n_scans * K])
    assert rec.shape[0] == 256 * 256 * ppl
    sv.set_launch(0, -1)
    sv.upload(rec)
    assert _on_chip(sv) and sv.path_info().coop_points_per_lane == ppl
    x0 = sv.pose_plus(sd.pose7_from_T(sd.tlc_to_tcl(sd.GT_RLC, sd.GT_TLC))[None, :], np.array([[0.02, -0.01, 0.02, 0.01, -0.02, 0.01]]))[0]
    r = sv.solve(x0, trace_cap=0)
    ref = oracle_mod.solve(rec, x0, linear_solver="qr", threads=oracle_mod.max_threads())
    assert r.summary.termination == ref.summary.termination and r.summary.num_iterations == ref.summary.num_iterations
    assert _dT(r.pose, ref.pose) <= T_TOL and abs(r.summary.final_cost - ref.summary.final_cost) <= COST_TOL
    # one observation more needs a 27th slot in some lane: not held — the step chain on the rows that carry z
    more = np.ascontiguousarray(full[: n_scans * K + 1])
    sv.upload(more)
    assert sv.path_info().coop_resident == 0 and sv.rows_carry_z()[0]
    r = sv.solve(x0, trace_cap=0)
    ref = oracle_mod.solve(more, x0, linear_solver="qr", threads=oracle_mod.max_threads())
    assert r.summary.num_iterations == ref.summary.num_iterations and _dT(r.pose, ref.pose) <= T_TOL


@pytest.mark.parametrize("use_loss", [1, 0])
@pytest.mark.parametrize("cut", [False, True])
@pytest.mark.parametrize("wgs", [32, 256])
@pytest.mark.parametrize("ppl", [9, 10])
def test_cooperative_kernel_z_with_9_and_10_points_per_lane(sv, oracle_mod, ppl, wgs, cut, use_loss):
    """The z form keeps 10 points per lane in registers, and a pass walks blocks of 8 slots: at 9 or 10 points per lane the block that
    holds slots 8..15 runs on into LDS-held slots 10..15, which must hold zeros (round 4 left them unwritten: whatever the previous
    launch had there — here a 26-points-per-lane problem with coordinates of hundreds of metres — entered the moments).  Both launch
    forms (32 workgroups one-hop, 256 workgroups), loss on and off, against the oracle's DENSE_QR solve; `cut`: the array ends inside a
    scan, so chunks cut scans — lanes with fewer points, idle lanes (9 points per scan-lane then become 10 per lane: the other bug case)."""
    o, oo = clc.default_options(), oracle_mod.default_options()
    o.use_loss = oo.use_loss = use_loss
    sv.set_launch(0, -1)
    # a previous launch that fills every LDS slot with large non-zero points
    dirty = _off_plane(clc.flatten_observations(sd.sim_fixed_count(33, 256 * 2, 26 * 128, noise_sigma=0.01), False), 11, sigma=0.5)
    dirty[:, 4:7] *= 300.0
    sv.upload(dirty)
    assert _on_chip(sv) and sv.path_info().coop_points_per_lane == 26
    sv.solve(X0, o, trace_cap=0)
    K = 16 * ppl
    n_scans = wgs * 16
    S = sd.sim_fixed_count(40 + ppl, n_scans, K, noise_sigma=0.01)
    rec = _off_plane(clc.flatten_observations(S, False), 12, sigma=0.02)
    if cut:
        rec = np.ascontiguousarray(rec[: n_scans * K - 3 * ppl - 1])
    sv.upload(rec)
    pi = sv.path_info()
    assert _on_chip(sv)
    if not cut:  # (a cut array may need one point per lane more, and then the next launch form)
        assert pi.coop_points_per_lane == ppl and pi.coop_workgroups == wgs, (pi.coop_points_per_lane, pi.coop_workgroups)
    x0 = sv.pose_plus(sd.pose7_from_T(sd.tlc_to_tcl(sd.GT_RLC, sd.GT_TLC))[None, :], np.array([[0.03, -0.02, 0.02, 0.02, -0.02, 0.01]]))[0]
    n0 = pi.coop_solves
    r = sv.solve(x0, o, trace_cap=0)
    assert sv.path_info().coop_solves == n0 + 1 and sv.path_info().coop_timeouts == 0
    ref = oracle_mod.solve(rec, x0, options=oo, linear_solver="qr", threads=oracle_mod.max_threads())
    assert r.summary.termination == ref.summary.termination and r.summary.num_iterations == ref.summary.num_iterations
    assert _dT(r.pose, ref.pose) <= T_TOL and abs(r.summary.final_cost - ref.summary.final_cost) <= COST_TOL
