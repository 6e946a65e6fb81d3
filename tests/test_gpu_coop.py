"""GPU tests of the cooperative whole-GPU solve (csrc/clc_coop.hpp): one problem of 1e4..2e6 observations dealt once to 256
co-resident workgroups (registers + LDS), every LM pass run from there, the per-pass totals exchanged through tagged words in
device memory instead of a kernel boundary.  Replaces ceres::Solve on one problem (src/LaseCamCalCeres.cpp:301-307) for the
sizes between what one workgroup holds (clc_resident.hpp) and what the chip holds.

The sums are taken in another order than on the streaming layouts: same LM decisions (termination, iteration and evaluation
counts), pose and cost inside the BASELINE gates against the oracle's DENSE_QR solve, and to rounding against the step chain."""
import numpy as np
import pytest

import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd

pytestmark = pytest.mark.gpu

X0 = sd.pose7_from_T(np.eye(4))
T_TOL = 1e-6
COST_TOL = 1e-8
STEP_CHAIN = 2 | 16 | 32 | 128 | 256 | 512   # the explicit default flag set: clc_solve as the step chain


@pytest.fixture(scope="module")
def sv():
    s = clc.Solver(0)
    yield s
    s.close()


def _dT(a, b):
    return np.abs(sd.T_from_pose7(a) - sd.T_from_pose7(b)).max()


def _key(s):
    return (s.termination, s.num_iterations, s.num_evaluations, s.num_successful_steps, s.num_unsuccessful_steps)


def _cases():
    yield "c2 shape, 400 scans x 500", clc.flatten_observations(sd.sim_fixed_count(3, 400, 500, noise_sigma=0.01), False)
    yield "ragged scans 0..180 (C1 generator)", clc.flatten_observations(sd.GenerateSimData(4, n_poses=400, noise_sigma=0.02), False)
    yield "24 scans x 5000: chunks cut scans", clc.flatten_observations(sd.sim_fixed_count(5, 24, 5000, noise_sigma=0.01), False)
    yield "board-edge terms (C5)", clc.flatten_observations(sd.sim_board_edges(6, n_poses=300, pts_per_pose=137, noise_sigma=0.002), True, True)
    yield "just above one workgroup", clc.flatten_observations(sd.sim_fixed_count(7, 23, 500, noise_sigma=0.01), False)


def test_cooperative_solve_is_the_default_and_matches_oracle_and_step_chain(sv, oracle_mod):
    for name, rec in _cases():
        sv.set_launch(0, -1)
        sv.upload(rec)
        built, ppl, solves0, aborts, off = sv.debug_coop()
        assert built and 1 <= ppl <= 40 and not off, name
        assert not sv.debug_resident_single()[0], name
        r = sv.solve(X0)
        assert sv.debug_coop()[2] == solves0 + 1 and sv.debug_coop()[3] == aborts, name  # it ran there, and did not time out
        ref = oracle_mod.solve(rec, X0, linear_solver="qr")
        assert r.summary.termination == ref.summary.termination and r.summary.num_iterations == ref.summary.num_iterations, name
        assert _dT(r.pose, ref.pose) <= T_TOL and abs(r.summary.final_cost - ref.summary.final_cost) <= COST_TOL, name
        assert abs(r.summary.initial_cost - ref.summary.initial_cost) <= 1e-11 * abs(ref.summary.initial_cost), name
        sv.set_launch(0, STEP_CHAIN)
        r2 = sv.solve(X0)
        assert sv.debug_coop()[2] == solves0 + 1, name  # explicit flags: the step chain
        assert _key(r.summary) == _key(r2.summary), name
        assert np.abs(r.pose - r2.pose).max() <= 1e-9 and abs(r.summary.final_cost - r2.summary.final_cost) <= 1e-11 * abs(r2.summary.final_cost), name
        # the iteration trace, record by record
        assert len(r.trace) == len(r2.trace) == r.summary.num_iterations + 1, name
        for a, b in zip(r.trace, r2.trace):
            assert a.step_is_successful == b.step_is_successful and abs(a.cost - b.cost) <= 1e-10 * abs(b.cost), name
            assert abs(a.trust_region_radius - b.trust_region_radius) <= 1e-9 * abs(b.trust_region_radius), name
    sv.set_launch(0, -1)


def test_cooperative_solve_is_repeatable_bit_for_bit(sv):
    """Same lane -> point map, same exchange order every launch; 40 solves in a row advance the pass tags on the same boards."""
    rec = clc.flatten_observations(sd.sim_fixed_count(9, 300, 400, noise_sigma=0.01), False)
    sv.set_launch(0, -1)
    sv.upload(rec)
    assert sv.debug_coop()[0]
    first = sv.solve(X0)
    n0, aborts_before = sv.debug_coop()[2], sv.debug_coop()[3]
    for _ in range(40):
        r = sv.solve(X0)
        assert np.array_equal(r.pose, first.pose) and r.summary.final_cost == first.summary.final_cost
        assert _key(r.summary) == _key(first.summary)
    built, _, solves, aborts, off = sv.debug_coop()
    assert solves == n0 + 40 and aborts == aborts_before and not off
    # another start pose in between leaves no trace on the boards
    other = sd.pose7_from_T(sd.tlc_to_tcl(sd.GT_RLC, sd.GT_TLC))
    sv.solve(other)
    r = sv.solve(X0)
    assert np.array_equal(r.pose, first.pose)


def test_cooperative_solve_options_and_limits(sv, oracle_mod):
    """Solver options reach the in-kernel controller; flag 4096 at upload or a problem one workgroup holds leave
    the problem to the other paths (p.z != 0: the kernel's 24-byte-slot form); profile_events = 1 asks for per-pass events and gets the launch pair, 2 times the one launch."""
    rec = clc.flatten_observations(sd.sim_fixed_count(11, 100, 500, noise_sigma=0.01), False)
    sv.set_launch(0, -1)
    sv.upload(rec)
    assert sv.debug_coop()[0]
    for kw in (dict(use_loss=0), dict(max_num_iterations=3), dict(function_tolerance=1e-12), dict(max_num_iterations=0)):
        o, oo = clc.default_options(), oracle_mod.default_options()
        for k, v in kw.items():
            setattr(o, k, v)
            setattr(oo, k, v)
        n0 = sv.debug_coop()[2]
        r = sv.solve(X0, o)
        assert sv.debug_coop()[2] == n0 + 1
        ref = oracle_mod.solve(rec, X0, options=oo, linear_solver="qr")
        assert r.summary.termination == ref.summary.termination and r.summary.num_iterations == ref.summary.num_iterations, kw
        assert _dT(r.pose, ref.pose) <= T_TOL and abs(r.summary.final_cost - ref.summary.final_cost) <= COST_TOL, kw
    o = clc.default_options()
    o.profile_events = 1
    n0 = sv.debug_coop()[2]
    r = sv.solve(X0, o)
    assert sv.debug_coop()[2] == n0 and r.summary.eval_kernel_launches == r.summary.num_evaluations and r.summary.eval_kernel_ms > 0
    o.profile_events = 2
    r = sv.solve(X0, o)
    assert sv.debug_coop()[2] == n0 + 1 and r.summary.eval_kernel_launches == 1 and 0 < r.summary.eval_kernel_ms < r.summary.solve_ms + 1.0
    recz = rec.copy()
    recz[7, 6] = 1e-3  # one point off the lidar plane: the 24-byte-slot form of the kernel (tests/test_gpu_rows_z.py)
    sv.upload(recz)
    assert sv.debug_coop()[0] and sv.path_info().coop_points_carry_z == 1
    sv.set_launch(0, STEP_CHAIN | 4096)
    sv.upload(rec)
    assert not sv.debug_coop()[0]
    sv.set_launch(0, -1)
    sv.upload(rec[:11000])
    assert not sv.debug_coop()[0] and sv.debug_resident_single()[0]
    sv.upload(rec)
    assert sv.debug_coop()[0]


def test_cooperative_solve_at_c2_full_size(sv, oracle_mod):
    """BASELINE configs[1]: 2 000 poses x 500 points = 1e6 observations, 8 points per lane; against the oracle and the step chain."""
    rec = clc.flatten_observations(sd.sim_fixed_count(1000, 2000, 500, noise_sigma=0.01), False)
    sv.set_launch(0, -1)
    sv.upload(rec)
    built, ppl, _, aborts0, _ = sv.debug_coop()
    assert built and ppl in (8, 16)
    r = sv.solve(X0)
    ref = oracle_mod.solve(rec, X0, linear_solver="qr")
    assert r.summary.termination == ref.summary.termination and r.summary.num_iterations == ref.summary.num_iterations
    assert _dT(r.pose, ref.pose) <= T_TOL and abs(r.summary.final_cost - ref.summary.final_cost) <= COST_TOL
    sv.set_launch(0, STEP_CHAIN)
    r2 = sv.solve(X0)
    sv.set_launch(0, -1)
    assert _key(r.summary) == _key(r2.summary) and np.abs(r.pose - r2.pose).max() <= 1e-9
    assert sv.debug_coop()[3] == aborts0


def test_cooperative_solve_that_cannot_complete_falls_back_and_is_disabled(sv, oracle_mod):
    """The safety net on hardware: a launch that is one workgroup short (test hook) cannot complete its first exchange — the leader that
    misses a row gives up after the census timeout (200 us), raises the launch's abort word, which every waiting workgroup reads, and the
    whole grid leaves; nothing is written, clc_solve answers through the step chain within milliseconds and the handle rests the path
    (16 solves, doubling); re-enabled, it works again on the same boards (the aborted launch's tags are behind the next solve's)."""
    import time
    rec = clc.flatten_observations(sd.sim_fixed_count(13, 120, 400, noise_sigma=0.01), False)
    sv.set_launch(0, -1)
    sv.upload(rec)
    assert sv.debug_coop()[0]
    good = sv.solve(X0)
    _, _, solves0, aborts0, off0 = sv.debug_coop()
    assert not off0
    for drop in (1, 200):
        sv.debug_coop_control(drop_next=drop)
        t0 = time.perf_counter()
        r = sv.solve(X0)
        dt = time.perf_counter() - t0
        _, _, solves, aborts, off = sv.debug_coop()
        assert aborts == aborts0 + 1 and off and solves == solves0, drop
        assert 0.0002 < dt < 0.005, dt  # the census timeout + the step chain, not 20-60 ms of chained timeouts
        assert r.summary.termination == good.summary.termination and r.summary.num_iterations == good.summary.num_iterations
        assert np.abs(r.pose - good.pose).max() <= 1e-9  # the step chain's answer
        r = sv.solve(X0)  # disabled: the step chain again, at once
        assert sv.debug_coop()[2] == solves0
        sv.debug_coop_control(reenable=True)
        r = sv.solve(X0)
        _, _, solves0, aborts0, off0 = sv.debug_coop()
        assert solves0 == solves + 1 and not off0
        assert np.array_equal(r.pose, good.pose) and r.summary.final_cost == good.summary.final_cost
    # without the hook: the path comes back by itself after 16 eligible solves
    sv.debug_coop_control(drop_next=1)
    sv.solve(X0)
    assert sv.debug_coop()[4]
    n = 0
    while sv.debug_coop()[4] and n < 64:
        sv.solve(X0)
        n += 1
    assert n <= 17
    r = sv.solve(X0)
    assert np.array_equal(r.pose, good.pose)
    sv.debug_coop_control(reenable=True)


def test_two_handles_solving_at_once(oracle_mod):
    """Two handles (two streams) on the one GPU, driven from two threads: their cooperative launches would compete for the same 256 CUs —
    the library lets one at a time onto the device (per process), so both threads' solves all run on chip, with zero timeouts; both
    answers are right, nothing hangs, and the whole exercise stays within a second."""
    import threading
    rec = clc.flatten_observations(sd.sim_fixed_count(17, 100, 500, noise_sigma=0.01), False)
    ref = oracle_mod.solve(rec, X0, linear_solver="qr")
    out = {}

    ready = threading.Barrier(2, timeout=60)

    def work(name):
        s = clc.Solver(0)
        try:
            s.upload(rec)
            ready.wait()  # (both uploads are done: what overlaps from here on is cooperative solves only — what the gate is about)
            out[name] = [s.solve(X0) for _ in range(30)] + [s.debug_coop() + (s.path_info().coop_gate_waits_expired,)]
        finally:
            s.close()

    import time
    th = [threading.Thread(target=work, args=(k,)) for k in ("a", "b")]
    t_all = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    t_all = time.perf_counter() - t_all
    assert all(not t.is_alive() for t in th)
    print("two handles: wall %.3f s, aborts a/b %d/%d" % (t_all, out["a"][-1][3], out["b"][-1][3]))
    assert t_all < 5.0  # (uploads and context creation included)
    # one cooperative launch at a time per device in this process (the gate of solve_coop): no launch meets another one's workgroups on
    # the CUs, so none times out (round 4: the two threads alternated aborts) and every solve ran as the one-launch kernel
    assert out["a"][-1][3] == 0 and out["b"][-1][3] == 0, (out["a"][-1], out["b"][-1])
    # (a wait for the gate that outlasts its 5 ms bound — a descheduled holder — sends that one solve down the step chain, counted in
    # clc_path_info.coop_gate_waits_expired: every solve is accounted for either way)
    assert out["a"][-1][2] + out["a"][-1][5] == 30 and out["b"][-1][2] + out["b"][-1][5] == 30
    assert out["a"][-1][5] + out["b"][-1][5] <= 2
    for k in ("a", "b"):
        *res, dbg = out[k]
        for r in res:
            assert r.summary.termination == ref.summary.termination and r.summary.num_iterations == ref.summary.num_iterations
            assert _dT(r.pose, ref.pose) <= T_TOL and abs(r.summary.final_cost - ref.summary.final_cost) <= COST_TOL


def test_cooperative_solve_from_a_far_start(sv, oracle_mod):
    """Start 1e4 m off: every factor 1 + r0^2/lf^2 of the cost product is ~1e13-1e15 and the product is renormalised once per eight
    points; the solve must take the step chain's decisions record by record (and the oracle's)."""
    rec = clc.flatten_observations(sd.sim_fixed_count(19, 150, 400, noise_sigma=0.01), False)
    far = X0.copy()
    far[:3] = (1e4, -2e4, 5e3)
    sv.set_launch(0, -1)
    sv.upload(rec)
    n0 = sv.debug_coop()[2]
    r = sv.solve(far)
    assert sv.debug_coop()[2] == n0 + 1
    sv.set_launch(0, STEP_CHAIN)
    r2 = sv.solve(far)
    sv.set_launch(0, -1)
    ref = oracle_mod.solve(rec, far, linear_solver="qr")
    assert np.isfinite(r.summary.initial_cost) and abs(r.summary.initial_cost - ref.summary.initial_cost) <= 1e-11 * abs(ref.summary.initial_cost)
    assert _key(r.summary) == _key(r2.summary)
    assert r.summary.termination == ref.summary.termination and r.summary.num_iterations == ref.summary.num_iterations
    for a, b in zip(r.trace, r2.trace):
        assert a.step_is_successful == b.step_is_successful and abs(a.cost - b.cost) <= 1e-9 * abs(b.cost)
    assert _dT(r.pose, ref.pose) <= 1e-5 * max(1.0, np.abs(sd.T_from_pose7(ref.pose)).max())


def test_cooperative_solve_pass_tags_wrap(sv):
    """The boards carry 32-bit (solve, pass) tags that only grow; when they run out the boards are cleared and the tags start over.
    Solves right below the limit, across it and after it give the same bits."""
    rec = clc.flatten_observations(sd.sim_fixed_count(23, 90, 400, noise_sigma=0.01), False)
    sv.set_launch(0, -1)
    sv.upload(rec)
    assert sv.debug_coop()[0]
    first = sv.solve(X0)
    aborts0 = sv.debug_coop()[3]
    for tag in (0xFFFFFFFF - 1000, 0xFFFFFFFF - 150, 0xFFFFFFFF - 60, 0xFFFFFFF0):
        sv.debug_coop_set_tag(tag)
        for _ in range(4):  # (a solve reserves max_num_iterations + 4 = 104 tags)
            r = sv.solve(X0)
            assert np.array_equal(r.pose, first.pose) and r.summary.final_cost == first.summary.final_cost
    assert sv.debug_coop()[3] == aborts0 and not sv.debug_coop()[4]


def test_small_problems_run_the_one_hop_form_on_32_workgroups(sv, oracle_mod):
    """Up to 32 x 256 x 13 = 106 496 observations the cooperative solve runs on 32 workgroups, every one of which gathers all 32 rows
    itself: one store-to-load hop per pass instead of two.  Same termination, iteration count and trace decisions as the
    256-workgroup form (clc_set_auto_paths bit 8 at upload) and as the oracle; results to rounding (another summation order);
    bit-repeatable; a launch a workgroup short times out and the step chain answers."""
    import time
    sv.debug_coop_control(reenable=True)
    t_outs = sv.path_info().coop_timeouts
    for n_poses, pts, loss in ((24, 500, 1), (100, 500, 1), (163, 500, 0), (300, 137, 1), (200, 500, 1)):
        rec = clc.flatten_observations(sd.sim_fixed_count(31 + n_poses, n_poses, pts, noise_sigma=0.01), False)
        o, oo = clc.default_options(), oracle_mod.default_options()
        o.use_loss = oo.use_loss = loss
        sv.set_launch(0, -1)
        sv.set_auto_paths(0)
        sv.upload(rec)
        pi = sv.path_info()
        assert pi.coop_resident == 1 and pi.coop_workgroups == 32 and pi.coop_points_per_lane <= 13, (n_poses, pts)
        n0 = pi.coop_solves
        r = sv.solve(X0, o)
        r2 = sv.solve(X0, o)
        assert sv.path_info().coop_solves == n0 + 2 and sv.path_info().coop_timeouts == t_outs
        assert np.array_equal(r.pose, r2.pose) and r.summary.final_cost == r2.summary.final_cost
        sv.set_auto_paths(8)  # (takes effect at upload: the 256-workgroup layout)
        sv.upload(rec)
        assert sv.path_info().coop_workgroups == 256
        w = sv.solve(X0, o)
        sv.set_auto_paths(0)
        ref = oracle_mod.solve(rec, X0, options=oo, linear_solver="qr")
        for x in (r, w):
            assert x.summary.termination == ref.summary.termination and x.summary.num_iterations == ref.summary.num_iterations
            assert _dT(x.pose, ref.pose) <= T_TOL and abs(x.summary.final_cost - ref.summary.final_cost) <= COST_TOL
        assert _key(r.summary) == _key(w.summary) and np.abs(r.pose - w.pose).max() <= 1e-9
        assert [(a.step_is_valid, a.step_is_successful) for a in r.trace] == [(a.step_is_valid, a.step_is_successful) for a in w.trace]
    # 107 000 observations: some lane would need a 14th point -> 256 workgroups
    rec = clc.flatten_observations(sd.sim_fixed_count(5, 214, 500, noise_sigma=0.01), False)
    sv.upload(rec)
    assert sv.path_info().coop_workgroups == 256
    # the one-hop launch a workgroup short: bounded time-out, the step chain's answer, the path rests and comes back
    rec = clc.flatten_observations(sd.sim_fixed_count(9, 60, 500, noise_sigma=0.01), False)
    sv.upload(rec)
    assert sv.path_info().coop_workgroups == 32
    good = sv.solve(X0)
    sv.debug_coop_control(drop_next=1)
    t0 = time.perf_counter()
    bad = sv.solve(X0)
    dt = time.perf_counter() - t0
    assert sv.path_info().coop_timeouts == t_outs + 1 and dt < 0.005
    assert _key(bad.summary) == _key(good.summary) and np.abs(bad.pose - good.pose).max() <= 1e-9
    sv.debug_coop_control(reenable=True)
    again = sv.solve(X0)
    assert np.array_equal(again.pose, good.pose)


def test_small_problems_on_32_workgroups_by_request(sv, oracle_mod):
    """clc_set_small_on_coop (at upload): a problem one workgroup holds ALSO gets the cooperative layout and clc_solve runs it on 32
    co-resident workgroups (the faster pass); same decisions as the single-workgroup kernel and the oracle; when the cooperative launch
    cannot complete (a workgroup short) the single-workgroup kernel answers — not the step chain; without the bit nothing changes."""
    S = sd.GenerateSimData(1, noise_sigma=0.01)  # the reference's C1 size
    rec = clc.flatten_observations(S, False)
    ref = oracle_mod.solve(rec, X0, linear_solver="qr")
    sv.debug_coop_control(reenable=True)
    sv.set_launch(0, -1)
    sv.set_auto_paths(0)
    sv.upload(rec)
    assert sv.path_info().single_resident == 1 and sv.path_info().coop_resident == 0
    single = sv.solve(X0)
    try:
        sv.set_small_on_coop(True)
        sv.upload(rec)
        pi = sv.path_info()
        assert pi.single_resident == 1 and pi.coop_resident == 1 and pi.coop_workgroups == 32
        n0, t0 = pi.coop_solves, pi.coop_timeouts
        r = sv.solve(X0)
        assert sv.path_info().coop_solves == n0 + 1 and sv.path_info().coop_timeouts == t0
        for x in (single, r):
            assert x.summary.termination == ref.summary.termination and x.summary.num_iterations == ref.summary.num_iterations
            assert _dT(x.pose, ref.pose) <= T_TOL and abs(x.summary.final_cost - ref.summary.final_cost) <= COST_TOL
        assert _key(r.summary) == _key(single.summary) and np.abs(r.pose - single.pose).max() <= 1e-9
        assert len(r.trace) == len(single.trace)
        # the cooperative launch a workgroup short: it times out, and the SINGLE-WORKGROUP kernel answers, bit for bit its own result
        sv.debug_coop_control(drop_next=1)
        bad = sv.solve(X0)
        assert sv.path_info().coop_timeouts == t0 + 1
        assert np.array_equal(bad.pose, single.pose) and bad.summary.final_cost == single.summary.final_cost
        # while the cooperative path rests, the single-workgroup kernel keeps answering
        again = sv.solve(X0)
        assert np.array_equal(again.pose, single.pose)
        sv.debug_coop_control(reenable=True)
        assert np.array_equal(sv.solve(X0).pose, r.pose)
        # the switch cleared at solve time: the single-workgroup kernel, although both layouts are there
        sv.set_small_on_coop(False)
        n1 = sv.path_info().coop_solves
        assert np.array_equal(sv.solve(X0).pose, single.pose) and sv.path_info().coop_solves == n1
        with pytest.raises(clc.ClcError):
            sv.set_auto_paths(16)  # (no enable bit hides in the disable mask any more)
    finally:
        sv.set_small_on_coop(False)
        sv.set_auto_paths(0)
        sv.debug_coop_control(reenable=True)
