"""GPU edge cases of the point-to-plane path: degenerate geometry, gross outliers, empty and
non-finite inputs, a large array, device-resident input on a caller stream, concurrent handles."""
import ctypes as C
import threading

import numpy as np
import pytest

import camlasercalibratool_amd as clc
from camlasercalibratool_amd import _capi, simdata as sd

pytestmark = pytest.mark.gpu
X0 = sd.pose7_from_T(np.eye(4))


@pytest.fixture(scope="module")
def sv():
    s = clc.Solver(0)
    yield s
    s.close()


def _dT(a, b):
    return np.abs(sd.T_from_pose7(a) - sd.T_from_pose7(b)).max()


def test_unobservable_only_pitch_configuration(sv, oracle_mod):
    """'ONLY pitch' experiment of main/calibr_simulation.cpp:50-51: all boards rotate about one
    camera axis -> the information matrix has a null direction.  Both implementations must report
    it (n_null >= 1, same singular spectrum) and the damped LM must still terminate identically."""
    rng = np.random.default_rng(3)
    P = 40
    ang = (rng.random(P) * 2 - 1) * np.pi / 6
    Rca = sd.rot_zyx(np.zeros(P), ang, np.zeros(P))
    tca = np.stack([np.zeros(P), np.zeros(P), rng.uniform(1, 5, P)], 1)
    Rlc, tlc = sd.GT_RLC, sd.GT_TLC
    n = (Rlc @ Rca)[:, :, 2]
    d = -np.sum(n * ((Rlc @ tca[..., None])[..., 0] + tlc), axis=1)
    theta = np.linspace(-0.6, 0.6, 60)
    pts = []
    off = [0]
    for i in range(P):
        den = np.cos(theta) * n[i, 0] + np.sin(theta) * n[i, 1]
        depth = -d[i] / den
        ok = (depth > 0) & (depth < 8)
        pts.append(np.stack([depth[ok] * np.cos(theta[ok]), depth[ok] * np.sin(theta[ok]), np.zeros(ok.sum())], 1))
        off.append(off[-1] + int(ok.sum()))
    S = sd.ObservationSet(sd.rot_to_quat_wxyz(Rca), tca, np.array(off, dtype=np.int64), np.concatenate(pts),
                          np.array(off, dtype=np.int64), np.concatenate(pts))
    rec = clc.flatten_observations(S, False)
    sv.upload(rec)
    gt = sd.pose7_from_T(sd.tlc_to_tcl(Rlc, tlc))
    H, b, chi2, s6, V, nn = sv.information(gt)
    H0, b0, chi0, s60, V0, nn0 = oracle_mod.information(rec, gt)
    assert nn == nn0 and nn >= 1
    assert np.allclose(s6, s60, rtol=1e-6, atol=1e-9)
    x0 = oracle_mod.pose_plus(gt, np.array([0.02, 0.01, -0.02, 0.01, 0.02, -0.01]))
    ref = oracle_mod.solve(rec, x0, linear_solver="qr")
    sv.set_launch(0, 2 | 16 | 32 | 128 | 256 | 512)  # the step-kernel chain
    res = sv.solve(x0)
    sv.set_launch(0, -1)
    assert res.summary.termination == ref.summary.termination
    assert abs(res.summary.final_cost - ref.summary.final_cost) <= 1e-8
    assert res.summary.num_iterations == ref.summary.num_iterations
    # default flags: 2 400 points fit one workgroup -> the single-workgroup resident solve, which sums in another order.  The
    # data is noise-free, so the solve ends at a cost of ~1e-20 where WHICH convergence test fires first (gradient <= 1e-10 or
    # cost change <= 1e-6 cost) is decided by rounding, and so is the number of iterations spent drifting along the null
    # direction before it does: same minimum (cost), a convergence termination, a bounded number of iterations.
    assert sv.debug_resident_single()[0]
    res1 = sv.solve(x0)
    assert res1.summary.termination in (1, 2, 3) and res1.summary.num_iterations <= ref.summary.num_iterations + 12
    assert abs(res1.summary.final_cost - ref.summary.final_cost) <= 1e-8
    # ... and the same pose (the gate of BASELINE.json) — up to the drift along the unobservable direction(s), which the data does not
    # constrain: the tangent-space difference to the oracle's pose [dt, 2 vec(q_ref^-1 q)] has no component > 1e-6 outside the null
    # space of H — and the same decisions as the step chain for as long as the cost is still far above rounding (>= 1e-10: before
    # the tolerance tests can fire): a controller that diverged could not hide behind the iteration allowance
    def qmul(a, b):  # (x, y, z, w)
        ax, ay, az, aw = a
        bx, by, bz, bw = b
        return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                         aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])
    qr = ref.pose[3:] * np.array([-1, -1, -1, 1])
    dq = qmul(qr, res1.pose[3:])
    delta = np.concatenate([res1.pose[:3] - ref.pose[:3], 2 * dq[:3] * np.sign(dq[3])])
    Vn = V[:, 6 - nn:]
    assert np.abs(delta - Vn @ (Vn.T @ delta)).max() <= 1e-6, (delta, Vn)
    k = 0
    for a, b in zip(res1.trace, res.trace):
        if min(a.cost, b.cost) < 1e-10:
            break
        assert (a.iteration, a.step_is_valid, a.step_is_successful) == (b.iteration, b.step_is_valid, b.step_is_successful)
        assert abs(a.cost - b.cost) <= 1e-6 * b.cost and abs(a.trust_region_radius - b.trust_region_radius) <= 1e-6 * b.trust_region_radius
        k += 1
    assert k >= 3


@pytest.mark.parametrize("max_invalid", [5, 2])
def test_invalid_steps_take_the_same_path_in_the_wavefront_and_the_serial_controller(sv, oracle_mod, max_invalid):
    """Rank-deficient problems (k distinct points of one scan, repeated) with the LM diagonal clamp opened to [0, 0]:
    the undamped normal matrix is exactly singular up to rounding, so step computations fail (non-positive pivot,
    non-finite solution or non-positive model cost change), the radius shrinks, and after
    max_num_consecutive_invalid_steps the solve stops with FAILURE.  The step kernel's wavefront controller hands an
    invalid step to the serial loop (lm_iterate): pose, summary and the whole iteration trace must equal the
    [evaluation, lm_kernel] launch pair's (serial controller throughout) bit for bit."""
    S = sd.sim_fixed_count(5, 1, 300, noise_sigma=0.0)
    base = clc.flatten_observations(S, False)
    o = clc.default_options()
    o.min_lm_diagonal = 0.0
    o.max_lm_diagonal = 0.0
    o.max_num_consecutive_invalid_steps = max_invalid
    seen_invalid = 0
    for k in (1, 2, 3, 5):
        rec = np.ascontiguousarray(np.tile(base[:k], (200 // k, 1)))
        sv.upload(rec)
        res = {}
        for name, fl in (("pair_compact", 2 | 16 | 32), ("step_compact", 2 | 16 | 32 | 128), ("pair_rows", 2 | 32 | 256), ("step_rows", 2 | 32 | 128 | 256)):
            sv.set_launch(0, fl)
            res[name] = sv.solve(X0, o)
        sv.set_launch(0, -1)
        for a, b in (("step_compact", "pair_compact"), ("step_rows", "pair_rows")):
            ra, rb = res[a], res[b]
            assert np.array_equal(ra.pose, rb.pose, equal_nan=True) and ra.summary.termination == rb.summary.termination, (k, a)
            assert ra.summary.num_iterations == rb.summary.num_iterations and ra.summary.num_evaluations == rb.summary.num_evaluations, (k, a)
            assert ra.summary.final_cost == rb.summary.final_cost
            assert [(t.iteration, t.cost, t.step_is_valid, t.step_is_successful, t.trust_region_radius, t.gradient_max_norm) for t in ra.trace] == \
                   [(t.iteration, t.cost, t.step_is_valid, t.step_is_successful, t.trust_region_radius, t.gradient_max_norm) for t in rb.trace], (k, a)
        seen_invalid += sum(t.step_is_valid == 0 for t in res["step_rows"].trace) + sum(t.step_is_valid == 0 for t in res["step_compact"].trace)
    assert seen_invalid > 0, "the configurations are meant to produce invalid steps"


def test_gross_outliers_cauchy_loss(sv, oracle_mod):
    """10 % of the scan points are off by up to 1 m: the Cauchy loss (LaseCamCalCeres.cpp:249)
    keeps the estimate at the ground truth; with the loss disabled it is pulled away."""
    S = sd.GenerateSimData(12, noise_sigma=0.005)
    rng = np.random.default_rng(5)
    bad = rng.random(S.pts.shape[0]) < 0.10
    S.pts[bad] *= (1.0 + rng.uniform(0.1, 0.4, size=bad.sum()))[:, None]
    S.ptl = S.pts.copy()
    rec = clc.flatten_observations(S, False)
    sv.upload(rec)
    res = sv.solve(X0)
    ref = oracle_mod.solve(rec, X0, linear_solver="qr")
    assert _dT(res.pose, ref.pose) <= 1e-6 and abs(res.summary.final_cost - ref.summary.final_cost) <= 1e-8
    assert res.summary.num_iterations == ref.summary.num_iterations
    gt = sd.pose7_from_T(sd.tlc_to_tcl(sd.GT_RLC, sd.GT_TLC))
    err_robust = _dT(res.pose, gt)
    o = clc.default_options()
    o.use_loss = 0
    err_l2 = _dT(sv.solve(X0, o).pose, gt)
    assert err_robust < 0.02 and err_l2 > 3 * err_robust


def test_empty_problem_and_nonfinite_data(sv, oracle_mod):
    sv.upload(np.zeros((0, 8)))
    res = sv.solve(X0)
    ref = oracle_mod.solve(np.zeros((0, 8)), X0, linear_solver="ne")
    assert res.summary.termination == ref.summary.termination == 1  # gradient tolerance at iteration 0
    assert res.summary.num_iterations == 0 and np.array_equal(res.pose, X0) and res.summary.final_cost == 0.0
    rec = clc.flatten_observations(sd.GenerateSimData(1, n_poses=5), False)
    rec[17, 4] = np.nan
    sv.upload(rec)
    res = sv.solve(X0)
    assert res.summary.termination == 6  # FAILURE: the initial evaluation is not finite
    assert np.array_equal(res.pose, X0)
    with pytest.raises(clc.ClcError):
        sv.solve(np.full(7, np.inf))
    o = clc.default_options()
    o.loss_scale_factor = 0.0
    with pytest.raises(clc.ClcError):
        sv.solve(X0, o)


def test_large_array_16m_observations(sv, oracle_mod):
    """16 x 10^6 observations (1 GB of records): linearity against 16 copies of the base array and
    spot parity with the oracle's OpenMP evaluation."""
    S = sd.sim_fixed_count(31, 2000, 500, noise_sigma=0.01)
    rec = clc.flatten_observations(S, False)
    pose = oracle_mod.pose_plus(X0, np.array([0.1, 0.0, -0.1, 0.05, 0.1, -0.2]))
    sv.upload(rec)
    c1, g1, H1 = sv.eval(pose)
    big = np.ascontiguousarray(np.tile(rec, (16, 1)))
    sv.upload(big)
    assert sv.num_observations == 16_000_000
    c, g, H = sv.eval(pose)
    assert abs(c - 16 * c1) <= 1e-11 * c and np.allclose(H, 16 * H1, rtol=1e-11) and np.allclose(g, 16 * g1, rtol=1e-9, atol=1e-9)
    c0, g0, H0 = oracle_mod.evaluate_ne(big, pose, threads=max(1, oracle_mod.max_threads()))
    assert abs(c - c0) <= 1e-11 * c0 and np.abs(H - H0).max() <= 1e-11 * np.abs(H0).max()
    del big
    sv.upload(rec)


def test_device_resident_input_on_a_torch_stream(sv, oracle_mod):
    torch = pytest.importorskip("torch")
    rec = clc.flatten_observations(sd.GenerateSimData(2, noise_sigma=0.01), False)
    t = torch.from_numpy(rec).to("cuda:0")
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        sv.set_stream(stream.cuda_stream)
        sv.upload_device(t.data_ptr(), t.shape[0])
        res = sv.solve(X0)
        sv.set_stream(None)
    ref = oracle_mod.solve(rec, X0, linear_solver="qr")
    assert _dT(res.pose, ref.pose) <= 1e-6 and abs(res.summary.final_cost - ref.summary.final_cost) <= 1e-8


def test_device_side_problem_assembly_is_bitwise_the_host_flatten(sv, oracle_mod):
    """clc_store_observations + clc_select_observations: the residual blocks of every (linefit, boundary) selection built
    on the device from the resident pose-major scans are bitwise the records clc_flatten_observations / the oracle build
    on the host (plane :227-231, scale :239-240, board-edge planes :258-288), ragged scans and empty poses included; and
    the whole Session flow equals the upload-per-call flow."""
    S = sd.sim_board_edges(5, 60, 137, noise_sigma=0.002)
    S.ptl = S.ptl[::-1].copy()  # points_on_line differ from points
    cases = [(S, [(False, False), (True, False), (True, True), (False, True)])]
    S1 = sd.GenerateSimData(0, n_poses=50, noise_sigma=0.01)  # ragged, possibly empty scans
    cases.append((S1, [(False, False), (True, False)]))
    for obs, sels in cases:
        sv.store_observations(obs)
        for lf, bd in sels:
            host = clc.flatten_observations(obs, lf, bd)
            dev = sv.debug_flatten_device(lf, bd)
            assert dev.shape == host.shape and np.array_equal(dev, host), (lf, bd)
            assert sv.select_observations(lf, bd) == host.shape[0] == sv.num_observations
            c, g, H = sv.eval(X0)
            sv.upload(host)
            c2, g2, H2 = sv.eval(X0)
            assert c == c2 and np.array_equal(H, H2) and np.array_equal(g, g2)
    # boundary terms on an empty scan: same error as the host path (the reference throws at :278)
    S2 = sd.sim_board_edges(3, 5, 12)
    S2.pts_off[:] = 0
    sv.store_observations(S2)
    with pytest.raises(clc.ClcError) as e:
        sv.select_observations(True, True)
    assert e.value.code == -4
    # Session == per-call flow (closed form -> refinement with boundary terms -> analysis pass)
    S3 = sd.sim_board_edges(11, 40, 30, noise_sigma=0.002)
    ses = clc.Session(S3, solver=sv)
    Tlc = np.eye(4)
    ses.CamLaserCalClosedSolution(Tlc, verbose=False)
    Tcl = np.linalg.inv(Tlc)
    rep = ses.CamLaserCalibration(Tcl, True, True, verbose=False)
    rec_cf = clc.flatten_observations(S3, True, False)
    sv.upload(rec_cf)
    T2, _, _ = sv.closed_form()
    assert np.array_equal(T2, Tlc)
    sv.upload(clc.flatten_observations(S3, True, True))
    r2 = sv.solve(sd.pose7_from_T(np.linalg.inv(T2)))
    assert np.array_equal(r2.pose, rep.result.pose) and r2.summary.final_cost == rep.result.summary.final_cost
    sv.upload(rec_cf)
    H2 = sv.information(r2.pose)[0]
    assert np.array_equal(H2, rep.H)


def test_device_pointer_front_end_and_batched_upload(sv, oracle_mod):
    """clc_scan_to_points_device -> clc_line_fit_batched_device on torch tensors (nothing crosses PCIe between the steps),
    against the host-pointer entry points; clc_upload_batched_device against clc_upload_batched."""
    import torch
    rng = np.random.default_rng(4)
    n_sc, rays = 300, 360
    ranges = rng.uniform(0.3, 12.0, n_sc * rays).astype(np.float32)
    ranges[::17] = np.inf
    off = np.arange(n_sc + 1, dtype=np.int64) * rays
    am = rng.uniform(-2.4, -2.2, n_sc).astype(np.float32); ai = np.full(n_sc, 0.0131, np.float32); rm = np.full(n_sc, 0.1, np.float32)
    ref_pts = sv.scan_to_points(ranges, off, am, ai, rm)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    d_r, d_off, d_am, d_ai, d_rm = t(ranges), t(off), t(am), t(ai), t(rm)
    d_pts = torch.empty((n_sc * rays, 3), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    sv.scan_to_points_device(d_r.data_ptr(), d_off.data_ptr(), n_sc, n_sc * rays, d_am.data_ptr(), d_ai.data_ptr(), d_rm.data_ptr(),
                             d_pts.data_ptr())
    assert np.array_equal(d_pts.cpu().numpy(), ref_pts)
    # line fit on device-resident (x, y): the first 120 rays of every scan, sentinels replaced
    xy = ref_pts.reshape(n_sc, rays, 3)[:, :120, :2].copy()
    th = rng.uniform(-1.2, 1.2, n_sc); c = rng.uniform(0.8, 5.0, n_sc); tt = np.sort(rng.uniform(-0.5, 0.5, (n_sc, 120)), axis=1)
    xy[:, :, 0] = c[:, None] * np.cos(th)[:, None] - tt * np.sin(th)[:, None] + rng.normal(size=(n_sc, 120)) * 0.004
    xy[:, :, 1] = c[:, None] * np.sin(th)[:, None] + tt * np.cos(th)[:, None] + rng.normal(size=(n_sc, 120)) * 0.004
    xy = np.ascontiguousarray(xy.reshape(-1, 2))
    loff = np.arange(n_sc + 1, dtype=np.int64) * 120
    lines_ref, sms = sv.line_fit_batched(xy, loff, np.zeros((n_sc, 2)))
    d_xy, d_loff = t(xy), t(loff)
    d_lines = torch.zeros((n_sc, 2), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    sv.line_fit_batched_device(d_xy.data_ptr(), d_loff.data_ptr(), n_sc, d_lines.data_ptr())
    assert np.array_equal(d_lines.cpu().numpy(), lines_ref)
    # batched upload from device memory
    rec, boff, xb, gt = sd.sim_shard_records(21, 0, 12, 6, 64, 0.01)
    sv.upload_batched(rec, boff)
    p1, s1 = sv.solve_batched(xb)
    d_rec = t(rec)
    torch.cuda.synchronize()
    sv.upload_batched_device(d_rec.data_ptr(), boff)
    p2, s2 = sv.solve_batched(xb)
    assert np.array_equal(p1, p2) and [s.final_cost for s in s1] == [s.final_cost for s in s2]


def test_two_handles_on_two_threads(oracle_mod):
    """Handles are independent (own stream, own LM state): concurrent solves on two host threads
    give exactly the sequential results."""
    recs = [clc.flatten_observations(sd.sim_fixed_count(s, 300, 200, noise_sigma=0.01), False) for s in (1, 2)]
    seq = []
    for r in recs:
        with clc.Solver(0) as s:
            s.upload(r)
            seq.append(s.solve(X0))
    out = [None, None]

    def work(i):
        with clc.Solver(0) as s:
            s.upload(recs[i])
            for _ in range(20):
                out[i] = s.solve(X0)

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    for i in range(2):
        assert np.array_equal(out[i].pose, seq[i].pose) and out[i].summary.final_cost == seq[i].summary.final_cost


def test_abi_argument_validation(sv):
    L = _capi.lib()
    assert L.clc_create(None, 0) == -1
    h = C.c_void_p()
    assert L.clc_create(C.byref(h), 9999) == -1
    assert L.clc_solve(sv._h, None, None, None, None, 0) == -1
    assert L.clc_upload(sv._h, None, C.c_size_t(5)) == -1
    assert L.clc_set_launch(sv._h, -1, 0) == -1 and L.clc_set_launch(sv._h, 0, 16384) == -1
    assert b"clc_set_launch" in L.clc_last_error()
    o = clc.default_options()
    o.max_num_iterations = -1
    with pytest.raises(clc.ClcError):
        sv.solve(X0, o)


def test_sessions_on_the_shared_handle_do_not_solve_each_others_scans(oracle_mod):
    """Two Sessions on the process-wide solver (calib._shared_solver): the scans stored on a handle belong to the handle,
    so the second Session replaces the first one's — which must put its own back (store generation stamp,
    clc_store_generation) instead of silently calibrating on the other's observations."""
    from camlasercalibratool_amd import calib
    A = sd.GenerateSimData(11, noise_sigma=0.01)
    B = sd.GenerateSimData(12, n_poses=30, noise_sigma=0.02)
    sa = calib.Session(A)
    g0 = sa.sv.store_generation
    sb = calib.Session(B)  # same shared handle: replaces A's scans
    assert sb.sv is sa.sv and sa.sv.store_generation == g0 + 1
    Ta, Tb = np.eye(4), np.eye(4)
    ra = sa.CamLaserCalibration(Ta, False, False, verbose=False)   # must re-store A first
    rb = sb.CamLaserCalibration(Tb, False, False, verbose=False)   # ... and B again
    for S, r in ((A, ra), (B, rb)):
        ref = oracle_mod.solve(clc.flatten_observations(S, False), sd.pose7_from_T(np.eye(4)), linear_solver="qr")
        assert np.abs(sd.T_from_pose7(r.result.pose) - sd.T_from_pose7(ref.pose)).max() <= 1e-6
        assert abs(r.result.summary.final_cost - ref.summary.final_cost) <= 1e-8


def test_repeated_selection_is_built_once_and_forgotten_by_any_upload(sv):
    """clc_select_observations of the selection the observation array already IS returns at once (the flow of main/calibr_offline.cpp
    :166-170 selects points_on_line twice; the simulation node's points_on_line == points, so its (linefit) and (no linefit) selections are
    the same records): same answers, and any upload / new store / upload-time setting in between makes it build again."""
    import time
    S = sd.GenerateSimData(3, noise_sigma=0.01)          # points_on_line == points (main/calibr_simulation.cpp:102)
    host = clc.flatten_observations(S, True, False)
    assert np.array_equal(host, clc.flatten_observations(S, False, False))
    sv.set_launch(0, -1)
    sv.store_observations(S)
    n = sv.select_observations(True, False)
    c0, g0, H0 = sv.eval(X0)
    t = []
    for lf in (True, False, False, True):
        t0 = time.perf_counter()
        assert sv.select_observations(lf, False) == n == host.shape[0] == sv.num_observations
        t.append(time.perf_counter() - t0)
        c, g, H = sv.eval(X0)
        assert c == c0 and np.array_equal(H, H0)
    assert max(t) < 50e-6, t                              # (a cached selection is a few host instructions, not a dozen launches)
    sv.upload(host[: n // 2])                             # any upload forgets the selection
    assert sv.select_observations(True, False) == n and sv.num_observations == n
    assert sv.eval(X0)[0] == c0
    S2 = sd.sim_board_edges(5, 30, 40, noise_sigma=0.002)
    S2.ptl = S2.ptl[::-1].copy()                          # points_on_line differ from points: the two selections are different records
    sv.store_observations(S2)                             # a new store is a new generation
    a = sv.select_observations(True, False); ca = sv.eval(X0)[0]
    b = sv.select_observations(False, False); cb = sv.eval(X0)[0]
    assert a == b and ca != cb
    assert sv.select_observations(True, True) == a + 2 * S2.n_poses
    assert sv.select_observations(False, True) == b and sv.eval(X0)[0] == cb   # (no board-edge terms without points_on_line, :258)


def test_small_problems_planned_on_the_host_build_the_same_layouts_as_the_device_pipeline(sv, oracle_mod):
    """Reference-size problems (<= 512 x 22 records, p.z == 0) are uploaded by a host-planned path: the host derives the scan structure
    (from the stored scans' offsets and tag poses, or from the records), plans rows and points per lane, and enqueues the same build
    kernels without a read-back.  Against the generic device pipeline (hook: clc_debug_fast_small off), on the shapes of the reference's
    flows — every evaluation, closed form, analysis pass and solve must be BIT-IDENTICAL (same tables -> same layouts -> same sums):
    C1 (ragged scans, some empty), the offline shape (two points per scan: sparse rows), board-edge terms, one scan, adjacent poses
    with the SAME plane (merged into one scan by both), and what the path must refuse (z != 0, too many records) running generically."""
    import copy
    sv.set_launch(0, -1)
    x_near = oracle_mod.pose_plus(sd.pose7_from_T(sd.tlc_to_tcl(sd.GT_RLC, sd.GT_TLC)), np.array([0.03, -0.02, 0.02, 0.03, -0.04, 0.02]))

    def everything():
        out = [sv.path_info().single_resident, sv.path_info().rows_layout, sv.path_info().n_rows, sv.num_observations]
        for x in (X0, x_near):
            c, g, H = sv.eval(x)
            out += [c, g.tobytes(), H.tobytes()]
            c2, _, _ = sv.eval(x, with_loss=False)
            out.append(c2)
        T, un, sv9 = sv.closed_form()
        out += [T.tobytes(), un, sv9.tobytes()]
        r = sv.solve(x_near)
        out += [r.pose.tobytes(), r.summary.final_cost, r.summary.num_iterations, r.summary.termination, len(r.trace)]
        Hm, b, chi2, svals, V, nn = sv.information(r.pose)
        out += [Hm.tobytes(), b.tobytes(), chi2, nn]
        return out

    S_c1 = sd.GenerateSimData(0, n_poses=50, noise_sigma=0.01)
    S_edges = sd.sim_board_edges(5, 40, 60, noise_sigma=0.002)
    S_edges.ptl = S_edges.ptl[::-1].copy()
    S_same = sd.sim_fixed_count(4, 6, 50, noise_sigma=0.01)      # poses 2 and 3 get the same tag pose: one scan of 100 records
    S_same = copy.deepcopy(S_same); S_same.tag_q[3] = S_same.tag_q[2]; S_same.tag_t[3] = S_same.tag_t[2]
    S_one = sd.sim_fixed_count(6, 1, 300, noise_sigma=0.01)
    S_line = clc.points_on_fitted_lines(sd.GenerateSimData(2, n_poses=100, noise_sigma=0.01), sv) if hasattr(clc, "points_on_fitted_lines") else None
    if S_line is None:
        from camlasercalibratool_amd import calib
        S_line = calib.points_on_fitted_lines(sd.GenerateSimData(2, n_poses=100, noise_sigma=0.01), sv)
    cases = [(S_c1, [(False, False), (True, False)]), (S_edges, [(True, False), (True, True), (False, False)]), (S_same, [(False, False)]),
             (S_one, [(False, False)]), (S_line, [(True, False)])]
    n0 = sv.debug_fast_small(True)
    taken = 0
    for S, sels in cases:
        for lf, bd in sels:
            res = {}
            for fast in (True, False):
                sv.debug_fast_small(fast)
                sv.store_observations(S)
                n = sv.select_observations(lf, bd)
                res[("select", fast)] = everything()
                sv.upload(clc.flatten_observations(S, lf, bd))   # the same records through clc_upload
                res[("upload", fast)] = everything()
                assert n == sv.num_observations
            taken += 2
            assert res[("select", True)] == res[("select", False)], (lf, bd)
            assert res[("upload", True)] == res[("upload", False)], (lf, bd)
            assert res[("select", True)] == res[("upload", True)], (lf, bd)
    assert sv.debug_fast_small(True) - n0 == taken   # every fast-mode upload above really took the host-planned path
    # refused: points off the lidar plane, and more records than one workgroup holds -> the generic pipeline, same answers as ever
    n1 = sv.debug_fast_small()
    Sz = copy.deepcopy(S_c1); Sz.pts[::7, 2] = 0.01
    sv.store_observations(Sz)
    sv.select_observations(False, False)
    assert sv.path_info().rows_layout == 2
    big = clc.flatten_observations(sd.sim_fixed_count(8, 30, 500, noise_sigma=0.01), False)
    sv.upload(big)
    assert sv.path_info().single_resident == 0 and sv.debug_fast_small() == n1
