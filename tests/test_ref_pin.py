"""Parity pin against the REFERENCE'S OWN code.

tests/golden/ref_vectors.json holds outputs of the reference's translation units
(src/LaseCamCalCeres.cpp, src/pose_local_parameterization.cpp, src/utilities.cpp) compiled from
/root/reference against the stand-in Eigen/Ceres/sensor_msgs headers of oracle/ref_shim/ and run in
the build container (tests/golden/make_ref_golden.py).  Three layers:

  * oracle  vs the committed vectors     (CPU, always)           — pins the checker;
  * oracle  vs libref.so live            (CPU, when oracle/_ref/libref.so exists or can be built:
                                          fresh random inputs, not just the frozen ones);
  * HIP path vs the committed vectors    (-m gpu, through the C-ABI) — no oracle in the loop.

What this pins: the factor + Jacobian, Plus, pi_from_ppp, TranScanToPoints (exact code paths), the
closed form and the assembly loops of CamLaserCalibration / LineFittingCeres (scales, loss scales,
boundary planes, which points are used).  What it does not: Ceres' own minimiser — absent from this
image; the stand-in ceres::Solve runs the oracle's restatement of it.
"""
import json
import os

import numpy as np
import pytest

import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd

HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, "golden", "ref_vectors.json")))


def _f(v):
    return float(v) if isinstance(v, str) else v


def _sim(case):
    if case["generator"] == "GenerateSimData":
        return sd.GenerateSimData(case["seed"], noise_sigma=case["noise"])
    if case["generator"] == "sim_board_edges":
        return sd.sim_board_edges(case["seed"], 40, 30, noise_sigma=case["noise"])
    return sd.sim_fixed_count(case["seed"], 64, 200, noise_sigma=case["noise"])


def _cal_id(c):
    return f"{c['generator']}{c['seed']}_lf{int(c['linefit'])}_bd{int(c['boundary'])}_{c['init']}"


# ------------------------------------------------------------------------------------------
# oracle vs the committed reference vectors
# ------------------------------------------------------------------------------------------
def test_oracle_factor_matches_reference_code(oracle_mod):
    for c in G["factor"]:
        r, j = oracle_mod.factor_evaluate(c["plane"], c["point"], c["scale"], c["pose"])
        assert r == pytest.approx(c["residual"], rel=1e-14, abs=1e-15)
        assert np.allclose(j, c["jacobian"], rtol=1e-14, atol=1e-15)
        assert j[6] == 0.0 and c["jacobian"][6] == 0.0  # :60


def test_oracle_plus_matches_reference_code(oracle_mod):
    for c in G["plus"]:
        assert np.allclose(oracle_mod.pose_plus(c["x"], c["delta"]), c["out"], rtol=1e-15, atol=1e-16)
    assert np.array_equal(oracle_mod.pose_plus_jacobian(np.r_[np.zeros(6), 1.0]), np.array(G["plus_jacobian"]))
    assert G["pose_sizes"] == [7, 6]


def test_oracle_pi_from_ppp_and_scan_to_points_match_reference_code(oracle_mod):
    for c in G["pi_from_ppp"]:
        assert np.allclose(oracle_mod.pi_from_ppp(c["x1"], c["x2"], c["x3"]), c["pi"], rtol=1e-15, atol=1e-16)
    for c in G["scan_to_points"]:
        r = np.array([_f(v) for v in c["ranges"]], dtype=np.float32)
        P = oracle_mod.scan_to_points(r, c["angle_min"], c["angle_increment"], c["range_min"])
        assert np.array_equal(P, np.array([[_f(v) for v in row] for row in c["points"]]))


@pytest.mark.parametrize("case", G["closed_solution"], ids=lambda c: f"seed{c['seed']}")
def test_oracle_closed_form_matches_reference_function(oracle_mod, case):
    S = _sim(case)
    T, _, _ = oracle_mod.closed_form(oracle_mod.flatten(S, True, False))
    assert np.abs(T - np.array(case["Tlc"])).max() < 1e-11  # different SVD / LDLT algorithms behind the same formulas


@pytest.mark.parametrize("case", G["closed_solution_degenerate"], ids=lambda c: c["kind"])
def test_oracle_closed_form_on_unobservable_input_behaves_like_the_reference(oracle_mod, case):
    """src/LaseCamCalCeres.cpp:164-200 on a rank-deficient normal matrix: the reference prints its "system unobservable"
    notice and still returns LDLT + SVD output.  The oracle flags the same, returns a finite Tlc with an orthogonal
    rotation block; what the data determine (the translation h3-part and the rows of R not completed by the SVD of a
    rank-deficient matrix) agrees with the reference's output."""
    S = sd.sim_degenerate(case["kind"])
    T, unobservable, sv9 = oracle_mod.closed_form(oracle_mod.flatten(S, True, False))
    Tref = np.array([[_f(v) for v in row] for row in case["Tlc"]])
    assert case["notice_printed"] and unobservable and (sv9 < 1e-10).any()
    assert np.isfinite(T).all() and np.isfinite(Tref).all()
    for M in (T, Tref):
        assert np.abs(M[:3, :3] @ M[:3, :3].T - np.eye(3)).max() < 1e-12
    assert np.abs(T[:, 3] - Tref[:, 3]).max() < 1e-9
    assert np.abs(np.abs(T[:3, :3]) - np.abs(Tref[:3, :3])).max() < 1e-9  # completed columns may differ in sign


@pytest.mark.parametrize("case", G["calibration"], ids=_cal_id)
def test_oracle_solve_matches_reference_function(oracle_mod, case):
    """CamLaserCalibration as the reference wrote it (assembly loop, factor, loss objects, Plus) vs
    oracle.flatten + oracle.solve: same residual count, same LM trajectory, same Tcl."""
    S = _sim(case)
    rec = oracle_mod.flatten(S, case["linefit"], case["boundary"])
    assert rec.shape[0] == case["n_blocks"]
    r = oracle_mod.solve(rec, sd.pose7_from_T(np.array(case["Tcl0"])), linear_solver="qr")
    assert r.summary.num_iterations == case["num_iterations"] and r.summary.termination == case["termination"]
    assert r.summary.num_successful_steps == case["num_successful_steps"]
    assert r.summary.initial_cost == pytest.approx(case["initial_cost"], rel=1e-12)
    assert r.summary.final_cost == pytest.approx(case["final_cost"], rel=1e-10, abs=1e-18)
    assert np.abs(sd.T_from_pose7(r.pose) - np.array(case["Tcl"])).max() < 1e-12
    # analysis pass (:316-381: no loss, no board-edge terms); the reference only prints it, with 6 significant digits
    rec_pts = oracle_mod.flatten(S, case["linefit"], False)
    H, b, chi2, sv6, V, n_null = oracle_mod.information(rec_pts, r.pose)
    assert np.allclose(sv6, case["analysis_singular_values"], rtol=2e-5)
    assert chi2 / 2.0 == pytest.approx(case["analysis_chi2_half"], rel=2e-5, abs=1e-12)


def test_oracle_line_fit_matches_reference_function(oracle_mod):
    for c in G["line_fitting"]:
        P = np.array(c["points"])
        r = oracle_mod.line_fit(P[:, :2], c["line0"])
        assert r.summary.num_iterations == c["num_iterations"] and r.summary.termination == c["termination"]
        assert np.abs(r.pose - np.array(c["line"])).max() < 1e-13
        assert r.summary.final_cost == pytest.approx(c["final_cost"], rel=1e-11)


@pytest.mark.parametrize("case", G["simulation"], ids=lambda c: f"seed{c['seed']}")
def test_sim_generator_and_whole_reference_program(oracle_mod, case):
    """The reference's GenerateSimData on its own (pinned) tag poses vs simdata.points_from_tag_poses, and the reference's
    whole simulation node — its main(): generate, CamLaserCalibration from the identity, print Tlc — vs oracle.solve on
    the same observations (the program prints with 6 significant digits)."""
    q, t = np.array(case["tag_q_wxyz"]), np.array(case["tag_t"])
    S = sd.points_from_tag_poses(sd.quat_wxyz_to_rot(q), t)
    assert np.array_equal(np.diff(S.pts_off), np.array(case["counts"]))
    sums = np.array([S.pts[S.pts_off[i]:S.pts_off[i + 1]].sum(axis=0) for i in range(50)])
    assert np.abs(sums - np.array(case["point_sums"])).max() < 1e-10
    assert np.abs(S.pts[0] - np.array(case["first_point"])).max() < 1e-12 and np.abs(S.pts[-1] - np.array(case["last_point"])).max() < 1e-12
    r = oracle_mod.solve(oracle_mod.flatten(S, False, False), sd.pose7_from_T(np.eye(4)))
    Tlc = np.linalg.inv(sd.T_from_pose7(r.pose))
    P = np.array(case["program_Tlc_printed"])
    assert np.abs(Tlc - P).max() < 1e-5 * max(1.0, np.abs(P).max())
    assert np.abs(P[:3, :3] - sd.GT_RLC).max() < 1e-5 and np.abs(P[:3, 3] - sd.GT_TLC).max() < 1e-5  # it finds the ground truth


# ------------------------------------------------------------------------------------------
# oracle vs the reference library, live (fresh inputs)
# ------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ref_mod():
    import oracle.ref as ref

    if not ref.available():
        pytest.skip("oracle/_ref/libref.so not present and /root/reference not available to build it")
    return ref


def test_live_reference_elementwise(oracle_mod, ref_mod):
    rng = np.random.default_rng(int.from_bytes(os.urandom(4), "little"))
    for _ in range(300):
        n = rng.normal(size=3)
        plane = np.r_[n / np.linalg.norm(n) * rng.uniform(0.2, 2.0), rng.normal()]
        pt = rng.normal(size=3) * 4
        s = rng.uniform(0.02, 1.0)
        q = rng.normal(size=4)
        pose = np.r_[rng.normal(size=3), q / np.linalg.norm(q)]
        r0, j0 = oracle_mod.factor_evaluate(plane, pt, s, pose)
        r1, j1 = ref_mod.factor_evaluate(plane, pt, s, pose)
        assert abs(r0 - r1) <= 1e-14 * max(1.0, abs(r1)) and np.allclose(j0, j1, rtol=1e-14, atol=1e-15)
        d = rng.normal(size=6) * 10.0 ** rng.uniform(-8, 0)
        assert np.allclose(oracle_mod.pose_plus(pose, d), ref_mod.pose_plus(pose, d), rtol=1e-15, atol=1e-16)
        a, b, c = rng.normal(size=(3, 3))
        assert np.allclose(oracle_mod.pi_from_ppp(a, b, c), ref_mod.pi_from_ppp(a, b, c), rtol=1e-15, atol=1e-16)


def test_live_reference_whole_functions(oracle_mod, ref_mod):
    seed = int.from_bytes(os.urandom(2), "little")
    S = sd.GenerateSimData(seed, noise_sigma=0.01)
    rec = oracle_mod.flatten(S, False, False)
    r = oracle_mod.solve(rec, sd.pose7_from_T(np.eye(4)))
    Tcl, record, nb = ref_mod.calibration(S, np.eye(4), False, False)
    assert nb == rec.shape[0] and record["num_iterations"] == r.summary.num_iterations, f"seed {seed}"
    assert np.abs(Tcl - sd.T_from_pose7(r.pose)).max() < 1e-12, f"seed {seed}"
    T, _, _ = oracle_mod.closed_form(oracle_mod.flatten(S, True, False))
    assert np.abs(T - ref_mod.closed_solution(S)).max() < 1e-10, f"seed {seed}"
    # boundary mode on the raw simulation: a pose with an empty scan makes the reference throw
    # std::out_of_range at obi.points.at(0) (:278); the oracle must fail on exactly the same inputs
    def raises(fn):
        try:
            fn()
        except IndexError:
            return True
        return False
    assert raises(lambda: ref_mod.calibration(S, np.eye(4), True, True)) == raises(lambda: oracle_mod.flatten(S, True, True)), f"seed {seed}"
    S0 = sd.GenerateSimData(0, noise_sigma=0.0)  # known to contain empty scans
    assert raises(lambda: ref_mod.calibration(S0, np.eye(4), True, True)) and raises(lambda: oracle_mod.flatten(S0, True, True))
    Sb = sd.sim_board_edges(seed, 30, 40, noise_sigma=0.002)
    x0 = oracle_mod.pose_plus(sd.pose7_from_T(sd.tlc_to_tcl(sd.GT_RLC, sd.GT_TLC)), np.array([0.03, -0.02, 0.03, 0.04, -0.03, 0.02]))
    recb = oracle_mod.flatten(Sb, True, True)
    rb = oracle_mod.solve(recb, x0)
    Tb, recordb, nbb = ref_mod.calibration(Sb, sd.T_from_pose7(x0), True, True)
    assert nbb == recb.shape[0] and recordb["num_iterations"] == rb.summary.num_iterations, f"seed {seed}"
    assert np.abs(Tb - sd.T_from_pose7(rb.pose)).max() < 1e-12, f"seed {seed}"


def test_live_reference_simulation_program(oracle_mod, ref_mod):
    seed = int.from_bytes(os.urandom(3), "little") + 1
    q, t, cnt, pts = ref_mod.generate_sim_data(seed)
    S = sd.points_from_tag_poses(sd.quat_wxyz_to_rot(q), t)
    assert np.array_equal(np.diff(S.pts_off), cnt), f"seed {seed}"
    assert np.abs(S.pts - pts).max() < 1e-12, f"seed {seed}"
    P = ref_mod.parse_simulation_tlc(ref_mod.simulation_program(seed))
    r = oracle_mod.solve(oracle_mod.flatten(S, False, False), sd.pose7_from_T(np.eye(4)))
    assert np.abs(np.linalg.inv(sd.T_from_pose7(r.pose)) - P).max() < 1e-5, f"seed {seed}"


# ------------------------------------------------------------------------------------------
# HIP path vs the committed reference vectors (through the C-ABI; no oracle involved)
# ------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def sv():
    s = clc.Solver(0)
    yield s
    s.close()


@pytest.mark.gpu
def test_gpu_factor_and_plus_match_reference_code(sv):
    for c in G["factor"]:
        rec = np.array([c["plane"] + c["point"] + [c["scale"]]])
        sv.upload(rec)
        r, j = sv.factor_evaluate(np.array(c["pose"]))
        assert abs(r[0] - c["residual"]) <= 4e-15 * max(1.0, abs(c["residual"]))
        assert np.allclose(j[0], c["jacobian"], rtol=1e-13, atol=4e-15)
    x = np.array([c["x"] for c in G["plus"]])
    d = np.array([c["delta"] for c in G["plus"]])
    assert np.abs(sv.pose_plus(x, d) - np.array([c["out"] for c in G["plus"]])).max() < 4e-16


@pytest.mark.gpu
@pytest.mark.parametrize("case", G["calibration"], ids=_cal_id)
def test_gpu_calibration_matches_reference_function(sv, case):
    """clc.CamLaserCalibration (the drop-in mirror: host flatten + clc_upload + clc_solve) against the
    reference's CamLaserCalibration run on the same observations.  Bars of BASELINE.json."""
    S = _sim(case)
    Tcl = np.array(case["Tcl0"])
    rep = clc.CamLaserCalibration(S, Tcl, case["linefit"], case["boundary"], solver=sv, verbose=False)
    assert np.abs(Tcl - np.array(case["Tcl"])).max() <= 1e-6
    assert abs(rep.result.summary.final_cost - case["final_cost"]) <= 1e-8
    assert rep.result.summary.num_iterations == case["num_iterations"]
    assert np.abs(Tcl - np.array(case["Tcl"])).max() <= 1e-10  # what is actually achieved
    assert np.allclose(rep.singular_values, case["analysis_singular_values"], rtol=2e-5)  # printed with 6 digits
    assert rep.chi2 / 2.0 == pytest.approx(case["analysis_chi2_half"], rel=2e-5, abs=1e-12)


@pytest.mark.gpu
def test_gpu_closed_form_line_fit_scan_match_reference_functions(sv):
    for c in G["closed_solution"]:
        Tlc = np.eye(4)
        clc.CamLaserCalClosedSolution(_sim(c), Tlc, solver=sv, verbose=False)
        assert np.abs(Tlc - np.array(c["Tlc"])).max() < 1e-10  # same factorisations as the reference now (pivoted LDL^T, SVD)
    pts = [np.array(c["points"]) for c in G["line_fitting"]]
    off = np.zeros(len(pts) + 1, dtype=np.int64)
    off[1:] = np.cumsum([p.shape[0] for p in pts])
    lines, sm = sv.line_fit_batched(np.concatenate(pts)[:, :2], off, np.array([c["line0"] for c in G["line_fitting"]]))
    for k, c in enumerate(G["line_fitting"]):
        assert np.abs(lines[k] - np.array(c["line"])).max() < 1e-12 and sm[k].num_iterations == c["num_iterations"]
    for c in G["scan_to_points"]:
        r = np.array([_f(v) for v in c["ranges"]], dtype=np.float32)
        P = sv.scan_to_points(r, np.array([0, r.shape[0]]), c["angle_min"], c["angle_increment"], c["range_min"])
        assert np.abs(P - np.array([[_f(v) for v in row] for row in c["points"]])).max() < 2e-14  # device cos/sin


@pytest.mark.gpu
@pytest.mark.parametrize("case", G["closed_solution_degenerate"], ids=lambda c: c["kind"])
def test_gpu_closed_form_on_unobservable_input(sv, oracle_mod, case, capsys):
    """The HIP path on the same inputs: notice printed, a finite Tlc written (not an error return), equal to the oracle's
    (same pivoted LDLT / SVD completion rules) and to what the reference's output determines."""
    S = sd.sim_degenerate(case["kind"])
    Tlc = np.full((4, 4), 7.0)
    unobservable, sv9 = clc.CamLaserCalClosedSolution(S, Tlc, solver=sv, verbose=True)
    out = capsys.readouterr().out
    assert unobservable and "Notice Notice Notice: system unobservable" in out and "Closed-form solution Tlc" in out
    To, uo, svo = oracle_mod.closed_form(oracle_mod.flatten(S, True, False))
    assert uo and np.isfinite(Tlc).all()
    assert np.abs(Tlc - To).max() < 1e-9
    assert np.allclose(sv9, svo, rtol=1e-9, atol=1e-6 * svo[0])
    Tref = np.array([[_f(v) for v in row] for row in case["Tlc"]])
    assert np.abs(Tlc[:, 3] - Tref[:, 3]).max() < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("case", G["simulation"], ids=lambda c: f"seed{c['seed']}")
def test_gpu_dropin_matches_whole_reference_program(sv, case):
    """The reference's simulation node end to end (what its main() printed) vs the drop-in flow of
    main/calibr_simulation.cpp:126-133 on the HIP backend, same observations."""
    S = sd.points_from_tag_poses(sd.quat_wxyz_to_rot(np.array(case["tag_q_wxyz"])), np.array(case["tag_t"]))
    Tcl = np.linalg.inv(np.eye(4))
    clc.CamLaserCalibration(S, Tcl, False, solver=sv, verbose=False)
    P = np.array(case["program_Tlc_printed"])
    assert np.abs(np.linalg.inv(Tcl) - P).max() < 1e-5 * max(1.0, np.abs(P).max())
