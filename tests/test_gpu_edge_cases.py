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
    res = sv.solve(x0)
    ref = oracle_mod.solve(rec, x0, linear_solver="qr")
    assert res.summary.termination == ref.summary.termination
    assert abs(res.summary.final_cost - ref.summary.final_cost) <= 1e-8
    assert res.summary.num_iterations == ref.summary.num_iterations


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
    assert L.clc_set_launch(sv._h, -1, 0) == -1 and L.clc_set_launch(sv._h, 0, 1024) == -1
    assert b"clc_set_launch" in L.clc_last_error()
    o = clc.default_options()
    o.max_num_iterations = -1
    with pytest.raises(clc.ClcError):
        sv.solve(X0, o)
