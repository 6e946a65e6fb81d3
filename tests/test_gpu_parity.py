"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI,
against the CPU oracle on identical seeded inputs.

Tolerances (BASELINE.json north_star): recovered T_cl within 1e-6 max-abs on the 4x4, final
cost within 1e-8, both against the oracle's DENSE_QR Ceres restatement.  Reductions are
compared at 1e-11 relative (FP64 sums of up to 1e6 terms in a different association order)."""
import ctypes as C

import numpy as np
import pytest

import camlasercalibratool_amd as clc
from camlasercalibratool_amd import _capi, simdata as sd

pytestmark = pytest.mark.gpu

X0 = sd.pose7_from_T(np.eye(4))  # calibr_simulation.cpp:126-129
T_TOL = 1e-6
COST_TOL = 1e-8


@pytest.fixture(scope="module")
def sv():
    s = clc.Solver(0)
    yield s
    s.close()


def _rand_pose(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    return np.concatenate([rng.normal(size=3), q])


def _mirror(oracle_mod, o):
    oo = oracle_mod.default_options()
    for f, _ in oo._fields_:
        setattr(oo, f, getattr(o, f))
    return oo


def _dT(a, b):
    return np.abs(sd.T_from_pose7(a) - sd.T_from_pose7(b)).max()


# ---------------------------------------------------------------------------------------
def test_extension_loaded_is_the_in_tree_hip_library(sv):
    name, cus = sv.device_info()
    assert "gfx950" in name, name
    assert cus >= 64
    maps = open("/proc/self/maps").read()
    assert "camlasercalibratool_amd/csrc/libclc_hip" in maps  # (the hooks build under tests/conftest.py; the product build: test_gpu_product_library.py)


@pytest.mark.parametrize("mode", [0, 1])
def test_wave_reduction_exact_on_integers(sv, mode):
    """Butterfly (permlane32/16 swap + DPP) and shuffle reductions are exact on integer-valued
    data, whatever the association order."""
    rng = np.random.default_rng(0)
    lanes = rng.integers(-1000, 1000, size=(64, 28)).astype(np.float64)
    lanes[5, 3] = 2.0 ** 40  # asymmetric spike: catches a lane/column permutation
    out = sv.debug_wave_reduce(lanes, mode)
    assert np.array_equal(out, lanes.sum(axis=0))


def test_factor_evaluate_elementwise(sv, oracle_mod):
    """PointInPlaneFactor::Evaluate, record by record (LaseCamCalCeres.cpp:43-66)."""
    rng = np.random.default_rng(1)
    S = sd.GenerateSimData(2, n_poses=12, noise_sigma=0.01)
    rec = clc.flatten_observations(S, False)
    sv.upload(rec)
    for _ in range(3):
        pose = _rand_pose(rng)
        r, J = sv.factor_evaluate(pose)
        r0, J0 = oracle_mod.factor_evaluate_batch(rec, pose)
        assert np.abs(r - r0).max() <= 1e-14 * max(1.0, np.abs(r0).max())
        assert np.abs(J - J0).max() <= 1e-14 * max(1.0, np.abs(J0).max())
        assert np.all(J[:, 6] == 0.0)
    r, J = sv.factor_evaluate(pose, want_jacobian=False)
    assert J is None and np.abs(r - r0).max() < 1e-13


def test_pose_plus_on_device(sv, oracle_mod):
    rng = np.random.default_rng(2)
    x = np.stack([_rand_pose(rng) for _ in range(300)])
    d = rng.normal(size=(300, 6)) * 0.3
    d[0] = 0.0
    out = sv.pose_plus(x, d)
    ref = np.stack([oracle_mod.pose_plus(x[i], d[i]) for i in range(300)])
    assert np.abs(out - ref).max() < 4e-16
    assert np.allclose(np.linalg.norm(out[:, 3:], axis=1), 1.0, atol=1e-15)


@pytest.mark.parametrize("n", [1, 2, 63, 127, 128, 129, 1000, 5317, 65536 + 77])
@pytest.mark.parametrize("with_loss", [True, False])
def test_eval_reduction_matches_oracle_ragged_sizes(sv, oracle_mod, n, with_loss):
    """cost, g, H of one pass for ragged N (partial tiles, fewer tiles than waves)."""
    rng = np.random.default_rng(n)
    S = sd.sim_fixed_count(n % 97, n_poses=max(1, (n + 499) // 500), pts_per_pose=500, noise_sigma=0.01)
    rec = clc.flatten_observations(S, False)[:n].copy()
    sv.upload(rec)
    assert sv.num_observations == n
    pose = sd.pose7_from_T(sd.tlc_to_tcl(sd.GT_RLC, sd.GT_TLC)) if n % 2 else X0
    pose = oracle_mod.pose_plus(pose, rng.normal(size=6) * 0.05)
    for flags in (0, 1, 2, 4, 6, 7, 16 | 2, 16 | 6, 16 | 7, 32 | 2, 32 | 6, 32 | 16 | 2, 32 | 16 | 7,
                  64 | 16, 64 | 32 | 16 | 4,  # reduction x prefetch x nt x compact x 512-thread WGs x deep pipeline
                  256, 256 | 1, 256 | 4, 256 | 32, 256 | 32 | 4, 256 | 32 | 512):  # row layout x nt x workgroup size x wave shares
        sv.set_launch(0, flags)
        c, g, H = sv.eval(pose, with_loss=with_loss)
        c0, g0, H0 = oracle_mod.evaluate_ne(rec, pose, with_loss=with_loss)
        assert abs(c - c0) <= 1e-11 * abs(c0) + 1e-300
        assert np.abs(g - g0).max() <= 1e-11 * np.abs(g0).max() + 1e-18
        assert np.abs(H - H0).max() <= 1e-11 * np.abs(H0).max()
    sv.set_launch(0, -1)  # library default
    c1, _, _ = sv.eval(pose, with_loss=with_loss, want_jacobian=False)  # cost-only variant
    assert abs(c1 - c0) <= 1e-11 * abs(c0) + 1e-300


@pytest.mark.parametrize("grid", [1, 2, 3, 7])
@pytest.mark.parametrize("n", [128 * 40, 128 * 97 + 5, 65536 + 77])
def test_deep_pipeline_long_tile_runs(sv, oracle_mod, n, grid):
    """flags bit 64: with few workgroups every wave walks many tiles, so the x6-unrolled
    rotation of the point / group-id / plane buffers wraps several times (and ends at every
    residue of the unroll)."""
    S = sd.sim_fixed_count(11, n_poses=(n + 499) // 500, pts_per_pose=500, noise_sigma=0.01)
    rec = clc.flatten_observations(S, False)[:n].copy()
    sv.upload(rec)
    pose = oracle_mod.pose_plus(X0, np.random.default_rng(grid).normal(size=6) * 0.05)
    c0, g0, H0 = oracle_mod.evaluate_ne(rec, pose, with_loss=True)
    for flags in (64 | 16, 64 | 32 | 16, 64 | 32 | 16 | 4):
        sv.set_launch(grid, flags)
        c, g, H = sv.eval(pose, with_loss=True)
        assert abs(c - c0) <= 1e-11 * abs(c0)
        assert np.abs(g - g0).max() <= 1e-11 * np.abs(g0).max()
        assert np.abs(H - H0).max() <= 1e-11 * np.abs(H0).max()
    sv.set_launch(0, -1)


def test_eval_is_bitwise_reproducible_and_grid_invariant_to_rounding(sv, oracle_mod):
    S = sd.sim_fixed_count(5, 200, 500, noise_sigma=0.01)
    rec = clc.flatten_observations(S, False)
    sv.upload(rec)
    a = sv.eval(X0)
    b = sv.eval(X0)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])  # fixed-shape reduction
    sv.set_launch(64, -1)
    c = sv.eval(X0)
    sv.set_launch(0, -1)
    assert abs(c[0] - a[0]) <= 1e-12 * abs(a[0]) and np.allclose(c[2], a[2], rtol=1e-12)


def _layout(sv):
    c, g, bc, bg = C.c_int(), C.c_longlong(), C.c_int(), C.c_longlong()
    _capi.check(_capi.lib().clc_debug_layout(sv._h, C.byref(c), C.byref(g), C.byref(bc), C.byref(bg)), "clc_debug_layout")
    return c.value, g.value, bc.value, bg.value


def test_compact_layout_is_lossless_and_optional(sv, oracle_mod):
    """The compact (28 B/obs) layout is a lossless re-encoding: same operands, same order ->
    bitwise the same reduction as the 64-byte tiles.  Arrays without scan structure (every record
    its own plane) do not compress and silently use the 64-byte path."""
    S = sd.sim_fixed_count(3, 300, 137, noise_sigma=0.01)  # 137 pts/scan: groups straddle tiles
    rec = clc.flatten_observations(S, False)
    sv.upload(rec)
    compact, n_groups, _, _ = _layout(sv)
    assert compact == 1 and n_groups == 300
    pose = oracle_mod.pose_plus(X0, np.array([0.1, -0.2, 0.05, 0.2, -0.1, 0.3]))
    sv.set_launch(0, 6)
    a = sv.eval(pose)
    sv.set_launch(0, 6 | 16)
    b = sv.eval(pose)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    ra = sv.solve(X0)
    sv.set_launch(0, 6)
    rb = sv.solve(X0)
    assert np.array_equal(ra.pose, rb.pose) and ra.summary.final_cost == rb.summary.final_cost
    # same for the 512-thread workgroups with the weighted tile map (their own fixed order)
    sv.set_launch(0, 32 | 6)
    c = sv.eval(pose)
    sv.set_launch(0, 32 | 16 | 2)
    d = sv.eval(pose)
    e = sv.eval(pose)
    sv.set_launch(0, -1)
    assert c[0] == d[0] and np.array_equal(c[2], d[2]) and d[0] == e[0] and np.array_equal(d[1], e[1])
    assert abs(c[0] - a[0]) <= 1e-12 * abs(a[0]) and np.allclose(c[2], a[2], rtol=1e-12)
    # boundary terms: two extra single-record groups per scan (src/LaseCamCalCeres.cpp:281-288)
    Sb = sd.sim_board_edges(5, n_poses=50, pts_per_pose=30)
    recb = clc.flatten_observations(Sb, True, True)
    sv.upload(recb)
    assert _layout(sv)[:2] == (1, 150)
    c0, g0, H0 = oracle_mod.evaluate_ne(recb, pose)
    c1, g1, H1 = sv.eval(pose)
    assert abs(c1 - c0) <= 1e-11 * abs(c0) and np.abs(H1 - H0).max() <= 1e-11 * np.abs(H0).max()
    # incompressible: every record its own plane
    rng = np.random.default_rng(0)
    junk = rec[:5000].copy()
    junk[:, 3] += rng.normal(size=5000) * 1e-3
    sv.upload(junk)
    assert _layout(sv)[0] == 0
    c0, g0, H0 = oracle_mod.evaluate_ne(junk, pose)
    c1, g1, H1 = sv.eval(pose)
    assert abs(c1 - c0) <= 1e-11 * abs(c0) and np.abs(H1 - H0).max() <= 1e-11 * np.abs(H0).max()


def test_row_layout_matches_oracle_and_the_per_point_paths(sv, oracle_mod):
    """The row layout (every scan padded to rows of 64 points, per-scan moments expanded once per scan segment) against
    the oracle and the per-point layouts: ragged C1 scans (0..180 points: partial rows, empty poses), C5 board-edge terms
    (single-record scans with un-normalised planes), a truncated array, loss on and off.  (Records with p.z != 0 get rows
    that carry z: tests/test_gpu_rows_z.py.)"""
    rng = np.random.default_rng(5)
    gt = sd.pose7_from_T(sd.tlc_to_tcl(sd.GT_RLC, sd.GT_TLC))
    cases = []
    S1 = sd.GenerateSimData(3, n_poses=50, noise_sigma=0.02)
    cases.append(clc.flatten_observations(S1, False))
    S5 = sd.sim_board_edges(5, n_poses=300, pts_per_pose=137, noise_sigma=0.002)
    cases.append(clc.flatten_observations(S5, True, True))
    cases.append(clc.flatten_observations(sd.sim_fixed_count(8, 120, 500, 0.01), False)[:57311].copy())
    for rec in cases:
        sv.upload(rec)
        ok, n_rows, _, _ = sv.debug_rows()
        assert ok and n_rows >= (rec.shape[0] + 63) // 64, (rec.shape, n_rows)
        for pose in (X0, oracle_mod.pose_plus(gt, rng.normal(size=6) * 0.03)):
            for with_loss in (True, False):
                c0, g0, H0 = oracle_mod.evaluate_ne(rec, pose, with_loss=with_loss)
                sv.set_launch(0, 2 | 16 | 32)
                c2, g2, H2 = sv.eval(pose, with_loss=with_loss)
                for grid, flags in ((0, -1), (0, 256 | 32 | 512), (5, 256 | 32), (0, 256), (3, 256 | 4)):
                    sv.set_launch(grid, flags)
                    c, g, H = sv.eval(pose, with_loss=with_loss)
                    for (a, b) in ((c, c0), (c, c2)):
                        assert abs(a - b) <= 1e-11 * abs(b)
                    for (a, b) in ((H, H0), (H, H2), (g, g0), (g, g2)):
                        assert np.abs(a - b).max() <= 1e-11 * np.abs(H0).max() if a is H else np.abs(a - b).max() <= 1e-10 * np.abs(g0).max() + 1e-16
                    assert sv.eval(pose, with_loss=with_loss)[0] == c  # fixed lane -> row map: bitwise repeatable
        sv.set_launch(0, -1)
    # z != 0: the rows carry z (tests/test_gpu_rows_z.py); a small array still runs on the compact layout by default
    rec = cases[0].copy()
    rec[::7, 6] = 0.01
    sv.upload(rec)
    assert sv.debug_rows()[0] and sv.rows_carry_z()[0] and _layout(sv)[0] == 1
    c0, g0, H0 = oracle_mod.evaluate_ne(rec, X0)
    c, g, H = sv.eval(X0)
    assert abs(c - c0) <= 1e-11 * abs(c0) and np.abs(H - H0).max() <= 1e-11 * np.abs(H0).max()
    res = sv.solve(X0)
    ref = oracle_mod.solve(rec, X0, linear_solver="qr")
    assert _dT(res.pose, ref.pose) <= T_TOL and res.summary.num_iterations == ref.summary.num_iterations


@pytest.mark.parametrize("n_poses,pts", [(2000, 500), (300, 130), (97, 333), (50, 30)])
def test_wave_split_table_moves_boundaries_to_scan_starts_only(sv, n_poses, pts):
    """Equal-shares mode of the row layout (flag 512, the default): wave w streams rows [split[w], split[w+1]).  The
    table must be a partition of [0, n_rows) in order, every boundary either where the arithmetic split puts it or on a
    scan start at most half a share (rounded up) away; with 500-point scans (8 rows) and 2 048 waves — C2 — every wave
    owns whole scans."""
    S = sd.sim_fixed_count(3, n_poses, pts, noise_sigma=0.01)
    rec = clc.flatten_observations(S, False)
    sv.upload(rec)
    ok, n_rows = sv.debug_rows()[:2]
    assert ok
    for grid in (1, 3, 64, 256, 1000):
        g = min(grid, (rec.shape[0] + 127) // 128)  # never more workgroups than 128-record tiles (eval_grid)
        split, first = sv.debug_wave_split(g)
        assert split[0] == 0 and split[-1] == n_rows and np.all(np.diff(split) >= 0), g
        q, r = divmod(n_rows, g)
        window = (q // 8 + 1) // 2
        b = np.repeat(np.arange(g), 8)
        w = np.tile(np.arange(8), g)
        wgn = q + (b < r)
        nominal = b * q + np.minimum(b, r) + (wgn * w) // 8
        s = split[:-1].astype(np.int64)
        moved = s != nominal
        assert np.all(np.abs(s[moved] - nominal[moved]) <= window) and np.all(first[s[moved]] == 1), g
        # kept in place although not on a scan start: there is none within the window
        for k in np.nonzero(~moved & (nominal > 0) & (nominal < n_rows))[0]:
            if first[nominal[k]] != 1:
                lo, hi = max(1, nominal[k] - window), min(n_rows - 1, nominal[k] + window)
                assert not first[lo:hi + 1].any(), (g, k)
        if (n_poses, pts, g) == (2000, 500, 256):
            inner = split[1:-1]
            assert np.all(first[inner[inner < n_rows]] == 1)  # whole scans only


@pytest.mark.parametrize("grid", [1, 2, 3, 7])
@pytest.mark.parametrize("n_poses,pts", [(40, 500), (97, 333), (260, 65)])
def test_row_layout_long_runs_per_wave(sv, oracle_mod, grid, n_poses, pts):
    """Few workgroups: every wave walks many rows, so the 8-deep rotation of the row buffers wraps several times and ends
    at every residue of the unroll, with scan changes at every position of it."""
    S = sd.sim_fixed_count(17, n_poses, pts, noise_sigma=0.01)
    rec = clc.flatten_observations(S, False)
    sv.upload(rec)
    pose = oracle_mod.pose_plus(X0, np.random.default_rng(grid).normal(size=6) * 0.05)
    c0, g0, H0 = oracle_mod.evaluate_ne(rec, pose)
    for flags in (256, 256 | 32, 256 | 32 | 4, 256 | 32 | 512):
        sv.set_launch(grid, flags)
        c, g, H = sv.eval(pose)
        assert abs(c - c0) <= 1e-11 * abs(c0)
        assert np.abs(g - g0).max() <= 1e-11 * np.abs(g0).max() and np.abs(H - H0).max() <= 1e-11 * np.abs(H0).max()
    sv.set_launch(0, -1)


def test_eval_empty_and_errors(sv):
    with pytest.raises(clc.ClcError) as e:
        sv.eval(np.array([0, 0, 0, 0, 0, 0, np.nan]))
    assert e.value.code == -3
    sv.upload(np.zeros((0, 8)))
    c, g, H = sv.eval(X0)
    assert c == 0.0 and not g.any() and not H.any()
    fresh = clc.Solver(0)
    with pytest.raises(clc.ClcError) as e:
        fresh.eval(X0)
    assert e.value.code == -5
    fresh.close()


# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(4))
def test_c1_noise_free_sim_recovers_ground_truth(sv, oracle_mod, seed):
    """configs[0]: simulation_lasercamcal_node default, Tcl = I start, (linefit=false)."""
    S = sd.GenerateSimData(seed)
    rec = clc.flatten_observations(S, False)
    sv.upload(rec)
    res = sv.solve(X0)
    ref = oracle_mod.solve(rec, X0, linear_solver="qr")
    Tlc = np.linalg.inv(sd.T_from_pose7(res.pose))
    assert np.abs(Tlc[:3, :3] - sd.GT_RLC).max() < 1e-7 and np.abs(Tlc[:3, 3] - sd.GT_TLC).max() < 1e-7
    assert _dT(res.pose, ref.pose) <= T_TOL
    assert abs(res.summary.final_cost - ref.summary.final_cost) <= COST_TOL
    assert res.summary.termination in (1, 2, 3)


@pytest.mark.parametrize("noise,seed", [(0.01, 1), (0.01, 7), (0.03, 2), (0.03, 9)])
def test_c1_noisy_solve_matches_oracle_trace(sv, oracle_mod, noise, seed):
    S = sd.GenerateSimData(seed, noise_sigma=noise)
    rec = clc.flatten_observations(S, False)
    sv.upload(rec)
    res = sv.solve(X0)
    ref = oracle_mod.solve(rec, X0, linear_solver="qr")
    assert _dT(res.pose, ref.pose) <= T_TOL
    assert abs(res.summary.final_cost - ref.summary.final_cost) <= COST_TOL
    assert res.summary.termination == ref.summary.termination
    assert res.summary.num_iterations == ref.summary.num_iterations
    assert len(res.trace) == len(ref.trace)
    for a, b in zip(res.trace, ref.trace):
        assert a.iteration == b.iteration and a.step_is_successful == b.step_is_successful
        assert a.cost == pytest.approx(b.cost, rel=1e-9, abs=1e-14)
        assert a.trust_region_radius == pytest.approx(b.trust_region_radius, rel=1e-7)
    # in/out semantics: restarting from the answer stays there
    again = sv.solve(res.pose)
    assert again.summary.num_iterations <= 2 and _dT(again.pose, res.pose) < 1e-5


def test_solver_options_are_honoured(sv, oracle_mod):
    S = sd.GenerateSimData(3, noise_sigma=0.02)
    rec = clc.flatten_observations(S, False)
    sv.upload(rec)
    for kw in [dict(max_num_iterations=3), dict(use_loss=0), dict(jacobi_scaling=0),
               dict(initial_trust_region_radius=1e12, max_num_iterations=40), dict(launch_ahead=1),
               dict(launch_ahead=7), dict(function_tolerance=1e-12, parameter_tolerance=1e-12)]:
        o = clc.default_options()
        for k, v in kw.items():
            setattr(o, k, v)
        res = sv.solve(X0, o)
        ref = oracle_mod.solve(rec, X0, options=_mirror(oracle_mod, o), linear_solver="qr")
        assert res.summary.termination == ref.summary.termination, kw
        assert res.summary.num_iterations == ref.summary.num_iterations, kw
        assert _dT(res.pose, ref.pose) <= T_TOL and abs(res.summary.final_cost - ref.summary.final_cost) <= COST_TOL, kw
    o = clc.default_options()
    o.profile_events = 1
    res = sv.solve(X0, o)
    assert res.summary.eval_kernel_launches == res.summary.num_evaluations and res.summary.eval_kernel_ms > 0


@pytest.mark.parametrize("base", [2 | 16 | 32, 2 | 32 | 256], ids=["compact", "rows"])
@pytest.mark.parametrize("grid", [0, 1, 7, 256, 300, 1000, 2048])
@pytest.mark.parametrize("n_poses,noise", [(400, 0.01), (3, 0.03)])
def test_step_kernel_solve_matches_two_kernel_path(sv, oracle_mod, grid, n_poses, noise, base):
    """flags bit 128 (the default): one step_kernel launch per LM iteration, every workgroup running the controller
    redundantly on the previous launch's rows.  Same grid => bit-identical to the [eval_kernel, lm_kernel] path,
    including the per-iteration trace, for any number of workgroups (fewer/more than CUs, more than one round of
    256 rows), repeated back to back (state / row double buffers, launches queued ahead of a finished solve); for the
    compact and for the row layout."""
    S = sd.sim_fixed_count(77, n_poses, 500, noise_sigma=noise)
    rec = clc.flatten_observations(S, False)
    sv.upload(rec)
    ref = oracle_mod.solve(rec, X0, linear_solver="qr")
    sv.set_launch(grid, base)
    two = sv.solve(X0)
    sv.set_launch(grid, base | 128)
    for rep in range(25):
        r = sv.solve(X0)
        assert np.array_equal(r.pose, two.pose) and r.summary.final_cost == two.summary.final_cost, rep
        assert r.summary.num_iterations == two.summary.num_iterations and r.summary.termination == two.summary.termination
        assert r.summary.num_evaluations == two.summary.num_evaluations
        assert [(t.cost, t.step_is_successful, t.trust_region_radius) for t in r.trace] == \
               [(t.cost, t.step_is_successful, t.trust_region_radius) for t in two.trace]
    o = clc.default_options()
    o.max_num_iterations = 3  # stopped by the iteration cap: NO_CONVERGENCE through the same path
    a = sv.solve(X0, o)
    sv.set_launch(grid, base)
    b = sv.solve(X0, o)
    sv.set_launch(0, -1)
    assert a.summary.termination == b.summary.termination == 5  # NO_CONVERGENCE and np.array_equal(a.pose, b.pose)
    assert a.summary.num_iterations == b.summary.num_iterations == 3
    o.max_num_iterations = 0  # one evaluation, no step
    sv.set_launch(grid, base | 128)
    z = sv.solve(X0, o)
    sv.set_launch(0, -1)
    assert z.summary.termination == 5 and z.summary.num_iterations == 0 and z.summary.num_evaluations == 1
    assert np.array_equal(z.pose, X0) and z.summary.final_cost == z.summary.initial_cost == two.summary.initial_cost
    assert _dT(two.pose, ref.pose) <= T_TOL and abs(two.summary.final_cost - ref.summary.final_cost) <= COST_TOL
    assert two.summary.num_iterations == ref.summary.num_iterations


def test_randomized_problems_step_vs_pair_vs_oracle(sv, oracle_mod):
    """60 random problems (1 ... 3x10^4 observations, ragged scans, noise, random start within the basin, loss on/off):
    the default step-kernel solve equals the launch-pair solve bit for bit and the oracle within the BASELINE gates,
    with the same iteration count and termination."""
    rng = np.random.default_rng(2024)
    worst = 0.0
    for case in range(60):
        n_poses = int(rng.integers(3, 60))
        pts = int(rng.integers(2, 500))
        S = sd.sim_fixed_count(1000 + case, n_poses, pts, noise_sigma=float(rng.choice([0.0, 0.005, 0.02])))
        rec = clc.flatten_observations(S, False)
        n = int(rng.integers(max(8, rec.shape[0] // 2), rec.shape[0] + 1))
        rec = np.ascontiguousarray(rec[:n])
        gt = sd.pose7_from_T(sd.tlc_to_tcl(sd.GT_RLC, sd.GT_TLC))
        x0 = oracle_mod.pose_plus(gt, rng.normal(size=6) * np.array([0.05, 0.05, 0.05, 0.08, 0.08, 0.08])) if case % 3 else X0
        o = clc.default_options()
        o.use_loss = int(case % 5 != 0)
        oo = _mirror(oracle_mod, o)
        sv.upload(rec)
        sv.set_launch(0, 2 | 16 | 32 | 128)
        a = sv.solve(x0, o)
        sv.set_launch(0, 2 | 16 | 32)
        b = sv.solve(x0, o)
        sv.set_launch(0, 2 | 16 | 32 | 128 | 256)  # row layout (per-scan moments), step kernel — equal to the others up to rounding only
        d = sv.solve(x0, o)
        sv.set_launch(0, 2 | 32 | 256)  # row layout through the [evaluation, controller] launch pair
        e = sv.solve(x0, o)
        assert np.array_equal(d.pose, e.pose) and d.summary.final_cost == e.summary.final_cost, case
        sv.set_launch(0, -1)  # library default (layout chosen by size)
        f = sv.solve(x0, o)
        ref = oracle_mod.solve(rec, x0, options=oo, linear_solver="qr")
        assert f.summary.num_iterations == ref.summary.num_iterations and abs(f.summary.final_cost - ref.summary.final_cost) <= COST_TOL, case
        assert np.array_equal(a.pose, b.pose) and a.summary.final_cost == b.summary.final_cost, case
        assert a.summary.termination == b.summary.termination == d.summary.termination == ref.summary.termination, case
        assert a.summary.num_iterations == b.summary.num_iterations == d.summary.num_iterations == ref.summary.num_iterations, (case, n)
        assert abs(a.summary.final_cost - ref.summary.final_cost) <= COST_TOL and abs(d.summary.final_cost - ref.summary.final_cost) <= COST_TOL, case
        worst = max(worst, _dT(a.pose, ref.pose), _dT(d.pose, ref.pose))
        assert worst <= T_TOL, (case, n, worst)
    sv.set_launch(0, -1)


def test_c5_boundary_constraint_mixed_terms(sv, oracle_mod):
    """configs[4] (reduced): board-edge residuals (LaseCamCalCeres.cpp:258-294) mixed with the
    point residuals, same record type, same kernel."""
    S = sd.sim_board_edges(11, n_poses=60, pts_per_pose=40, noise_sigma=0.002)
    rec = clc.flatten_observations(S, True, True)
    assert rec.shape[0] == 60 * 42
    sv.upload(rec)
    Tcl_gt = sd.tlc_to_tcl(sd.GT_RLC, sd.GT_TLC)
    x0 = oracle_mod.pose_plus(sd.pose7_from_T(Tcl_gt), np.array([0.05, -0.04, 0.03, 0.05, -0.06, 0.04]))
    res = sv.solve(x0)
    ref = oracle_mod.solve(rec, x0, linear_solver="qr")
    assert _dT(res.pose, ref.pose) <= T_TOL
    assert abs(res.summary.final_cost - ref.summary.final_cost) <= COST_TOL
    assert res.summary.num_iterations == ref.summary.num_iterations
    assert _dT(res.pose, sd.pose7_from_T(Tcl_gt)) < 5e-2


def test_information_and_closed_form(sv, oracle_mod):
    S = sd.GenerateSimData(4, noise_sigma=0.01)
    rec = clc.flatten_observations(S, True)
    sv.upload(rec)
    T, unobs, sv9 = sv.closed_form()
    T0, unobs0, sv90 = oracle_mod.closed_form(rec)
    assert unobs == unobs0 and np.abs(T - T0).max() < 1e-9 and np.allclose(sv9, sv90, rtol=1e-9)
    pose = sd.pose7_from_T(np.linalg.inv(T))
    H, b, chi2, s6, V, nn = sv.information(pose)
    H0, b0, chi0, s60, V0, nn0 = oracle_mod.information(rec, pose)
    assert np.allclose(H, H0, rtol=1e-11) and np.allclose(b, b0, rtol=1e-9, atol=1e-14)
    assert abs(chi2 - chi0) <= 1e-11 * chi0 and np.allclose(s6, s60, rtol=1e-9) and nn == nn0 == 0
    # noise-free closed form is exact
    rec = clc.flatten_observations(sd.GenerateSimData(4), True)
    sv.upload(rec)
    T, unobs, _ = sv.closed_form()
    assert not unobs and np.abs(T[:3, :3] - sd.GT_RLC).max() < 1e-9 and np.abs(T[:3, 3] - sd.GT_TLC).max() < 1e-9


def test_reference_call_surface_end_to_end(oracle_mod, capsys):
    """calibr_offline.cpp:166-170 flow: closed form -> invert -> CamLaserCalibration(obs,Tcl,false)."""
    obs = sd.GenerateSimData(8, noise_sigma=0.01).to_list()  # std::vector<Oberserve>
    Tlc = np.eye(4)
    clc.CamLaserCalClosedSolution(obs, Tlc)
    Tcl = np.linalg.inv(Tlc)
    rep = clc.CamLaserCalibration(obs, Tcl, False)
    out = capsys.readouterr().out
    assert "Closed-form solution Tlc" in out and "H singular values" in out and "recover chi2" in out
    S = sd.ObservationSet.from_list(obs)
    rec = oracle_mod.flatten(S, False, False)
    ref = oracle_mod.solve(rec, sd.pose7_from_T(np.linalg.inv(oracle_mod.closed_form(oracle_mod.flatten(S, True, False))[0])))
    assert np.abs(Tcl - sd.T_from_pose7(ref.pose)).max() <= T_TOL
    assert abs(rep.result.summary.final_cost - ref.summary.final_cost) <= COST_TOL
    assert rep.null_space.shape == (6, 0)
    Tlc_est = np.linalg.inv(Tcl)
    assert np.abs(Tlc_est[:3, 3] - sd.GT_TLC).max() < 0.02


# ---------------------------------------------------------------------------------------
def test_c3_batched_problems_match_oracle(sv, oracle_mod):
    """configs[2] (reduced): independent T_cl problems, each with its own ground truth."""
    probs, gts = sd.sim_batch(21, n_problems=24, n_poses=12, pts_per_pose=97, noise_sigma=0.01)
    recs = [clc.flatten_observations(p, False) for p in probs]
    recs[3] = recs[3][:130]   # ragged sizes
    recs[7] = recs[7][:1]
    off = np.zeros(len(recs) + 1, dtype=np.int64)
    off[1:] = np.cumsum([r.shape[0] for r in recs])
    sv.upload_batched(np.concatenate(recs), off)
    assert sv.num_problems == 24
    x0 = np.stack([oracle_mod.pose_plus(sd.pose7_from_T(g), np.array([0.1, -0.1, 0.1, 0.1, -0.1, 0.05])) for g in gts])
    poses, sms = sv.solve_batched(x0)
    for k in range(24):
        ref = oracle_mod.solve(recs[k], x0[k], linear_solver="qr")
        assert sms[k].termination == ref.summary.termination, k
        assert sms[k].num_iterations == ref.summary.num_iterations, k
        assert _dT(poses[k], ref.pose) <= T_TOL, k
        assert abs(sms[k].final_cost - ref.summary.final_cost) <= COST_TOL, k
        if k not in (3, 7):
            assert _dT(poses[k], sd.pose7_from_T(gts[k])) < 0.05


def test_wide_ragged_batch_one_wave_per_problem(sv, oracle_mod):
    """8 192 small problems (>= 32 per CU: the lockstep solver gives each problem ONE single-wave workgroup), ragged: 6
    poses x 80 points cut to random lengths, some down to one observation, some empty.  The one-wave form must agree
    with the 256-thread-workgroup form (flag 1024) to rounding on every problem — same termination and iteration
    count, pose and cost within the BASELINE gates — and with the oracle on a sample; the default (whole solve per
    launch) must equal the 256-thread lockstep form bit for bit."""
    P, n_poses, K = 8192, 6, 80
    rec, off, x0, gt = sd.sim_shard_records(4242, 0, P, n_poses, K, 0.01)
    per = n_poses * K
    rng = np.random.default_rng(5)
    keep = rng.integers(per // 2, per + 1, size=P)
    keep[::97] = 1
    keep[5::211] = 0
    idx = np.concatenate([np.arange(k * per, k * per + keep[k]) for k in range(P)])
    rec2 = np.ascontiguousarray(rec[idx])
    off2 = np.zeros(P + 1, dtype=np.int64)
    off2[1:] = np.cumsum(keep)
    sv.upload_batched(rec2, off2)
    assert sv.debug_rows()[2]  # batched row layout in use
    sv.set_launch(0, 2 | 16 | 32 | 128 | 256 | 512 | 2048)  # lockstep launches: one single-wave workgroup per problem
    pa, sa = sv.solve_batched(x0)
    sv.set_launch(0, 2 | 16 | 32 | 128 | 256 | 512 | 2048 | 1024)  # lockstep launches, 256-thread workgroups
    pb, sb = sv.solve_batched(x0)
    sv.set_launch(0, 2 | 16 | 32 | 128 | 256 | 512 | 4096)  # every problem's whole solve in one launch on the row layout
    pc, sc = sv.solve_batched(x0)
    sv.set_launch(0, -1)  # (the default is the on-chip resident kernel: tests/test_gpu_resident.py)
    # ... which sums like the 256-thread lockstep form: bit-identical to it
    assert np.array_equal(pc, pb, equal_nan=True)
    assert [(s.final_cost, s.num_iterations, s.termination, s.num_evaluations) for s in sc] == \
           [(s.final_cost, s.num_iterations, s.termination, s.num_evaluations) for s in sb]
    n_diff = 0
    for k in range(P):
        if sa[k].termination != sb[k].termination or sa[k].num_iterations != sb[k].num_iterations:
            n_diff += 1  # a tolerance test may flip on a last-bit difference of the sums; rare
            continue
        if keep[k] >= per // 2:
            assert _dT(pa[k], pb[k]) <= T_TOL and abs(sa[k].final_cost - sb[k].final_cost) <= COST_TOL, k
    assert n_diff <= P // 500, n_diff
    for k in list(range(0, P, 512)) + [97, 5]:
        r1 = rec2[off2[k]:off2[k + 1]]
        if r1.shape[0] == 0:
            continue
        ref = oracle_mod.solve(r1, x0[k], linear_solver="qr")
        assert sa[k].termination == ref.summary.termination and sa[k].num_iterations == ref.summary.num_iterations, k
        if keep[k] >= per // 2:
            assert _dT(pa[k], ref.pose) <= T_TOL and abs(sa[k].final_cost - ref.summary.final_cost) <= COST_TOL, k


def test_c3_full_size_batch(sv, oracle_mod):
    """configs[2] at full size: 1 024 independent T_cl problems x 10^4 observations (655 MB).
    Size-independent properties on all problems (every problem terminates by convergence at a
    cost no higher than where it started, recovers its own ground truth to the noise level, the
    batch is permutation-equivariant) + oracle parity on ALL 1 024 problems (~15 s of CPU)."""
    P = 1024
    probs, gts = sd.sim_batch(4242, P, 20, 500, noise_sigma=0.01)
    recs = [clc.flatten_observations(p, False) for p in probs]
    off = np.zeros(P + 1, dtype=np.int64)
    off[1:] = np.cumsum([r.shape[0] for r in recs])
    rng = np.random.default_rng(1)
    x0 = sv.pose_plus(np.stack([sd.pose7_from_T(g) for g in gts]), rng.normal(size=(P, 6)) * 0.05)
    sv.upload_batched(np.concatenate(recs), off)
    assert sv.debug_resident()[0]
    poses, sms = sv.solve_batched(x0)  # default: every problem resident on chip for its whole solve (clc_resident.hpp)
    # flag 4096 = the whole solve of every problem in one launch on the row layout (batched_solve_kernel); flag
    # 2048 = the same workgroups in lockstep launches with the serial controller: bit-identical to each other
    sv.set_launch(0, 2 | 16 | 32 | 128 | 256 | 512 | 4096)
    poses_w, sms_w = sv.solve_batched(x0)
    sv.set_launch(0, 2 | 16 | 32 | 128 | 256 | 512 | 2048)
    poses_l, sms_l = sv.solve_batched(x0)
    sv.set_launch(0, -1)
    assert np.array_equal(poses_w, poses_l)
    assert [(s.final_cost, s.initial_cost, s.num_iterations, s.termination, s.num_evaluations, s.num_successful_steps) for s in sms_w] == \
           [(s.final_cost, s.initial_cost, s.num_iterations, s.termination, s.num_evaluations, s.num_successful_steps) for s in sms_l]
    # the resident kernel sums in another order: same decisions, results to rounding
    for k in range(P):
        assert (sms[k].num_iterations, sms[k].termination, sms[k].num_evaluations) == \
               (sms_l[k].num_iterations, sms_l[k].termination, sms_l[k].num_evaluations), k
        assert np.abs(poses[k] - poses_l[k]).max() <= 1e-9 and abs(sms[k].final_cost - sms_l[k].final_cost) <= 1e-11, k
    for k in range(P):
        assert sms[k].termination in (1, 2, 3), k
        assert sms[k].final_cost <= sms[k].initial_cost
        assert np.abs(sd.T_from_pose7(poses[k]) - gts[k]).max() < 0.02, k
        assert abs(np.linalg.norm(poses[k, 3:]) - 1) < 1e-14
    for k in range(P):  # every problem against the oracle's DENSE_QR solve
        ref = oracle_mod.solve(recs[k], x0[k], linear_solver="qr")
        assert sms[k].num_iterations == ref.summary.num_iterations and sms[k].termination == ref.summary.termination, k
        assert _dT(poses[k], ref.pose) <= T_TOL and abs(sms[k].final_cost - ref.summary.final_cost) <= COST_TOL, k
    # permutation equivariance: reversing the problem order reverses the results bit for bit
    perm = np.arange(P)[::-1]
    off2 = np.zeros(P + 1, dtype=np.int64)
    off2[1:] = np.cumsum([recs[i].shape[0] for i in perm])
    sv.upload_batched(np.concatenate([recs[i] for i in perm]), off2)
    poses2, sms2 = sv.solve_batched(x0[perm])
    assert np.array_equal(poses2[::-1], poses)
    assert [sms2[P - 1 - k].final_cost for k in range(P)] == [sms[k].final_cost for k in range(P)]


def test_c2_full_size_properties_and_parity(sv, oracle_mod):
    """configs[1]: single T_cl, 10^6 observations (2000 poses x 500 points)."""
    S = sd.sim_fixed_count(2024, 2000, 500, noise_sigma=0.0)
    rec = clc.flatten_observations(S, False)
    assert rec.shape == (1_000_000, 8)
    sv.upload(rec)
    # linearity of the reduction: eval(all) == eval(first half) + eval(second half)
    c, g, H = sv.eval(X0)
    sv.upload(rec[:500_000]); c1, g1, H1 = sv.eval(X0)
    sv.upload(rec[500_000:]); c2, g2, H2 = sv.eval(X0)
    assert abs(c - (c1 + c2)) <= 1e-12 * c and np.allclose(H, H1 + H2, rtol=1e-12) and np.allclose(g, g1 + g2, rtol=1e-10, atol=1e-12)
    # noise-free: unique zero-cost minimum at the ground truth (the reference's known answer)
    sv.upload(rec)
    res = sv.solve(X0)
    Tlc = np.linalg.inv(sd.T_from_pose7(res.pose))
    assert np.abs(Tlc[:3, :3] - sd.GT_RLC).max() < 1e-7 and np.abs(Tlc[:3, 3] - sd.GT_TLC).max() < 1e-7
    assert res.summary.final_cost < 1e-10
    # noisy: parity with the oracle's DENSE_QR solve at full size
    S = sd.sim_fixed_count(2025, 2000, 500, noise_sigma=0.01)
    rec = clc.flatten_observations(S, False)
    sv.upload(rec)
    res = sv.solve(X0)
    ref = oracle_mod.solve(rec, X0, linear_solver="qr")
    assert _dT(res.pose, ref.pose) <= T_TOL
    assert abs(res.summary.final_cost - ref.summary.final_cost) <= COST_TOL
    assert res.summary.num_iterations == ref.summary.num_iterations
    # idempotence: solving again from the solution does not move it
    again = sv.solve(res.pose)
    assert _dT(again.pose, res.pose) < 1e-5


def test_c5_full_size_board_edge_terms(sv, oracle_mod):
    """configs[4] at full size: 10^6 point residuals + 2 board-edge residuals per scan (1 004 000
    records, mixed unit / un-normalised planes), same kernel."""
    S5 = sd.sim_board_edges(5, 2000, 500, noise_sigma=0.002)
    rec5 = clc.flatten_observations(S5, True, True)
    assert rec5.shape[0] == 1_004_000
    gt = sd.pose7_from_T(sd.tlc_to_tcl(sd.GT_RLC, sd.GT_TLC))
    x0 = oracle_mod.pose_plus(gt, np.array([0.05, -0.04, 0.03, 0.05, -0.06, 0.04]))
    sv.upload(rec5)
    res = sv.solve(x0)
    ref = oracle_mod.solve(rec5, x0, linear_solver="qr")
    assert _dT(res.pose, ref.pose) <= T_TOL and abs(res.summary.final_cost - ref.summary.final_cost) <= COST_TOL
    assert res.summary.num_iterations == ref.summary.num_iterations
    assert _dT(res.pose, gt) < 1e-4
    # the analysis pass skips the edge terms (LaseCamCalCeres.cpp:316-362): upload the point records only
    recp = clc.flatten_observations(S5, True, False)
    sv.upload(recp)
    H, b, chi2, s6, V, nn = sv.information(res.pose)
    H0, b0, chi0, s60, V0, nn0 = oracle_mod.information(recp, res.pose)
    assert np.allclose(H, H0, rtol=1e-11) and abs(chi2 - chi0) <= 1e-11 * chi0 and nn == nn0


def _scans(n_scans, seed):
    rng = np.random.default_rng(seed)
    xs, truth, off = [], [], [0]
    for k in range(n_scans):
        n = int(rng.integers(0, 4)) if k % 50 == 7 else int(rng.integers(40, 400))
        th = rng.uniform(-1.3, 1.3); c = rng.uniform(0.8, 5.0)
        t = np.sort(rng.uniform(-0.5, 0.5, n))
        xy = np.stack([c * np.cos(th) - t * np.sin(th), c * np.sin(th) + t * np.cos(th)], 1) + rng.normal(size=(n, 2)) * 0.004
        bad = rng.random(n) < 0.08
        xy[bad] += rng.normal(size=(int(bad.sum()), 2)) * 0.3
        xs.append(xy); truth.append([-np.cos(th) / c, -np.sin(th) / c]); off.append(off[-1] + n)
    return np.concatenate(xs), np.array(off, dtype=np.int64), np.array(truth)


def test_line_fit_batched_matches_oracle(sv, oracle_mod):
    """LineFittingCeres (LaseCamCalCeres.cpp:385-433) for 600 ragged scans in one launch, against
    the oracle's DENSE_QR restatement scan by scan (incl. empty / 1-3 point scans)."""
    xy, off, truth = _scans(600, 11)
    lines0 = np.zeros((600, 2))
    lines, sms = sv.line_fit_batched(xy, off, lines0)
    worst = 0.0
    for k in range(600):
        ref = oracle_mod.line_fit(xy[off[k]:off[k + 1]], lines0[k], linear_solver="qr")
        assert sms[k].termination == ref.summary.termination, k
        assert sms[k].num_iterations == ref.summary.num_iterations, k
        assert np.abs(lines[k] - ref.pose).max() <= 1e-9 * max(1.0, np.abs(ref.pose).max()), k
        assert abs(sms[k].final_cost - ref.summary.final_cost) <= 1e-12, k
        if off[k + 1] - off[k] >= 40:
            worst = max(worst, np.abs(lines[k] - truth[k]).max() / np.abs(truth[k]).max())
    assert worst < 0.05
    # options are honoured (no loss, iteration cap) and the single-scan mirror is in/out
    o = clc.default_line_options(); o.use_loss = 0; o.max_num_iterations = 3
    l2, s2 = sv.line_fit_batched(xy, off, lines0, o)
    oo = oracle_mod.default_line_options(); oo.use_loss = 0; oo.max_num_iterations = 3
    for k in (0, 5, 100):
        ref = oracle_mod.line_fit(xy[off[k]:off[k + 1]], lines0[k], options=oo)
        assert s2[k].num_iterations == ref.summary.num_iterations and np.abs(l2[k] - ref.pose).max() < 1e-9
    L = np.zeros(2)
    P3 = np.concatenate([xy[off[0]:off[1]], np.zeros((off[1] - off[0], 1))], 1)
    clc.LineFittingCeres(P3, L, solver=sv)
    assert np.array_equal(L, lines[0])


def test_offline_front_end_flow_with_line_fit(sv, oracle_mod):
    """main/calibr_offline.cpp:121-170: fit a line per scan -> two points_on_line per scan ->
    closed form on them -> nonlinear refinement.  Everything O(points) on the GPU."""
    S = sd.GenerateSimData(21, noise_sigma=0.005)
    keep = (S.pts_off[1:] - S.pts_off[:-1]) >= 10
    S = sd.ObservationSet.from_list([o for o, k in zip(S.to_list(), keep) if k])
    S2 = clc.points_on_fitted_lines(S, solver=sv)
    assert S2.ptl.shape[0] == 2 * S2.n_poses
    # the two points lie on the board plane (up to noise) at the scan's end abscissae
    rec = clc.flatten_observations(S2, True)
    gt = sd.pose7_from_T(sd.tlc_to_tcl(sd.GT_RLC, sd.GT_TLC))
    r, _ = oracle_mod.factor_evaluate_batch(rec, gt, want_jac=False)
    assert np.abs(r).max() < 0.02
    Tlc = np.eye(4)
    clc.CamLaserCalClosedSolution(S2, Tlc, solver=sv, verbose=False)
    Tcl = np.linalg.inv(Tlc)
    rep = clc.CamLaserCalibration(S2, Tcl, True, False, solver=sv, verbose=False)
    ref = oracle_mod.solve(rec, sd.pose7_from_T(np.linalg.inv(oracle_mod.closed_form(rec)[0])))
    assert np.abs(Tcl - sd.T_from_pose7(ref.pose)).max() <= T_TOL
    assert np.abs(np.linalg.inv(Tcl)[:3, 3] - sd.GT_TLC).max() < 0.03


def test_scan_to_points_matches_oracle(sv, oracle_mod):
    """TranScanToPoints (src/utilities.cpp:181-215): 40 scans of 1081 rays (0.25 deg), with
    out-of-range returns (inf / 0 / > 30 m) mapped to the (1000, 1000, 0) sentinel."""
    rng = np.random.default_rng(4)
    S, n = 40, 1081
    ranges = rng.uniform(0.05, 35.0, size=(S, n)).astype(np.float32)
    ranges[:, ::97] = np.inf
    ranges[:, 5::113] = 0.0
    off = np.arange(S + 1, dtype=np.int64) * n
    amin = rng.uniform(-2.4, -2.3, S).astype(np.float32)
    ainc = np.full(S, np.float32(np.deg2rad(0.25)))
    rmin = np.full(S, np.float32(0.1))
    pts = sv.scan_to_points(ranges.reshape(-1), off, amin, ainc, rmin)
    for k in (0, 7, 39):
        ref = oracle_mod.scan_to_points(ranges[k], amin[k], ainc[k], rmin[k])
        got = pts[off[k]:off[k + 1]]
        assert np.array_equal(got[:, 2], ref[:, 2]) and np.array_equal(got == 1000.0, ref == 1000.0)
        assert np.abs(got - ref).max() <= 4e-15 * 35.0
    assert (pts[:, 0] == 1000.0).sum() > S  # sentinels present


def test_bench_two_rank_path_dry_run(tmp_path):
    """bench.py's N>1 code path (every rank generates and solves only its own shard of the batch, gather of all result
    records, max-over-ranks timing) with 2 ranks oversubscribing the one visible GPU; gloo stands in for the process
    group and the records travel through torch.distributed (RCCL refuses two ranks on one GPU).  Started the way the driver starts
    the N=1 run — `python bench.py --gpus 2 ...`, NO launcher: bench.py launches the two ranks itself."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    detail = str(tmp_path / "detail.json")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--blocks", "2",
           "--backend", "gloo", "--oversubscribe", "--problems-per-gpu", "96", "--shard-poses", "6", "--shard-pts", "100", "--detail-file", detail]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096  # rank 0 prints ONE short JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["timed_blocks"] == 2
    assert d["ms_per_step_min"] <= d["ms_per_step"] <= d["ms_per_step_max"]
    rf = d["roofline"]
    assert rf["bound"] == "valu_f64" and 0 < rf["frac_moved"] <= 1  # no HBM fraction above 1 anywhere
    if rf["valu_issue"]["current"]:  # the instruction counts belong to the sources in the tree (profiles/valu_counts.json)
        assert 0 < rf["frac"] <= 1
    else:
        assert rf["frac"] is None
    assert abs(d["per_gpu_value"] * 2 - d["value"]) <= 1e-5 * d["value"]
    full = json.load(open(detail))
    assert full["value"] == pytest.approx(d["value"], rel=1e-6)
    c4 = full["batched_c4_shard"]
    assert c4["problems"] == 192 and c4["problems_per_gpu"] == 96
    first, last = c4["first_and_last_record"]
    assert first[11] == 0.0 and last[11] == 191.0 and first[:7] != last[:7]
    assert all(abs(sum(x * x for x in r[3:7]) - 1) < 1e-12 for r in (first, last))
    assert c4["max_abs_T_err_vs_ground_truth_sampled"] < 0.05 and c4["T_cl_max_abs_err_vs_oracle_sample"] <= T_TOL
    assert full["roofline"]["streaming_eval_kernel"]["bound"] == "hbm" and 0 < full["roofline"]["streaming_eval_kernel"]["frac"] <= 1
    assert "scale_base" in full["scale_base_field"]


def test_bench_eight_rank_path_uneven_shards_dry_run(tmp_path):
    """What C4 looks like on 8 ranks, through bench.py's own N>1 path: 8 processes (oversubscribing the one visible GPU, gloo
    process group), a problem count that 8 does not divide — shards of 13 and 12 problems, padding records in the gather —
    and the rank-major gathered buffer put back into global order by bench.py (order_records).  Launched the way the driver launches
    N>1: python -m torch.distributed.run ... bench.py --gpus 8."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    detail = str(tmp_path / "detail.json")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--blocks", "1",
           "--backend", "gloo", "--oversubscribe", "--problems-total", "100", "--problems-per-gpu", "13", "--shard-poses", "6",
           "--shard-pts", "100", "--detail-file", detail]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["value"] > 0
    c4 = json.load(open(detail))["batched_c4_shard"]
    g = c4["gather"]
    assert c4["problems"] == 100
    assert g["shard_sizes"] == [13, 13, 13, 13, 12, 12, 12, 12] and g["records_per_rank"] == 13
    assert g["first_global_index_per_rank"] == [0, 13, 26, 39, 52, 64, 76, 88]
    assert g["padding_records"] == 8 * 13 - 100
    assert g["first_record_index_of_each_rank_block"] == [0.0, 13.0, 26.0, 39.0, 52.0, 64.0, 76.0, 88.0]
    first, last = c4["first_and_last_record"]
    assert first[11] == 0.0 and last[11] == 99.0
    assert c4["max_abs_T_err_vs_ground_truth_sampled"] < 0.05 and c4["T_cl_max_abs_err_vs_oracle_sample"] <= T_TOL
    assert g["rccl_ranks"] is None  # (gloo dry run; with --backend nccl bench.py asserts rccl_ranks == WORLD_SIZE)


def test_rccl_world1_sharded_solve_through_the_c_abi(sv, oracle_mod):
    """RCCL for real on the one GPU: clc_comm_create (ncclCommInitRank, world size 1) + clc_gather_results
    (ncclAllGather from the device buffer of clc_solve_batched) through dist.solve_sharded / ShardSolver; the gathered
    records equal what clc_solve_batched returned and the oracle's solves."""
    from camlasercalibratool_amd import dist as cdist

    P = 37
    calls = []

    def load_shard(lo, hi):
        calls.append((lo, hi))
        return sd.sim_shard_records(11, lo, hi, 6, 64, 0.01)[:3]

    full = cdist.solve_sharded(load_shard, P, device_index=0)
    assert calls == [(0, P)] and full.shape == (P, 12)
    assert np.array_equal(full[:, 11], np.arange(P))
    rec, off, x0, gt = sd.sim_shard_records(11, 0, P, 6, 64, 0.01)
    for k in range(P):
        ref = oracle_mod.solve(rec[off[k]:off[k + 1]], x0[k], linear_solver="qr")
        assert _dT(full[k, :7], ref.pose) <= T_TOL and abs(full[k, 7] - ref.summary.final_cost) <= COST_TOL, k
        assert full[k, 9] == ref.summary.num_iterations and full[k, 10] == ref.summary.termination, k
    # persistent form with padding: capacity larger than the shard, raw rank-major buffer
    with cdist.ShardSolver(P, device_index=0, rank=0, world=1) as ss:
        assert "rccl" in ss.comm.library
        ss.upload(rec, off)
        raw = ss.solve(x0, ordered=False)
        poses, sms = ss.last_poses, ss.last_summaries
        assert np.array_equal(raw[:, :7], poses) and np.array_equal(raw, full)
        big = ss.comm.gather_results(1000, P + 5)  # padding records carry global index -1
        assert big.shape == (P + 5, 12) and np.array_equal(big[:P, 11], 1000 + np.arange(P))
        assert np.all(big[P:, 11] == -1) and np.all(big[P:, :11] == 0) and np.array_equal(big[:P, :11], full[:, :11])
        view = ss.comm.gather_results(0, P, copy=False)  # zero-copy view of the pinned buffer
        assert np.array_equal(view, full)
        with pytest.raises(clc.ClcError):
            ss.comm.gather_results(0, P - 1)  # capacity below the shard size


def test_solve_batched_gather_one_call_equals_the_two_calls(oracle_mod):
    """clc_solve_batched_gather (the kernel's epilogue writes the records into the gather buffer; in-place ncclAllGather; one copy):
    the same records, bit for bit, as clc_solve_batched + clc_gather_results; the shard's totals equal the sums over the summaries;
    padding records when the capacity exceeds the shard; call after call (the device totals are running sums), mixed with the two-call
    form (which rewrites this rank's segment), with the handle's pinned pose buffer as input; a batch that does not run on chip
    (points with z) takes the two-call fall-back; a non-finite start pose is reported AFTER the collective, with padding only."""
    from camlasercalibratool_amd import dist as cdist

    P = 61
    rec, off, x0, gt = sd.sim_shard_records(23, 100, 100 + P, 8, 90, 0.01)
    with cdist.ShardSolver(P, device_index=0, rank=0, world=1) as ss:
        ss.upload(rec, off)
        assert ss.solver.path_info().batched_resident == 1
        two = ss.solve(x0, ordered=False)
        sms = ss.last_summaries
        ev, it = sum(sms[k].num_evaluations for k in range(P)), sum(sms[k].num_iterations for k in range(P))
        for rep in range(3):
            one = ss.solve_gather(x0, ordered=False)
            st = ss.last_stats
            assert np.array_equal(one, two), rep
            assert (st.problems, st.evaluations, st.iterations, st.not_converged, st.fused) == (P, ev, it, 0, 1), rep
        # capacity above the shard + another index base: padding records, global indices
        big, st = ss.comm.solve_gather(x0, 5000, P + 7)
        assert big.shape == (P + 7, 12) and np.array_equal(big[:P, :11], two[:, :11]) and np.array_equal(big[:P, 11], 5000 + np.arange(P))
        assert np.all(big[P:, 11] == -1) and np.all(big[P:, :11] == 0) and st.problems == P
        # the two-call form in between rewrites this rank's segment of the gather buffer; the one-call form pads again
        assert np.array_equal(ss.solve(x0, ordered=False), two)
        big2, _ = ss.comm.solve_gather(x0, 5000, P + 7, copy=False)
        assert np.array_equal(big2, big)
        # start poses already in the handle's pinned buffer (poses0 = None)
        ss.solver.batched_buffers()[0][:] = x0
        view, st = ss.comm.solve_gather(None, 0, P, copy=False)
        assert np.array_equal(view, two) and st.evaluations == ev
        # profile_events = 1: the launch's HIP event pair comes back in the stats
        o = clc.default_options()
        o.profile_events = 1
        _, st = ss.comm.solve_gather(x0, 0, P, o)
        assert 0 < st.kernel_ms < st.solve_ms
        # a non-finite start pose: the rank still takes part (padding only) and reports afterwards
        bad = x0.copy()
        bad[3, 1] = np.nan
        with pytest.raises(clc.ClcError, match="non-finite"):
            ss.comm.solve_gather(bad, 0, P)
        assert np.array_equal(ss.solve_gather(x0, ordered=False), two)  # and the next call is whole again
        # points off the lidar plane: still one launch (the 24-byte-slot form of the on-chip kernel), still the one-call form
        recz = rec.copy()
        recz[::5, 6] = 0.01
        ss.upload(recz, off)
        assert ss.solver.path_info().batched_resident == 1 and ss.solver.path_info().batched_points_carry_z == 1
        twoz = ss.solve(x0, ordered=False)
        evz = sum(ss.last_summaries[k].num_evaluations for k in range(P))
        onez = ss.solve_gather(x0, ordered=False)
        assert np.array_equal(onez, twoz) and ss.last_stats.fused == 1 and ss.last_stats.evaluations == evz
    # a batch that does not run as the on-chip solve (problems of 15 000 observations: more than a workgroup holds): the two-call
    # fall-back, same contract
    P3 = 5
    rec3, off3, x03, _ = sd.sim_shard_records(29, 0, P3, 30, 500, 0.01)
    with cdist.ShardSolver(P3, device_index=0, rank=0, world=1) as ss:
        ss.upload(rec3, off3)
        assert ss.solver.path_info().batched_resident == 0
        two3 = ss.solve(x03, ordered=False)
        ev3 = sum(ss.last_summaries[k].num_evaluations for k in range(P3))
        one3 = ss.solve_gather(x03, ordered=False)
        assert np.array_equal(one3, two3) and ss.last_stats.fused == 0 and ss.last_stats.problems == P3
        assert ss.last_stats.evaluations == ev3
    k = 17
    ref = oracle_mod.solve(rec[off[k]:off[k + 1]], x0[k], linear_solver="qr")
    assert _dT(two[k, :7], ref.pose) <= T_TOL and two[k, 9] == ref.summary.num_iterations


def test_gather_buffer_arithmetic_at_rank_1_of_3():
    """What cannot be run for real on one GPU — a rank other than 0 of a world larger than 1 — through the hooks build's layout-only
    communicator (no RCCL behind it: its all-gather moves this rank's segment into place, the other ranks' segments stay as they are).
    Both gather forms at rank 1 of 3 with a capacity above the shard: this rank's records in the SECOND segment with their global
    indices, its padding records behind them, the other two segments copied to the host untouched (zeros), totals of the shard."""
    from camlasercalibratool_amd import dist as cdist
    from camlasercalibratool_amd.solver import Comm

    P, cap, lo = 23, 29, 29  # rank 1's shard of 3 x 29 problems would start at global index 29
    rec, off, x0, gt = sd.sim_shard_records(31, lo, lo + P, 7, 80, 0.01)
    with clc.Solver(0, library="hooks") as s0:
        s0.upload_batched(rec, off)
        poses, sms = s0.solve_batched(x0)
        c = Comm(s0, None, 1, 3)
        try:
            two = c.gather_results(lo, cap)
            one, st = c.solve_gather(x0, lo, cap)
            one2, st2 = c.solve_gather(x0, lo, cap, copy=False)
            one2 = one2.copy()  # (a view of the communicator's pinned buffer: gone with the communicator)
        finally:
            c.close()
    for name, g in (("two-call", two), ("one-call", one), ("one-call again", one2)):
        assert g.shape == (3 * cap, 12), name
        assert np.all(g[:cap] == 0) and np.all(g[2 * cap:] == 0), name               # ranks 0 and 2: nothing arrived (no RCCL), nothing overwritten
        mine = g[cap:2 * cap]
        assert np.array_equal(mine[:P, :7], poses) and np.array_equal(mine[:P, 11], lo + np.arange(P)), name
        assert np.all(mine[P:, 11] == -1) and np.all(mine[P:, :11] == 0), name        # this rank's padding records
        assert np.array_equal(mine[:P, 9], [sms[k].num_iterations for k in range(P)]), name
    ev = sum(sms[k].num_evaluations for k in range(P))
    assert (st.problems, st.evaluations, st.fused) == (P, ev, 1) and (st2.problems, st2.evaluations) == (P, ev)


def test_gather_to_root_copies_to_the_host_at_the_root_only():
    """clc_comm_set_root (a gather to rank 0): at rank 1 of 3 with root 0 the gather calls issue NO
    device-to-host copy of other ranks' segments (the one-call form: none at all — the kernel itself wrote this rank's records to the
    host), at root 1 they copy both other segments; this rank's segment is the same either way.  The layout-only communicator of the
    hooks build stands in for ranks RCCL cannot be given on one GPU; the byte counts are clc_comm_get_info's."""
    from camlasercalibratool_amd.solver import Comm

    P, cap, lo = 23, 29, 29
    rec, off, x0, gt = sd.sim_shard_records(31, lo, lo + P, 7, 80, 0.01)
    seg = 96 * cap
    with clc.Solver(0, library="hooks") as s0:
        s0.upload_batched(rec, off)
        poses, sms = s0.solve_batched(x0)
        c = Comm(s0, None, 1, 3)
        try:
            i0 = c.info()
            assert (i0.struct_size, i0.rank, i0.world, i0.root, i0.copies_other_ranks_to_host) == (C.sizeof(_capi.CommInfo), 1, 3, -1, 1)
            all_ranks, _ = c.solve_gather(x0, lo, cap)
            i1 = c.info()
            assert (i1.host_copies - i0.host_copies, i1.host_copy_bytes - i0.host_copy_bytes) == (2, 2 * seg)  # segments 0 and 2
            c.set_root(0)                                   # this rank (1) is NOT the root
            assert c.info().copies_other_ranks_to_host == 0
            not_root, st = c.solve_gather(x0, lo, cap)
            i2 = c.info()
            assert (i2.host_copies, i2.host_copy_bytes) == (i1.host_copies, i1.host_copy_bytes)                # nothing crossed PCIe
            s0.solve_batched(x0)
            not_root_two = c.gather_results(lo, cap)        # the two-call form copies its own segment only
            i3 = c.info()
            assert (i3.host_copies - i2.host_copies, i3.host_copy_bytes - i2.host_copy_bytes) == (1, seg)
            c.set_root(1)                                   # this rank IS the root
            at_root, _ = c.solve_gather(x0, lo, cap)
            i4 = c.info()
            assert (i4.host_copies - i3.host_copies, i4.host_copy_bytes - i3.host_copy_bytes) == (2, 2 * seg)
            assert i4.collectives == 4 and i4.rooted_collectives == 0  # (no RCCL behind this communicator: nothing to call ncclGather on)
            with pytest.raises(_capi.ClcError):
                c.set_root(3)
        finally:
            c.close()
    for g in (all_ranks, not_root, not_root_two, at_root):
        mine = g[cap:2 * cap]
        assert np.array_equal(mine[:P, :7], poses) and np.array_equal(mine[:P, 11], lo + np.arange(P))
        assert np.all(mine[P:, 11] == -1)
    assert st.problems == P and st.fused == 1


def test_pipelined_steps_return_the_previous_step_and_leave_it_untouched():
    """clc_solve_batched_gather_pipelined: call k enqueues step k and returns step k-1's records and totals; the device-to-host copy of
    step k-1's other-rank segments goes to the copy stream and overlaps step k's kernel; the host buffers alternate, so the records
    handed back stay untouched while step k's kernel writes (they are valid until the next call).  Steps with DIFFERENT start poses tell the steps apart.  On the
    layout-only communicator at rank 1 of 3 (copies really issued) and on real RCCL at world size 1."""
    from camlasercalibratool_amd import dist as cdist
    from camlasercalibratool_amd.solver import Comm

    P, cap, lo = 40, 41, 41
    rec, off, x0, gt = sd.sim_shard_records(77, lo, lo + P, 7, 80, 0.01)
    rng = np.random.default_rng(5)
    starts = [x0] + [np.array([sd.pose7_from_T(g) for g in gt]) for _ in range(1)]
    with clc.Solver(0, library="hooks") as s0:
        starts.append(s0.pose_plus(starts[1], rng.normal(size=(P, 6)) * 0.02))
        s0.upload_batched(rec, off)
        want = [s0.solve_batched(x)[0] for x in starts]   # the poses each step must deliver
        for rank, world, comm_of in ((1, 3, lambda: Comm(s0, None, 1, 3)), (0, 1, lambda: Comm(s0, cdist.exchange_unique_id(0, 1), 0, 1))):
            c = comm_of()
            try:
                c.set_root(rank)  # the root: copies issued (world 3), nothing to copy (world 1)
                r0, st0 = c.solve_gather_pipelined(starts[0], lo, cap)
                assert r0 is None and st0 is None and c.info().step_in_flight == 1
                with pytest.raises(_capi.ClcError):
                    c.solve_gather(starts[0], lo, cap)    # refused while a step is in flight
                import time
                r1, st1 = c.solve_gather_pipelined(starts[1], lo, cap)
                keep1 = r1.copy()
                time.sleep(0.05)                           # (step 1's kernel has run by now — into the OTHER twin)
                assert np.array_equal(r1, keep1)           # step 0's records untouched by step 1's kernel: valid until the next call
                r2, st2 = c.solve_gather_pipelined(starts[2], lo, cap)
                keep2 = r2.copy()
                time.sleep(0.05)
                assert np.array_equal(r2, keep2)
                r3, st3 = c.flush(cap)
                assert c.flush(cap) == (None, None) and c.info().step_in_flight == 0
                for k, (r, st) in enumerate(((keep1, st1), (keep2, st2), (r3, st3))):
                    mine = r[rank * cap:(rank + 1) * cap]
                    assert np.array_equal(mine[:P, :7], want[k]), (world, k)
                    assert np.array_equal(mine[:P, 11], lo + np.arange(P)) and np.all(mine[P:, 11] == -1)
                    assert st.problems == P and st.fused == 1 and st.iterations == int(mine[:P, 9].sum())
                ci = c.info()
                assert ci.pipelined_steps == 3 and ci.host_copy_bytes == (0 if world == 1 else 3 * 2 * 96 * cap)
                one, st = c.solve_gather(starts[0], lo, cap)  # the plain call works again after the flush
                assert np.array_equal(one[rank * cap:rank * cap + P, :7], want[0])
            finally:
                c.close()


def test_a_rank_with_an_empty_shard_still_takes_part_in_every_gather_form():
    """13 problems over 8 ranks leave some ranks one problem, and a job of fewer problems than ranks leaves some NONE: such a rank enters the
    collective with padding records only — the plain call, the pipelined calls and the flush — and reports zero problems (layout-only
    communicator, rank 2 of 3, capacity 4)."""
    from camlasercalibratool_amd.solver import Comm

    cap = 4
    with clc.Solver(0, library="hooks") as s0:
        s0.upload_batched(np.zeros((0, 8)), np.array([0], dtype=np.int64))   # this rank's shard: no problems
        assert s0.num_problems == 0
        c = Comm(s0, None, 2, 3)
        try:
            g, st = c.solve_gather(None, 8, cap)
            assert st.problems == 0 and st.evaluations == 0
            r0, _ = c.solve_gather_pipelined(None, 8, cap)
            r1, st1 = c.solve_gather_pipelined(None, 8, cap)
            r2, st2 = c.flush(cap)
            assert r0 is None and st1.problems == 0 and st2.problems == 0
            for rec_ in (g, r1, r2):
                assert rec_.shape == (3 * cap, 12)
                mine = rec_[2 * cap:]
                assert np.all(mine[:, 11] == -1) and np.all(mine[:, :11] == 0)
        finally:
            c.close()


def test_c4_full_size_shard(sv, oracle_mod):
    """configs[3], one GPU's share at full size: 8 192 independent T_cl problems x 10^4 observations (5.2 GB of
    records), solved by clc_solve_batched and gathered through RCCL.  Size-independent properties on ALL problems
    (termination by convergence at a cost no higher than the start, unit quaternions, own ground truth recovered to the
    noise level, solving twice is bitwise repeatable) + oracle parity (DENSE_QR Ceres restatement) on ALL 8 192 problems (the oracle's
    answers frozen in tests/golden/oracle_c4_shard.npz) and on 16 of them run live, regenerated one by one from (seed, global index)."""
    from camlasercalibratool_amd import dist as cdist

    P, n_poses, K, seed, lo = 8192, 20, 500, 65536, 3 * 8192  # the shard rank 3 of 8 would own
    rec, off, x0, gt = sd.sim_shard_records(seed, lo, lo + P, n_poses, K, 0.01)
    assert rec.shape == (P * n_poses * K, 8)
    with cdist.ShardSolver(P, device_index=0, rank=0, world=1) as ss:  # one rank standing in for rank 3 of 8
        ss.upload(rec, off)
        del rec
        ss.solver.solve_batched(x0)
        raw = ss.comm.gather_results(lo, P)
        ss.solver.solve_batched(x0)
        raw2 = ss.comm.gather_results(lo, P)
        raw3, st = ss.comm.solve_gather(x0, lo, P)  # the one-call step bench.py times
        assert st.fused == 1 and st.problems == P and st.not_converged == 0
    assert np.array_equal(raw, raw2) and np.array_equal(raw, raw3)
    assert st.iterations == int(raw[:, 9].sum())
    assert np.array_equal(raw[:, 11], lo + np.arange(P))
    assert np.isin(raw[:, 10], (1, 2, 3)).all()
    assert np.all(raw[:, 7] <= raw[:, 8])
    assert np.abs(np.linalg.norm(raw[:, 3:7], axis=1) - 1).max() < 1e-14
    worst = max(np.abs(sd.T_from_pose7(raw[k, :7]) - gt[k]).max() for k in range(P))
    assert worst < 0.02, worst
    # ALL 8 192 problems against the oracle's frozen answers (tests/golden/oracle_c4_shard.npz, made by tests/golden/make_c4_golden.py; the
    # file itself is checked against the live oracle on the CPU: tests/test_golden_regression.py) ...
    G = _golden_c4()
    assert (G["meta"]["seed"], G["meta"]["lo"], G["meta"]["P"]) == (seed, lo, P)
    assert np.array_equal(raw[:, 9], G["iterations"]) and np.array_equal(raw[:, 10], G["termination"])
    dT = np.abs(_T_batch(raw[:, :7]) - _T_batch(G["pose"])).reshape(P, -1).max(axis=1)
    assert dT.max() <= T_TOL and np.abs(raw[:, 7] - G["final_cost"]).max() <= COST_TOL, (dT.max(), np.abs(raw[:, 7] - G["final_cost"]).max())
    assert np.abs(raw[:, 8] - G["initial_cost"]).max() <= 1e-11 * G["initial_cost"].max()
    # ... and 16 of them against the oracle run live (every 512th)
    for k in range(0, P, 512):
        one = sd.sim_shard(seed, lo + k, lo + k + 1, n_poses, K, 0.01)
        r1, _ = one.records()
        ref = oracle_mod.solve(r1, x0[k], linear_solver="qr")
        assert np.array_equal(one.start_poses()[0], x0[k])
        assert raw[k, 9] == ref.summary.num_iterations and raw[k, 10] == ref.summary.termination, k
        assert _dT(raw[k, :7], ref.pose) <= T_TOL and abs(raw[k, 7] - ref.summary.final_cost) <= COST_TOL, k


def _golden_c4():
    import json, os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_c4_shard.npz"))
    G = {k: z[k] for k in ("iterations", "termination", "final_cost", "initial_cost", "pose")}
    G["meta"] = json.loads(str(z["meta"]))
    return G


def _T_batch(p):
    """[P, 7] poses -> [P, 3, 4] of R | t (Appendix B of SURVEY.md: toRotationMatrix, no normalisation)."""
    t, x, y, z, w = p[:, :3], p[:, 3], p[:, 4], p[:, 5], p[:, 6]
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                  2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                  2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], axis=1).reshape(-1, 3, 3)
    return np.concatenate([R, t[:, :, None]], axis=2)


def _golden_variants():
    import json, os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_variants.json")))


@pytest.mark.parametrize("seed", range(10))
@pytest.mark.parametrize("sigma", [0.0, 0.01])
def test_c1_all_ten_seeds_against_frozen_oracle(sv, seed, sigma):
    """SURVEY.md 8(d) C1 in full: the simulation node's default input, seeds 0..9, noise-free and sigma = 0.01 m, init Tcl = I — iterations,
    termination, accepted / rejected steps, T_cl and cost against the oracle's frozen answers (tests/golden/oracle_variants.json)."""
    g = next(v for v in _golden_variants()["c1"] if v["seed"] == seed and v["sigma"] == sigma)
    rec = clc.flatten_observations(sd.GenerateSimData(seed, noise_sigma=sigma), False)
    assert rec.shape[0] == g["n"]
    sv.set_launch(0, -1)
    sv.upload(rec)
    res = sv.solve(np.array(g["start"]))
    assert (res.summary.num_iterations, res.summary.termination) == (g["iterations"], g["termination"])
    assert [t.step_is_successful for t in res.trace] == g["accepted"]
    assert _dT(res.pose, np.array(g["pose"])) <= T_TOL and abs(res.summary.final_cost - g["final_cost"]) <= COST_TOL
    if sigma == 0.0:
        assert np.abs(sd.T_from_pose7(res.pose) - sd.tlc_to_tcl(sd.GT_RLC, sd.GT_TLC)).max() < 1e-6 and res.summary.final_cost < 1e-12


@pytest.mark.parametrize("sigma", [0.0, 0.01])
@pytest.mark.parametrize("init", ["identity", "closed_form"])
def test_c2_variants_against_frozen_oracle(sv, sigma, init):
    """SURVEY.md 8(d) C2 in full: 1e6 observations, noise-free and sigma = 0.01 m, init Tcl = I and init = the closed form (the frozen
    start pose, so that both solves start from identical bits; the GPU's own closed form is held to 1e-9 of the frozen Tlc beside it)."""
    g = next(v for v in _golden_variants()["c2"] if v["sigma"] == sigma and v["init"] == init)
    rec = clc.flatten_observations(sd.sim_fixed_count(g["seed"], 2000, 500, noise_sigma=sigma), False)
    assert rec.shape[0] == g["n"] == 1_000_000
    sv.set_launch(0, -1)
    sv.upload(rec)
    if init == "closed_form":
        Tlc, unobs, _ = sv.closed_form()
        assert not unobs and np.abs(Tlc - np.array(g["Tlc"]).reshape(4, 4)).max() < 1e-9
    res = sv.solve(np.array(g["start"]))
    assert sv.path_info().coop_resident == 1
    assert (res.summary.num_iterations, res.summary.termination) == (g["iterations"], g["termination"])
    assert [t.step_is_successful for t in res.trace] == g["accepted"]
    assert _dT(res.pose, np.array(g["pose"])) <= T_TOL and abs(res.summary.final_cost - g["final_cost"]) <= COST_TOL


# ---------------------------------------------------------------------------------------
# the reference iterates ONE ceres::Solve over whatever N it is given
# (src/LaseCamCalCeres.cpp:299-309).  Beyond what the chip holds (> 256 x 256 x 40 observations) clc_solve with the DEFAULT flags runs the
# step chain (csrc/clc_kernels.hpp step_kernel: one launch per LM iteration, the rows re-streamed from the Infinity Cache / from HBM).
@pytest.mark.parametrize("n_poses, beyond_cache", [(8000, False), (64000, True)])
def test_default_path_beyond_chip_capacity(oracle_mod, n_poses, beyond_cache):
    """4e6 observations (70 MB of rows: cache-resident) and 3.2e7 (557 MB: beyond the 256 MiB Infinity Cache, non-temporal row loads), default
    flags, against the oracle: same termination and iteration count, T_cl <= 1e-6, cost <= 1e-8, every trace record's decision equal, and
    the solve repeats bit for bit.  Oracle: DENSE_QR at 4e6; at 3.2e7 the normal-equation form with threads (the dense 3.2e7 x 6 Jacobian +
    Householder QR is 1.5 GB and ~100 s per solve on one core; test_oracle_solver.py pins 'ne' against 'qr' to 1e-15 on the CPU)."""
    rec = clc.flatten_observations(sd.sim_fixed_count(1000 + n_poses, n_poses, 500, noise_sigma=0.01), False)
    assert rec.shape[0] == n_poses * 500
    with clc.Solver(0) as s:
        s.upload(rec)
        pi = s.path_info()
        assert pi.coop_resident == 0 and pi.single_resident == 0  # neither on-chip form holds it: the step chain is the default path
        assert pi.rows_layout == 1 and pi.n_rows == n_poses * 8   # 500-point scans = 8 rows of 64
        layout_bytes = pi.n_rows * (64 * 16 + 64)
        assert (layout_bytes > (256 << 20) * 3 // 2) == beyond_cache
        res = s.solve(X0)
        again = s.solve(X0)
        assert s.path_info().coop_solves == 0
    ref = oracle_mod.solve(rec, X0, linear_solver="ne" if beyond_cache else "qr",
                           threads=min(64, oracle_mod.max_threads()) if beyond_cache else 1)
    assert res.summary.termination == ref.summary.termination == 3  # CONVERGENCE(function)
    assert res.summary.num_iterations == ref.summary.num_iterations
    assert _dT(res.pose, ref.pose) <= T_TOL
    assert abs(res.summary.final_cost - ref.summary.final_cost) <= COST_TOL
    assert abs(res.summary.initial_cost - ref.summary.initial_cost) <= 1e-11 * ref.summary.initial_cost
    assert len(res.trace) == len(ref.trace)
    for a, b in zip(res.trace, ref.trace):
        assert (a.step_is_valid, a.step_is_successful) == (b.step_is_valid, b.step_is_successful)
        assert abs(a.cost - b.cost) <= 1e-11 * max(b.cost, 1e-300)
    assert np.array_equal(res.pose, again.pose) and res.summary.final_cost == again.summary.final_cost  # fixed summation order
    assert np.abs(sd.T_from_pose7(res.pose) - sd.tlc_to_tcl(sd.GT_RLC, sd.GT_TLC)).max() < 2e-3  # ground truth; sigma = 0.01 m over >= 4e6 points
