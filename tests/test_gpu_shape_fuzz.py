"""Seeded shape fuzz of the default clc_solve against the oracle (DENSE_QR Ceres restatement): 200 single problems whose
shapes sit on the edges of the on-chip paths' capacity logic —

  * the single-workgroup resident kernel (csrc/clc_resident.hpp) holds up to 512 lanes x 22 points = 11 264 observations:
    n = 11 263 .. 11 266, n = 512 j and 512 j +- 1 (a lane more, a point per lane more);
  * the cooperative kernel (csrc/clc_coop.hpp) takes over at 11 265 — or earlier, when the scans' lengths leave too many half-filled
    lanes or there are more than 512 scans — and holds 65 536 lanes x 40 points: 16 / 17 points per
    lane (the register part / the first LDS slot) and 40 / 41 (the last problem it holds / the first one the step chain keeps);
  * about one problem in seven has points off the lidar plane (p.z != 0): never one workgroup's layout, always the cooperative kernel's
    24-byte-slot form;
  * scans: one scan only, 511 / 512 / 513 scans (a scan per lane of the single workgroup, +- 1), scans of 1-3 points, scans
    longer than a lane's chunk (chunks cut scans), ragged mixtures.

Every problem: same termination and iteration count as the oracle, T_cl within 1e-6 and final cost within 1e-8 (the gates of
BASELINE.json); the path that ran is checked where the shape decides it."""
import time

import numpy as np
import pytest

import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd

pytestmark = pytest.mark.gpu

T_TOL = 1e-6
COST_TOL = 1e-8
SINGLE_CAP = 512 * 22
COOP_LANES = 256 * 256


@pytest.fixture(scope="module")
def sv():
    s = clc.Solver(0)
    yield s
    s.close()


class Pool:
    """Scans of K points each (records [S*K, 8]) around one ground truth; a problem = a prefix of some of them."""

    def __init__(self, seed, n_scans, K, noise):
        self.K, self.S = K, n_scans
        self.rec = clc.flatten_observations(sd.sim_fixed_count(seed, n_scans, K, noise_sigma=noise), False).reshape(n_scans, K, 8)
        self.gt = sd.pose7_from_T(sd.tlc_to_tcl(sd.GT_RLC, sd.GT_TLC))

    def take(self, lens, rng):
        lens = np.asarray(lens, dtype=np.int64)
        assert len(lens) <= self.S and lens.max() <= self.K and lens.min() >= 1
        scans = rng.permutation(self.S)[: len(lens)]
        return np.ascontiguousarray(np.concatenate([self.rec[s, :l] for s, l in zip(scans, lens)]))


def _split(n, parts, rng, cap):
    """`parts` positive integers <= cap that sum to n."""
    parts = int(max(-(-n // cap), min(parts, n)))
    w = rng.random(parts) + 0.05
    lens = np.maximum(1, np.floor(w / w.sum() * n).astype(np.int64))
    lens = np.minimum(lens, cap)
    d = n - int(lens.sum())
    while d != 0:
        i = int(rng.integers(parts))
        step = int(np.sign(d)) * min(abs(d), max(1, abs(d) // parts))
        new = int(np.clip(lens[i] + step, 1, cap))
        d -= new - int(lens[i])
        lens[i] = new
    return lens


def _shapes(rng):
    """(label, scan lengths) of the 196 small problems."""
    out = []
    for n in (SINGLE_CAP - 1, SINGLE_CAP, SINGLE_CAP + 1, SINGLE_CAP + 2):  # 16 on the single-workgroup capacity edge
        for parts in (23, 60, 300, 512):
            out.append((f"n={n} in {parts} scans", _split(n, parts, rng, 500)))
    for j in (1, 2, 7, 16, 21, 22):  # 18: a point per lane more
        for dn in (-1, 0, 1):
            n = 512 * j + dn
            if n <= SINGLE_CAP + 1:
                out.append((f"n=512*{j}{dn:+d}", _split(n, int(rng.integers(3, 120)), rng, 500)))
    for n in (6, 7, 63, 64, 65, 200, 499):  # 7: one scan only (rank-deficient: one plane)
        out.append((f"one scan of {n}", np.array([n])))
    for ns in (511, 512, 513):  # 12: a scan per lane of the single workgroup, +- 1
        for lo, hi in ((1, 1), (1, 3), (5, 22), (1, 40)):
            lens = rng.integers(lo, hi + 1, size=ns)
            if lens.sum() > SINGLE_CAP and hi <= 22:
                lens = np.minimum(lens, 21)
            out.append((f"{ns} scans of {lo}..{hi}", lens))
    for _ in range(20):  # short scans, many of them
        ns = int(rng.integers(30, 600))
        out.append(("short scans", rng.integers(1, 4, size=ns)))
    for _ in range(30):  # scans longer than a lane's chunk in the single workgroup (chunks cut scans)
        ns = int(rng.integers(3, 24))
        out.append(("long scans", rng.integers(200, 501, size=ns)))
    for _ in range(35):  # the cooperative kernel's small end: 1-2 points per lane, most lanes empty
        n = int(rng.integers(SINGLE_CAP + 1, 90000))
        out.append((f"coop n={n}", _split(n, int(rng.integers(23, 600)), rng, 500)))
    for j in (1, 2):  # 6: cooperative kernel, a point per lane more
        for dn in (-1, 0, 1):
            n = COOP_LANES * j + dn
            out.append((f"coop n=65536*{j}{dn:+d}", _split(n, 600, rng, 500)))
    while len(out) < 196:  # ragged mixtures
        ns = int(rng.integers(3, 400))
        lens = np.where(rng.random(ns) < 0.3, rng.integers(1, 4, size=ns), rng.integers(4, 200, size=ns))
        if lens.sum() > 150000:
            continue
        out.append(("ragged mixture", lens))
    return out[:196]


def _check(sv, oracle_mod, rec, x0, label, use_loss=True, threads=1):
    o = clc.default_options()
    oo = oracle_mod.default_options()
    if not use_loss:
        o.use_loss = 0
        oo.use_loss = 0
    sv.upload(rec)
    r = sv.solve(x0, o, trace_cap=0)
    ref = oracle_mod.solve(rec, x0, oo, linear_solver="qr", threads=threads)
    n = rec.shape[0]
    tag = (label, n, use_loss)
    assert r.summary.termination == ref.summary.termination, tag
    assert r.summary.num_iterations == ref.summary.num_iterations, tag
    dT = np.abs(sd.T_from_pose7(r.pose) - sd.T_from_pose7(ref.pose)).max()
    assert dT <= T_TOL and abs(r.summary.final_cost - ref.summary.final_cost) <= COST_TOL, tag + (dT,)
    return r


def test_shape_fuzz_200_single_problems_around_the_capacity_edges(sv, oracle_mod):
    rng = np.random.default_rng(20260925)
    pool = Pool(11, 640, 500, 0.01)
    sv.set_launch(0, -1)
    t0 = time.perf_counter()
    paths = {"single": 0, "coop": 0, "chain": 0}
    shapes = _shapes(rng)
    assert len(shapes) == 196
    n_z = 0
    for label, lens in shapes:
        rec = pool.take(lens, rng)
        n = rec.shape[0]
        with_z = bool(rng.random() < 0.15)  # points off the lidar plane: never one workgroup's layout, the cooperative kernel's z form
        if with_z:
            idx = np.arange(0, n, int(rng.integers(1, 4)))
            rec[idx, 6] = rng.normal(size=idx.shape[0]) * 0.02
            n_z += 1
        x0 = sv.pose_plus(pool.gt[None, :], rng.normal(size=(1, 6)) * (0.03 if len(lens) > 2 else 0.005))[0]
        _, _, solves0, aborts0, off = sv.debug_coop()
        assert not off
        r = _check(sv, oracle_mod, rec, x0, label, use_loss=bool(rng.random() < 0.8))
        single_ok = sv.debug_resident_single()[0]
        built, _, solves1, aborts1, _ = sv.debug_coop()
        assert aborts1 == aborts0
        # exactly one of the two on-chip layouts exists: the single workgroup's when the problem fits it (a lane holds points of ONE
        # scan: n <= 11 264 is necessary, not sufficient), the cooperative kernel's otherwise — never the step chain at these sizes
        assert single_ok != built, (label, n)
        if n > SINGLE_CAP or len(lens) > 512 or with_z:
            assert built and sv.path_info().coop_points_carry_z == int(with_z), (label, n)
        if single_ok:
            assert solves1 == solves0, (label, n)
            paths["single"] += 1
        else:
            assert solves1 == solves0 + 1, (label, n)  # ONE launch of the cooperative kernel
            paths["coop"] += 1
    # the four big ones: 16 / 17 and 40 / 41 points per lane of the cooperative kernel.  A lane holds points of ONE scan, so the
    # exact-fit shapes have scans of a whole number of full lanes: 2 048 scans x 512 = 65 536 lanes x 16, 4 096 scans x 640 =
    # 65 536 lanes x 40 (the last problem the kernel holds); one observation more needs a 17th point in some lane / does not
    # fit and keeps the step chain
    big = Pool(12, 4097, 640, 0.01)
    for ppl, K, dn in ((16, 512, 0), (16, 512, 1), (40, 640, 0), (40, 640, 1)):
        lens = np.full(COOP_LANES * ppl // K + dn, K)
        if dn:
            lens[-1] = 1
        rec = big.take(lens, rng)
        assert rec.shape[0] == COOP_LANES * ppl + dn
        x0 = sv.pose_plus(big.gt[None, :], rng.normal(size=(1, 6)) * 0.03)[0]
        _, _, solves0, aborts0, _ = sv.debug_coop()
        r = _check(sv, oracle_mod, rec, x0, f"coop {ppl} points per lane {dn:+d}", threads=oracle_mod.max_threads())
        built, ppl_built, solves1, aborts1, _ = sv.debug_coop()
        assert aborts1 == aborts0
        if (ppl, dn) == (40, 1):
            assert not built and solves1 == solves0  # the step chain
            paths["chain"] += 1
        else:
            assert built and ppl_built == ppl + dn and solves1 == solves0 + 1, (ppl, dn, built, ppl_built)
            paths["coop"] += 1
    dt = time.perf_counter() - t0
    print(f"shape fuzz: {paths}, {n_z} with z, {dt:.1f} s")
    assert n_z >= 15
    assert paths["single"] >= 60 and paths["coop"] >= 40 and paths["chain"] == 1
    assert dt < 60.0, dt
