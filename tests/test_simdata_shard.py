"""The vectorised C3/C4 shard generator: every problem is a pure function of (seed, global index), and its flat
records are bitwise what the reference's residual-block loop (through clc_flatten_observations / the oracle's
flatten) builds from the same Oberserve data."""
import numpy as np

import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd


def test_shard_records_equal_flatten_of_each_problem(oracle_mod):
    sh = sd.sim_shard(7, 100, 140, 6, 50, noise_sigma=0.01)
    rec, off = sh.records()
    assert rec.shape == (40 * 6 * 50, 8) and off[-1] == rec.shape[0]
    for k in (0, 17, 39):
        S = sh.problem(k)
        for lf in (False, True):
            assert np.array_equal(clc.flatten_observations(S, lf, False), rec[off[k]:off[k + 1]])
        assert np.array_equal(oracle_mod.flatten(S, False, False), rec[off[k]:off[k + 1]])
    assert np.all(rec[:, 6] == 0.0)  # scan points live in the lidar plane z = 0 (src/utilities.cpp:203)


def test_problem_is_a_pure_function_of_seed_and_global_index():
    a = sd.sim_shard(7, 100, 140, 6, 50, noise_sigma=0.01)
    b = sd.sim_shard(7, 117, 118, 6, 50, noise_sigma=0.01)
    assert np.array_equal(b.pts[0], a.pts[17]) and np.array_equal(b.tag_q[0], a.tag_q[17])
    assert np.array_equal(b.gt_Tcl[0], a.gt_Tcl[17]) and np.array_equal(b.start_poses()[0], a.start_poses()[17])
    c = sd.sim_shard(8, 117, 118, 6, 50, noise_sigma=0.01)
    assert not np.array_equal(c.pts[0], b.pts[0])
    rec, off, x0, gt = sd.sim_shard_records(7, 100, 140, 6, 50, 0.01, chunk=16)
    r2, o2 = a.records()
    assert np.array_equal(rec, r2) and np.array_equal(off, o2) and np.array_equal(gt, a.gt_Tcl)


def test_shard_problems_are_solvable(oracle_mod):
    sh = sd.sim_shard(3, 0, 4, 20, 100, noise_sigma=0.01)
    rec, off = sh.records()
    x0 = sh.start_poses()
    for k in range(4):
        ref = oracle_mod.solve(rec[off[k]:off[k + 1]], x0[k], linear_solver="qr")
        assert ref.summary.termination in (1, 2, 3)
        assert np.abs(sd.T_from_pose7(ref.pose) - sh.gt_Tcl[k]).max() < 0.02
