#!/usr/bin/env python3
"""Rows that carry z (p.z != 0; csrc/clc_rows.hpp rows3_*) beside the 16-byte rows, on the GPU box:
   C2 shape (1e6 observations): evaluation kernel, clc_solve, step period;  3.2e7 observations (beyond the Infinity Cache):
   evaluation kernel and the fraction of 8 TB/s its moved bytes reach.  Prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd

x0 = sd.pose7_from_T(np.eye(4))
rec = clc.flatten_observations(sd.sim_fixed_count(1000, 2000, 500, noise_sigma=0.01), False)
recz = rec.copy()
recz[:, 6] = np.random.default_rng(0).normal(size=rec.shape[0]) * 0.01
sv = clc.Solver(0)
out = {}
for name, r in (("flat", rec), ("z", recz)):
    sv.upload(r)
    assert sv.rows_carry_z()[0] == (name == "z")
    n_rows = sv.debug_rows()[1]
    ev = min(sv.time_eval(x0, reps=20) for _ in range(3))
    t = []
    for _ in range(7):
        t0 = time.perf_counter(); res = sv.solve(x0); t.append(time.perf_counter() - t0)
    step, passes = sv.time_steps(x0, 2, res.summary.num_evaluations - 1)
    big = np.ascontiguousarray(np.tile(r, (32, 1)))
    sv.upload(big)
    nb = big.shape[0]; del big
    nrb = sv.debug_rows()[1]
    evb = min(sv.time_eval(x0, reps=12) for _ in range(3))
    moved = nrb * (64 * (24 if name == "z" else 16) + 64)
    out[name] = {"c2_rows": n_rows, "c2_eval_us": 1e3 * ev, "c2_solve_ms_median": 1e3 * float(np.median(t)), "c2_step_us": 1e3 * step,
                 "iterations": res.summary.num_iterations, "big_observations": nb, "big_eval_us": 1e3 * evb,
                 "big_bytes_per_obs": moved / nb, "big_frac_moved": moved / (evb * 1e-3) / 8e12}
print(json.dumps(out))
