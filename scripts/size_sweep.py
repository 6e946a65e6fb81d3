#!/usr/bin/env python3
"""Evaluation-kernel and whole-solve time across problem sizes (default launch settings)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
x0 = sd.pose7_from_T(np.eye(4))
sv = clc.Solver(0)
for poses, pts in ((50, 110), (200, 500), (1000, 500), (2000, 500), (8000, 500), (32000, 500)):
    S = sd.sim_fixed_count(7, poses, pts, noise_sigma=0.01)
    rec = clc.flatten_observations(S, False)
    sv.upload(rec)
    k = min(sv.time_eval(x0, reps=100) for _ in range(3)) * 1e3
    for _ in range(5): r = sv.solve(x0, trace_cap=0)
    t = time.perf_counter()
    for _ in range(30): r = sv.solve(x0, trace_cap=0)
    dt = (time.perf_counter() - t) / 30
    n = rec.shape[0]
    print(f"N={n:>9d}: eval kernel {k:7.2f} us ({64*n/k/1e6:7.0f} GB/s algorithmic), solve {dt*1e3:.3f} ms, {r.summary.num_evaluations} passes, {r.summary.num_evaluations*n/dt:.3e} evals/s")
