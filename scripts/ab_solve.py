#!/usr/bin/env python3
"""A/B of two builds of the library on the same box: CLC_LIBRARY=<path> python scripts/ab_solve.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
x0 = sd.pose7_from_T(np.eye(4))
sv = clc.Solver(0)
for seed, poses, pts in ((1000, 2000, 500), (7, 2000, 500), (7, 50, 110)):
    S = sd.sim_fixed_count(seed, poses, pts, noise_sigma=0.01)
    sv.upload(clc.flatten_observations(S, False))
    for _ in range(10): r = sv.solve(x0, trace_cap=0)
    best = 1e9
    for rep in range(5):
        t = time.perf_counter()
        for _ in range(50): r = sv.solve(x0, trace_cap=0)
        best = min(best, (time.perf_counter() - t) / 50)
    print(f"{os.environ.get('CLC_LIBRARY','default'):>28s} seed {seed} N={poses*pts}: {best*1e6:.1f} us/solve, {r.summary.num_evaluations} passes, {best*1e6/r.summary.num_evaluations:.2f} us/pass")
