#!/usr/bin/env python3
"""Evaluation-kernel and whole-solve time across problem sizes.

usage: size_sweep.py [launch_flags ...]    (default: -1 = library default; see clc_set_launch)
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd

x0 = sd.pose7_from_T(np.eye(4))
sv = clc.Solver(0)
flag_sets = [int(a) for a in sys.argv[1:]] or [-1]
for poses, pts in ((50, 110), (200, 500), (1000, 500), (2000, 500), (8000, 500), (32000, 500), (64000, 500)):
    S = sd.sim_fixed_count(7, poses, pts, noise_sigma=0.01)
    rec = clc.flatten_observations(S, False)
    n = rec.shape[0]
    sv.upload(rec)
    for fl in flag_sets:
        sv.set_launch(0, fl)
        k = min(sv.time_eval(x0, reps=100) for _ in range(3)) * 1e3          # us per launch
        for _ in range(5):
            r = sv.solve(x0, trace_cap=0)
        t = time.perf_counter()
        for _ in range(30):
            r = sv.solve(x0, trace_cap=0)
        dt = (time.perf_counter() - t) / 30
        print(f"N={n:>9d} flags={fl:>3d}: eval kernel {k:7.2f} us ({64 * n / k / 1e3:7.0f} GB/s algorithmic), "
              f"solve {dt * 1e3:.3f} ms, {r.summary.num_evaluations} passes, "
              f"{r.summary.num_evaluations * n / dt:.3e} evals/s", flush=True)
