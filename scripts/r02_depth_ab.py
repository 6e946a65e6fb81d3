#!/usr/bin/env python3
"""Single-problem evaluation at large sizes + C2 solve for the library named by CLC_LIBRARY (A/B of ROWS_DEPTH builds)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
sv = clc.Solver(0)
x0 = sd.pose7_from_T(np.eye(4))
tag = os.path.basename(os.environ.get("CLC_LIBRARY", "default"))
for poses in (2000, 32000, 64000):
    S = sd.sim_fixed_count(1000 if poses == 2000 else 7, poses, 500, noise_sigma=0.01)
    sv.upload(clc.flatten_observations(S, False)); del S
    k = min(sv.time_eval(x0, reps=100 if poses == 2000 else 20) for _ in range(4)) * 1e3
    line = f"{tag} N={poses*500}: eval {k:.2f} us"
    if poses == 2000:
        for _ in range(5): r = sv.solve(x0, trace_cap=0)
        st = min(sv.time_steps(x0, 2, r.summary.num_evaluations - 1)[0] for _ in range(5)) * 1e3
        line += f", step {st:.2f} us"
    print(line, flush=True)
