#!/usr/bin/env python3
"""What the Python / ctypes wrapper adds to a C2 solve: wall time per solve of the bench's loop against clc_solve's own wall time
(clc_summary.solve_ms).  Round 5, MI355X: 84.5-85.1 us against 82.4-83.0 us = 2.1 us of Python per step (the kernel is 76.5 us: launch +
completion flag cost the other 6 us)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
x0 = sd.pose7_from_T(np.eye(4))
sv = clc.Solver(0)
rec = clc.flatten_observations(sd.sim_fixed_count(1000, 2000, 500, noise_sigma=0.01), False)
sv.upload(rec)
for _ in range(50): r = sv.solve(x0, trace_cap=0)
for rep in range(3):
    t0 = time.perf_counter(); inner = 0.0
    for _ in range(200):
        r = sv.solve(x0, trace_cap=0); inner += r.summary.solve_ms
    wall = (time.perf_counter() - t0) / 200 * 1e3
    print("python wall per solve %.4f ms, clc_solve's own wall %.4f ms, python overhead %.2f us" % (wall, inner / 200, 1e3 * (wall - inner / 200)))
