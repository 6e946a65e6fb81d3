#!/usr/bin/env python3
"""Step period / solve time vs number of workgroups of the step kernel (fewer workgroups = fewer partial rows for every
workgroup to re-read before its controller; more rows of points per wave)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
x0 = sd.pose7_from_T(np.eye(4))
sv = clc.Solver(0)
for poses, pts in ((50, 110), (200, 500), (2000, 500), (8000, 500)):
    S = sd.sim_fixed_count(1000 if poses == 2000 else 7, poses, pts, noise_sigma=0.01)
    sv.upload(clc.flatten_observations(S, False))
    for grid in (0, 224, 192, 160, 128, 96, 64, 32, 0):
        sv.set_launch(grid, -1)
        for _ in range(4): r = sv.solve(x0, trace_cap=0)
        p = r.summary.num_evaluations
        st = min(sv.time_steps(x0, 2, p - 1)[0] for _ in range(5)) * 1e3
        best = 1e9
        for rep in range(4):
            t = time.perf_counter()
            for _ in range(30): sv.solve(x0, trace_cap=0)
            best = min(best, (time.perf_counter() - t) / 30)
        print(f"N={sv.num_observations} grid={grid or 'default'}: step {st:.2f} us, solve {best*1e6:.1f} us ({p} passes)", flush=True)
    sv.set_launch(0, -1)
