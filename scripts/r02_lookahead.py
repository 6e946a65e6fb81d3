#!/usr/bin/env python3
"""Launch-ahead depth vs step period / solve time at C2 (run once per CLC_LAUNCH_AHEAD value)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
x0 = sd.pose7_from_T(np.eye(4))
sv = clc.Solver(0)
for poses in (50, 2000, 8000):
    S = sd.sim_fixed_count(1000 if poses == 2000 else 7, poses, 500 if poses > 50 else 110, noise_sigma=0.01)
    sv.upload(clc.flatten_observations(S, False))
    for _ in range(5): r = sv.solve(x0, trace_cap=0)
    p = r.summary.num_evaluations
    st = min(sv.time_steps(x0, 2, p - 1)[0] for _ in range(5)) * 1e3
    best = 1e9
    for rep in range(5):
        t = time.perf_counter()
        for _ in range(40): sv.solve(x0, trace_cap=0)
        best = min(best, (time.perf_counter() - t) / 40)
    print(f"launch_ahead={os.environ.get('CLC_LAUNCH_AHEAD','default')} N={sv.num_observations}: step {st:.2f} us, solve {best*1e6:.1f} us ({p} passes, {best*1e6/p:.2f} us/pass)", flush=True)
