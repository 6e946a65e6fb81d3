#!/usr/bin/env python3
"""Stepped solve (flag 128: one step_kernel launch per LM iteration) vs the two-kernel path: identical results, timing."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
x0 = sd.pose7_from_T(np.eye(4))
sv = clc.Solver(0)
for seed, poses, pts in ((1000, 2000, 500), (7, 2000, 500), (7, 200, 500), (7, 50, 110), (3, 8000, 500)):
    S = sd.sim_fixed_count(seed, poses, pts, noise_sigma=0.01)
    sv.upload(clc.flatten_observations(S, False))
    out = {}
    for fl in (50, 178, 50, 178):
        sv.set_launch(0, fl)
        for _ in range(10): r = sv.solve(x0)
        best = 1e9
        for rep in range(5):
            t = time.perf_counter()
            for _ in range(40): r2 = sv.solve(x0, trace_cap=0)
            best = min(best, (time.perf_counter() - t) / 40)
        out.setdefault(fl, r)
        same = np.array_equal(r.pose, out[50].pose) and r.summary.final_cost == out[50].summary.final_cost and \
            [t.cost for t in r.trace] == [t.cost for t in out[50].trace]
        print(f"seed {seed} N={poses*pts} flags={fl}: {best*1e6:7.1f} us/solve, {r.summary.num_evaluations} passes, {best*1e6/r.summary.num_evaluations:.2f} us/pass, "
              f"iters {r.summary.num_iterations} term {r.summary.termination} identical-to-two-kernel {same}", flush=True)
