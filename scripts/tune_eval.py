#!/usr/bin/env python3
"""GPU tuning sweep of the evaluation kernel (grid size x variant flags x N) and of the
solve loop's launch-ahead depth.  Run on the GPU box:  python scripts/tune_eval.py"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd

S = sd.sim_fixed_count(1000, 2000, 500, noise_sigma=0.01)
rec = clc.flatten_observations(S, False)
x0 = sd.pose7_from_T(np.eye(4))
sv = clc.Solver(0)
print(sv.device_info())
res = {}
for n_mult in (1, 8):
    big = np.ascontiguousarray(np.tile(rec, (n_mult, 1)))
    sv.upload(big)
    n = big.shape[0]
    for flags in (18, 50):
        row = []
        for grid in (192, 256, 512):
            sv.set_launch(grid, flags)
            ms = min(sv.time_eval(x0, reps=100) for _ in range(3))
            row.append((grid, ms, 64 * n / ms / 1e6))
        res[(n, flags)] = row
        print(f"N={n} flags={flags}: " + "  ".join(f"{g}:{ms*1e3:.1f}us/{gb:.0f}GB/s" for g, ms, gb in row), flush=True)
# cost-only and no-loss variants at default geometry
sv.upload(rec)
sv.set_launch(0, -1)
print("default  jac+loss %.2f us | cost-only %.2f us | no-loss %.2f us" % (
    1e3 * sv.time_eval(x0, reps=100), 1e3 * sv.time_eval(x0, reps=100, with_jacobian=False),
    1e3 * sv.time_eval(x0, reps=100, with_loss=False)))
# solve loop: launch-ahead depth
for flags, grid in ((18, 256), (50, 256)):
    sv.set_launch(grid, flags)
    for la in (2, 3):
        o = clc.default_options(); o.launch_ahead = la
        for _ in range(5): sv.solve(x0, o)
        t = time.perf_counter()
        for _ in range(50): r = sv.solve(x0, o)
        dt = (time.perf_counter() - t) / 50
        print(f"solve grid={grid} flags={flags} lookahead={la}: {dt*1e3:.3f} ms/solve, {r.summary.num_evaluations} passes, {dt*1e6/r.summary.num_evaluations:.1f} us/pass", flush=True)
o = clc.default_options(); o.profile_events = 1
sv.set_launch(0, -1)
for _ in range(3): r = sv.solve(x0, o)
print("profiled solve: kernel avg %.2f us over %d launches, solve %.3f ms" % (1e3 * r.summary.eval_kernel_ms / r.summary.eval_kernel_launches, r.summary.eval_kernel_launches, r.summary.solve_ms))
