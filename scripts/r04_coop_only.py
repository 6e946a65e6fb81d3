#!/usr/bin/env python3
"""C2 cooperative solve: kernel time (HIP events) and wall time, for A/B runs of library variants (CLC_LIBRARY)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import camlasercalibratool_amd as clc
from camlasercalibratool_amd import simdata as sd
x0 = sd.pose7_from_T(np.eye(4))
sv = clc.Solver(0)
n_poses, pts = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2000, 500)
rec = clc.flatten_observations(sd.sim_fixed_count(1000, n_poses, pts, noise_sigma=0.01), False)
sv.upload(rec)
for _ in range(5):
    res = sv.solve(x0, trace_cap=0)
t = []
for _ in range(9):
    t0 = time.perf_counter()
    for _ in range(20):
        res = sv.solve(x0, trace_cap=0)
    t.append((time.perf_counter() - t0) / 20)
o = clc.default_options(); o.profile_events = 2
k = [sv.solve(x0, o, trace_cap=0).summary.eval_kernel_ms for _ in range(9)]
print(json.dumps({"lib": os.path.basename(os.environ.get("CLC_LIBRARY", "default")), "obs": int(rec.shape[0]), "ppl": sv.debug_coop()[1], "aborts": sv.debug_coop()[3],
                  "solve_ms": 1e3 * float(np.median(t)), "kernel_ms_min": min(k), "kernel_ms_med": float(np.median(k)), "passes": int(res.summary.num_evaluations),
                  "us_per_pass": 1e3 * min(k) / res.summary.num_evaluations, "iters": int(res.summary.num_iterations), "final_cost": res.summary.final_cost}))
